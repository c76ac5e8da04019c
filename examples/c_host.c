/*
 * A host without Python: Serra09 scores for all pairs of a small synthetic pool through the C ABI of libacx
 * (include/acx.h) -- what acoss's Serra09.similarity(idxs) does per pair with two essentia objects
 * (rqa_serra09.py:55-69), for the whole index array in one call.
 *
 *   gcc -std=c99 -Iinclude examples/c_host.c -Lacoss_amd/csrc -lacx -Wl,-rpath,$PWD/acoss_amd/csrc -lm -o /tmp/c_host
 *   /tmp/c_host            (needs a gfx950 GPU; without one acx_create fails and the program says why, exit code 3)
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "acx.h"

int main(void)
{
    enum { N = 4, T = 120 };
    static float frames[N * T * 12];
    int64_t offsets[N + 1];
    /* tracks 0 / 1 and 2 / 3 are "covers": the same chord walk, the second one transposed by 5 bins */
    for (int n = 0; n <= N; ++n) offsets[n] = (int64_t)n * T;
    unsigned s = 12345u;
    for (int n = 0; n < N; ++n)
        for (int t = 0; t < T; ++t) {
            const int chord = ((t / 8) * (n / 2 ? 5 : 7)) % 12, shift = (n & 1) ? 5 : 0;
            float mx = 0.0f, *f = frames + ((size_t)n * T + t) * 12;
            for (int c = 0; c < 12; ++c) {
                s = s * 1664525u + 1013904223u;
                const int src = (c - shift + 12) % 12;
                f[c] = (src == chord ? 1.0f : (src == (chord + 4) % 12 ? 0.8f : (src == (chord + 7) % 12 ? 0.9f : 0.0f))) + 0.05f * (float)(s >> 8) / 16777216.0f;
                if (f[c] > mx) mx = f[c];
            }
            for (int c = 0; c < 12; ++c) f[c] /= mx;
        }
    if (acx_abi_version() != ACX_ABI_VERSION) { fprintf(stderr, "libacx ABI %d, header %d\n", acx_abi_version(), ACX_ABI_VERSION); return 2; }
    int err = 0;
    acx_ctx *ctx = acx_create(0, &err);
    if (!ctx) { fprintf(stderr, "acx_create failed (%d): %s\n", err, acx_last_error(NULL)); return 3; }
    if (acx_upload_pool(ctx, frames, offsets, N, 12) != ACX_OK) { fprintf(stderr, "%s\n", acx_last_error(ctx)); return 4; }
    acx_serra09_params p;
    acx_serra09_default_params(&p);
    int32_t pairs[2 * N * (N - 1) / 2];
    int K = 0;
    for (int i = 0; i < N; ++i) for (int j = i + 1; j < N; ++j) { pairs[2 * K] = i; pairs[2 * K + 1] = j; ++K; }
    float score[N * (N - 1) / 2];
    if (acx_serra09_pairs(ctx, pairs, K, &p, score) != ACX_OK) { fprintf(stderr, "%s\n", acx_last_error(ctx)); return 5; }
    for (int k = 0; k < K; ++k) printf("pair (%d, %d): Qmax %.1f\n", pairs[2 * k], pairs[2 * k + 1], score[k]);
    /* the whole N x N grid in one call, mirrored, then the reference's column normalisation (rqa_serra09.py:71-83) */
    static float D[N * N];
    float *planes[1] = { D };
    acx_grid_spec spec = { ACX_ALGO_SERRA09, 1, 0, 1 };
    if (acx_pair_grid(ctx, &spec, &p, planes, N, 1) != ACX_OK) { fprintf(stderr, "%s\n", acx_last_error(ctx)); return 6; }
    for (int i = 0; i < N; ++i) for (int j = 0; j < N; ++j) D[i * N + j] = (float)((double)D[i * N + j] / sqrt((double)T));
    const int covers_win = D[0 * N + 1] > D[0 * N + 2] && D[0 * N + 1] > D[0 * N + 3] && D[2 * N + 3] > D[2 * N + 0] && D[2 * N + 3] > D[2 * N + 1];
    printf("covers rank first: %s\n", covers_win ? "yes" : "no");
    acx_destroy(ctx);
    return covers_win ? 0 : 7;
}
