/*
 * acx_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the per-pair arithmetic on acoss's all-pairwise
 * cover-song hot path.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product path (acoss_amd/) never
 * does.
 *
 * PARITY STATUS
 *   - Serra09 (OTI + delay embedding + Euclidean CSM + mutual-kappa
 *     binarisation + Qmax): the reference delegates this arithmetic to the
 *     third-party, UN-PINNED dependency `essentia`
 *     (reference setup.py:53; call sites acoss/algorithms/rqa_serra09.py:9,60-67),
 *     which is absent from /root/reference and cannot be installed here, and
 *     the reference holds no test/golden vector for it (test/basetest.py:1-21).
 *     ==> Serra09 PARITY IS UNPINNED.  This file restates the published
 *     algorithm (Serra, Serra & Andrzejak 2009, NJP 11 093017) following the
 *     reference call sites and their parameters; every detail that is only
 *     recalled from essentia is a switchable parameter of acx_o_serra09_params.
 *   - smith_waterman_constrained (acoss/algorithms/utils/alignment_tools.py:7-46)
 *     is pinned by golden vectors generated from the reference itself
 *     (tests/golden/make_goldens.py).
 *
 * Build: oracle/Makefile  (gcc -O3 -ffp-contract=off; no fast-math: the f32
 * operation order below IS the specification the HIP kernels are checked
 * against bit-for-bit).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* The two big per-pair arrays -- the T x T f32 distance matrix and its byte-sized recurrence plot -- live in
 * per-thread buffers that grow on demand and are reused from pair to pair (the frame Gram is a ring of m rows,
 * the alignment three rolling rows): a fresh 16 MB malloc per pair is an mmap / munmap and a page fault per
 * 4 KB, which with one worker per host core dominates everything else.  (Round 2 retuned glibc's malloc
 * process-wide from a library constructor instead; that also changed the process that loaded the oracle.) */
static __thread float *tl_d = NULL;
static __thread size_t tl_d_cap = 0;
static __thread uint8_t *tl_r = NULL;
static __thread size_t tl_r_cap = 0;
static void *tl_grow(void *p, size_t *cap, size_t need)
{
    if (need <= *cap) return p;
    free(p);
    *cap = need + need / 8;
    return malloc(*cap);
}


#define NB 12 /* chroma bins */

typedef struct {
    int32_t m;            /* frameStackSize   (rqa_serra09.py:32  m=9)        */
    int32_t tau;          /* frameStackStride (rqa_serra09.py:32  tau=1)      */
    float kappa;          /* binarizePercentile (rqa_serra09.py:32 0.095)     */
    int32_t oti;          /* rqa_serra09.py:32 oti=True                       */
    float gamma_o;        /* disOnset     (essentia default 0.5; not overridden, rqa_serra09.py:64) */
    float gamma_e;        /* disExtension (essentia default 0.5)              */
    /* --- recalled-essentia switches --- */
    int32_t embed_full;   /* 0: M = T - m*tau (essentia, recalled)   1: M = T-(m-1)*tau (paper) */
    int32_t pct_mode;     /* 0: linear interpolation, exact-integer k -> sorted[k]
                             1: essentia "d0+d1" form (exact-integer k -> 0)
                             2: lower  3: nearest                              */
    int32_t oti_target;   /* 0: rotate the REFERENCE toward the query (essentia)  1: rotate the query */
    int32_t dp_start;     /* 2: cell (i,j) reads R[i][j], i,j>=2 (essentia)  3: reads R[i-1][j-1], i,j>=3 */
    int32_t inclusive;    /* 1: R = [d <= eps]   0: R = [d < eps]             */
    int32_t arith;        /* 0: "tree"  -- per-frame fmaf chains + doubling-tree window sums
                                           (the arithmetic the HIP kernels reproduce bit-for-bit)
                             1: "seq108" -- sequential f32 inner products over the stacked
                                           108-dim vectors, (xx - 2xy) + yy, as essentia's
                                           pairwiseDistance is recalled to do          */
    int32_t dmax;         /* 0: Qmax ('serra09')   1: Dmax ('chen17', latefusion_chen.py:68) */
} acx_o_serra09_params;

/* ------------------------------------------------------------------ */
/* helpers                                                            */
/* ------------------------------------------------------------------ */

/* Global chroma profile: sum over frames (sequential f32), divided by its max.
 * essentia globalAverageChroma + normalize (recalled); SURVEY App. C step 1. */
void acx_o_global_chroma(const float *X, int32_t T, float *g)
{
    for (int c = 0; c < NB; ++c) g[c] = 0.0f;
    for (int t = 0; t < T; ++t)
        for (int c = 0; c < NB; ++c) g[c] = g[c] + X[(size_t)t * NB + c];
    float mx = g[0];
    for (int c = 1; c < NB; ++c) if (g[c] > mx) mx = g[c];
    if (mx > 0.0f)
        for (int c = 0; c < NB; ++c) g[c] = g[c] / mx;
}

/* OTI: argmax_s <ga, roll(gb, s)>, s = 0..12 inclusive (s=12 == s=0), first max
 * wins; separate f32 multiply and add.  SURVEY App. C step 1. */
int32_t acx_o_oti(const float *ga, const float *gb)
{
    int best = 0;
    float bestv = 0.0f;
    for (int s = 0; s <= NB; ++s) {
        float acc = 0.0f;
        for (int c = 0; c < NB; ++c) {
            float p = ga[c] * gb[((c - s) % NB + NB) % NB];
            acc = acc + p;
        }
        if (s == 0 || acc > bestv) { bestv = acc; best = s; }
    }
    return best % NB;
}

/* doubling-tree window sum of m terms s[0], s[stride], ... (see DESIGN.md
 * "arithmetic spec"): W1=s, W2[t]=W1[t]+W1[t+1], W4[t]=W2[t]+W2[t+2], ...;
 * result = W_hb[0] + W_b1[hb] + W_b2[hb+b1] + ... over the set bits of m,
 * high to low. */
static float tree_w(const float *s, size_t stride, int w)
{
    /* pairwise reduction, bottom-up: the same association as the recursive definition
     * tree_w(first half) + tree_w(second half) */
    float v[64];
    v[0] = s[0];
    for (int t = 1; t < w; ++t) v[t] = s[(size_t)t * stride];
    for (int step = 1; step < w; step *= 2)
        for (int t = 0; t + step < w; t += 2 * step) v[t] = v[t] + v[t + step];
    return v[0];
}
static float tree_sum(const float *s, size_t stride, int m)
{
    int hb = 1;
    while (hb * 2 <= m) hb *= 2;
    float acc = tree_w(s, stride, hb);
    int off = hb;
    for (int b = hb / 2; b >= 1; b /= 2) {
        if (m & b) {
            acc = acc + tree_w(s + (size_t)off * stride, stride, b);
            off += b;
        }
    }
    return acc;
}

/* k-th smallest (0-based) of w[0..n) by quickselect (Hoare partition, median-of-three pivot); on return
 * w[k] is that value, everything left of it is <= and everything right of it >= .  Order statistics do
 * not depend on how they are found: same values as sorting the row. */
static float select_kth(float *w, int n, int k)
{
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        int mid = lo + (hi - lo) / 2;
        float a = w[lo], b = w[mid], c = w[hi];
        float pv = (a < b) ? ((b < c) ? b : (a < c ? c : a)) : ((a < c) ? a : (b < c ? c : b));
        int i = lo, j = hi;
        while (i <= j) {
            /* bounded scans: a NaN in the row (every comparison false) cannot run them off the array */
            while (i <= hi && w[i] < pv) ++i;
            while (j >= lo && w[j] > pv) --j;
            if (i <= j) { float t = w[i]; w[i] = w[j]; w[j] = t; ++i; --j; }
        }
        if (k <= j) hi = j;
        else if (k >= i) lo = i;
        else break;
    }
    return w[k];
}

/* percentile of v[0..n) at fraction q.  f32 arithmetic
 * throughout; `work` is scratch of n floats. */
static float percentile_f32(const float *v, int n, float q, int mode, float *work)
{
    memcpy(work, v, (size_t)n * sizeof(float));
    float k = (n > 1) ? (float)(n - 1) * q : (float)n * q;
    float fl = floorf(k), ce = ceilf(k);
    int ilo = (int)fl, ihi = (int)ce;
    if (ilo < 0) ilo = 0;
    if (ihi < 0) ihi = 0;
    if (ilo > n - 1) ilo = n - 1;
    if (ihi > n - 1) ihi = n - 1;
    if (mode == 3) {
        int r = (int)floorf(k + 0.5f);
        if (r > n - 1) r = n - 1;
        if (r < 0) r = 0;
        return select_kth(work, n, r);
    }
    /* sorted[ilo], and sorted[ihi] = the smallest element to its right when ihi == ilo + 1 */
    float vlo = select_kth(work, n, ilo), vhi = vlo;
    if (ihi > ilo) {
        vhi = work[ilo + 1];
        for (int t = ilo + 2; t < n; ++t) if (work[t] < vhi) vhi = work[t];
    }
    switch (mode) {
    case 2: return vlo;
    case 1: {
        float d0 = vlo * (ce - k);
        float d1 = vhi * (k - fl);
        return d0 + d1;
    }
    default:
        if (ihi == ilo) return vlo;
        {
            float d0 = vlo * (ce - k);
            float d1 = vhi * (k - fl);
            return d0 + d1;
        }
    }
}

/* number of embedded frames for a track of T pooled frames */
int32_t acx_o_embed_len(int32_t T, const acx_o_serra09_params *p)
{
    int span = p->embed_full ? (p->m - 1) * p->tau : p->m * p->tau;
    int L = T - span;
    if (L <= 0) return 0;
    return (L + p->tau - 1) / p->tau; /* loop i = 0, tau, 2tau, ... < L */
}

/* ------------------------------------------------------------------ */
/* Serra09 per-pair chain                                              */
/* ------------------------------------------------------------------ */

/*
 * Q (Tq,12), Rf (Tr,12): pooled chroma (f32, row-major).
 * Optional outputs (may be NULL): d (Mq*Mr distances), epsq (Mq), epsr (Mr),
 * bin (Mq*Mr uint8), oti_out.
 * Returns the raw score max(Q) ('symmetric' distanceType), or -1 on
 * bad input (track shorter than the stack).
 */
float acx_o_serra09_pair(const float *Q, int32_t Tq, const float *Rf, int32_t Tr,
                         const acx_o_serra09_params *p,
                         float *d_out, float *epsq_out, float *epsr_out,
                         uint8_t *bin_out, int32_t *oti_out)
{
    const int m = p->m, tau = p->tau;
    /* tree_w sums windows of up to 64 terms on the stack (the device supports m <= 33) */
    if (m < 1 || m > 64 || tau < 1) return -1.0f;
    const int Mq = acx_o_embed_len(Tq, p), Mr = acx_o_embed_len(Tr, p);
    if (Mq <= 0 || Mr <= 0) return -1.0f;

    /* 1. OTI */
    float gq[NB], gr[NB];
    int s = 0;
    float *A = (float *)malloc((size_t)Tq * NB * sizeof(float));
    float *B = (float *)malloc((size_t)Tr * NB * sizeof(float));
    memcpy(A, Q, (size_t)Tq * NB * sizeof(float));
    memcpy(B, Rf, (size_t)Tr * NB * sizeof(float));
    if (p->oti) {
        acx_o_global_chroma(Q, Tq, gq);
        acx_o_global_chroma(Rf, Tr, gr);
        if (p->oti_target == 0) {
            s = acx_o_oti(gq, gr);
            for (int t = 0; t < Tr; ++t)
                for (int c = 0; c < NB; ++c)
                    B[(size_t)t * NB + c] = Rf[(size_t)t * NB + ((c - s) % NB + NB) % NB];
        } else {
            s = acx_o_oti(gr, gq);
            for (int t = 0; t < Tq; ++t)
                for (int c = 0; c < NB; ++c)
                    A[(size_t)t * NB + c] = Q[(size_t)t * NB + ((c - s) % NB + NB) % NB];
        }
    }
    if (oti_out) *oti_out = s;

    /* 2+3. distances */
    float *d = tl_d = (float *)tl_grow(tl_d, &tl_d_cap, (size_t)Mq * Mr * sizeof(float));
    if (p->arith == 0) {
        /* frame-level Gram (fmaf chain over the 12 bins), frame norms.  Row i of d needs the Gram rows
         * i tau .. (i + m - 1) tau only: they live in a ring of `span` rows instead of a Tq x Tr matrix. */
        const int span = (m - 1) * tau + 1;
        float *G = (float *)malloc((size_t)span * Tr * sizeof(float));
        const float **grow = (const float **)malloc((size_t)m * sizeof(float *));
        float *gd = (float *)malloc((size_t)m * sizeof(float));
        float *nq = (float *)malloc((size_t)Tq * sizeof(float));
        float *nr = (float *)malloc((size_t)Tr * sizeof(float));
        for (int a = 0; a < Tq; ++a) {
            float acc = 0.0f;
            for (int c = 0; c < NB; ++c) acc = fmaf(A[(size_t)a * NB + c], A[(size_t)a * NB + c], acc);
            nq[a] = acc;
        }
        for (int b = 0; b < Tr; ++b) {
            float acc = 0.0f;
            for (int c = 0; c < NB; ++c) acc = fmaf(B[(size_t)b * NB + c], B[(size_t)b * NB + c], acc);
            nr[b] = acc;
        }
        float *xx = (float *)malloc((size_t)Mq * sizeof(float));
        float *yy = (float *)malloc((size_t)Mr * sizeof(float));
        for (int i = 0; i < Mq; ++i) xx[i] = tree_sum(nq + (size_t)i * tau, (size_t)tau, m);
        for (int j = 0; j < Mr; ++j) yy[j] = tree_sum(nr + (size_t)j * tau, (size_t)tau, m);
        int have = 0;                      /* Gram rows [0, have) have been computed (row a sits in ring slot a % span) */
        for (int i = 0; i < Mq; ++i) {
            const int last = i * tau + (m - 1) * tau;
            for (; have <= last; ++have) {
                float *g = G + (size_t)(have % span) * Tr;
                const float *xa = A + (size_t)have * NB;
                for (int b = 0; b < Tr; ++b) {
                    float acc = 0.0f;
                    for (int c = 0; c < NB; ++c) acc = fmaf(xa[c], B[(size_t)b * NB + c], acc);
                    g[b] = acc;
                }
            }
            for (int k = 0; k < m; ++k) grow[k] = G + (size_t)((i * tau + k * tau) % span) * Tr;
            for (int j = 0; j < Mr; ++j) {
                for (int k = 0; k < m; ++k) gd[k] = grow[k][(size_t)j * tau + (size_t)k * tau];   /* the diagonal G[i tau + k tau][j tau + k tau] */
                float xy = tree_sum(gd, 1, m);
                float t1 = 2.0f * xy;
                float t2 = xx[i] - t1;
                float t3 = t2 + yy[j];
                if (!(t3 > 0.0f)) t3 = 0.0f;
                d[(size_t)i * Mr + j] = sqrtf(t3);
            }
        }
        free((void *)grow); free(gd);
        free(G); free(nq); free(nr); free(xx); free(yy);
    } else {
        /* sequential 108-dim inner products (mul then add, f32) */
        const int K = m * NB;
        float *X = (float *)malloc((size_t)Mq * K * sizeof(float));
        float *Y = (float *)malloc((size_t)Mr * K * sizeof(float));
        for (int i = 0; i < Mq; ++i)
            for (int k = 0; k < m; ++k)
                memcpy(X + (size_t)i * K + (size_t)k * NB, A + (size_t)(i * tau + k * tau) * NB, NB * sizeof(float));
        for (int j = 0; j < Mr; ++j)
            for (int k = 0; k < m; ++k)
                memcpy(Y + (size_t)j * K + (size_t)k * NB, B + (size_t)(j * tau + k * tau) * NB, NB * sizeof(float));
        float *xx = (float *)malloc((size_t)Mq * sizeof(float));
        float *yy = (float *)malloc((size_t)Mr * sizeof(float));
        for (int i = 0; i < Mq; ++i) {
            float acc = 0.0f;
            for (int k = 0; k < K; ++k) { float pr = X[(size_t)i * K + k] * X[(size_t)i * K + k]; acc = acc + pr; }
            xx[i] = acc;
        }
        for (int j = 0; j < Mr; ++j) {
            float acc = 0.0f;
            for (int k = 0; k < K; ++k) { float pr = Y[(size_t)j * K + k] * Y[(size_t)j * K + k]; acc = acc + pr; }
            yy[j] = acc;
        }
        for (int i = 0; i < Mq; ++i)
            for (int j = 0; j < Mr; ++j) {
                float acc = 0.0f;
                const float *x = X + (size_t)i * K, *y = Y + (size_t)j * K;
                for (int k = 0; k < K; ++k) { float pr = x[k] * y[k]; acc = acc + pr; }
                float t1 = 2.0f * acc;
                float t2 = xx[i] - t1;
                float t3 = t2 + yy[j];
                if (!(t3 > 0.0f)) t3 = 0.0f; /* essentia has no clamp (NaN for tiny negatives); acoss's own get_csm clamps, cross_recurrence.py:47 */
                d[(size_t)i * Mr + j] = sqrtf(t3);
            }
        free(X); free(Y); free(xx); free(yy);
    }
    free(A); free(B);

    /* 4. thresholds: kappa-percentile of every row and every column */
    float *epsq = (float *)malloc((size_t)Mq * sizeof(float));
    float *epsr = (float *)malloc((size_t)Mr * sizeof(float));
    {
        int nmax = Mq > Mr ? Mq : Mr;
        float *work = (float *)malloc((size_t)nmax * sizeof(float));
        enum { CB = 16 };                /* columns gathered per pass over d: one cache line of a row feeds 16 columns */
        float *col = (float *)malloc((size_t)CB * Mq * sizeof(float));
        for (int i = 0; i < Mq; ++i)
            epsq[i] = percentile_f32(d + (size_t)i * Mr, Mr, p->kappa, p->pct_mode, work);
        for (int j0 = 0; j0 < Mr; j0 += CB) {
            const int nc = (Mr - j0 < CB) ? Mr - j0 : CB;
            for (int i = 0; i < Mq; ++i)
                for (int jj = 0; jj < nc; ++jj) col[(size_t)jj * Mq + i] = d[(size_t)i * Mr + j0 + jj];
            for (int jj = 0; jj < nc; ++jj)
                epsr[j0 + jj] = percentile_f32(col + (size_t)jj * Mq, Mq, p->kappa, p->pct_mode, work);
        }
        free(work); free(col);
    }

    /* 5. cross recurrence plot */
    uint8_t *R = tl_r = (uint8_t *)tl_grow(tl_r, &tl_r_cap, (size_t)Mq * Mr);
    for (int i = 0; i < Mq; ++i)
        for (int j = 0; j < Mr; ++j) {
            float v = d[(size_t)i * Mr + j];
            int a = p->inclusive ? (v <= epsq[i]) : (v < epsq[i]);
            int b = p->inclusive ? (v <= epsr[j]) : (v < epsr[j]);
            R[(size_t)i * Mr + j] = (uint8_t)(a && b);
        }

    /* 6. Qmax / Dmax in f32.  Cell (i, j) reads rows i - 1 and i - 2 of the score matrix only: three
     * rolling rows (rows and columns below dp_start stay 0). */
    float best = 0.0f;
    {
        float *rows = (float *)calloc((size_t)3 * Mr, sizeof(float));
        const int st = p->dp_start;
        const int o = (st == 3) ? 1 : 0; /* R index offset */
        const float go = p->gamma_o, ge = p->gamma_e;
#define RR(i, j) R[(size_t)(i) * Mr + (j)]
#define GAM(v) ((v) ? go : ge)
        for (int i = st; i < Mq; ++i) {
            float *cur = rows + (size_t)(i % 3) * Mr;
            const float *p1 = rows + (size_t)((i + 2) % 3) * Mr;     /* row i - 1 */
            const float *p2 = rows + (size_t)((i + 1) % 3) * Mr;     /* row i - 2 */
            for (int j = 0; j < st && j < Mr; ++j) cur[j] = 0.0f;
            for (int j = st; j < Mr; ++j) {
                int ri = i - o, rj = j - o;
                float c2 = p1[j - 1], c3 = p2[j - 1], c4 = p1[j - 2];
                if (p->dmax) {
                    c3 = c3 + (float)RR(ri - 1, rj);
                    c4 = c4 + (float)RR(ri, rj - 1);
                }
                float v;
                if (RR(ri, rj)) {
                    float mx = c2; if (c3 > mx) mx = c3; if (c4 > mx) mx = c4;
                    v = mx + 1.0f;
                } else {
                    float a2 = c2 - GAM(RR(ri - 1, rj - 1));
                    float a3 = c3 - GAM(RR(ri - 2, rj - 1));
                    float a4 = c4 - GAM(RR(ri - 1, rj - 2));
                    float mx = 0.0f; if (a2 > mx) mx = a2; if (a3 > mx) mx = a3; if (a4 > mx) mx = a4;
                    v = mx;
                }
                cur[j] = v;
                if (v > best) best = v;
            }
        }
#undef RR
#undef GAM
        free(rows);
    }

    if (d_out) memcpy(d_out, d, (size_t)Mq * Mr * sizeof(float));
    if (epsq_out) memcpy(epsq_out, epsq, (size_t)Mq * sizeof(float));
    if (epsr_out) memcpy(epsr_out, epsr, (size_t)Mr * sizeof(float));
    if (bin_out) memcpy(bin_out, R, (size_t)Mq * Mr);
    free(epsq); free(epsr);           /* (d and R stay with the thread) */
    return best;
}

/* batch over a packed pool: frames (sum T,12), offsets (n+1), pairs (K,2) */
int acx_o_serra09_pairs(const float *frames, const int64_t *offsets, int32_t n_tracks,
                        const int32_t *pairs, int64_t K,
                        const acx_o_serra09_params *p, float *out)
{
    for (int64_t k = 0; k < K; ++k) {
        int i = pairs[2 * k], j = pairs[2 * k + 1];
        if (i < 0 || j < 0 || i >= n_tracks || j >= n_tracks) return -2;
        out[k] = acx_o_serra09_pair(frames + offsets[i] * NB, (int32_t)(offsets[i + 1] - offsets[i]),
                                    frames + offsets[j] * NB, (int32_t)(offsets[j + 1] - offsets[j]),
                                    p, NULL, NULL, NULL, NULL, NULL);
    }
    return 0;
}

/* ------------------------------------------------------------------ */
/* Qmax/Dmax on a given binary matrix (for DP-only tests)              */
/* ------------------------------------------------------------------ */
float acx_o_qmax_binary(const uint8_t *R, int32_t M, int32_t N, float go, float ge, int32_t dmax)
{
    float best = 0.0f;
    float *C = (float *)calloc((size_t)M * N, sizeof(float));
    for (int i = 2; i < M; ++i)
        for (int j = 2; j < N; ++j) {
            float c2 = C[(size_t)(i - 1) * N + j - 1], c3 = C[(size_t)(i - 2) * N + j - 1], c4 = C[(size_t)(i - 1) * N + j - 2];
            if (dmax) { c3 += (float)R[(size_t)(i - 1) * N + j]; c4 += (float)R[(size_t)i * N + j - 1]; }
            float v;
            if (R[(size_t)i * N + j]) {
                float mx = c2; if (c3 > mx) mx = c3; if (c4 > mx) mx = c4;
                v = mx + 1.0f;
            } else {
                float a2 = c2 - (R[(size_t)(i - 1) * N + j - 1] ? go : ge);
                float a3 = c3 - (R[(size_t)(i - 2) * N + j - 1] ? go : ge);
                float a4 = c4 - (R[(size_t)(i - 1) * N + j - 2] ? go : ge);
                float mx = 0.0f; if (a2 > mx) mx = a2; if (a3 > mx) mx = a3; if (a4 > mx) mx = a4;
                v = mx;
            }
            C[(size_t)i * N + j] = v;
            if (v > best) best = v;
        }
    free(C);
    return best;
}

/* ------------------------------------------------------------------ */
/* constrained Smith-Waterman                                          */
/*   follows acoss/algorithms/utils/alignment_tools.py:7-46            */
/* ------------------------------------------------------------------ */
/* delta_func (alignment_tools.py:8-14): 0 if value_a > 0 else gap_extension
 * (-0.7); the gap_opening branch is unreachable.  match (:17-23): +1 / -1,
 * anything else is an error (returns NaN here; the Python shim raises IOError).
 * f64 as in the reference (np.zeros default dtype). */
double acx_o_sw_constrained(const uint8_t *B, int32_t M, int32_t N)
{
    double best = 0.0;
    if (N < 4 || M < 4) return best;
    for (size_t k = 0; k < (size_t)M * N; ++k)
        if (B[k] > 1) return NAN;
    double *S = (double *)calloc((size_t)M * N, sizeof(double));
#define BB(i, j) B[(size_t)(i) * N + (j)]
#define SS(i, j) S[(size_t)(i) * N + (j)]
#define DEL(v) ((v) > 0 ? 0.0 : -0.7)
    for (int i = 3; i < M; ++i)
        for (int j = 3; j < N; ++j) {
            double mv = BB(i - 1, j - 1) ? 1.0 : -1.0;
            double d1 = SS(i - 1, j - 1) + mv + DEL(BB(i - 2, j - 2));
            double d2 = SS(i - 2, j - 1) + mv + DEL(BB(i - 3, j - 2));
            double d3 = SS(i - 1, j - 2) + mv + DEL(BB(i - 2, j - 3));
            double v = 0.0;
            if (d1 > v) v = d1;
            if (d2 > v) v = d2;
            if (d3 > v) v = d3;
            SS(i, j) = v;
            if (v > best) best = v;
        }
#undef BB
#undef SS
#undef DEL
    free(S);
    return best;
}

/* Exact integer restatement of the same DP in tenths (+10/-10, delta -7):
 * this is the arithmetic the HIP kernel uses; the test-suite checks that it
 * equals the f64 version rounded to 0.1 on every golden. */
int32_t acx_o_sw_constrained_i32(const uint8_t *B, int32_t M, int32_t N)
{
    int32_t best = 0;
    if (N < 4 || M < 4) return best;
    int32_t *S = (int32_t *)calloc((size_t)M * N, sizeof(int32_t));
    for (int i = 3; i < M; ++i)
        for (int j = 3; j < N; ++j) {
            int32_t mv = B[(size_t)(i - 1) * N + j - 1] ? 10 : -10;
            int32_t d1 = S[(size_t)(i - 1) * N + j - 1] + mv + (B[(size_t)(i - 2) * N + j - 2] ? 0 : -7);
            int32_t d2 = S[(size_t)(i - 2) * N + j - 1] + mv + (B[(size_t)(i - 3) * N + j - 2] ? 0 : -7);
            int32_t d3 = S[(size_t)(i - 1) * N + j - 2] + mv + (B[(size_t)(i - 2) * N + j - 3] ? 0 : -7);
            int32_t v = 0;
            if (d1 > v) v = d1;
            if (d2 > v) v = d2;
            if (d3 > v) v = d3;
            S[(size_t)i * N + j] = v;
            if (v > best) best = v;
        }
    free(S);
    return best;
}
