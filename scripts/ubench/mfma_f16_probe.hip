// Probe for the f16x2 Gram of the band kernel: (1) do v_mfma_f32_16x16x32_f16 / 16x16x16_f16 keep fp16 SUBNORMAL inputs,
// (2) the k-slot layout (lane l: row l % 16, k-group l / 16), (3) how exact a 2-term fp16 split of chroma-like values is.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/mfma_f16_probe scripts/ubench/mfma_f16_probe.hip && /tmp/mfma_f16_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void probe(const float *x, const float *y, float *out32, float *out16, float *sub)
{
    // x, y: 16 rows x 12 bins each.  lane l: row r = l % 16, class g = l / 16 -> bins g, g + 4, g + 8
    const int l = threadIdx.x, r = l & 15, g = l >> 4;
    _Float16 x1[3], x2[3], y1[3], y2[3];
    for (int j = 0; j < 3; ++j) {
        const float xv = x[r * 12 + g + 4 * j], yv = y[r * 12 + g + 4 * j];
        x1[j] = (_Float16)xv; x2[j] = (_Float16)(xv - (float)x1[j]);
        y1[j] = (_Float16)yv; y2[j] = (_Float16)(yv - (float)y1[j]);
    }
    half8 a0 = {x1[0], x1[1], x1[2], x1[0], x1[1], x1[2], x2[0], x2[1]};
    half8 b0 = {y1[0], y1[1], y1[2], y2[0], y2[1], y2[2], y1[0], y1[1]};
    half4 a1 = {x2[2], 0, 0, 0};
    half4 b1 = {y1[2], 0, 0, 0};
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    // first operand = "column frames" (y rows), second = "row frames" (x rows): acc[i] = C[y row 4 g + i][x row r]
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(b0, a0, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x16f16(b1, a1, acc, 0, 0, 0);
    for (int i = 0; i < 4; ++i) out16[(4 * g + i) * 16 + r] = acc[i];
    if (x[0] == 12345.0f) { sub[2 + l] = (float)a0[0] + (float)b0[0] + (float)a1[0] + (float)b1[0] + (float)a0[7] + (float)b0[7]; }   // keep the operands alive: no dst / src overlap
    // the exact f32 chain for comparison (3 k-steps of 16x16x4)
    f32x4 e = {0.f, 0.f, 0.f, 0.f};
    for (int j = 0; j < 3; ++j) e = __builtin_amdgcn_mfma_f32_16x16x4f32(y[r * 12 + g + 4 * j], x[r * 12 + g + 4 * j], e, 0, 0, 0);
    for (int i = 0; i < 4; ++i) out32[(4 * g + i) * 16 + r] = e[i];
    // subnormal test: a = 2^-20 (fp16 subnormal) x 8 slots, b = 1.0 -> 8 * 2^-20 if kept, 0 if flushed
    if (blockIdx.x == 0) {
        const _Float16 tiny = (_Float16)9.5367431640625e-07f;
        half8 ta = {tiny, tiny, tiny, tiny, tiny, tiny, tiny, tiny}, tb = {1, 1, 1, 1, 1, 1, 1, 1};
        f32x4 z = {0.f, 0.f, 0.f, 0.f};
        z = __builtin_amdgcn_mfma_f32_16x16x32_f16(ta, tb, z, 0, 0, 0);
        half4 ta4 = {tiny, tiny, tiny, tiny}, tb4 = {1, 1, 1, 1};
        f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
        z4 = __builtin_amdgcn_mfma_f32_16x16x16f16(ta4, tb4, z4, 0, 0, 0);
        if (l == 0) { sub[0] = z[0]; sub[1] = z4[0]; }
    }
}

int main(int argc, char **argv)
{
    const int ones = argc > 1;
    std::vector<float> x(16 * 12), y(16 * 12);
    unsigned s = 7u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)(s >> 8) / 16777216.0f; };
    for (int r = 0; r < 16; ++r) {
        float mx = 0, my = 0;
        for (int c = 0; c < 12; ++c) { x[r * 12 + c] = rnd() * (c % 3 ? 1.0f : 0.02f); y[r * 12 + c] = rnd() * (c % 4 ? 1.0f : 0.003f); mx = fmaxf(mx, x[r * 12 + c]); my = fmaxf(my, y[r * 12 + c]); }
        for (int c = 0; c < 12; ++c) { x[r * 12 + c] /= mx; y[r * 12 + c] /= my; if (ones) { x[r * 12 + c] = 1.0f + r; y[r * 12 + c] = 1.0f; } }
    }
    float *dx, *dy, *d32, *d16, *ds;
    hipMalloc(&dx, 768); hipMalloc(&dy, 768); hipMalloc(&d32, 1024); hipMalloc(&d16, 1024); hipMalloc(&ds, 8 + 256);
    hipMemcpy(dx, x.data(), 768, hipMemcpyHostToDevice); hipMemcpy(dy, y.data(), 768, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dx, dy, d32, d16, ds);
    std::vector<float> o32(256), o16(256); float sub[2];
    hipMemcpy(o32.data(), d32, 1024, hipMemcpyDeviceToHost); hipMemcpy(o16.data(), d16, 1024, hipMemcpyDeviceToHost); hipMemcpy(sub, ds, 8, hipMemcpyDeviceToHost);
    double worst_rel = 0, worst_abs = 0, worst32 = 0;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
        double t = 0; for (int c = 0; c < 12; ++c) t += (double)y[i * 12 + c] * (double)x[j * 12 + c];
        worst_abs = fmax(worst_abs, fabs(o16[i * 16 + j] - t)); worst_rel = fmax(worst_rel, fabs(o16[i * 16 + j] - t) / t);
        worst32 = fmax(worst32, fabs(o32[i * 16 + j] - t) / t);
    }
    for (int i = 0; i < 2; ++i) { for (int j = 0; j < 6; ++j) { double t = 0; for (int c = 0; c < 12; ++c) t += (double)y[i * 12 + c] * (double)x[j * 12 + c]; printf("[%d][%d] f16 %.6f f32 %.6f true %.6f | ", i, j, o16[i * 16 + j], o32[i * 16 + j], t); } printf("\n"); }
    printf("f16x2 Gram vs f64: max abs %.3g  max rel %.3g   (f32 MFMA chain: max rel %.3g)\n", worst_abs, worst_rel, worst32);
    printf("subnormal inputs: 16x16x32 -> %.6g (kept: %.6g), 16x16x16 -> %.6g (kept: %.6g)\n", sub[0], 8 * 9.5367431640625e-07, sub[1], 4 * 9.5367431640625e-07);
    return 0;
}
