// libacx: launcher of the Serra09 band kernel (serra09_kernels.hpp, K1').  A translation unit of its
// own: the Makefile compiles it with -mllvm -amdgpu-sched-strategy=max-ilp, which gives this kernel
// +1 % and would cost simple_kernel 26 % and ef_rowstat_kernel 20 %.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "serra09_kernels.hpp"
#include "serra09_band2_kernels.hpp"

namespace acx {

namespace {

template <int M>
bool launch_band_m(const BandLaunch &L, const PairDesc *dpd, int B, int maxRows, int maxCols, int role, int write_d2, int want_eps, int arith)
{
    const dim3 grid((maxRows + BAND - 1) / BAND, B, 1);
    const int ndata = (maxCols + BAND - 1 + 63) / 64;      // tiles per band that hold matrix cells
#define ACX_BAND_K(V4_, R_, W_, A_) hipLaunchKernelGGL((band_kernel<M, V4_, R_, W_, A_>), grid, dim3(BAND_THREADS), 0, L.stream, \
                                                  L.frot, L.toff, L.normtab, L.noff, dpd, L.scratch, L.thr, L.bits, L.kappa, \
                                                  L.pct_mode, L.inclusive, L.oti_target, want_eps)
    // (the variant that also writes D2 exists for the row pass only: the debug entry point)
#define ACX_BAND(V4_, A_) do { if (role) ACX_BAND_K(V4_, 1, false, A_); else if (write_d2) ACX_BAND_K(V4_, 0, true, A_); else ACX_BAND_K(V4_, 0, false, A_); } while (0)
    if (arith == 0) {
        // rows of <= 505 cells, m <= 9: the two-rows-per-wave kernel (serra09_band2_kernels.hpp); ACX_BAND2=0 keeps band_kernel<M, 2>
        if constexpr (M <= 9) {
            static const bool two_rows = [] { const char *e = getenv("ACX_BAND2"); return !(e && e[0] == '0'); }();
            // (a second class of it, 24 positions per lane: rows of <= 761 cells -- ACX_BAND2=1 keeps that one on band_kernel<M, 4>)
            static const bool mid_two_rows = [] { const char *e = getenv("ACX_BAND2"); return !(e && (e[0] == '0' || e[0] == '1')); }();
#define ACX_BAND2_K(R_, W_, NV_, GL_) hipLaunchKernelGGL((band2_kernel<M, R_, W_, NV_, GL_>), grid, dim3(64 * B2Geom<NV_, GL_>::WAVES), 0, L.stream, L.frot, L.normtab, \
                                                         dpd, L.scratch, L.thr, L.bits, L.pct_mode, L.inclusive, L.oti_target, want_eps)
#define ACX_BAND2(NV_, GL_) do { if (role) ACX_BAND2_K(1, false, NV_, GL_); else if (write_d2) ACX_BAND2_K(0, true, NV_, GL_); else ACX_BAND2_K(0, false, NV_, GL_); } while (0)
            // (rows of <= 249 cells: FOUR rows per wave -- ACX_BAND2=2 keeps them on the two-row kernel)
            static const bool four_rows = [] { const char *e = getenv("ACX_BAND2"); return !(e && (e[0] == '0' || e[0] == '1' || e[0] == '2')); }();
            if (ndata <= 4 && two_rows && four_rows) { ACX_BAND2(B2_NV, 16); return true; }
            if (ndata <= 8 && two_rows) { ACX_BAND2(B2_NV, 32); return true; }
            if (ndata > 8 && ndata <= 12 && two_rows && mid_two_rows) { ACX_BAND2(B2_NV_MID, 32); return true; }
#undef ACX_BAND2
#undef ACX_BAND2_K
        }
        if (ndata <= 8) ACX_BAND(2, 0);
        else if (ndata <= 16) ACX_BAND(4, 0);
        else ACX_BAND(8, 0);
        return true;
    }
    // the opt-in f16x2 Gram: the default stack size only
    if constexpr (M == 9) {
        if (arith != 1) return false;
        if (ndata <= 8) ACX_BAND(2, 1);
        else if (ndata <= 16) ACX_BAND(4, 1);
        else ACX_BAND(8, 1);
        return true;
    }
    return false;
#undef ACX_BAND_K
#undef ACX_BAND
}

}  // namespace

bool launch_band_kernel(const BandLaunch &L, int m, const PairDesc *dpd, int B, int maxRows, int maxCols, int role, int write_d2, int want_eps, int arith)
{
    switch (m) {
#define ACX_CASE(M_) case M_: return launch_band_m<M_>(L, dpd, B, maxRows, maxCols, role, write_d2, want_eps, arith);
#ifdef ACX_FAST_BUILD   /* development builds: only the default stack size */
#ifndef ACX_FAST_BUILD_M
#define ACX_FAST_BUILD_M 9
#endif
        ACX_CASE(ACX_FAST_BUILD_M)
#else
        ACX_CASE(1) ACX_CASE(2) ACX_CASE(3) ACX_CASE(4) ACX_CASE(5) ACX_CASE(6) ACX_CASE(7) ACX_CASE(8)
        ACX_CASE(9) ACX_CASE(10) ACX_CASE(11) ACX_CASE(12) ACX_CASE(13) ACX_CASE(14) ACX_CASE(15) ACX_CASE(16)
#endif
#undef ACX_CASE
    }
    return false;
}


}  // namespace acx

#ifdef ACX_TIMING
// development builds: per-phase clock totals of band_kernel (slots 0-7; slot 15 = waves); reset != 0 clears them
extern "C" int acx_dev_band_timing(unsigned long long *out, int reset)
{
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(acx::g_band_clk), sizeof(unsigned long long) * 32) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[32] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(acx::g_band_clk), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif
