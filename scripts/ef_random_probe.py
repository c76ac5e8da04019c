"""EarlyFusion on an ARBITRARY pair list (acx_earlyfusion_pairs: sparse rectangles) next to the dense grid, 1500 tracks
of 300-500 blocks generated on the device: pairs/s including the host side (python scripts/ef_random_probe.py [pairs])."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")


def main():
    import torch
    from acoss_amd import _lib
    npairs = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
    rng = np.random.default_rng(3)
    nb = rng.integers(300, 501, N).astype(np.int64)
    ctx = _lib.Context(0)
    ctx.ef_pool_begin(nb, (650, 1225, 480))
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev)
    off = np.concatenate([[0], np.cumsum(nb)])
    for a in range(0, N, 250):
        b = min(N, a + 250)
        rows = int(off[b] - off[a])
        gen.manual_seed(1000 + a)
        mf = torch.randn((rows, 650), generator=gen, device=dev)
        ss = 2 * torch.rand((rows, 1225), generator=gen, device=dev)
        ch = torch.rand((rows, 480), generator=gen, device=dev)
        med = torch.rand((b - a, 12), generator=gen, device=dev, dtype=torch.float64)
        ctx.ef_pool_tracks(a, b - a, mf, ss, ch, med)
    ctx.ef_pool_end()
    pairs = rng.integers(0, N, (npairs, 2)).astype(np.int32)
    pairs = pairs[pairs[:, 0] != pairs[:, 1]]
    ctx.earlyfusion_pairs(pairs[:2000])
    for mode in ("bf16x3", "bf16x3_pairwise"):
      ctx.set_ef_gemm(mode)
      for rep in range(2):
        ctx.profile_enable(True)
        ctx.profile_reset()
        t0 = time.time()
        out = ctx.earlyfusion_pairs(pairs)
        dt = time.time() - t0
        prof = ctx.profile()
        print("%s random list of %d tracks: %d pairs %.2f s = %.0f pairs/s; kernels ms: %s; checksum %.1f" % (
            mode, N, len(pairs), dt, len(pairs) / dt, {k: round(v["ms"], 1) for k, v in prof.items() if v["launches"]}, float(out.sum())))
    ctx.close()


if __name__ == "__main__":
    main()
