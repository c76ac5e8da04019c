"""Phase clocks of ef_gemm_rect_bf16x3_kernel (development build: scripts/ab_build_acx.sh timing -DACX_EF_TIMING, run with
ACX_LIB=build_ab/libacx_timing.so).  Per wave and k chunk: cycles from stamp to stamp (s_memtime), summed by lane 0."""
import ctypes
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 48
    import torch
    from acoss_amd import _lib, synth
    lib = ctypes.CDLL(_lib.LIB_PATH)
    lib.acx_ef_clk.argtypes = [ctypes.c_void_p, ctypes.c_int]
    ctx = _lib.Context(0)
    tracks = synth.earlyfusion_set(n, seed=1, nb_range=(300, 500))
    ctx.ef_upload_pool(tracks)
    nb = np.array([t["mfccs"].shape[0] for t in tracks])
    plan = _lib.grid_plan(nb, _lib.ALGO_EARLYFUSION, True, world=1, tile=128, want_tiles=True)
    buf = torch.zeros(int(plan["floats_per_rank"][0]), dtype=torch.float32, device="cuda:0")
    ep = _lib.EfParams(0.1, 10)
    ctx.grid_run(plan["spec"], ep, 0, buf.data_ptr())
    assert lib.acx_ef_clk(None, 1) == 0
    ctx.grid_run(plan["spec"], ep, 0, buf.data_ptr())
    out = (ctypes.c_ulonglong * 16)()
    assert lib.acx_ef_clk(out, 0) == 0
    v = [int(x) for x in out]
    steps = max(1, v[15])
    names = ["wait+lds_store", "global_load_issue", "lds_read+mfma_issue", "barrier", "epilogue"]
    print(json.dumps({"wave_k_steps": steps, "cycles_per_wave_k_step": {nm: round(v[i] / steps, 1) for i, nm in enumerate(names)},
                      "sum": round(sum(v[:5]) / steps, 1)}))
    ctx.close()


if __name__ == "__main__":
    main()
