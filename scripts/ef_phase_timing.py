"""Where a tile of ef_gemm_rect_bf16x3_kernel spends its time (development build: scripts/ab_build_acx.sh timing
-DACX_EF_TIMING, run with ACX_LIB=build_ab/libacx_timing.so): s_memtime at the start of a wave, before and behind the k loop
and behind the last store's acknowledgement, summed over the waves that hold a pair."""
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
    w = max(1, v[15])
    print(json.dumps({"waves_with_a_pair": w, "cycles_per_wave": {"start_up": round(v[0] / w), "k_loop": round(v[1] / w),
                                                                  "epilogue_to_last_store_issued": round(v[2] / w),
                                                                  "last_store_issued_to_acknowledged": round(v[3] / w)},
                      "start_up_parts": {"groups_and_pairs": round(v[5] / w), "staging_rows": round(v[6] / w),
                                         "first_chunk_arrives": round(v[7] / w), "lds_store_second_loads_barrier": round(v[8] / w)}}))
    ctx.close()


if __name__ == "__main__":
    main()
