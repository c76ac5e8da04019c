"""Development aid (GPU box): race screen of the EarlyFusion GEMMs that stage their operands by LDS-DMA (ef_gemm_dma_kernels.hpp) -- an
early read of a DMA'd buffer passes whenever the data happens to land first, so one clean run proves little.  Several pools (dense
grid tiles, sparse pair lists, tracks of 1-40 blocks: rims and idle waves everywhere), each run REPS times in the default build
(persistent + DMA) and compared bit for bit with ONE run of the register-staging one-tile kernel in a child process
(ACX_EF_PERSIST=0 ACX_EF_DMA=0).   python scripts/ef_dma_soak.py [reps]"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def cases():
    from acoss_amd import synth
    rng = np.random.default_rng(11)
    out = []
    tracks = synth.earlyfusion_set(96, seed=21, nb_range=(300, 500))
    iu, ju = np.triu_indices(96, 1)
    out.append(("grid tile of 96 tracks", tracks, np.stack([iu, ju], 1).astype(np.int32)))
    tracks = synth.earlyfusion_set(300, seed=22, nb_range=(40, 700))
    out.append(("3000 random pairs of 300 tracks", tracks, rng.integers(0, 300, (3000, 2)).astype(np.int32)))
    tracks = synth.earlyfusion_set(120, seed=23, nb_range=(1, 40))
    iu, ju = np.triu_indices(120, 1)
    out.append(("tracks of 1-40 blocks", tracks, np.stack([iu, ju], 1).astype(np.int32)))
    return out


def run(reps):
    from acoss_amd import _lib
    ctx = _lib.Context(0)
    res = []
    for name, tracks, pairs in cases():
        ctx.ef_upload_pool(tracks)
        first = ctx.earlyfusion_pairs(pairs)
        same = 1
        for _ in range(reps - 1):
            same += int(np.array_equal(ctx.earlyfusion_pairs(pairs), first))
        res.append((name, first, same))
    ctx.close()
    return res


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        r = run(1)
        np.savez(sys.argv[2], *[x[1] for x in r])
        sys.exit(0)
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    tmp = "/tmp/ef_soak_ref.npz"
    env = dict(os.environ, ACX_EF_PERSIST="0", ACX_EF_DMA="0")
    subprocess.check_call([sys.executable, os.path.abspath(__file__), "--child", tmp], env=env)
    ref = np.load(tmp)
    bad = 0
    for k, (name, first, same) in enumerate(run(reps)):
        ok_ref = np.array_equal(first, ref["arr_%d" % k])
        print("%-36s %6d pairs: %d / %d runs identical, equal to the register-staging kernel: %s" % (name, len(first), same, reps, ok_ref))
        bad += (same != reps) + (not ok_ref)
    sys.exit(1 if bad else 0)
