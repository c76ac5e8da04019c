"""Worker of tests/test_gpu_grid.py::test_two_ranks_share_one_gpu (gloo, every rank on GPU 0) and
::test_two_gpus_exchange_over_rccl (ACX_TEST_BACKEND=nccl: rank r on GPU r, the tile exchange over RCCL): launched
under torch.distributed.run; runs the REAL Serra09 / ChenFusion classes over a synthetic cover set and stores
rank 0's matrices."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    workdir, out = sys.argv[1], sys.argv[2]
    os.chdir(workdir)
    import torch.distributed as dist
    from acoss_amd import synth
    from acoss_amd.algorithms.rqa_serra09 import Serra09
    from acoss_amd.algorithms.latefusion_chen import ChenFusion
    world = int(os.environ.get("WORLD_SIZE", "1"))
    backend = os.environ.get("ACX_TEST_BACKEND", "gloo")
    dev = 0
    if world > 1:
        if backend == "nccl":
            import torch
            dev = int(os.environ["LOCAL_RANK"])
            torch.cuda.set_device(dev)
        dist.init_process_group(backend)
    rank = dist.get_rank() if world > 1 else 0
    d = synth.cover_set(clique_sizes=[2] * 12 + [3, 3, 1, 1], seed=77, t_range=(60, 700))
    n = len(d["offsets"]) - 1
    tracks = [d["frames"][d["offsets"][i]:d["offsets"][i + 1]] for i in range(n)]
    res = {}
    for cls, name in ((Serra09, "serra09"), (ChenFusion, "chen")):
        algo = cls("grid.csv", workdir + "/", shortname="w%d" % world, device=dev)
        algo.set_pooled_features(tracks, d["labels"])
        algo.all_pairwise(symmetric=True)
        algo.normalize_by_length()
        for key in list(algo.Ds.keys()):
            st = algo.getEvalStatistics(key, topsidx=[1, 10])
            res["%s_%s_stats" % (name, key)] = np.array(st[:4])
            if rank == 0:
                res["%s_%s" % (name, key)] = np.array(algo.Ds[key])
        algo.cleanup_memmap()
    if world > 1:
        from acoss_amd import dist as adist
        kind = adist.exchange_in_use()          # (every rank: the probe behind it is a collective the first time)
        if rank == 0:
            res["exchange"] = np.array(kind)    # which collective carried the tiles: "gather" unless the probe refused it
    np.savez(os.path.join(out, "world%d_rank%d.npz" % (world, rank)), **res)
    if world > 1:
        dist.barrier(device_ids=[dev]) if backend == "nccl" else dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
