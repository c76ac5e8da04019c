"""End-to-end run of BASELINE configs[2] on ONE GPU through the host surface, what acoss.coverid.benchmark does
for Serra09 (coverid.py:57-70): Serra09.all_pairwise(symmetric=True) + normalize_by_length() +
getEvalStatistics() on the synthetic 5 000-track x 2000-frame pool of bench.py (12 497 500 pairs), wall time
per phase -> profiles/r03_end_to_end.json (via gpurun_out/).

    python scripts/end_to_end.py [n_tracks] [frames | covers]        (covers: lengths uniform in 150 .. 650 pooled frames, the
                                                                       shape of covers80 / DA-TACOS tracks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 scripts/end_to_end.py [n_tracks] [frames]

The i.i.d. pool has no cover structure; every 5 consecutive tracks are labelled as one work so that the
evaluation does its full amount of work (its MAP is that of random scores)."""
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from acoss_amd.algorithms.rqa_serra09 import Serra09  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else bench.N_TRACKS
COVERS = len(sys.argv) > 2 and sys.argv[2] == "covers"
T = 0 if COVERS else (int(sys.argv[2]) if len(sys.argv) > 2 else bench.T_FRAMES)
# one process per GPU under torch.distributed.run (RANK / WORLD_SIZE / LOCAL_RANK from the launcher): the classes find the
# process group on their own (acoss_amd/dist.py); every rank keeps its own phase clock, rank 0 reports all of them
rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
dist = None
if world > 1 or os.environ.get("ACX_GRID_VIA_COLLECTIVE", "") not in ("", "0"):
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29544")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    backend = os.environ.get("ACX_BENCH_BACKEND", "nccl")
    if backend == "nccl":
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    dist.init_process_group(backend)
if COVERS:
    _rng = np.random.default_rng(4321)
    _len = _rng.integers(150, 651, N)
    offsets = np.concatenate([[0], np.cumsum(_len)]).astype(np.int64)
    frames = _rng.random((int(offsets[-1]), 12), dtype=np.float32)
    frames /= frames.max(axis=1, keepdims=True)
else:
    frames, offsets = bench.make_pool(N, T)
labels = ["w%d" % (i // 5) for i in range(N)]
out_dir = os.path.join(ROOT, "gpurun_out")
os.makedirs(out_dir, exist_ok=True)
tmp = tempfile.mkdtemp()                   # per rank: every rank has its own memmaps, rank 0's hold the result
os.chdir(tmp)
with open("ds.csv", "w") as f:
    f.write("work_id,track_id\n")
    for i, l in enumerate(labels):
        f.write("%s,t%d\n" % (l, i))
ph = {}
t0 = time.time()
dev = int(os.environ.get("LOCAL_RANK", "0"))
if dist is not None and dist.get_backend() != "nccl":
    import torch
    dev = dev % max(1, torch.cuda.device_count())          # development: gloo ranks sharing the GPUs of the box
alg = Serra09("ds.csv", "feat/", shortname="e2e", device=dev)
alg.set_pooled_features([frames[offsets[i]:offsets[i + 1]] for i in range(N)], labels)
ctx = alg._context()                       # pool upload (H2D 1 x, rotated copy, norm table on first use)
ph["setup_and_pool_upload_s"] = time.time() - t0
t0 = time.time()
alg.all_pairwise(symmetric=True)           # one GPU: acx_pair_grid straight into the memmap; N GPUs: tiles, ONE all-gather, rank-0 scatter
ph["all_pairwise_s"] = time.time() - t0
t0 = time.time()
alg.normalize_by_length()
ph["normalize_by_length_s"] = time.time() - t0
t0 = time.time()
res = alg.getEvalStatistics("main")
ph["getEvalStatistics_s"] = time.time() - t0
sys.stderr.write("[end_to_end rank %d/%d] %s\n" % (rank, world, json.dumps({k: round(v, 2) for k, v in ph.items()})))
per_rank = [ph]
if dist is not None:
    per_rank = [None] * world
    dist.all_gather_object(per_rank, ph)
if rank == 0:
    pairs = N * (N - 1) // 2
    total = sum(ph.values())
    host_phases = ph["normalize_by_length_s"] + ph["getEvalStatistics_s"]
    ap = [p["all_pairwise_s"] for p in per_rank]
    shape = "150-650 frames each (uniform: the covers80 / DA-TACOS shape; %d frames in all)" % int(offsets[-1]) if COVERS else "%d frames" % T
    cells = None
    if COVERS:
        M = (np.diff(offsets) - 8).astype(np.float64)           # embedded frames (m = 9, tau = 1)
        cells = float((M.sum() ** 2 - (M ** 2).sum()) / 2)
    rec = {"workload": "configs[2]: %d tracks x %s, Serra09 Qmax, all %d unordered pairs, %d rank(s), one MI355X each" % (N, shape, pairs, world),
           "gcells_per_s_all_pairwise": (round(cells / ph["all_pairwise_s"] / 1e9, 1) if cells else None),
           "n_gpus": world, "collectives": (dist.get_backend() if dist is not None else None),
           "phases_s": {k: round(v, 2) for k, v in ph.items()}, "total_s": round(total, 2),
           "phases_s_per_rank": [{k: round(v, 2) for k, v in p.items()} for p in per_rank],
           "all_pairwise_imbalance_max_over_mean": round(max(ap) / (sum(ap) / len(ap)), 4),
           "pairs_per_s_all_pairwise": round(pairs / ph["all_pairwise_s"], 1),
           "pairs_per_s_end_to_end": round(pairs / total, 1),
           "host_post_phases_fraction": round(host_phases / total, 4),
           "note": "all_pairwise includes the tile plan, every kernel, the device-to-host copy of the tile scores (one GPU: 256 MB "
                   "slices; N GPUs: after the one all-gather), the scatter + mirror into the N x N float32 memmap and writing the "
                   "<prefix>_Ds.npz cache; rank 0's clock (it owns the result), every rank's clock in phases_s_per_rank",
           "stats": {"MR": res[0], "MRR": res[1], "MDR": res[2], "MAP": res[3]}}
    print(json.dumps(rec))
    with open(os.path.join(out_dir, "end_to_end.json"), "w") as f:
        json.dump(rec, f, indent=1)
alg.cleanup_memmap()
if dist is not None:
    dist.barrier(device_ids=[dev]) if dist.get_backend() == "nccl" else dist.barrier()
    dist.destroy_process_group()
