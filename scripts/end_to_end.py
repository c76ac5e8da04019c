"""End-to-end run of BASELINE configs[2] on ONE GPU through the host surface, what acoss.coverid.benchmark does
for Serra09 (coverid.py:57-70): Serra09.all_pairwise(symmetric=True) + normalize_by_length() +
getEvalStatistics() on the synthetic 5 000-track x 2000-frame pool of bench.py (12 497 500 pairs), wall time
per phase -> profiles/r03_end_to_end.json (via gpurun_out/).

    python scripts/end_to_end.py [n_tracks] [frames]

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
T = int(sys.argv[2]) if len(sys.argv) > 2 else bench.T_FRAMES
frames, offsets = bench.make_pool(N, T)
labels = ["w%d" % (i // 5) for i in range(N)]
out_dir = os.path.join(ROOT, "gpurun_out")
os.makedirs(out_dir, exist_ok=True)
tmp = tempfile.mkdtemp()
os.chdir(tmp)
with open("ds.csv", "w") as f:
    f.write("work_id,track_id\n")
    for i, l in enumerate(labels):
        f.write("%s,t%d\n" % (l, i))
ph = {}
t0 = time.time()
alg = Serra09("ds.csv", "feat/", shortname="e2e")
alg.set_pooled_features([frames[offsets[i]:offsets[i + 1]] for i in range(N)], labels)
ctx = alg._context()                       # pool upload (H2D 1 x, rotated copy, norm table on first use)
ph["setup_and_pool_upload_s"] = time.time() - t0
t0 = time.time()
alg.all_pairwise(symmetric=True)           # acx_pair_grid: plan + kernels + D2H slices + scatter/mirror into the memmap + cache files
ph["all_pairwise_s"] = time.time() - t0
prof_before = None
t0 = time.time()
alg.normalize_by_length()
ph["normalize_by_length_s"] = time.time() - t0
t0 = time.time()
res = alg.getEvalStatistics("main")
ph["getEvalStatistics_s"] = time.time() - t0
pairs = N * (N - 1) // 2
total = sum(ph.values())
host_phases = ph["normalize_by_length_s"] + ph["getEvalStatistics_s"]
rec = {"workload": "configs[2]: %d tracks x %d frames, Serra09 Qmax, all %d unordered pairs, one MI355X" % (N, T, pairs),
       "phases_s": {k: round(v, 2) for k, v in ph.items()}, "total_s": round(total, 2),
       "pairs_per_s_all_pairwise": round(pairs / ph["all_pairwise_s"], 1),
       "pairs_per_s_end_to_end": round(pairs / total, 1),
       "host_post_phases_fraction": round(host_phases / total, 4),
       "note": "all_pairwise includes the tile plan, every kernel, the device-to-host copy of the tile scores in 256 MB slices, "
               "the scatter + mirror into the N x N float32 memmap and writing the <prefix>_Ds.npz cache",
       "stats": {"MR": res[0], "MRR": res[1], "MDR": res[2], "MAP": res[3]}}
print(json.dumps(rec))
with open(os.path.join(out_dir, "end_to_end.json"), "w") as f:
    json.dump(rec, f, indent=1)
alg.cleanup_memmap()
