"""End-to-end run of BASELINE configs[4] at DA-TACOS size on ONE GPU through the host surface, what
acoss.coverid.benchmark does for EarlyFusionTraile (coverid.py:72-80): EarlyFusion.all_pairwise(symmetric=True) +
do_late_fusion() + getEvalStatistics() for all six similarity types, on a synthetic pool of N tracks with 300-500
blocks each (block features i.i.d., generated on the device slice by slice: 56 GB at N = 15 000, which no host array
ever holds) -> gpurun_out/end_to_end_ef.json.

    python scripts/end_to_end_ef.py [n_tracks]

The i.i.d. pool has no cover structure (every 5 consecutive tracks are labelled as one work so that the evaluation does
its full amount of work); what is measured is time per phase."""
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from acoss_amd import _lib  # noqa: E402
from acoss_amd.algorithms.earlyfusion_traile import EarlyFusion  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 15000
out_dir = os.path.join(ROOT, "gpurun_out")
os.makedirs(out_dir, exist_ok=True)
tmp = tempfile.mkdtemp()
os.chdir(tmp)
labels = ["w%d" % (i // 5) for i in range(N)]
with open("ds.csv", "w") as f:
    f.write("work_id,track_id\n")
    for i, l in enumerate(labels):
        f.write("%s,t%d\n" % (l, i))
ph = {}
t0 = time.time()
ef = EarlyFusion("ds.csv", "feat/", shortname="e2e")
for i, l in enumerate(labels):
    ef._register_label(i, l)
import torch  # noqa: E402
ctx = _lib.Context(0)
rng = np.random.default_rng(15)
nb = rng.integers(300, 501, N).astype(np.int64)
off = np.concatenate([[0], np.cumsum(nb)])
ctx.ef_pool_begin(nb, (650, 1225, 480))
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev)
SL = 250
for a in range(0, N, SL):
    b = min(N, a + SL)
    rows = int(off[b] - off[a])
    gen.manual_seed(1000 + a)
    mf = torch.randn((rows, 650), generator=gen, device=dev, dtype=torch.float32)
    mf /= torch.linalg.vector_norm(mf, dim=1, keepdim=True)
    ss = 2 * torch.rand((rows, 1225), generator=gen, device=dev, dtype=torch.float32)
    ch = torch.rand((rows, 480), generator=gen, device=dev, dtype=torch.float32)
    med = torch.rand((b - a, 12), generator=gen, device=dev, dtype=torch.float64)
    torch.cuda.synchronize()
    ctx.ef_pool_tracks(a, b - a, mf, ss, ch, med)
    del mf, ss, ch, med
torch.cuda.empty_cache()
ctx.ef_pool_end()
ef._ctx, ef._pool_ready = ctx, True                     # the class finds its pool on the device
ph["setup_and_pool_on_device_s"] = time.time() - t0
t0 = time.time()
ef.all_pairwise(symmetric=True)
ph["all_pairwise_s"] = time.time() - t0
t0 = time.time()
ef.do_late_fusion()
ph["do_late_fusion_s"] = time.time() - t0
t0 = time.time()
stats = {k: ef.getEvalStatistics(k)[:4] for k in list(ef.Ds.keys())}
ph["getEvalStatistics_x6_s"] = time.time() - t0
pairs = N * (N - 1) // 2
total = sum(ph.values())
rec = {"workload": "configs[4]: %d tracks of 300-500 blocks (%d blocks, %.1f GB of block features on the device), EarlyFusionTraile, all %d "
                   "unordered pairs, one MI355X" % (N, int(off[-1]), off[-1] * 2355 * 4 / 1e9, pairs),
       "phases_s": {k: round(v, 2) for k, v in ph.items()}, "total_s": round(total, 2),
       "pairs_per_s_all_pairwise": round(pairs / ph["all_pairwise_s"], 1), "pairs_per_s_end_to_end": round(pairs / total, 1),
       "similarity_types": list(ef.Ds.keys()), "MAP": {k: v[3] for k, v in stats.items()}}
print(json.dumps(rec))
with open(os.path.join(out_dir, "end_to_end_ef.json"), "w") as f:
    json.dump(rec, f, indent=1)
ef.cleanup_memmap()
