"""Development aid: end-to-end timing of the host surface (all_pairwise + normalisation + evaluation) on a
synthetic cover set.   usage: scale_host.py [n_works] [versions] [t_lo] [t_hi]"""
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, ".")
from acoss_amd import synth  # noqa: E402
from acoss_amd.algorithms.rqa_serra09 import Serra09  # noqa: E402

n_works = int(sys.argv[1]) if len(sys.argv) > 1 else 500
versions = int(sys.argv[2]) if len(sys.argv) > 2 else 4
t_range = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (300, 600)
d = synth.cover_set(n_works=n_works, versions=versions, seed=7, t_range=t_range)
n = len(d["offsets"]) - 1
tmp = tempfile.mkdtemp()
os.chdir(tmp)
with open("ds.csv", "w") as f:
    f.write("work_id,track_id\n")
    for i, l in enumerate(d["labels"]):
        f.write("%s,t%d\n" % (l, i))
t0 = time.time()
alg = Serra09("ds.csv", "feat/", shortname="scale")
alg.set_pooled_features([d["frames"][d["offsets"][i]:d["offsets"][i + 1]] for i in range(n)], d["labels"])
t1 = time.time()
alg.all_pairwise(symmetric=True)
t2 = time.time()
alg.normalize_by_length()
t3 = time.time()
res = alg.getEvalStatistics("main")
t4 = time.time()
pairs = n * (n - 1) // 2
print("N=%d tracks, %d pairs: setup %.2f s, all_pairwise %.2f s (%.0f pairs/s incl. host), normalise %.2f s, evaluation %.2f s" % (
    n, pairs, t1 - t0, t2 - t1, pairs / (t2 - t1), t3 - t2, t4 - t3))
print("MR %.3f MRR %.4f MDR %.2f MAP %.4f" % tuple(res[:4]))
