"""Development aid: per-phase shader-clock breakdown of band_kernel from an -DACX_TIMING build.
   scripts/ab_build.sh timing -DACX_TIMING
   ACX_LIB=build_ab/libacx_timing.so python experiments/phase_timing.py [n_tracks] [T] [covers]"""
import ctypes
import sys

import numpy as np

sys.path.insert(0, ".")
from acoss_amd import _lib, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 48
T = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
if len(sys.argv) > 3 and sys.argv[3] == "covers":      # structured (cover-song-like) tracks of about T frames
    d = synth.cover_set(n_works=max(1, n // 2), versions=2, seed=4321, t_range=(int(0.7 * T), int(1.3 * T)))
    n = len(d["offsets"]) - 1
else:
    d = synth.rand_set(n, T=T, seed=1234)
ctx = _lib.Context(0)
ctx.upload_pool(d["frames"], d["offsets"])
i, j = np.triu_indices(n, 1)
pairs = np.stack([i, j], 1).astype(np.int32)
ctx.serra09_pairs(pairs[:64])
L = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_ulonglong * 32)()
L.acx_dev_band_timing(buf, 1)
ctx.serra09_pairs(pairs)
L.acx_dev_band_timing(buf, 1)
t = np.array(list(buf), dtype=np.float64)
names = ["operands + sweep / gather", "wait B1", "exchange writes", "wait B2", "row read (+ Z store)", "selection", "eps + d2 threshold",
         "threshold store + bitmap (row pass)"]
for title, o in (("band_kernel", 0), ("band_read_kernel", 16)):
    w = t[o + 15]
    if w == 0:
        continue
    tot = t[o:o + 8].sum()
    print("%s: waves %d, mean clock ticks per wave %.0f" % (title, w, tot / w))
    for k, nm in enumerate(names):
        print("  %-36s %8.1f  %5.1f %%" % (nm, t[o + k] / w, 100 * t[o + k] / tot))
