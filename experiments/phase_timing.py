"""Development aid: per-phase shader-clock breakdown of band_kernel from an -DACX_TIMING build.
usage: ACX_LIB=build_ab/timing.so python scripts/phase_timing.py [n_tracks] [T]"""
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
L = _lib.load()
buf = (ctypes.c_ulonglong * 32)()
L.acx_debug_timing.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
L.acx_debug_timing(ctx._h, buf, 1)
ctx.serra09_pairs(pairs)
L.acx_debug_timing(ctx._h, buf, 1)
t = np.array(list(buf), dtype=np.float64)
w = t[31]
names = ["row operands + norms", "sweep", "wait B1", "exchange write + hist clear", "wait B2", "row read",
         "selection", "eps + d2 threshold", "bitmap (role 0 only)"]
tot = t[:9].sum()
print("waves %d, mean cycles per wave %.0f" % (w, tot / w))
for k, nm in enumerate(names):
    print("  %-28s %8.0f  %5.1f %%" % (nm, t[k] / w, 100 * t[k] / tot))
print("  sweep split: dma wait %.0f  gram %.0f  walk %.0f" % (t[16] / w, t[17] / w, t[18] / w))
print("  fast selection split: range %.0f  bin+atomics %.0f  scan+find %.0f  gather %.0f  rank %.0f" % tuple(t[20:25] / w))
print("  fast selection: %d rows decided, %d rows fell back to the generic selection" % (t[26], t[25]))
