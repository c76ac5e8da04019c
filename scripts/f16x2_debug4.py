import sys
import numpy as np
sys.path.insert(0, ".")
from acoss_amd import _lib
T = 520
t0 = 200
q = np.zeros((T, 12), np.float32); q[:, 0] = 1.0
r = np.zeros((T, 12), np.float32); r[:, 1] = 1.0; r[t0, 0] = 0.5
ctx = _lib.Context(0)
ctx.upload_pool(np.concatenate([q, r]), np.array([0, T, 2 * T]))
e = ctx.serra09_debug_pair(0, 1, _lib.serra09_params(oti=False))["d2"]
f = ctx.serra09_debug_pair(0, 1, _lib.serra09_params(oti=False, arith="f16x2"))["d2"]
print("median exact %.4f f16 %.4f" % (np.median(e), np.median(f)))
for row in (0, 9):
    dev = np.nonzero(np.abs(f[row] - np.median(f)) > 1e-3)[0]
    print("row", row, "f16 deviating cols:", dev.tolist())
    print("   f16 values:", np.round(f[row][dev], 3).tolist())
    print("   exact there:", np.round(e[row][dev], 3).tolist())
ctx.close()
