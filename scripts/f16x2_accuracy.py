"""Serra09 Gram, exact f32 chain vs the opt-in f16x2 arithmetic: both d^2 matrices of one 300-frame i.i.d. pair against
the f64-evaluated truth (max / rms error), as cited in profiles/r04_f16x2.md.  (python scripts/f16x2_accuracy.py, on a GPU)"""
import sys
import numpy as np
sys.path.insert(0, ".")
from acoss_amd import _lib
rng = np.random.default_rng(1)
T = 300
q = rng.random((T, 12), dtype=np.float32); q /= q.max(axis=1, keepdims=True)
r = rng.random((T, 12), dtype=np.float32); r /= r.max(axis=1, keepdims=True)
ctx = _lib.Context(0)
ctx.upload_pool(np.concatenate([q, r]), np.array([0, T, 2 * T]))
e = ctx.serra09_debug_pair(0, 1, _lib.serra09_params(oti=False))["d2"].astype(np.float64)
f = ctx.serra09_debug_pair(0, 1, _lib.serra09_params(oti=False, arith="f16x2"))["d2"].astype(np.float64)
M = T - 9
X = np.concatenate([q[k:k + M] for k in range(9)], 1).astype(np.float64)
Y = np.concatenate([r[k:k + M] for k in range(9)], 1).astype(np.float64)
t = (X ** 2).sum(1)[:, None] + (Y ** 2).sum(1)[None, :] - 2 * X @ Y.T
print("abs error vs f64: exact max %.3g rms %.3g | f16x2 max %.3g rms %.3g | (xx + yy ~ %.1f, d2 ~ %.1f)" % (
    np.abs(e - t).max(), np.sqrt(np.mean((e - t) ** 2)), np.abs(f - t).max(), np.sqrt(np.mean((f - t) ** 2)), ((X ** 2).sum(1).mean() + (Y ** 2).sum(1).mean()), t.mean()))
# what a 2-term fp16 representation of the inputs alone costs
def split(x):
    h1 = x.astype(np.float16); h2 = (x - h1.astype(np.float32)).astype(np.float16)
    return h1.astype(np.float64) + h2.astype(np.float64)
Xs = np.concatenate([split(q)[k:k + M] for k in range(9)], 1); Ys = np.concatenate([split(r)[k:k + M] for k in range(9)], 1)
ts = (X ** 2).sum(1)[:, None] + (Y ** 2).sum(1)[None, :] - 2 * Xs @ Ys.T
print("representation error alone (exact arithmetic on h1 + h2): max %.3g rms %.3g" % (np.abs(ts - t).max(), np.sqrt(np.mean((ts - t) ** 2))))
h1 = q.astype(np.float16); h2 = (q - h1.astype(np.float32)).astype(np.float16)
print("max |x - (h1 + h2)| = %.3g, max |h2| = %.3g, min nonzero |h2| = %.3g" % (np.abs(q.astype(np.float64) - h1.astype(np.float64) - h2.astype(np.float64)).max(), np.abs(h2.astype(np.float64)).max(), np.abs(h2[h2 != 0].astype(np.float64)).min()))
