"""
CPU ORACLE -- test infrastructure, NOT product code.

A restatement of the arithmetic on acoss's all-pairwise hot path, used only as
the checker by tests/, __graft_entry__.smoke() and the cpu_baseline legs of
bench.py / bench_other.py.  Nothing under acoss_amd/ imports this package.

Pinning (see DESIGN.md "Oracle"):
  * SiMPle, EarlyFusion per-pair kernels, the harness (pair grid, symmetrise,
    evaluation statistics) and SNF are pinned by golden vectors generated in
    the authoring container by importing the reference itself
    (tests/golden/make_goldens.py -> tests/golden/*.npz).
  * Serra09: PARITY UNPINNED.  Its arithmetic lives in essentia (un-pinned,
    absent: reference setup.py:53, rqa_serra09.py:9,60-67) and the reference
    holds no vector for it.  oracle/acx_oracle.c restates the published
    algorithm; recalled essentia details are switchable parameters.

Each function cites the reference file:line it follows (paths relative to
/root/reference).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libacx_oracle.so")
_lib = None


def build(force=False):
    """Compile oracle/acx_oracle.c with gcc (recipe: oracle/Makefile)."""
    src = os.path.join(_HERE, "acx_oracle.c")
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= os.path.getmtime(src)):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


class Serra09Params(ctypes.Structure):
    """Mirror of acx_o_serra09_params (oracle/acx_oracle.c)."""
    _fields_ = [
        ("m", ctypes.c_int32), ("tau", ctypes.c_int32), ("kappa", ctypes.c_float),
        ("oti", ctypes.c_int32), ("gamma_o", ctypes.c_float), ("gamma_e", ctypes.c_float),
        ("embed_full", ctypes.c_int32), ("pct_mode", ctypes.c_int32),
        ("oti_target", ctypes.c_int32), ("dp_start", ctypes.c_int32),
        ("inclusive", ctypes.c_int32), ("arith", ctypes.c_int32), ("dmax", ctypes.c_int32),
    ]


def serra09_params(m=9, tau=1, kappa=0.095, oti=True, gamma_o=0.5, gamma_e=0.5,
                   embed_full=0, pct_mode=0, oti_target=0, dp_start=2, inclusive=1,
                   arith="tree", dmax=0):
    """Defaults = acoss call site (rqa_serra09.py:31-32,60-64) + recalled essentia defaults."""
    ar = {"tree": 0, "seq108": 1}[arith] if isinstance(arith, str) else int(arith)
    return Serra09Params(int(m), int(tau), float(kappa), int(bool(oti)), float(gamma_o),
                         float(gamma_e), int(embed_full), int(pct_mode), int(oti_target),
                         int(dp_start), int(inclusive), ar, int(dmax))


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        fp = ctypes.POINTER(ctypes.c_float)
        L.acx_o_serra09_pair.restype = ctypes.c_float
        L.acx_o_serra09_pair.argtypes = [fp, ctypes.c_int32, fp, ctypes.c_int32,
                                         ctypes.POINTER(Serra09Params), fp, fp, fp,
                                         ctypes.POINTER(ctypes.c_uint8),
                                         ctypes.POINTER(ctypes.c_int32)]
        L.acx_o_serra09_pairs.restype = ctypes.c_int
        L.acx_o_serra09_pairs.argtypes = [fp, ctypes.POINTER(ctypes.c_int64), ctypes.c_int32,
                                          ctypes.POINTER(ctypes.c_int32), ctypes.c_int64,
                                          ctypes.POINTER(Serra09Params), fp]
        L.acx_o_embed_len.restype = ctypes.c_int32
        L.acx_o_embed_len.argtypes = [ctypes.c_int32, ctypes.POINTER(Serra09Params)]
        L.acx_o_global_chroma.restype = None
        L.acx_o_global_chroma.argtypes = [fp, ctypes.c_int32, fp]
        L.acx_o_oti.restype = ctypes.c_int32
        L.acx_o_oti.argtypes = [fp, fp]
        L.acx_o_qmax_binary.restype = ctypes.c_float
        L.acx_o_qmax_binary.argtypes = [ctypes.POINTER(ctypes.c_uint8), ctypes.c_int32, ctypes.c_int32,
                                        ctypes.c_float, ctypes.c_float, ctypes.c_int32]
        L.acx_o_sw_constrained.restype = ctypes.c_double
        L.acx_o_sw_constrained.argtypes = [ctypes.POINTER(ctypes.c_uint8), ctypes.c_int32, ctypes.c_int32]
        L.acx_o_sw_constrained_i32.restype = ctypes.c_int32
        L.acx_o_sw_constrained_i32.argtypes = [ctypes.POINTER(ctypes.c_uint8), ctypes.c_int32, ctypes.c_int32]
        _lib = L
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _fptr(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


# ----------------------------------------------------------------------------
# Serra09 (rqa_serra09.py:44-83 + essentia)
# ----------------------------------------------------------------------------

def sync_median(chroma, fac):
    """librosa.util.sync(chroma.T, arange(0, T0, fac), aggregate=np.median).T as
    called at rqa_serra09.py:51: segment medians over boundaries
    unique({0, fac, 2fac, ..., T0}); dtype preserved (f32 in -> f32 out;
    the median of an even count is the mean of the two middle values)."""
    chroma = np.asarray(chroma)
    T0 = chroma.shape[0]
    bounds = np.unique(np.concatenate([[0], np.arange(0, T0, fac), [T0]]))
    out = np.empty((len(bounds) - 1, chroma.shape[1]), dtype=chroma.dtype)
    for k in range(len(bounds) - 1):
        out[k] = np.median(chroma[bounds[k]:bounds[k + 1]], axis=0)
    return out


def serra09_embed_len(T, params=None):
    p = params or serra09_params()
    return int(lib().acx_o_embed_len(int(T), ctypes.byref(p)))


def serra09_pair(query, reference, params=None, want_intermediates=False):
    """One pair through the full chain.  query (Tq,12), reference (Tr,12) pooled chroma.
    Returns score, or (score, dict(d, eps_q, eps_r, R, oti)) with intermediates."""
    p = params or serra09_params()
    q = _f32(query)
    r = _f32(reference)
    assert q.ndim == 2 and q.shape[1] == 12 and r.ndim == 2 and r.shape[1] == 12
    L = lib()
    if not want_intermediates:
        s = L.acx_o_serra09_pair(_fptr(q), q.shape[0], _fptr(r), r.shape[0], ctypes.byref(p),
                                 None, None, None, None, None)
        if s < 0:
            raise RuntimeError("track shorter than the delay-embedding stack")
        return float(s)
    Mq, Mr = serra09_embed_len(q.shape[0], p), serra09_embed_len(r.shape[0], p)
    if Mq <= 0 or Mr <= 0:
        raise RuntimeError("track shorter than the delay-embedding stack")
    d = np.empty((Mq, Mr), np.float32)
    eq = np.empty(Mq, np.float32)
    er = np.empty(Mr, np.float32)
    R = np.empty((Mq, Mr), np.uint8)
    oti = ctypes.c_int32(0)
    s = L.acx_o_serra09_pair(_fptr(q), q.shape[0], _fptr(r), r.shape[0], ctypes.byref(p),
                             _fptr(d), _fptr(eq), _fptr(er),
                             R.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), ctypes.byref(oti))
    return float(s), dict(d=d, eps_q=eq, eps_r=er, R=R, oti=int(oti.value))


def serra09_pairs(frames, offsets, pairs, params=None):
    """Batch over a packed pool.  frames (sum T,12) f32, offsets (n+1) int64, pairs (K,2) int32."""
    p = params or serra09_params()
    frames = _f32(frames)
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    pairs = np.ascontiguousarray(pairs, dtype=np.int32).reshape(-1, 2)
    out = np.empty(len(pairs), np.float32)
    rc = lib().acx_o_serra09_pairs(_fptr(frames), offsets.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)),
                                   len(offsets) - 1, pairs.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
                                   len(pairs), ctypes.byref(p), _fptr(out))
    if rc != 0:
        raise RuntimeError("oracle serra09_pairs failed rc=%d" % rc)
    return out


def serra09_pairs_mt(frames, offsets, pairs, params=None, workers=None, chunk=8):
    """serra09_pairs over a thread pool (ctypes releases the GIL; the C routine keeps no global
    state): same results, in the order of `pairs`."""
    from concurrent.futures import ThreadPoolExecutor
    pairs = np.ascontiguousarray(pairs, dtype=np.int32).reshape(-1, 2)
    workers = workers or max(1, min(64, os.cpu_count() or 1))
    lib()
    frames = _f32(frames)
    chunks = [pairs[a:a + chunk] for a in range(0, len(pairs), chunk)]
    with ThreadPoolExecutor(workers) as ex:
        parts = list(ex.map(lambda ch: serra09_pairs(frames, offsets, ch, params), chunks))
    return np.concatenate(parts) if parts else np.empty(0, np.float32)


def qmax_binary(R, gamma_o=0.5, gamma_e=0.5, dmax=False):
    R = np.ascontiguousarray(R, dtype=np.uint8)
    return float(lib().acx_o_qmax_binary(R.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)),
                                         R.shape[0], R.shape[1], gamma_o, gamma_e, int(dmax)))


def serra09_normalize_by_length(D, lengths):
    """rqa_serra09.py:71-83: D[i, j] /= sqrt(T_j) (column-wise, f32 in place on a copy)."""
    D = np.array(D, dtype=np.float32)
    norm = np.sqrt(np.asarray(lengths, dtype=np.float64))
    for j in range(D.shape[1]):
        D[:, j] /= norm[j]
    return D


# ----------------------------------------------------------------------------
# EarlyFusion per-pair kernels (cross_recurrence.py, alignment_tools.py,
# similarity_fusion.py:38-54, earlyfusion_traile.py:157-198)
# ----------------------------------------------------------------------------

def get_csm(X, Y):
    """cross_recurrence.py:30-48."""
    C = np.sum(X ** 2, 1)[:, None] + np.sum(Y ** 2, 1)[None, :] - 2 * X.dot(Y.T)
    C[C < 0] = 0
    return np.sqrt(C)


def get_csm_cosine(X, Y):
    """cross_recurrence.py:53-73."""
    XN = np.sqrt(np.sum(X ** 2, 1))
    XN[XN == 0] = 1
    YN = np.sqrt(np.sum(Y ** 2, 1))
    YN[YN == 0] = 1
    return 1 - (X / XN[:, None]).dot((Y / YN[:, None]).T)


def get_oti(C1, C2):
    """cross_recurrence.py:75-103: argmax_s sum(roll(C1, s) * C2), first max wins."""
    n = len(C1)
    sc = np.zeros(n)
    for i in range(n):
        sc[i] = np.sum(np.roll(C1, i) * C2)
    return int(np.argmax(sc))


def get_csm_blocked_oti(X, Y, C1, C2, csm_fn=get_csm_cosine):
    """cross_recurrence.py:105-134: roll the bins of the FIRST song by get_oti(C1, C2)."""
    nb = len(C1)
    per = X.shape[1] // nb
    oti = get_oti(C1, C2)
    X1 = np.reshape(X, (X.shape[0], per, nb))
    X1 = np.roll(X1, oti, axis=2)
    X1 = np.reshape(X1, (X.shape[0], per * nb))
    return csm_fn(X1, Y)


def binary_k(kappa, ncols):
    """cross_recurrence.py:149-154: neighbours per row."""
    if kappa == 0:
        return ncols
    if kappa < 1:
        return int(np.round(kappa * ncols))
    return int(kappa)


def csm_to_binary(D, kappa):
    """cross_recurrence.py:136-161: the k smallest of each ROW (argpartition; ties at the
    boundary arbitrary in the reference -- here: stable, lowest column index first)."""
    M, N = D.shape
    if kappa == 0:
        return np.ones_like(D)
    k = binary_k(kappa, N)
    B = np.zeros((M, N), np.uint8)
    if k <= 0:
        return B
    J = np.argsort(D, axis=1, kind="stable")[:, :k]
    B[np.arange(M)[:, None], J] = 1
    return B


def sw_constrained(B):
    """alignment_tools.py:26-46 (f64).  Raises IOError on non-binary input like match(), :23."""
    B = np.ascontiguousarray(B, dtype=np.uint8) if np.all((np.asarray(B) == 0) | (np.asarray(B) == 1)) else None
    if B is None:
        raise IOError("Non-binary elements found in input")
    return float(lib().acx_o_sw_constrained(B.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), B.shape[0], B.shape[1]))


def sw_constrained_i32(B):
    B = np.ascontiguousarray(B, dtype=np.uint8)
    return int(lib().acx_o_sw_constrained_i32(B.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), B.shape[0], B.shape[1]))


def get_wcsm(C, k1, k2, mu=0.5):
    """similarity_fusion.py:38-54."""
    n1 = np.partition(C, k2, 1)[:, 0:k2]
    m1 = np.mean(n1, 1)
    n2 = np.partition(C, k1, 0)[0:k1, :]
    m2 = np.mean(n2, 0)
    eps = m1[:, None] + m2[None, :] + C
    eps = eps / 3
    return np.exp(-C ** 2 / (2 * (mu * eps) ** 2))


def earlyfusion_pair(f1, f2, kappa=0.1, K=10, csm_f64=False):
    """earlyfusion_traile.py:157-198 (the arithmetic only).  f1/f2: dicts with
    mfccs (nb,650) f32, ssms (nb,1225) f32, chromas (nb,480) f32, chroma_med (12,).
    Returns dict(mfccs, ssms, chromas, early) of SW scores + the intermediates.
    csm_f64 (a yardstick, not the reference's arithmetic): the three cross-similarity matrices are
    evaluated in f64 and rounded to f32 once -- what ANY f32 evaluation (numpy's sgemm, an MFMA) approximates.
    Scores that differ between two f32 evaluations sit on row-kappa ties of those matrices; how many pairs
    the reference's own f32 arithmetic moves against this yardstick is the natural size of that effect."""
    csms = {}
    scores = {}
    if csm_f64:
        g1 = {s: np.asarray(f1[s], dtype=np.float64) for s in ("mfccs", "ssms", "chromas")}
        g2 = {s: np.asarray(f2[s], dtype=np.float64) for s in ("mfccs", "ssms", "chromas")}
        cast = lambda C: C.astype(np.float32)
    else:
        g1, g2 = f1, f2
        cast = lambda C: C
    csms["mfccs"] = cast(get_csm(g1["mfccs"], g2["mfccs"]))
    scores["mfccs"] = sw_constrained(csm_to_binary(csms["mfccs"], kappa))
    csms["ssms"] = cast(get_csm(g1["ssms"], g2["ssms"]))
    scores["ssms"] = sw_constrained(csm_to_binary(csms["ssms"], kappa))
    csms["chromas"] = cast(get_csm_blocked_oti(g1["chromas"], g2["chromas"], f1["chroma_med"], f2["chroma_med"]))
    scores["chromas"] = sw_constrained(csm_to_binary(csms["chromas"], kappa))
    wsum = np.zeros_like(csms["mfccs"])
    for s in ("mfccs", "ssms", "chromas"):
        wsum += get_wcsm(csms[s], K, K)
    fused = np.exp(-wsum)
    scores["early"] = sw_constrained(csm_to_binary(fused, kappa))
    return scores, dict(csms=csms, fused=fused)


# ----------------------------------------------------------------------------
# SiMPle (simple_silva.py)
# ----------------------------------------------------------------------------

def simple_pool(feat, WIN=200, SKIP=100):
    """simple_silva.py:34-41: feat (T0,12) -> (12, floor(T0/SKIP)) f64 window means."""
    fo = np.asarray(feat).T
    n = int(fo.shape[1] / SKIP)
    out = np.zeros((fo.shape[0], n))
    for i in range(n):
        out[:, i] = np.mean(fo[:, i * SKIP:i * SKIP + WIN], axis=1)
    return out


def simple_smooth(feat, win_len_smooth=4):
    """simple_silva.py:56-66: hann(win+2, sym) / sum, 'same' zero-fill convolution along
    time, then L2-normalise every column (librosa.util.normalize: columns whose norm is
    below the dtype's tiny are left unscaled)."""
    n = win_len_smooth + 2
    win = 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / (n - 1))  # scipy get_window('hann', n, fftbins=False)
    win = win / np.sum(win)
    out = np.empty_like(feat, dtype=np.float64)
    for c in range(feat.shape[0]):
        out[c] = np.convolve(feat[c], win, mode="full")[(n - 1) // 2:(n - 1) // 2 + feat.shape[1]]
    nrm = np.sqrt(np.sum(out ** 2, axis=0, keepdims=True))
    tiny = np.finfo(out.dtype).tiny
    nrm = np.where(nrm < tiny, 1.0, nrm)
    return out / nrm


def simple_features(feat, WIN=200, SKIP=100):
    """simple_silva.py:34-43."""
    return simple_smooth(simple_pool(feat, WIN, SKIP))


def simple_oti(seq_a, seq_b):
    """simple_silva.py:45-54: shift = argsort(<pa, roll(pb, i)>)[-1]; returns rolled seq_b."""
    pa = np.sum(seq_a, 1)
    pb = np.sum(seq_b, 1)
    v = np.zeros(12)
    for i in range(12):
        v[i] = np.dot(pa, np.roll(pb, i, axis=0))
    si = np.argsort(v)
    return np.roll(seq_b, si[-1], axis=0), int(si[-1])


def simple_sim(seq_a, seq_b, SSLEN=10):
    """simple_silva.py:68-118 restated by its definition (probed equal to the
    reference's incremental STOMP form to 1e-15):
        median_i  min_j  sum_{c,k} (A[c,i+k] - B[c,j+k])^2,
    evaluated as |a_i|^2 + |b_j|^2 - 2 <a_i, b_j> in f64 like the reference (:111)."""
    A = np.asarray(seq_a, dtype=np.float64)
    B = np.asarray(seq_b, dtype=np.float64)
    na, nb = A.shape[1] - SSLEN + 1, B.shape[1] - SSLEN + 1
    wa = np.lib.stride_tricks.sliding_window_view(A, SSLEN, axis=1)  # (12, na, L)
    wb = np.lib.stride_tricks.sliding_window_view(B, SSLEN, axis=1)
    wa = np.transpose(wa, (1, 0, 2)).reshape(na, -1)
    wb = np.transpose(wb, (1, 0, 2)).reshape(nb, -1)
    a2 = np.sum(wa ** 2, 1)
    b2 = np.sum(wb ** 2, 1)
    dist = b2[None, :] + a2[:, None] - 2.0 * wa.dot(wb.T)
    return float(np.median(np.min(dist, axis=1)))


def simple_pair(Si, Sj_raw, SSLEN=10):
    """simple_silva.py:120-126: D[i, j] = -simple_sim(Si, oti(Si, Sj))."""
    Sj, _ = simple_oti(Si, Sj_raw)
    return -simple_sim(Si, Sj, SSLEN)


# ----------------------------------------------------------------------------
# Harness (algorithm_template.py)
# ----------------------------------------------------------------------------

def all_pairs(N, symmetric):
    """algorithm_template.py:168-171: combinations (i<j) or permutations (i!=j), in
    itertools order."""
    if symmetric:
        i, j = np.triu_indices(N, 1)
    else:
        i, j = np.nonzero(~np.eye(N, dtype=bool))
    return np.stack([i, j], 1).astype(np.int64)


def eval_statistics(D, cliques, topsidx=(1, 10, 100, 1000), stable=True):
    """algorithm_template.py:205-290 restated (vectorised over k).  `cliques`: list of
    lists of track indices in dict insertion order.  np.argsort(-D, 1) in the reference is
    the default unstable sort (:234); `stable=True` uses a stable sort so that both sides
    of a parity check share one tie rule."""
    D = np.array(D, dtype=np.float32)
    N = D.shape[0]
    cl = [list(c) for c in cliques]
    Ks = np.array([len(c) for c in cl])
    order = np.argsort(-Ks, kind="stable")   # reference: default sort (:222); equal-size cliques may come in any order
    Ks = Ks[order]
    cl = [cl[i] for i in order]
    idx = np.array([t for c in cl for t in c], dtype=int)
    D = D[idx, :][:, idx]
    np.fill_diagonal(D, -np.inf)
    srt = np.argsort(-D, 1, kind="stable" if stable else None)
    ranks = np.nan * np.ones(N)
    allmap = np.nan * np.ones(N)
    start = 0
    kidx = 0
    for i in range(N):
        if i >= start + Ks[kidx]:
            start += Ks[kidx]
            kidx += 1
            if Ks[kidx] < 2:
                break
        pos = np.nonzero((srt[i] >= start) & (srt[i] < start + Ks[kidx]))[0] + 1
        ir = pos[:-1]
        if len(ir) == 0:
            break
        ranks[i] = ir[0]
        allmap[i] = np.mean(np.arange(1, Ks[kidx]) / ir.astype(np.float64))
    MAP = float(np.nanmean(allmap))
    ranks = ranks[~np.isnan(ranks)]
    MR = float(np.mean(ranks))
    MRR = float(1.0 / N * np.sum(1.0 / ranks))
    MDR = float(np.median(ranks))
    tops = np.array([np.sum(ranks <= t) for t in topsidx], dtype=np.float64)
    return MR, MRR, MDR, MAP, tops


# ----------------------------------------------------------------------------
# SNF late fusion (similarity_fusion.py:15-36, 101-196)
# ----------------------------------------------------------------------------

def snf_getW(D, K, Mu=0.5):
    """similarity_fusion.py:15-36."""
    DSym = 0.5 * (D + D.T)
    np.fill_diagonal(DSym, 0)
    Neighbs = np.partition(DSym, K + 1, 1)[:, 0:K + 1]
    MeanDist = np.mean(Neighbs, 1) * float(K + 1) / float(K)
    Eps = (MeanDist[:, None] + MeanDist[None, :] + DSym) / 3
    Denom = 2 * (Mu * Eps) ** 2
    Denom[Denom == 0] = 1
    return np.exp(-DSym ** 2 / Denom)


def snf_knn_lists(W, K):
    """getS (similarity_fusion.py:124-144) as K (column, weight) pairs per row: the K largest of a
    row, ties in column order (stable descending sort), weights divided by their sum."""
    J = np.argsort(-W, 1, kind="stable")[:, :K]
    V = np.take_along_axis(W, J, 1)
    sn = np.sum(V, 1)
    sn[sn == 0] = 1
    return J.astype(np.int32), V / sn[:, None]


def snf_fuse(Scores, K=5, niters=5, reg_diag=1):
    """similarity_fusion.py:146-196 (doSimilarityFusion -> fused matrix).  Dense
    restatement of the kNN-truncated cross-diffusion; the neighbour set of each row is
    the K largest entries (ties: argpartition order in the reference, stable here)."""
    Ws = [snf_getW(np.array(D, dtype=np.float64), K) for D in Scores]
    n = Ws[0].shape[0]

    def getP(W):
        rs = np.sum(W, 1)
        rs[rs == 0] = 1
        return W / rs[:, None]

    def getS(W):
        J = np.argsort(-W, 1, kind="stable")[:, :K]
        V = np.take_along_axis(W, J, 1)
        sn = np.sum(V, 1)
        sn[sn == 0] = 1
        V = V / sn[:, None]
        S = np.zeros((n, n))
        np.put_along_axis(S, J, V, 1)
        return S

    Ps = [getP(W) for W in Ws]
    Ss = [getS(W) for W in Ws]
    Pts = [np.array(P) for P in Ps]
    nxt = [np.zeros((n, n)) for _ in Pts]
    m = len(Pts)
    pix = np.arange(n)
    for _ in range(niters):
        for i in range(m):
            nxt[i] = nxt[i] * 0
            for k in range(m):
                if i != k:
                    nxt[i] = nxt[i] + Pts[k]
            nxt[i] = nxt[i] / float(m - 1)
            nxt[i] = Ss[i].dot((Ss[i].dot(nxt[i].T)).T)
            if reg_diag > 0:
                nxt[i][pix, pix] += reg_diag
        Pts = nxt  # NB: the reference aliases the two lists from here on (:179) -- kept
    F = np.zeros((n, n))
    for Pt in Pts:
        F += Pt
    return Ws, F / m


# ----------------------------------------------------------------------------
# EarlyFusion block features (earlyfusion_traile.py:100-140, resize_block :214-247)
# ----------------------------------------------------------------------------

def ef_resize(x, rows):
    """skimage.transform.resize(x, (rows, d), anti_aliasing=True, mode='constant') along the first
    axis (resize_block, earlyfusion_traile.py:241-244): scipy.ndimage.gaussian_filter with
    sigma = max(0, (n / rows - 1) / 2) truncated at 4 sigma, zeros outside; linear interpolation on the
    pixel-centre grid, zeros outside; skimage's clip=True: the result is clipped to [min, max] of the
    input block, values equal to the fill value 0 stay 0 when 0 lies outside that range.  PINNED by
    tests/golden/efprep_skimage.npz (the reference run with scikit-image 0.18.3,
    tests/golden/make_efprep_goldens.py).  NaN / inf results -> 0 (:245-246)."""
    x = np.asarray(x, dtype=np.float64)
    n, d = x.shape
    if n == 0:
        return np.zeros((rows, d))
    factor = n / float(rows)
    sigma = max(0.0, (factor - 1.0) / 2.0)
    r = int(4.0 * sigma + 0.5)
    filt = x
    if r > 0:
        k = np.arange(-r, r + 1, dtype=np.float64)
        w = np.exp(-0.5 * k * k / (sigma * sigma))
        w /= w.sum()
        padded = np.concatenate([np.zeros((r, d)), x, np.zeros((r, d))])
        filt = np.stack([np.convolve(padded[:, c], w[::-1], mode="valid") for c in range(d)], axis=1)
    pos = (np.arange(rows) + 0.5) * factor - 0.5
    t0 = np.floor(pos).astype(np.int64)
    f = (pos - t0)[:, None]
    ext = np.concatenate([np.zeros((2, d)), filt, np.zeros((2, d))])       # index t -> ext[t + 2], t in [-2, n + 1]
    a = ext[np.clip(t0, -2, n + 1) + 2]
    b = ext[np.clip(t0 + 1, -2, n + 1) + 2]
    out = (1.0 - f) * a + f * b
    mn, mx = x.min(), x.max()                                # skimage warp(..., clip=True): _clip_warp_output
    keep_fill = not (mn <= 0.0 <= mx)
    fill = out == 0.0
    out = np.minimum(np.maximum(out, mn), mx)
    if keep_fill:
        out[fill] = 0.0
    out[~np.isfinite(out)] = 0
    return out


def ef_block_features(chroma, mfcc_tm, onsets, blocksize=20, mfccs_per_block=50, chromas_per_block=40):
    """EarlyFusion.load_features (earlyfusion_traile.py:100-140): chroma (T, 12), mfcc_tm (T', ncoef)
    time-major (= feats['mfcc_htk'].T), onsets = beat frame indices."""
    chroma = np.asarray(chroma)
    mfcc = np.array(mfcc_tm, dtype=np.float64)
    mfcc[np.isnan(mfcc)] = 0
    onsets = np.asarray(onsets).astype(np.int64)
    nb = max(0, len(onsets) - blocksize)
    R = mfccs_per_block
    rr, cc = np.nonzero(np.tri(R, k=-1, dtype=bool))           # cells (r, c) with c < r, row-major
    out = dict(mfccs=np.zeros((nb, R * mfcc.shape[1]), np.float32), ssms=np.zeros((nb, R * (R - 1) // 2), np.float32),
               chromas=np.zeros((nb, chromas_per_block * chroma.shape[1]), np.float32),
               chroma_med=np.median(chroma, axis=0))
    for b in range(nb):
        blk = ef_resize(mfcc[onsets[b]:onsets[b + blocksize - 1]], R)
        blk = blk - blk.mean(axis=0, keepdims=True)
        nrm = np.linalg.norm(blk, axis=1, keepdims=True)
        blk = blk / np.where(nrm == 0, 1.0, nrm)
        out["mfccs"][b] = blk.ravel()
        sq = np.sum(blk * blk, axis=1)
        d2 = np.maximum(sq[:, None] + sq[None, :] - 2.0 * blk.dot(blk.T), 0.0)
        out["ssms"][b] = np.sqrt(d2)[rr, cc]
        out["chromas"][b] = ef_resize(np.asarray(chroma[onsets[b]:onsets[b + blocksize]], np.float64), chromas_per_block).ravel()
    return out
