"""
CPU suite, part 2: the Serra09 oracle.  PARITY UNPINNED against essentia (see
oracle/acx_oracle.c header) -- these tests pin the restatement to hand-checkable
answers, to its own committed vectors, and to the switchable recalled details.
"""
import numpy as np
import pytest

import oracle
from acoss_amd import synth


def test_qmax_known_answers():
    assert oracle.qmax_binary(np.eye(8, dtype=np.uint8)) == 6.0          # cells (2,2)..(7,7)
    assert oracle.qmax_binary(np.zeros((8, 8), np.uint8)) == 0.0
    assert oracle.qmax_binary(np.eye(2, dtype=np.uint8)) == 0.0          # first two rows/cols stay 0
    e = np.eye(10, dtype=np.uint8)
    e[5, 5] = 0                                                          # one disruption: -gamma_o
    assert oracle.qmax_binary(e) == 3.0 - 0.5 + 4.0
    e[6, 6] = 0                                                          # then an extension: -gamma_e
    # (4,4)=3 -> (5,6)=3-gamma_o=2.5 via the (i-1,j-2) predecessor -> (7,7)=3.5 -> (9,9)=5.5:
    # the skewed path beats staying on the diagonal (2.5-gamma_e=2.25 at (6,6))
    assert oracle.qmax_binary(e, gamma_o=0.5, gamma_e=0.25) == 5.5
    # the two skewed predecessors (i-2,j-1) and (i-1,j-2)
    B = np.zeros((9, 9), np.uint8)
    B[2, 2] = B[4, 3] = B[5, 5] = 1
    assert oracle.qmax_binary(B) == 3.0
    # Dmax adds R[i-1][j] / R[i][j-1] to the skewed predecessors
    assert oracle.qmax_binary(np.ones((6, 6), np.uint8), dmax=True) > oracle.qmax_binary(np.ones((6, 6), np.uint8))


def test_sync_median_segments():
    x = np.arange(95 * 12, dtype=np.float32).reshape(95, 12)
    out = oracle.sync_median(x, 40)
    assert out.shape == (3, 12) and out.dtype == np.float32           # [0,40) [40,80) [80,95)
    assert np.array_equal(out[0], np.median(x[0:40], axis=0))
    assert np.array_equal(out[2], np.median(x[80:95], axis=0))
    assert oracle.sync_median(x[:80], 40).shape == (2, 12)


def test_embed_len_switch():
    assert oracle.serra09_embed_len(2000) == 1991                       # T - m*tau (essentia, recalled)
    assert oracle.serra09_embed_len(2000, oracle.serra09_params(embed_full=1)) == 1992   # paper
    assert oracle.serra09_embed_len(9) == 0
    with pytest.raises(RuntimeError):
        oracle.serra09_pair(np.ones((9, 12)), np.ones((30, 12)))


def test_self_transposed_pair(golden):
    g = golden("serra09_selfpinned")
    q = g["self_q"]
    r = np.roll(q, 5, axis=1)
    s, inter = oracle.serra09_pair(q, r, want_intermediates=True)
    # rolling the reference RIGHT by 7 undoes a roll of 5
    assert inter["oti"] == 7 == int(g["self_oti"])
    M = q.shape[0] - 9
    assert s == float(g["self_score"]) == M - 2                         # the full main diagonal
    assert np.all(np.diag(inter["d"]) == 0)
    # per-row/column quantile property: roughly kappa of each row is recurrent before the AND
    frac = (inter["d"] <= inter["eps_q"][:, None]).mean()
    assert 0.09 < frac < 0.125


def test_selfpinned_vectors(golden):
    g = golden("serra09_selfpinned")
    tree = oracle.serra09_pairs(g["frames"], g["offsets"], g["pairs"])
    assert np.array_equal(tree, g["scores_tree"])
    seq = oracle.serra09_pairs(g["frames"], g["offsets"], g["pairs"], oracle.serra09_params(arith="seq108"))
    assert np.array_equal(seq, g["scores_seq108"])
    # the two arithmetics differ only through borderline threshold cells
    assert np.max(np.abs(tree - seq)) <= 2.0
    assert np.mean(tree == seq) > 0.6


def test_percentile_modes_and_intermediates():
    d = synth.cover_set(n_works=1, versions=2, seed=5, t_range=(50, 60))
    q = d["frames"][d["offsets"][0]:d["offsets"][1]]
    r = d["frames"][d["offsets"][1]:d["offsets"][2]]
    s, it = oracle.serra09_pair(q, r, want_intermediates=True)
    dm = it["d"]
    n = dm.shape[1]
    k = np.float32(n - 1) * np.float32(0.095)
    lo, hi = int(np.floor(k)), int(np.ceil(k))
    srt = np.sort(dm[3])
    want = np.float32(srt[lo] * np.float32(np.float32(hi) - k)) + np.float32(srt[hi] * np.float32(k - np.float32(lo)))
    assert it["eps_q"][3] == np.float32(want)
    Rw = (dm <= it["eps_q"][:, None]) & (dm <= it["eps_r"][None, :])
    assert np.array_equal(Rw.astype(np.uint8), it["R"])
    assert oracle.qmax_binary(it["R"]) == s
    s_lower, it2 = oracle.serra09_pair(q, r, oracle.serra09_params(pct_mode=2), want_intermediates=True)
    assert np.all(it2["eps_q"] <= it["eps_q"])
    # OTI off / rotating the query instead: the chain still runs and OTI is reported
    s0 = oracle.serra09_pair(q, r, oracle.serra09_params(oti=False))
    s1, it3 = oracle.serra09_pair(q, r, oracle.serra09_params(oti_target=1), want_intermediates=True)
    assert s0 >= 0 and s1 >= 0 and (it3["oti"] + it["oti"]) % 12 == 0


def test_map_identical_between_arithmetics():
    d = synth.cover_set(n_works=6, versions=3, seed=11, t_range=(70, 110))
    n = len(d["offsets"]) - 1
    pairs = oracle.all_pairs(n, True).astype(np.int32)
    lens = np.diff(d["offsets"])
    labels = d["labels"]
    cl = {}
    for i, l in enumerate(labels):
        cl.setdefault(l, []).append(i)
    res = []
    for ar in ("tree", "seq108"):
        sc = oracle.serra09_pairs(d["frames"], d["offsets"], pairs, oracle.serra09_params(arith=ar))
        D = np.zeros((n, n), np.float32)
        D[pairs[:, 0], pairs[:, 1]] = sc
        D += D.T
        D = oracle.serra09_normalize_by_length(D, lens)
        res.append(oracle.eval_statistics(D, list(cl.values()), topsidx=(1, 5)))
    assert res[0][3] == res[1][3] and res[0][0] == res[1][0]           # MAP, MR identical
    assert res[0][3] > 0.8                                             # and the set is recoverable


def test_qmax_analytic_known_answers():
    """Hand-derived known answers of the alignment (gamma_o = gamma_e = 0.5; derivations in
    tests/test_gpu_serra09.py::test_qmax_analytic_known_answers, which holds the device to the
    same numbers): they pin the oracle's DP independently of any implementation."""
    import oracle

    def planted(M, N, gaps=(), shift=0):
        R = np.zeros((M, N), np.uint8)
        for i in range(M):
            j = i + shift
            if 0 <= j < N and i not in gaps:
                R[i, j] = 1
        return R
    for M, N in [(40, 40), (300, 280), (120, 400)]:
        L = min(M, N)
        assert oracle.qmax_binary(planted(M, N)) == L - 2
        assert oracle.qmax_binary(planted(M, N, (L // 2,))) == L - 3.5
        assert oracle.qmax_binary(planted(M, N, (L // 2, L // 2 + 1))) == L - 4.5
        assert oracle.qmax_binary(planted(M, N, (L // 2, L // 2 + 1)), 1.0, 0.25) == L - 5
        assert oracle.qmax_binary(np.zeros((M, N), np.uint8)) == 0.0
        assert oracle.qmax_binary(planted(M, N), dmax=True) == L - 2
        assert oracle.qmax_binary(np.ones((M, N), np.uint8)) == L - 2
        if N >= M + 3:
            assert oracle.qmax_binary(planted(M, N, shift=3)) == M - 2
    one = np.zeros((30, 30), np.uint8)
    one[5, 7] = 1
    assert oracle.qmax_binary(one) == 1.0


def test_self_pair_scores_full_diagonal():
    """Chain-level known answer: a track against itself scores M - 2 (full main diagonal)."""
    import oracle
    from acoss_amd import synth
    rng = np.random.default_rng(404)
    for T in (60, 333):
        x = synth._frame_max_normalise(rng.random((T, 12)))
        assert oracle.serra09_pair(x, x) == T - 9 - 2
        assert oracle.serra09_pair(x, x, oracle.serra09_params(arith="seq108")) == T - 9 - 2


def test_integer_percentile_position():
    """kappa = 0.095 and n - 1 = 200 cells per row put the percentile position k = 19.0 on an order
    statistic.  pct_mode 0 (default) takes d_(19); pct_mode 1 evaluates the interpolation formula as
    recalled from essentia, whose two weights are both 0 there: threshold 0, an empty recurrence plot
    for rows of that length (include/acx.h documents the deliberate deviation of the default)."""
    import oracle
    from acoss_amd import synth
    rng = np.random.default_rng(200)
    q = synth._frame_max_normalise(rng.random((210, 12)))      # 201 embedded frames
    r = synth._frame_max_normalise(rng.random((210, 12)))
    s0, i0 = oracle.serra09_pair(q, r, oracle.serra09_params(pct_mode=0), want_intermediates=True)
    s1, i1 = oracle.serra09_pair(q, r, oracle.serra09_params(pct_mode=1), want_intermediates=True)
    d = np.sort(i0["d"], axis=1)
    assert np.array_equal(i0["eps_q"], d[:, 19])
    assert np.all(i1["eps_q"] == 0.0) and np.all(i1["eps_r"] == 0.0) and i1["R"].sum() == 0 and s1 == 0.0
    assert i0["R"].sum() > 0 and s0 > 0.0
    # one frame more: k = 19.095, both modes interpolate alike
    q2 = synth._frame_max_normalise(rng.random((211, 12)))
    a = oracle.serra09_pair(q2, q2[::-1].copy(), oracle.serra09_params(pct_mode=0))
    b = oracle.serra09_pair(q2, q2[::-1].copy(), oracle.serra09_params(pct_mode=1))
    assert a == b
