"""
GPU parity tests (run with -m gpu on a real MI355X): the HIP Serra09 chain, called through
the C ABI (libacx.so via ctypes), against the CPU oracle on the same seeded inputs.

Bar: BIT-EXACT.  The oracle's default "tree" arithmetic is the arithmetic spec of the
kernels (per-frame fmaf chains == v_mfma_f32_16x16x4_f32, doubling-tree window sums, f32
percentile interpolation), so every intermediate (distances, thresholds, recurrence plot)
and the final Qmax score must be identical.  Against the oracle's "seq108" arithmetic
(sequential 108-dim inner products as essentia is recalled to do) the stated tolerance is:
scores within +-2.0 and identical MAP / MR.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from acoss_amd import _lib
    c = _lib.Context(0)
    yield c
    c.close()


def _oracle():
    import oracle
    return oracle


def _track(d, i):
    return d["frames"][d["offsets"][i]:d["offsets"][i + 1]]


def _compare_pair(ctx, d, i, j, gp, op, tag=""):
    oracle = _oracle()
    g = ctx.serra09_debug_pair(i, j, gp)
    s, it = oracle.serra09_pair(_track(d, i), _track(d, j), op, want_intermediates=True)
    assert g["oti"] == it["oti"], "%s oti %d vs %d" % (tag, g["oti"], it["oti"])
    dg = np.sqrt(g["d2"])
    nbad = int(np.sum(dg != it["d"]))
    assert nbad == 0, "%s distances differ in %d / %d cells, max |diff| %g" % (
        tag, nbad, dg.size, float(np.max(np.abs(dg - it["d"]))))
    bq = np.nonzero(g["eps_q"] != it["eps_q"])[0]
    br = np.nonzero(g["eps_r"] != it["eps_r"])[0]
    assert len(bq) == 0, "%s row thresholds differ at %s: %s vs %s" % (tag, bq[:5], g["eps_q"][bq[:5]], it["eps_q"][bq[:5]])
    assert len(br) == 0, "%s col thresholds differ at %s: %s vs %s" % (tag, br[:5], g["eps_r"][br[:5]], it["eps_r"][br[:5]])
    # the device compares SQUARED distances against thresholds moved to the d2 domain: thr = the
    # largest f32 x with sqrt(x) <= eps (inclusive) or sqrt(x) < eps (exclusive), so `d2 <= thr` is the
    # test in both modes; it must reproduce the oracle's comparison of d = sqrt(d2) with eps
    Rg = (g["d2"] <= g["thr_q"][:, None]) & (g["d2"] <= g["thr_r"][None, :])
    if op.inclusive:
        Rd = (dg <= g["eps_q"][:, None]) & (dg <= g["eps_r"][None, :])
    else:
        Rd = (dg < g["eps_q"][:, None]) & (dg < g["eps_r"][None, :])
    assert np.array_equal(Rg, Rd), "%s d2-domain thresholds disagree with d-domain comparison in %d cells" % (
        tag, int(np.sum(Rg != Rd)))
    assert np.array_equal(Rg.astype(np.uint8), it["R"]), "%s recurrence plot differs in %d cells" % (
        tag, int(np.sum(Rg.astype(np.uint8) != it["R"])))
    assert g["score"] == s, "%s score %r vs %r" % (tag, g["score"], s)
    return g, it


def test_device_sqrt_is_correctly_rounded(ctx):
    rng = np.random.default_rng(3)
    x = np.concatenate([rng.random(200000).astype(np.float32) * 100,
                        np.float32(2.0) ** rng.integers(-60, 60, 20000).astype(np.float32) * rng.random(20000).astype(np.float32),
                        np.array([0.0, 1e-45, 1e-40, 1.0, 4.0, 3.4e38], np.float32)])
    assert np.array_equal(ctx.debug_sqrt(x), np.sqrt(x))


def test_intermediates_bit_exact_small(ctx):
    from acoss_amd import synth, _lib
    oracle = _oracle()
    d = synth.cover_set(n_works=3, versions=2, seed=21, t_range=(40, 150))
    ctx.upload_pool(d["frames"], d["offsets"])
    n = len(d["offsets"]) - 1
    for i in range(n):
        for j in range(n):
            if i != j:
                _compare_pair(ctx, d, i, j, _lib.serra09_params(), oracle.serra09_params(), "pair(%d,%d)" % (i, j))


def test_intermediates_bit_exact_ragged_and_large(ctx):
    from acoss_amd import synth, _lib
    oracle = _oracle()
    rng = np.random.default_rng(5)
    lens = [10, 11, 64 + 9, 65 + 9, 128 + 9, 513 + 9, 700, 1033, 2000]
    tracks = [synth._frame_max_normalise(rng.random((T, 12))) for T in lens]
    frames, offsets = synth.pack(tracks)
    d = dict(frames=frames, offsets=offsets)
    ctx.upload_pool(frames, offsets)
    for (i, j) in [(0, 1), (1, 0), (0, 8), (2, 3), (3, 4), (4, 2), (5, 6), (6, 5), (7, 5), (8, 7), (7, 8)]:
        _compare_pair(ctx, d, i, j, _lib.serra09_params(), oracle.serra09_params(), "len(%d,%d)" % (lens[i], lens[j]))


def test_ties_and_degenerate_rows(ctx):
    """Heavy exact ties (piecewise-constant tracks without noise, repeated frames, an all-equal
    track): the histogram selection's candidate bins overflow and the generic narrowing path /
    the all-equal shortcut must give the oracle's thresholds bit for bit."""
    from acoss_amd import synth, _lib
    oracle = _oracle()
    rng = np.random.default_rng(77)
    protos = synth._frame_max_normalise(rng.random((5, 12)))
    def steps(T, seg):
        idx = np.repeat(rng.integers(0, len(protos), T // seg + 1), seg)[:T]
        return protos[idx].astype(np.float32)
    tracks = [steps(300, 7), steps(257, 3), np.repeat(protos[:1], 120, axis=0).astype(np.float32),
              steps(900, 40), synth._frame_max_normalise(rng.random((400, 12))), steps(2000, 25)]
    frames, offsets = synth.pack(tracks)
    d = dict(frames=frames, offsets=offsets)
    ctx.upload_pool(frames, offsets)
    for (i, j) in [(0, 1), (1, 0), (2, 0), (0, 2), (2, 2), (3, 4), (4, 3), (3, 0), (5, 3), (3, 5), (5, 5)]:
        for kw in (dict(), dict(pct_mode=2), dict(kappa=0.5)):
            _compare_pair(ctx, d, i, j, _lib.serra09_params(**kw), oracle.serra09_params(**kw), "ties(%d,%d) %s" % (i, j, kw))


def test_pivot_filter_gives_up_on_periodic_rows(ctx):
    """Rows on which the pivot-filtered histogram must give up: tracks that repeat a 32-frame (16-frame, 8-frame)
    pattern put ONE near-zero distance into every owner lane's run of consecutive positions, so the largest
    lane minimum -- the pivot -- sits far below the kappa-percentile and fewer than k + 2 cells pass the
    filter.  The unfiltered pass takes over; thresholds, recurrence plot and score stay the oracle's bit for bit
    (debug entry point) and the production kernels return the same score."""
    from acoss_amd import synth, _lib
    oracle = _oracle()
    rng = np.random.default_rng(2024)
    def periodic(T, period, noise):
        base = synth._frame_max_normalise(rng.random((period, 12)))
        x = np.tile(base, (T // period + 1, 1))[:T] + noise * rng.random((T, 12))
        return x.astype(np.float32)
    tracks = [periodic(2000, 32, 0.02), periodic(1900, 32, 0.02), periodic(900, 16, 0.02), periodic(880, 32, 0.03),
              periodic(450, 8, 0.02), periodic(400, 16, 0.02)]
    # two tracks built on the SAME pattern, so that the near-zero cells exist between them
    tracks[1] = (tracks[0][:1900] + 0.01 * rng.random((1900, 12))).astype(np.float32)
    tracks[3] = (tracks[2][:880] + 0.01 * rng.random((880, 12))).astype(np.float32)
    tracks[5] = (tracks[4][:400] + 0.01 * rng.random((400, 12))).astype(np.float32)
    frames, offsets = synth.pack(tracks)
    d = dict(frames=frames, offsets=offsets)
    ctx.upload_pool(frames, offsets)
    pairs = [(0, 1), (1, 0), (2, 3), (3, 2), (4, 5), (5, 4), (0, 2)]
    scores = ctx.serra09_pairs(np.array(pairs, np.int32), _lib.serra09_params())
    for (i, j), sc in zip(pairs, scores):
        g, it = _compare_pair(ctx, d, i, j, _lib.serra09_params(), oracle.serra09_params(), "periodic(%d,%d)" % (i, j))
        assert sc == g["score"]


def test_full_size_pair_2000(ctx):
    from acoss_amd import synth, _lib
    oracle = _oracle()
    d = synth.rand_set(3, T=2000, seed=1234)
    ctx.upload_pool(d["frames"], d["offsets"])
    g, it = _compare_pair(ctx, d, 0, 1, _lib.serra09_params(), oracle.serra09_params(), "T=2000")
    assert g["d2"].shape == (1991, 1991)


@pytest.mark.parametrize("kw", [
    dict(pct_mode=1), dict(pct_mode=2), dict(pct_mode=3), dict(inclusive=0), dict(dp_start=3),
    dict(embed_full=1), dict(oti=False), dict(oti_target=1), dict(gamma_o=1.0, gamma_e=0.25),
    dict(gamma_o=0.25, gamma_e=1.5), dict(m=4), dict(m=12), dict(m=16), dict(m=1), dict(kappa=0.3),
    dict(kappa=0.0), dict(kappa=1.0), dict(dmax=1), dict(dmax=1, dp_start=3), dict(dmax=1, gamma_o=1.0, gamma_e=0.25),
    dict(dmax=1, gamma_o=1.0, gamma_e=0.5, m=1, kappa=0.3, pct_mode=2, inclusive=0, dp_start=3, oti=False),
    dict(dmax=1, gamma_o=0.25, gamma_e=2.0, kappa=0.7),
])
def test_parameter_switches_bit_exact(ctx, kw):
    from acoss_amd import synth, _lib
    oracle = _oracle()
    d = synth.cover_set(n_works=2, versions=2, seed=33, t_range=(70, 200))
    ctx.upload_pool(d["frames"], d["offsets"])
    for (i, j) in [(0, 1), (2, 1), (3, 0)]:
        _compare_pair(ctx, d, i, j, _lib.serra09_params(**kw), oracle.serra09_params(**kw), "%s (%d,%d)" % (kw, i, j))


def test_all_pairs_scores_and_map(ctx):
    """covers80-shaped plumbing case at reduced track length: every pair bit-exact, and the
    MAP / MR of the resulting matrix identical to the oracle's (tree) and to seq108's."""
    from acoss_amd import synth
    oracle = _oracle()
    d = synth.cover_set(clique_sizes=[2] * 10 + [3, 3, 4], seed=4321, t_range=(60, 140))
    n = len(d["offsets"]) - 1
    ctx.upload_pool(d["frames"], d["offsets"])
    pairs = oracle.all_pairs(n, True).astype(np.int32)
    got = ctx.serra09_pairs(pairs)
    ref = oracle.serra09_pairs(d["frames"], d["offsets"], pairs)
    assert np.array_equal(got, ref), "scores differ for %d / %d pairs" % (int(np.sum(got != ref)), len(ref))
    ref108 = oracle.serra09_pairs(d["frames"], d["offsets"], pairs, oracle.serra09_params(arith="seq108"))
    assert np.max(np.abs(got - ref108)) <= 2.0
    cl = {}
    for i, l in enumerate(d["labels"]):
        cl.setdefault(l, []).append(i)
    stats = []
    for sc in (got, ref, ref108):
        D = np.zeros((n, n), np.float32)
        D[pairs[:, 0], pairs[:, 1]] = sc
        D += D.T
        D = oracle.serra09_normalize_by_length(D, np.diff(d["offsets"]))
        stats.append(oracle.eval_statistics(D, list(cl.values()), topsidx=(1, 10)))
    assert stats[0][:4] == stats[1][:4]
    assert abs(stats[0][3] - stats[2][3]) <= 1e-4 and stats[0][0] == stats[2][0]


def test_chenfusion_pairs_qmax_and_dmax(ctx):
    """acx_chenfusion_pairs: one recurrence plot, both alignments (latefusion_chen.py:58-72);
    each column bit-exact vs the oracle run with dmax = 0 / 1, incl. a T = 2000 pair."""
    from acoss_amd import synth, _lib
    oracle = _oracle()
    d = synth.cover_set(n_works=3, versions=2, seed=12, t_range=(60, 260))
    n = len(d["offsets"]) - 1
    ctx.upload_pool(d["frames"], d["offsets"])
    pairs = oracle.all_pairs(n, False).astype(np.int32)
    got = ctx.chenfusion_pairs(pairs)
    q = oracle.serra09_pairs(d["frames"], d["offsets"], pairs)
    dm = oracle.serra09_pairs(d["frames"], d["offsets"], pairs, oracle.serra09_params(dmax=1))
    assert np.array_equal(got[:, 0], q) and np.array_equal(got[:, 1], dm)
    assert np.array_equal(ctx.serra09_pairs(pairs, _lib.serra09_params(dmax=1)), dm)
    assert np.all(dm >= q)
    big = synth.rand_set(2, T=2000, seed=5)
    ctx.upload_pool(big["frames"], big["offsets"])
    pr = np.array([[0, 1]], np.int32)
    got = ctx.chenfusion_pairs(pr)
    assert got[0, 0] == oracle.serra09_pairs(big["frames"], big["offsets"], pr)[0]
    assert got[0, 1] == oracle.serra09_pairs(big["frames"], big["offsets"], pr, oracle.serra09_params(dmax=1))[0]


def test_full_size_properties_without_the_oracle(ctx):
    """Size-independent properties at the benchmark's track length (T = 2000), no oracle involved:
    (1) OTI invariance -- rolling the chroma bins of either track by any number of semitones changes
    the transposition index but not one bit of the score (the rotated chain order is the same);
    (2) a pair scores the same alone, inside a large batch, and in any position of the pair list;
    (3) Dmax >= Qmax on the same recurrence plot."""
    from acoss_amd import synth, _lib
    rng = np.random.default_rng(2024)
    base = [synth._frame_max_normalise(rng.random((2000, 12))) for _ in range(3)]
    tracks = list(base)
    shifts = [1, 5, 7, 11]
    for s in shifts:
        tracks.append(np.ascontiguousarray(np.roll(base[1], s, axis=1)))      # 3 + k: base[1] transposed by s
    frames, offsets = synth.pack(tracks)
    ctx.upload_pool(frames, offsets)
    ref = ctx.serra09_pairs(np.array([[0, 1], [1, 0], [2, 1]], np.int32))
    for k, s in enumerate(shifts):
        got = ctx.serra09_pairs(np.array([[0, 3 + k], [3 + k, 0], [2, 3 + k]], np.int32))
        assert np.array_equal(got, ref), (s, got, ref)
    n = len(tracks)
    i, j = np.nonzero(~np.eye(n, dtype=bool))
    pairs = np.stack([i, j], 1).astype(np.int32)
    big = ctx.serra09_pairs(pairs)
    perm = rng.permutation(len(pairs))
    assert np.array_equal(ctx.serra09_pairs(pairs[perm]), big[perm])
    for q in (0, 7, len(pairs) - 1):
        assert ctx.serra09_pairs(pairs[q:q + 1])[0] == big[q]
    qd = ctx.chenfusion_pairs(pairs[:6])
    assert np.array_equal(qd[:, 0], big[:6]) and np.all(qd[:, 1] >= qd[:, 0])


def test_band_limit_and_long_tracks(ctx):
    """Rows of up to 2041 cells (T = 2050 with the default stack) run in the band kernel; longer
    tracks take the streaming kernels (serra09_long_kernels.hpp).  Both sides of the boundary, a
    mixed batch, and every intermediate of a long pair: bit-exact vs the oracle."""
    from acoss_amd import synth, _lib
    oracle = _oracle()
    rng = np.random.default_rng(17)
    lens = (2050, 2050, 2051, 1200, 2300, 100)
    tracks = [synth._frame_max_normalise(rng.random((T, 12))) for T in lens]
    frames, offsets = synth.pack(tracks)
    d = dict(frames=frames, offsets=offsets)
    ctx.upload_pool(frames, offsets)
    pairs = np.array([[0, 1], [1, 0], [3, 0], [0, 3], [2, 0], [0, 2], [2, 4], [4, 3], [5, 4], [4, 5], [3, 5]], np.int32)
    got = ctx.serra09_pairs(pairs)
    ref = oracle.serra09_pairs(frames, offsets, pairs)
    assert np.array_equal(got, ref), (got, ref)
    _compare_pair(ctx, d, 2, 4, _lib.serra09_params(), oracle.serra09_params(), "long(2051,2300)")
    _compare_pair(ctx, d, 5, 2, _lib.serra09_params(pct_mode=1, inclusive=0), oracle.serra09_params(pct_mode=1, inclusive=0), "long(100,2051)")


def test_long_tracks_5000(ctx):
    """T = 5 000 pooled frames (40 minutes of audio at the default profile): three DP strips of
    2048 columns, Qmax and Dmax, equal and distinct gap penalties, structured and i.i.d. chroma."""
    from acoss_amd import synth, _lib
    oracle = _oracle()
    rng = np.random.default_rng(5000)
    cov = synth.cover_set(n_works=1, versions=2, seed=50, t_range=(4300, 4400))
    tracks = [synth._frame_max_normalise(rng.random((5000, 12))), synth._frame_max_normalise(rng.random((4200, 12))),
              cov["frames"][cov["offsets"][0]:cov["offsets"][1]], cov["frames"][cov["offsets"][1]:cov["offsets"][2]],
              synth._frame_max_normalise(rng.random((700, 12)))]
    frames, offsets = synth.pack(tracks)
    ctx.upload_pool(frames, offsets)
    pairs = np.array([[0, 1], [1, 0], [2, 3], [3, 2], [4, 0], [0, 4], [2, 0]], np.int32)
    for kw in (dict(), dict(dmax=1), dict(gamma_o=1.0, gamma_e=0.25), dict(dmax=1, gamma_o=0.25, gamma_e=1.5, dp_start=3)):
        got = ctx.serra09_pairs(pairs, _lib.serra09_params(**kw))
        ref = oracle.serra09_pairs(frames, offsets, pairs, oracle.serra09_params(**kw))
        assert np.array_equal(got, ref), (kw, got, ref)
    both = ctx.chenfusion_pairs(pairs[:4])
    assert np.array_equal(both[:, 0], oracle.serra09_pairs(frames, offsets, pairs[:4]))
    assert np.array_equal(both[:, 1], oracle.serra09_pairs(frames, offsets, pairs[:4], oracle.serra09_params(dmax=1)))
    assert both[2, 0] > 50.0          # the two versions of one work align over a long stretch


@pytest.mark.parametrize("kw", [dict(tau=2), dict(tau=3, m=5), dict(tau=2, embed_full=1), dict(m=17), dict(m=24, kappa=0.2),
                                dict(m=33), dict(m=20, tau=2, dmax=1)])
def test_stack_stride_and_large_stacks(ctx, kw):
    """frameStackStride > 1 (the pool decimated on the device) and stacks of more than 16 frames
    (streaming kernels with a run-time m): every intermediate bit-exact vs the oracle."""
    from acoss_amd import synth, _lib
    oracle = _oracle()
    d = synth.cover_set(n_works=2, versions=2, seed=61, t_range=(150, 330))
    ctx.upload_pool(d["frames"], d["offsets"])
    for (i, j) in [(0, 1), (2, 1), (3, 0)]:
        _compare_pair(ctx, d, i, j, _lib.serra09_params(**kw), oracle.serra09_params(**kw), "%s (%d,%d)" % (kw, i, j))
    ctx.upload_pool(d["frames"], d["offsets"])        # a later default call sees the undecimated pool again
    pr = np.array([[0, 1], [1, 2]], np.int32)
    a = ctx.serra09_pairs(pr, _lib.serra09_params(**kw))
    b = ctx.serra09_pairs(pr)
    assert np.array_equal(a, oracle.serra09_pairs(d["frames"], d["offsets"], pr, oracle.serra09_params(**kw)))
    assert np.array_equal(b, oracle.serra09_pairs(d["frames"], d["offsets"], pr))


def test_integer_percentile_position(ctx):
    """n - 1 = 200 cells per row: the percentile position is an exact integer, where pct_mode 0 and 1
    differ (include/acx.h); both modes bit-exact vs the oracle, mode 1 yields empty plots."""
    from acoss_amd import synth, _lib
    oracle = _oracle()
    rng = np.random.default_rng(200)
    tracks = [synth._frame_max_normalise(rng.random((T, 12))) for T in (210, 210, 410, 211)]
    frames, offsets = synth.pack(tracks)
    d = dict(frames=frames, offsets=offsets)
    ctx.upload_pool(frames, offsets)
    for mode in (0, 1):
        for (i, j) in [(0, 1), (2, 0), (0, 2), (3, 0), (0, 3)]:
            _compare_pair(ctx, d, i, j, _lib.serra09_params(pct_mode=mode), oracle.serra09_params(pct_mode=mode), "pct%d(%d,%d)" % (mode, i, j))
    assert ctx.serra09_pairs(np.array([[0, 1]], np.int32), _lib.serra09_params(pct_mode=1))[0] == 0.0
    assert ctx.serra09_pairs(np.array([[0, 1]], np.int32))[0] > 0.0


def _planted(M, N, gaps=(), shift=0):
    R = np.zeros((M, N), np.uint8)
    for i in range(M):
        j = i + shift
        if 0 <= j < N and i not in gaps:
            R[i, j] = 1
    return R


@pytest.mark.parametrize("M,N", [(40, 40), (300, 280), (700, 900), (1500, 2041), (2600, 2300), (64, 4500)])
def test_qmax_analytic_known_answers(ctx, M, N):
    """Oracle-independent known answers of the alignment (hand-derived from the recurrences of
    Serra et al. 2009 eq. 8 / Chen et al. 2017 with gamma_o = gamma_e = 0.5; first two rows and
    columns of Q are zero):
      * full main diagonal of L = min(M, N) ones: Q[i][i] = i - 1  ->  L - 2;
      * the same with ONE missing cell in the interior: that cell scores Q - 0.5 instead of Q + 1
        ->  L - 3.5;
      * TWO adjacent missing cells (k, k), (k+1, k+1): the best path leaves the diagonal through the
        gap cell (k, k+1), reached from (k-1, k-1) by the (i-1, j-2) step at the price of one onset
        penalty (Q = k - 2 - 0.5), and rejoins it at (k+2, k+2) by the (i-2, j-1) step: 2 matches
        lost and 0.5 paid  ->  L - 4.5 (with gamma_o = 1: L - 5);
      * an empty plot -> 0; a single isolated one at (5, 7) -> 1;
      * Dmax on a plot whose only ones are a full diagonal equals Qmax (the extra terms read
        off-diagonal cells, all zero);
      * all ones: Q[i][j] = min(i, j) - 1 -> min(M, N) - 2 (Qmax)."""
    from acoss_amd import _lib
    L = min(M, N)
    P = _lib.serra09_params
    assert ctx.qmax_binary(_planted(M, N)) == L - 2
    assert ctx.qmax_binary(_planted(M, N, gaps=(L // 2,))) == L - 3.5
    assert ctx.qmax_binary(_planted(M, N, gaps=(L // 2, L // 2 + 1))) == L - 4.5
    assert ctx.qmax_binary(np.zeros((M, N), np.uint8)) == 0.0
    one = np.zeros((M, N), np.uint8)
    one[5, 7] = 1
    assert ctx.qmax_binary(one) == 1.0
    assert ctx.qmax_binary(_planted(M, N), P(dmax=1)) == L - 2
    assert ctx.qmax_binary(np.ones((M, N), np.uint8)) == L - 2
    # distinct penalties: the detour pays the onset penalty only
    assert ctx.qmax_binary(_planted(M, N, gaps=(L // 2, L // 2 + 1)), P(gamma_o=1.0, gamma_e=0.25)) == L - 5
    # a diagonal shifted off the main one, far enough from the first two columns to count fully
    if N >= M + 3:
        assert ctx.qmax_binary(_planted(M, N, shift=3)) == M - 2


def test_qmax_binary_random_vs_oracle(ctx):
    """The DP alone on random plots of every size class (8 / 16 / 32 columns per lane, strips)."""
    from acoss_amd import _lib
    oracle = _oracle()
    rng = np.random.default_rng(9)
    for (M, N, dens) in [(30, 50, 0.3), (400, 505, 0.1), (506, 300, 0.2), (1000, 1017, 0.1), (900, 1500, 0.15),
                         (2041, 700, 0.1), (1200, 2049, 0.1), (2500, 4100, 0.12), (3, 3, 0.5), (2, 9, 0.9), (1, 1, 1.0)]:
        R = (rng.random((M, N)) < dens).astype(np.uint8)
        for kw in (dict(), dict(dmax=1), dict(gamma_o=1.0, gamma_e=0.25), dict(dmax=1, gamma_o=0.25, gamma_e=2.0)):
            got = ctx.qmax_binary(R, _lib.serra09_params(**kw))
            ref = oracle.qmax_binary(R, kw.get("gamma_o", 0.5), kw.get("gamma_e", 0.5), bool(kw.get("dmax", 0)))
            assert got == ref, (M, N, kw, got, ref)


def test_self_pairs_score_full_diagonal(ctx):
    """Chain-level known answer, no oracle: a track against itself has d = 0 on the whole main
    diagonal (identical operands through identical arithmetic), every threshold is >= 0, so the
    diagonal is recurrent and the score is at least M - 2 = T - 11; i.i.d. tracks have no longer
    path, so it is exactly that."""
    from acoss_amd import synth
    rng = np.random.default_rng(404)
    lens = (60, 333, 1000, 2000, 2050, 2200)
    tracks = [synth._frame_max_normalise(rng.random((T, 12))) for T in lens]
    frames, offsets = synth.pack(tracks)
    ctx.upload_pool(frames, offsets)
    got = ctx.serra09_pairs(np.array([[i, i] for i in range(len(lens))], np.int32))
    assert np.array_equal(got, np.array([T - 9 - 2 for T in lens], np.float32)), got


def test_pairs_of_very_different_sizes_share_a_wave(ctx):
    """Round 5: the alignment of SHORT pairs runs two / four pairs to a wave (qmax_bits_h16_multi_kernel), the band kernels two /
    four rows to a wave.  A pair that runs out of rows idles beside its longer neighbours -- and must not pick anything up while it
    does (the fuzz aid found Dmax lifting the first idle row of a 3-row matrix from its last real row's bits).  Tiny and ordinary
    tracks mixed, every ordered pair, Qmax and Dmax, both start rows: bit-identical to the oracle."""
    import oracle
    from acoss_amd import _lib, synth
    rng = np.random.default_rng(998)
    lens = [146, 86, 7, 147, 146, 52, 12, 240, 11, 230, 300, 13]
    tracks = [synth._frame_max_normalise(rng.random((T, 12))) for T in lens]
    frames, offsets = synth.pack(tracks)
    ctx.upload_pool(frames, offsets)
    i, j = np.nonzero(~np.eye(len(lens), dtype=bool))
    pairs = np.stack([i, j], 1).astype(np.int32)
    for kw in (dict(m=4, kappa=1.0, pct_mode=2, inclusive=0, dp_start=3, oti=False, dmax=1), dict(m=4, kappa=0.3, dmax=1),
               dict(m=4, kappa=0.3, dmax=0), dict(m=9, dmax=1), dict(m=1, kappa=0.02, pct_mode=3, oti_target=1, dmax=1)):
        ok = np.array([min(lens[a], lens[b]) > kw["m"] + 1 for a, b in pairs])
        got = ctx.serra09_pairs(pairs[ok], _lib.serra09_params(**kw))
        ref = oracle.serra09_pairs(frames, offsets, pairs[ok], oracle.serra09_params(**kw))
        assert np.array_equal(got, ref), (kw, pairs[ok][got != ref][:5], got[got != ref][:5], ref[got != ref][:5])
    both = ctx.chenfusion_pairs(pairs[ok], _lib.serra09_params(m=1, kappa=0.02, pct_mode=3, oti_target=1))
    assert np.array_equal(both[:, 1], ref)


def test_batching_is_invisible(ctx):
    """A tiny scratch limit forces many batches; results must not change."""
    from acoss_amd import synth
    oracle = _oracle()
    d = synth.cover_set(n_works=4, versions=2, seed=8, t_range=(50, 90))
    n = len(d["offsets"]) - 1
    ctx.upload_pool(d["frames"], d["offsets"])
    pairs = oracle.all_pairs(n, False).astype(np.int32)       # ordered pairs, both orientations
    a = ctx.serra09_pairs(pairs)
    ctx.set_scratch_limit(3 * 90 * 128 * 4 * 2)
    b = ctx.serra09_pairs(pairs)
    ctx.set_scratch_limit(0)
    assert np.array_equal(a, b)
    assert np.array_equal(a, oracle.serra09_pairs(d["frames"], d["offsets"], pairs))


def test_batches_of_mixed_lengths_keep_their_scores(ctx):
    """Many batches of pairs from several size classes, one and two alignments per pair: the alignment sweeps of a batch run on the
    context's second stream beside the band kernels of the NEXT batch, whose row pass rewrites the shared bitmap arena -- the scores
    must be those of the one-batch run (and the oracle's on a sample), in any order of completion."""
    from acoss_amd import synth, _lib
    oracle = _oracle()
    rng = np.random.default_rng(77)
    lens = [int(v) for v in rng.integers(40, 700, 26)] + [150, 260, 520, 780]
    tracks = [synth._frame_max_normalise(rng.random((T, 12))) for T in lens]
    frames = np.concatenate(tracks).astype(np.float32)
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    n = len(lens)
    ctx.upload_pool(frames, offsets)
    pairs = oracle.all_pairs(n, True).astype(np.int32)
    pairs = pairs[rng.permutation(len(pairs))]
    one = ctx.serra09_pairs(pairs)
    two = ctx.chenfusion_pairs(pairs)
    assert np.array_equal(two[:, 0], one)
    ctx.set_scratch_limit(1 << 20)                          # the 435 pairs in ~10 batches
    try:
        for rep in range(3):
            assert np.array_equal(ctx.serra09_pairs(pairs), one), rep
            assert np.array_equal(ctx.chenfusion_pairs(pairs), two), rep
    finally:
        ctx.set_scratch_limit(0)
    sample = pairs[:48]
    assert np.array_equal(one[:48], oracle.serra09_pairs(frames, offsets, sample))
    assert np.array_equal(two[:48, 1], oracle.serra09_pairs(frames, offsets, sample, oracle.serra09_params(dmax=True)))


def test_error_behaviour(ctx):
    from acoss_amd import synth, _lib
    rng = np.random.default_rng(1)
    tracks = [synth._frame_max_normalise(rng.random((T, 12))) for T in (9, 40)]
    frames, offsets = synth.pack(tracks)
    ctx.upload_pool(frames, offsets)
    with pytest.raises(_lib.AcxError):                      # essentia raises on inputs shorter than the stack
        ctx.serra09_pairs(np.array([[0, 1]], np.int32))
    with pytest.raises(NotImplementedError):
        ctx.serra09_pairs(np.array([[1, 1]], np.int32), _lib.serra09_params(m=34))
    with pytest.raises(_lib.AcxError):                      # 40 frames, stride 5: shorter than the stack
        ctx.serra09_pairs(np.array([[1, 1]], np.int32), _lib.serra09_params(tau=5))
    with pytest.raises(ValueError):
        ctx.serra09_pairs(np.array([[1, 7]], np.int32))
    assert ctx.serra09_pairs(np.zeros((0, 2), np.int32)).shape == (0,)


def test_benchmark_end_to_end_from_feature_files(tmp_path, monkeypatch):
    """benchmark(algorithm="Serra09") over per-track feature files: raw chroma is pooled x40
    on load, every pair goes through libacx, D is mirrored and length-normalised, and the
    statistics equal the oracle's on the same data (configs[1] plumbing at reduced length)."""
    import acoss_amd
    from acoss_amd import synth
    from acoss_amd.featurestore import save_track
    oracle = _oracle()
    d = synth.cover_set(clique_sizes=[2, 2, 3, 2, 1, 1], seed=99, t_range=(45, 90))
    n = len(d["offsets"]) - 1
    root = str(tmp_path) + "/feat/"
    with open(tmp_path / "ds.csv", "w") as f:
        f.write("work_id,track_id\n")
        for i in range(n):
            pooled = _track(d, i)
            raw = np.repeat(pooled, 40, axis=0)[: 40 * len(pooled) - 7 * (i % 3)]   # ragged last segment
            save_track(root + "%s/t%d.h5" % (d["labels"][i], i),
                       {"label": d["labels"][i], "track_id": "t%d" % i, "hpcp": raw})
            f.write("%s,t%d\n" % (d["labels"][i], i))
    monkeypatch.chdir(tmp_path)
    res = acoss_amd.benchmark(str(tmp_path / "ds.csv"), root, feature_type="hpcp", algorithm="Serra09",
                              shortname="toy")
    pairs = oracle.all_pairs(n, True).astype(np.int32)
    sc = oracle.serra09_pairs(d["frames"], d["offsets"], pairs)
    D = np.zeros((n, n), np.float32)
    D[pairs[:, 0], pairs[:, 1]] = sc
    D += D.T
    D = oracle.serra09_normalize_by_length(D, np.diff(d["offsets"]))
    cl = {}
    for i, l in enumerate(d["labels"]):
        cl.setdefault(l, []).append(i)
    want = oracle.eval_statistics(D, list(cl.values()))
    got = res["main"]
    assert got[:4] == want[:4] and np.array_equal(got[4], want[4])
    assert np.array_equal(np.load("cache/Serra09_toy_Ds.npz")["main"] / 1.0, np.load("cache/Serra09_toy_Ds.npz")["main"])
    assert open("results_toy_Serra09.csv").read().startswith("name, MR, MRR, MDR, MAP,Top-1,Top-10,Top-100,Top-1000")


def test_f16x2_gram_is_an_equally_accurate_different_arithmetic(ctx):
    """The opt-in f16x2 Gram (acx_serra09_params.arith = ACX_ARITH_F16X2; Serra09(engine={"arith": "f16x2"})): two-term fp16
    splits, all four term products, on v_mfma_f32_16x16x32_f16.  NOT bit-identical to the f32 spec -- this test pins the
    envelope that was measured (profiles/r04_f16x2.md):
      * squared distances within 3e-5 of the exact mode's (both sit at ~4e-6 rms from f64) in every size class and with a
        transposition;
      * a few cells per 10 000 change side of a threshold; >= 80 % of the scores of a covers80-shaped set identical, >= 99 % within
        north_star's 2.0, none beyond 6.0; MR / Top-1 identical and |dMAP| <= 1e-4 on an easy AND a hard cover set;
      * stack sizes other than 9 are refused, pairs beyond the band kernel take the exact streaming kernels in either mode."""
    import oracle
    from acoss_amd import _lib, synth
    pe, pf = _lib.serra09_params(), _lib.serra09_params(arith="f16x2")
    rng = np.random.default_rng(1)
    for T, shift in ((80, 5), (300, 0), (600, 3), (1200, 2), (2000, 9)):
        q = rng.random((T, 12), dtype=np.float32); q /= q.max(axis=1, keepdims=True)
        r = np.roll(q, shift, axis=1) + 0.3 * rng.random((T, 12), dtype=np.float32); r /= r.max(axis=1, keepdims=True)
        ctx.upload_pool(np.concatenate([q, r]), np.array([0, T, 2 * T]))
        e, f = ctx.serra09_debug_pair(0, 1, pe), ctx.serra09_debug_pair(0, 1, pf)
        assert e["oti"] == f["oti"]
        assert np.max(np.abs(e["d2"].astype(np.float64) - f["d2"])) <= 3e-5, T
        Re = (e["d2"] <= e["thr_q"][:, None]) & (e["d2"] <= e["thr_r"][None, :])
        Rf = (f["d2"] <= f["thr_q"][:, None]) & (f["d2"] <= f["thr_r"][None, :])
        assert np.mean(Re != Rf) <= 1e-3, (T, float(np.mean(Re != Rf)))
        assert abs(e["score"] - f["score"]) <= 6.0

    def cliques(labels):
        cl = {}
        for i, l in enumerate(labels):
            cl.setdefault(l, []).append(i)
        return list(cl.values())
    for d, lo in ((synth.covers80_shaped(seed=100, t_range=(150, 650)), 0.99), (synth.covers80_shaped(seed=7, t_range=(300, 600), noise=1.0, segment_keep=0.5), 0.3)):
        n = len(d["offsets"]) - 1
        ctx.upload_pool(d["frames"], d["offsets"])
        pairs = oracle.all_pairs(n, True).astype(np.int32)
        se, sf = ctx.serra09_pairs(pairs, pe), ctx.serra09_pairs(pairs, pf)
        diff = np.abs(se - sf)
        assert np.mean(diff == 0) >= 0.80 and np.mean(diff <= 2.0) >= 0.99 and diff.max() <= 6.0, (np.mean(diff == 0), np.mean(diff <= 2.0), diff.max())
        st = []
        for s in (se, sf):
            D = np.zeros((n, n), np.float32)
            D[pairs[:, 0], pairs[:, 1]] = s
            D += D.T
            D = (D / np.sqrt(np.diff(d["offsets"]).astype(np.float64))[None, :]).astype(np.float32)
            st.append(oracle.eval_statistics(D, cliques(d["labels"]), topsidx=(1, 10)))
        assert st[0][3] >= lo and abs(st[0][3] - st[1][3]) <= 1e-4, (st[0][3], st[1][3])
        assert abs(st[0][0] - st[1][0]) <= 0.05 and st[0][4][0] == st[1][4][0], (st[0], st[1])
    with pytest.raises(NotImplementedError):
        ctx.serra09_pairs(pairs[:4], _lib.serra09_params(m=7, arith="f16x2"))
    #  features the two-term fp16 split cannot carry -- beyond fp16's range, or so small that the second term falls
    # into its subnormals -- are refused in this mode, and computed as ever in the exact one
    for factor in (2.0 ** 17, 2.0 ** -13):               # (powers of two: the exact chain is invariant to the bit)
        ctx.upload_pool(d["frames"] * np.float32(factor), d["offsets"])
        with pytest.raises(NotImplementedError):
            ctx.serra09_pairs(pairs[:4], pf)
        assert np.array_equal(ctx.serra09_pairs(pairs[:64], pe), se[:64])          # (distances scale, ranks and scores do not)
    ctx.upload_pool(d["frames"], d["offsets"])
    with pytest.raises(ValueError):
        ctx.serra09_pairs(pairs[:4], _lib.serra09_params(arith=5))
    # a pair beyond the band kernel (rows of more than 2041 cells) runs the exact streaming kernels in both modes
    T = 2100
    q = rng.random((T, 12), dtype=np.float32); r = rng.random((T, 12), dtype=np.float32)
    ctx.upload_pool(np.concatenate([q, r]), np.array([0, T, 2 * T]))
    one = np.array([[0, 1]], np.int32)
    assert np.array_equal(ctx.serra09_pairs(one, pe), ctx.serra09_pairs(one, pf))
