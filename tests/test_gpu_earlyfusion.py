"""
GPU parity tests for the EarlyFusion per-pair chain (earlyfusion_traile.py:157-198) through the
C ABI.

Bars:
  * smith_waterman_constrained: BIT-EXACT against the reference's own outputs (goldens) -- the
    device DP is an exact integer restatement in tenths;
  * binarisation / Smith-Waterman given a matrix: exact -- feeding the DEVICE's own CSMs and
    fused matrix to the oracle's csm_to_binary + SW must reproduce the device scores;
  * the f32 GEMM cross-similarity matrices: Euclidean |d^2 - d_ref^2| <= 4e-6 (|x|^2 + |y|^2)
    (about 32 ulp of the f32 norm sum; the order of the 650..1225-term sums differs from BLAS),
    against the reference AND against the f64 truth; cosine |diff| <= 5e-6, fused matrix rel 2e-3; end-to-end scores within +-2.0 of the oracle /
    reference (borderline neighbours can flip), and equal to the reference on the golden pairs
    when nothing flips.
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


def test_smith_waterman_goldens_bit_exact(ctx, golden):
    g = golden("ef_kernels")
    names = [k for k in g.files if k.startswith("sw_B_") or k.startswith("swk_B_")]
    assert len(names) >= 30
    for name in names:
        B = g[name]
        ref = float(g[name.replace("_B_", "_out_")])
        got = ctx.sw_binary(B)
        assert abs(got - ref) < 1e-5 and round(got * 10) == round(ref * 10), (name, got, ref)
    with pytest.raises(IOError):
        ctx.sw_binary(2 * np.ones((8, 8), np.uint8))


def test_smith_waterman_random_shapes_vs_oracle(ctx):
    import oracle
    rng = np.random.default_rng(9)
    for (m, n) in [(4, 4), (5, 64), (64, 5), (65, 65), (300, 511), (512, 512), (129, 8), (8, 500),
                   (513, 513), (700, 40), (40, 700), (1024, 1024), (600, 1023),        # > 512 columns: wide kernel
                   (700, 1025), (1500, 2100), (30, 3000), (2049, 600)]:                 # > 1024 columns: strips
        for dens in (0.05, 0.3, 0.8):
            B = (rng.random((m, n)) < dens).astype(np.uint8)
            assert round(ctx.sw_binary(B) * 10) == oracle.sw_constrained_i32(B), (m, n, dens)
    # long diagonal: the score grows to the matrix size
    assert ctx.sw_binary(np.eye(300, dtype=np.uint8)) == oracle.sw_constrained(np.eye(300, dtype=np.uint8))


def test_binarise_with_ties_keeps_k_cells_in_column_order(ctx):
    """csm_to_binary keeps exactly k cells per row; the oracle (and the device) take ties at the
    k-th value in column order.  Heavily tied matrices: few distinct values, constant rows, the
    saturated fused matrix (most cells exactly 1.0f)."""
    import oracle
    rng = np.random.default_rng(21)
    cases = []
    for (m, n) in [(40, 50), (64, 257), (130, 512), (75, 68), (9, 300), (33, 777), (600, 1024), (90, 1300), (1100, 2500)]:
        cases.append(rng.integers(0, 4, (m, n)).astype(np.float32))                       # 4 distinct values
        cases.append(np.ones((m, n), np.float32))                                         # one value
        D = np.ones((m, n), np.float32)
        hits = rng.random((m, n)) < 0.03
        D[hits] = rng.random(int(hits.sum())).astype(np.float32)                          # saturated + a few neighbours
        cases.append(D)
        cases.append(np.round(rng.random((m, n)) * 20).astype(np.float32) / 20)
    for D in cases:
        for kappa in (0.05, 0.1, 0.5, 0.0, 3):
            ref = oracle.sw_constrained(oracle.csm_to_binary(D, kappa))
            assert round(ctx.csm_binary_sw(D, kappa) * 10) == round(ref * 10), (D.shape, kappa)   # scores are tenths (f32 out)


def _golden_feats(g, pk):
    f1 = {s: g["p%d_f1_%s" % (pk, s)] for s in ("mfccs", "ssms", "chromas", "chroma_med")}
    f2 = {s: g["p%d_f2_%s" % (pk, s)] for s in ("mfccs", "ssms", "chromas", "chroma_med")}
    return f1, f2


def _check_pair(ctx, i, j, f1, f2, ref_csms=None, ref_fused=None, ref_scores=None):
    import oracle
    d = ctx.ef_debug_pair(i, j)
    assert d["oti"] == oracle.get_oti(f1["chroma_med"], f2["chroma_med"])
    oscores, inter = oracle.earlyfusion_pair(f1, f2, kappa=0.1, K=10)
    for k, s in enumerate(("mfccs", "ssms", "chromas")):
        ref = ref_csms[s] if ref_csms is not None else inter["csms"][s]
        if s == "chromas":
            np.testing.assert_allclose(d["csm"][k], ref, rtol=0, atol=5e-6)
        else:
            # Euclidean CSM = sqrt of a difference of ~1e3-sized f32 sums: the error lives in d^2
            scale = float(np.sum(f1[s].astype(np.float64) ** 2, 1).max() + np.sum(f2[s].astype(np.float64) ** 2, 1).max())
            assert np.max(np.abs(d["csm"][k].astype(np.float64) ** 2 - ref.astype(np.float64) ** 2)) <= 4e-6 * scale
            x64, y64 = f1[s].astype(np.float64), f2[s].astype(np.float64)
            true2 = np.maximum(0, np.sum(x64 ** 2, 1)[:, None] + np.sum(y64 ** 2, 1)[None, :] - 2 * x64.dot(y64.T))
            assert np.max(np.abs(d["csm"][k].astype(np.float64) ** 2 - true2)) <= 4e-6 * scale
        # exact downstream: oracle binarise + SW on the DEVICE matrix == device score
        assert round(oracle.sw_constrained(oracle.csm_to_binary(d["csm"][k], 0.1)) * 10) == round(float(d["scores"][k]) * 10)
    # fused matrix from the device CSMs with the oracle's getWCSM
    ws = np.zeros_like(d["csm"][0])
    for k in range(3):
        ws += oracle.get_wcsm(d["csm"][k], 10, 10)
    # f32 on both sides; the kernel weights are exp(-C^2 / (2 (eps / 2)^2)) with exponents of up to ~100, so a
    # 1e-7 relative difference in the neighbourhood means (summation order) shows as ~1e-5 in W
    np.testing.assert_allclose(d["fused"], np.exp(-ws), rtol=2e-4, atol=1e-6)
    assert round(oracle.sw_constrained(oracle.csm_to_binary(d["fused"], 0.1)) * 10) == round(float(d["scores"][3]) * 10)
    want = ref_scores if ref_scores is not None else np.array([oscores[s] for s in ("mfccs", "ssms", "chromas", "early")])
    assert np.all(np.abs(d["scores"] - want) <= 2.0), (d["scores"], want)
    return d


def test_chain_against_reference_goldens(ctx, golden):
    g = golden("ef_chain")
    for pk in (0, 1):
        f1, f2 = _golden_feats(g, pk)
        ctx.ef_upload_pool([f1, f2])
        ref_csms = {s: g["p%d_csm_%s" % (pk, s)] for s in ("mfccs", "ssms", "chromas")}
        d = _check_pair(ctx, 0, 1, f1, f2, ref_csms, g["p%d_fused" % pk], g["p%d_scores" % pk])
        assert d["scores"].max() > 5.0
        got = ctx.earlyfusion_pairs(np.array([[0, 1]], np.int32))
        assert np.array_equal(got[0], d["scores"])


def test_chain_synthetic_ragged(ctx):
    from acoss_amd import synth
    rng = np.random.default_rng(2)
    tracks = synth.earlyfusion_set(4, seed=5, nb_range=(30, 140))
    # plant shared structure so that alignments exist
    for key in ("mfccs", "ssms", "chromas"):
        n = 25
        tracks[1][key][3:3 + n] = tracks[0][key][1:1 + n] + 0.02 * rng.standard_normal((n, tracks[0][key].shape[1])).astype(np.float32)
    ctx.ef_upload_pool(tracks)
    for (i, j) in [(0, 1), (1, 0), (2, 3), (0, 3)]:
        _check_pair(ctx, i, j, tracks[i], tracks[j])
    pairs = np.array([[0, 1], [0, 2], [0, 3], [1, 2], [1, 3], [2, 3]], np.int32)
    a = ctx.earlyfusion_pairs(pairs)
    ctx.set_scratch_limit(4 * 140 * 192 * 4 * 8)          # force one pair per batch
    b = ctx.earlyfusion_pairs(pairs)
    ctx.set_scratch_limit(0)
    assert np.array_equal(a, b) and a.shape == (6, 4)


def test_chain_long_tracks_wide_kernels(ctx):
    """Tracks of 513..1024 blocks (songs beyond ~4.5 min at 120 bpm) run through the 16-values-per-lane
    row statistics and Smith-Waterman variants; mixed with a short track in the same batch."""
    from acoss_amd import synth
    rng = np.random.default_rng(6)
    tracks = synth.earlyfusion_set(2, seed=8, nb_range=(560, 700)) + synth.earlyfusion_set(1, seed=9, nb_range=(1024, 1024)) \
        + synth.earlyfusion_set(1, seed=10, nb_range=(60, 80))
    for key in ("mfccs", "ssms", "chromas"):
        n = 200
        tracks[1][key][300:300 + n] = tracks[0][key][250:250 + n] + 0.02 * rng.standard_normal((n, tracks[0][key].shape[1])).astype(np.float32)
    ctx.ef_upload_pool(tracks)
    d = _check_pair(ctx, 0, 1, tracks[0], tracks[1])
    assert d["scores"].max() > 50.0
    for (i, j) in [(1, 2), (2, 0), (3, 2), (2, 3)]:
        _check_pair(ctx, i, j, tracks[i], tracks[j])
    pairs = np.array([[0, 1], [1, 2], [2, 0], [3, 2], [2, 3], [3, 0]], np.int32)
    a = ctx.earlyfusion_pairs(pairs)
    assert np.array_equal(a[0], d["scores"])


def test_chain_tracks_of_2000_blocks(ctx):
    """More than 1024 blocks (a 17-minute track at 120 bpm): the streaming row statistics and the
    Smith-Waterman in column strips; every intermediate against the oracle, both orientations, mixed
    with short tracks in one batch."""
    from acoss_amd import synth
    rng = np.random.default_rng(16)
    tracks = synth.earlyfusion_set(1, seed=11, nb_range=(2000, 2000)) + synth.earlyfusion_set(1, seed=12, nb_range=(1025, 1025)) \
        + synth.earlyfusion_set(1, seed=13, nb_range=(300, 300))
    for key in ("mfccs", "ssms", "chromas"):
        n = 400
        tracks[1][key][500:500 + n] = tracks[0][key][1200:1200 + n] + 0.02 * rng.standard_normal((n, tracks[0][key].shape[1])).astype(np.float32)
    ctx.ef_upload_pool(tracks)
    d = _check_pair(ctx, 0, 1, tracks[0], tracks[1])
    assert d["scores"].max() > 100.0
    for (i, j) in [(1, 0), (2, 0), (0, 2), (1, 2)]:
        _check_pair(ctx, i, j, tracks[i], tracks[j])
    pairs = np.array([[0, 1], [1, 0], [2, 0], [2, 1]], np.int32)
    a = ctx.earlyfusion_pairs(pairs)
    assert np.array_equal(a[0], d["scores"])
    assert np.array_equal(a[3], ctx.earlyfusion_pairs(pairs[3:])[0])       # the short pair alone: narrow kernels, same scores


def test_errors(ctx):
    from acoss_amd import synth
    tracks = synth.earlyfusion_set(2, seed=1, nb_range=(20, 30))
    ctx.ef_upload_pool(tracks)
    with pytest.raises(ValueError):
        ctx.earlyfusion_pairs(np.array([[0, 2]], np.int32))
    assert ctx.earlyfusion_pairs(np.zeros((0, 2), np.int32)).shape == (0, 4)
    # the multi-pair debug entry: pair `which` of a list that runs as ONE batch; a list that needs two is refused, loudly
    pairs = np.array([[0, 1], [1, 0], [0, 0]], np.int32)
    d = ctx.ef_debug_pairs(pairs, 1)
    assert d["csm"].shape == (3, len(tracks[1]["mfccs"]), len(tracks[0]["mfccs"])) and np.array_equal(d["scores"], ctx.earlyfusion_pairs(pairs))
    with pytest.raises(ValueError):
        ctx.ef_debug_pairs(pairs, 3)
    ctx.set_scratch_limit(4 * 30 * 64 * 4 * 2)             # room for two of the three pairs
    try:
        with pytest.raises(NotImplementedError, match="one batch"):
            ctx.ef_debug_pairs(pairs, 0)
    finally:
        ctx.set_scratch_limit(0)


def test_benchmark_end_to_end(tmp_path, monkeypatch):
    """benchmark(algorithm="EarlyFusionTraile") (the name the reference advertises but never
    dispatches, coverid.py:19 vs :72) over files holding block features: four score matrices,
    mirrored, then SNF late fusion and the evaluation of all six."""
    import acoss_amd
    from acoss_amd import synth
    from acoss_amd.featurestore import save_track
    rng = np.random.default_rng(3)
    labels = ["a", "a", "b", "b", "c"] + ["s%d" % k for k in range(19)]      # SNF uses K = 20 neighbours
    tracks = synth.earlyfusion_set(len(labels), seed=11, nb_range=(24, 40))
    for (u, v) in [(0, 1), (2, 3)]:
        for key in ("mfccs", "ssms", "chromas"):
            n = 20
            tracks[v][key][2:2 + n] = tracks[u][key][1:1 + n] + 0.02 * rng.standard_normal((n, tracks[u][key].shape[1])).astype(np.float32)
    root = str(tmp_path) + "/feat/"
    with open(tmp_path / "ds.csv", "w") as f:
        f.write("work_id,track_id\n")
        for i, l in enumerate(labels):
            d = dict(tracks[i])
            d.update(label=l, track_id="t%d" % i)
            save_track(root + "%s/t%d.h5" % (l, i), d)
            f.write("%s,t%d\n" % (l, i))
    monkeypatch.chdir(tmp_path)
    res = acoss_amd.benchmark(str(tmp_path / "ds.csv"), root, algorithm="EarlyFusionTraile", shortname="toy")
    assert sorted(res.keys()) == ["chromas", "early", "early+late", "late", "mfccs", "ssms"]
    # the planted covers are found by the early-fusion score
    assert res["early"][0] == 1.0 and res["early"][3] == 1.0           # MR, MAP
    Ds = np.load("cache/EarlyFusionTraile_toy_hpcp_Ds.npz")
    assert np.array_equal(Ds["early"], Ds["early"].T) and Ds["early"][0, 1] > 10


def test_engine_options_of_the_class(ctx, tmp_path, monkeypatch):
    """EarlyFusion(engine={"gemm": ..., "fuse": ...}) reaches the library's switches: the class's scores are the
    context's in the same mode; unknown options are refused."""
    from acoss_amd import synth
    from acoss_amd.algorithms.earlyfusion_traile import EarlyFusion
    tracks = synth.earlyfusion_set(6, seed=31, nb_range=(40, 90))
    monkeypatch.chdir(tmp_path)
    with open("ds.csv", "w") as f:
        f.write("work_id,track_id\n")
        for i in range(6):
            f.write("w%d,t%d\n" % (i // 2, i))
    pairs = np.array([[0, 1], [2, 5], [3, 4]], np.int32)
    try:
        for eng in ({"gemm": "bf16x3"}, {"gemm": "f32", "fuse": "exact"}, {}):
            ef = EarlyFusion("ds.csv", "feat/", shortname="toy", engine=eng)
            ef.set_block_features(tracks)
            ctx.set_ef_gemm(eng.get("gemm", "default"))
            ctx.set_ef_fuse(eng.get("fuse", "fast"))
            ctx.ef_upload_pool(tracks)
            want = ctx.earlyfusion_pairs(pairs)
            for k, (i, j) in enumerate(pairs):
                ef.similarity([(int(i), int(j))])
                got = [ef.Ds[s][i, j] for s in ("mfccs", "ssms", "chromas", "early")]
                assert np.array_equal(np.asarray(got, np.float32), want[k]), (eng, i, j)
    finally:
        ctx.set_ef_gemm("default")
        ctx.set_ef_fuse("fast")
    with pytest.raises(ValueError):
        EarlyFusion("ds.csv", "feat/", shortname="toy", engine={"gem": "f32"})


def test_rectangle_gemm_equals_pairwise_gemm(ctx):
    """The dense-rectangle GEMM (ef_gemm_rect_bf16x3_kernel; chroma by f32 MFMAs as in the one-matrix-at-a-time
    kernels: mode 'bf16x3_chroma_f32') against the one-matrix-at-a-time kernels of
    round 2: the arithmetic of a cell is the same sequence of MFMAs, so the cross-similarity matrices -- and with
    them all four scores -- must be BIT-identical, whatever the shape of the pair list: full grid tiles, a
    triangular (diagonal) tile, repeated pairs, self pairs, pairs in both orders, tracks of 1 / 15 / 16 / 17 blocks,
    more than 128 distinct tracks on either side (several rectangles), a track of more than 1024 blocks."""
    from acoss_amd import _lib
    rng = np.random.default_rng(44)
    nbs = [1, 15, 16, 17, 33, 128, 129, 300, 47, 250] + [int(v) for v in rng.integers(20, 90, 150)]

    def track(nb):
        mf = rng.standard_normal((nb, 650)).astype(np.float32)
        mf /= np.linalg.norm(mf, axis=1, keepdims=True)
        return dict(mfccs=mf, ssms=(2 * rng.random((nb, 1225))).astype(np.float32), chromas=rng.random((nb, 480)).astype(np.float32),
                    chroma_med=rng.random(12))
    tracks = [track(nb) for nb in nbs]
    ctx.ef_upload_pool(tracks)
    n = len(tracks)
    lists = []
    ii, jj = np.meshgrid(np.arange(0, 12), np.arange(5, 30), indexing="ij")
    lists.append(np.stack([ii.ravel(), jj.ravel()], 1))                                  # a full rectangle (with i == j cells)
    iu, ju = np.triu_indices(20, 1)
    lists.append(np.stack([iu, ju], 1))                                                   # a diagonal tile
    rnd = rng.integers(0, n, (700, 2))
    lists.append(rnd)                                                                     # arbitrary, > 128 distinct tracks per side
    lists.append(np.array([[3, 4], [3, 4], [4, 3], [7, 7], [3, 4], [0, 1], [1, 0], [0, 0]]))   # repeats, both orders, self pairs
    try:
        for pairs in lists:
            pairs = np.ascontiguousarray(pairs, np.int32)
            ctx.set_ef_gemm("bf16x3_pairwise")
            want = ctx.earlyfusion_pairs(pairs)
            ctx.set_ef_gemm("bf16x3_chroma_f32")
            got = ctx.earlyfusion_pairs(pairs)
            assert np.array_equal(got, want), int(np.sum(np.any(got != want, axis=1)))
        # the matrices themselves, through the debug entry (a one-pair rectangle)
        for (i, j) in [(7, 128 % n), (5, 6), (0, 7), (3, 2)]:
            ctx.set_ef_gemm("bf16x3_pairwise")
            a = ctx.ef_debug_pair(i, j)
            ctx.set_ef_gemm("bf16x3_chroma_f32")
            b = ctx.ef_debug_pair(i, j)
            assert np.array_equal(a["csm"], b["csm"]) and np.array_equal(a["fused"], b["fused"])
        # through the pair grid (tile = 128 tracks: rectangles of 128 x 128 track slots; and a small tile)
        for tile in (0, 7):
            planes = {}
            for mode in ("bf16x3_pairwise", "bf16x3_chroma_f32"):
                ctx.set_ef_gemm(mode)
                planes[mode] = [np.zeros((n, n), np.float32) for _ in range(4)]
                ctx.pair_grid(_lib.ALGO_EARLYFUSION, True, _lib.EfParams(0.1, 10), planes[mode], mirror=True, tile=tile)
            for e in range(4):
                assert np.array_equal(planes["bf16x3_chroma_f32"][e], planes["bf16x3_pairwise"][e])
        # a track of more than 1024 blocks (streaming row statistics / Smith-Waterman behind the same GEMM)
        big = [track(1100), track(40), track(520)]
        ctx.ef_upload_pool(big)
        pr = np.array([[0, 1], [1, 0], [0, 2], [2, 1]], np.int32)
        ctx.set_ef_gemm("bf16x3_pairwise")
        want = ctx.earlyfusion_pairs(pr)
        ctx.set_ef_gemm("bf16x3_chroma_f32")
        assert np.array_equal(ctx.earlyfusion_pairs(pr), want)
    finally:
        ctx.set_ef_gemm("default")


def test_chroma_on_the_bf16_pipe(ctx):
    """The bf16x3 mode's chroma matrix (ef_gemm_rect_bf16x3_kernel<1>: bin-major three-term bf16 splits, the OTI roll as a
    shift of the staged row) against the f32-MFMA kernel and against the f64 truth: same bound for both; scores of
    whole pair lists (several tracks per tile, every roll, tracks shorter than a group, column chunks that end with
    their track) agree except on threshold ties."""
    from acoss_amd import _lib
    rng = np.random.default_rng(45)
    nbs = [1, 15, 16, 17, 33, 128, 129, 300, 47, 250] + [int(v) for v in rng.integers(20, 90, 60)]

    def track(nb):
        mf = rng.standard_normal((nb, 650)).astype(np.float32)
        ch = rng.random((nb, 480)).astype(np.float32) ** 3
        return dict(mfccs=mf, ssms=(2 * rng.random((nb, 1225))).astype(np.float32), chromas=ch, chroma_med=rng.random(12) ** 2)
    tracks = [track(nb) for nb in nbs]
    ctx.ef_upload_pool(tracks)
    n = len(tracks)
    try:
        otis = set()
        for (i, j) in [(7, 8), (5, 6), (0, 7), (3, 2), (9, 7), (8, 9), (20, 30), (31, 21), (5, 5), (40, 9)]:
            ctx.set_ef_gemm("bf16x3_chroma_f32")
            a = ctx.ef_debug_pair(i, j)
            ctx.set_ef_gemm("bf16x3")
            b = ctx.ef_debug_pair(i, j)
            assert a["oti"] == b["oti"]
            otis.add(int(a["oti"]))
            assert np.array_equal(a["csm"][0], b["csm"][0]) and np.array_equal(a["csm"][1], b["csm"][1])
            X = tracks[i]["chromas"].astype(np.float64)
            Y = tracks[j]["chromas"].astype(np.float64)
            X = np.roll(X.reshape(len(X), 40, 12), a["oti"], axis=2).reshape(len(X), 480)
            X /= np.linalg.norm(X, axis=1, keepdims=True)
            Y /= np.linalg.norm(Y, axis=1, keepdims=True)
            truth = 1.0 - X @ Y.T
            ea = float(np.max(np.abs(a["csm"][2] - truth))), float(np.max(np.abs(b["csm"][2] - truth)))
            assert ea[0] <= 2e-6 and ea[1] <= 2e-6 and ea[1] <= 2 * ea[0] + 2e-7, ea       # (f32 unit rows, 480-term f32 sums)
            assert float(np.max(np.abs(a["csm"][2] - b["csm"][2]))) <= 2e-6
        assert len(otis) >= 5, otis                               # several different rolls were exercised
        ii, jj = np.meshgrid(np.arange(0, 14), np.arange(5, 40), indexing="ij")
        iu, ju = np.triu_indices(30, 1)
        lists = [np.stack([ii.ravel(), jj.ravel()], 1), np.stack([iu, ju], 1), rng.integers(0, n, (500, 2)),
                 np.array([[3, 4], [3, 4], [4, 3], [7, 7], [0, 1], [1, 0], [0, 0]])]
        for pairs in lists:
            pairs = np.ascontiguousarray(pairs, np.int32)
            ctx.set_ef_gemm("bf16x3_chroma_f32")
            want = ctx.earlyfusion_pairs(pairs)
            ctx.set_ef_gemm("bf16x3")
            got = ctx.earlyfusion_pairs(pairs)
            assert np.array_equal(got[:, :2], want[:, :2])                                    # mfccs, ssms: the same kernel
            same = np.all(got == want, axis=1)
            assert same.mean() >= 0.97, same.mean()
            assert np.max(np.abs(got - want)) <= 3.0
        planes = {}
        for mode in ("bf16x3_chroma_f32", "bf16x3"):
            ctx.set_ef_gemm(mode)
            planes[mode] = [np.zeros((n, n), np.float32) for _ in range(4)]
            ctx.pair_grid(_lib.ALGO_EARLYFUSION, True, _lib.EfParams(0.1, 10), planes[mode], mirror=True, tile=0)
        for e in range(4):
            same = planes["bf16x3"][e] == planes["bf16x3_chroma_f32"][e]
            assert same.mean() >= (1.0 if e < 2 else 0.97), (e, same.mean())
        # blocks of a frame count the bin-major split does not cover (12 x 36 values): the f32 kernel takes over
        odd = [dict(mfccs=t["mfccs"][:30], ssms=t["ssms"][:30], chromas=t["chromas"][:30, :432], chroma_med=t["chroma_med"]) for t in tracks[10:14]]
        ctx.ef_upload_pool(odd)
        pr = np.array([[0, 1], [2, 3], [1, 3]], np.int32)
        ctx.set_ef_gemm("bf16x3_pairwise")
        want = ctx.earlyfusion_pairs(pr)
        ctx.set_ef_gemm("bf16x3")
        assert np.array_equal(ctx.earlyfusion_pairs(pr), want)
    finally:
        ctx.set_ef_gemm("default")


def test_f16x2_gemm_mode(ctx):
    """ACX_EF_GEMM_F16X2 (the default): two fp16 terms per value, three MFMAs per cell.  Matrices against the f64 truth
    under the bounds the other modes are held to; scores of pair lists against bf16x3's (threshold ties may move); the
    same scores after the whole pool is scaled by 2^10 and 2^-10 (every row carries its own power-of-two scale); each
    mode's bits return when the mode is switched back and forth (the pool is re-split both ways)."""
    rng = np.random.default_rng(46)
    nbs = [1, 15, 16, 17, 33, 128, 129, 300, 47, 250] + [int(v) for v in rng.integers(20, 90, 30)]

    def track(nb):
        mf = rng.standard_normal((nb, 650)).astype(np.float32)
        mf[:, :3] *= 30.0                                           # a few large coordinates, many small ones
        ch = rng.random((nb, 480)).astype(np.float32) ** 3
        return dict(mfccs=mf, ssms=(2 * rng.random((nb, 1225))).astype(np.float32), chromas=ch, chroma_med=rng.random(12) ** 2)
    tracks = [track(nb) for nb in nbs]
    n = len(tracks)
    iu, ju = np.triu_indices(24, 1)
    lists = [np.ascontiguousarray(np.stack([iu, ju], 1), np.int32), np.ascontiguousarray(rng.integers(0, n, (300, 2)), np.int32)]
    try:
        ctx.ef_upload_pool(tracks)
        ctx.set_ef_gemm("bf16x3")
        want = [ctx.earlyfusion_pairs(pr) for pr in lists]
        ctx.set_ef_gemm("f16x2")
        worst = [0.0, 0.0, 0.0]
        for (i, j) in [(7, 8), (5, 6), (0, 7), (3, 2), (9, 7), (8, 9), (20, 30), (31, 21), (5, 5)]:
            d = ctx.ef_debug_pair(i, j)
            for k, s in enumerate(("mfccs", "ssms")):
                x64, y64 = tracks[i][s].astype(np.float64), tracks[j][s].astype(np.float64)
                scale = float(np.sum(x64 ** 2, 1).max() + np.sum(y64 ** 2, 1).max())
                true2 = np.maximum(0, np.sum(x64 ** 2, 1)[:, None] + np.sum(y64 ** 2, 1)[None, :] - 2 * x64.dot(y64.T))
                err = float(np.max(np.abs(d["csm"][k].astype(np.float64) ** 2 - true2))) / scale
                worst[k] = max(worst[k], err)
                assert err <= 4e-6, (s, err)
            X = tracks[i]["chromas"].astype(np.float64)
            Y = tracks[j]["chromas"].astype(np.float64)
            X = np.roll(X.reshape(len(X), 40, 12), d["oti"], axis=2).reshape(len(X), 480)
            X /= np.linalg.norm(X, axis=1, keepdims=True)
            Y /= np.linalg.norm(Y, axis=1, keepdims=True)
            e2 = float(np.max(np.abs(d["csm"][2] - (1.0 - X @ Y.T))))
            worst[2] = max(worst[2], e2)
            assert e2 <= 2e-6, e2
        print("f16x2 worst errors (d^2 / scale mfccs, ssms; chroma abs):", worst)
        got = [ctx.earlyfusion_pairs(pr) for pr in lists]
        for g, w in zip(got, want):
            same = np.all(g == w, axis=1)
            print("f16x2 vs bf16x3: pairs with all four scores identical: %.4f of %d, max |d| %.1f" % (same.mean(), len(g), np.max(np.abs(g - w))))
            assert same.mean() >= 0.99, same.mean()
            assert np.max(np.abs(g - w)) <= 3.0
        # the pool scaled by powers of two: distances scale exactly, thresholds are ranks -> the mode's own scores again
        # (mfccs / ssms planes; chroma rows are normalised anyway)
        for f in (1024.0, 1.0 / 1024.0):
            scaled = [dict(mfccs=t["mfccs"] * np.float32(f), ssms=t["ssms"] * np.float32(f), chromas=t["chromas"], chroma_med=t["chroma_med"])
                      for t in tracks]
            ctx.ef_upload_pool(scaled)
            g2 = ctx.earlyfusion_pairs(lists[0])
            assert np.array_equal(g2[:, :3], got[0][:, :3]), f
        ctx.ef_upload_pool(tracks)
        assert np.array_equal(ctx.earlyfusion_pairs(lists[0]), got[0])
        ctx.set_ef_gemm("bf16x3")
        assert np.array_equal(ctx.earlyfusion_pairs(lists[0]), want[0])
        ctx.set_ef_gemm("bf16x3_pairwise")                          # (needs the bf16 pool as well)
        pw = ctx.earlyfusion_pairs(lists[0])
        assert np.array_equal(pw[:, :2], want[0][:, :2])
    finally:
        ctx.set_ef_gemm("default")


def test_every_gemm_instantiation_per_cell_in_multi_pair_rectangles(ctx):
    """Every instantiation of ef_gemm_rect_bf16x3_kernel<CH, F16> (mfcc / ssm and chroma, three bf16 terms and two fp16 terms)
    on rectangles that hold MANY pairs -- the product path's tiles: operands shared between the pairs of a track, sub-tiles
    dropped into their own pair's matrix, track ends inside a workgroup tile -- cell by cell against f64 products, and bit for bit
    against the one-pair rectangle of acx_ef_debug_pair.  (The fp16 / bf16 K = 32 MFMA returns garbage when its destination
    registers overlap a dying source operand, tests/test_isa_lint.py: whether the compiler allocates that way depends on the
    instantiation and on what surrounds the k loop, so each one is checked through the launch the product takes.)"""
    rng = np.random.default_rng(61)
    nbs = [1, 15, 16, 17, 33, 128, 129, 300, 47, 250, 64, 96] + [int(v) for v in rng.integers(20, 140, 12)]

    def track(nb):
        mf = rng.standard_normal((nb, 650)).astype(np.float32)
        mf[:, :3] *= 30.0
        ch = rng.random((nb, 480)).astype(np.float32) ** 3
        return dict(mfccs=mf, ssms=(2 * rng.random((nb, 1225))).astype(np.float32), chromas=ch, chroma_med=rng.random(12) ** 2)
    tracks = [track(nb) for nb in nbs]
    n = len(tracks)
    iu, ju = np.triu_indices(n, 1)
    pairs = np.ascontiguousarray(np.stack([iu, ju], 1), np.int32)                    # 276 pairs: one rectangle of 24 x 24 tracks
    pairs = np.concatenate([pairs, pairs[::7, ::-1]]).astype(np.int32)              # + some in the other orientation: a second rectangle
    index_of = {(int(i), int(j)): k for k, (i, j) in enumerate(pairs)}
    probes = [(0, 1), (1, 2), (2, 3), (3, 4), (5, 6), (6, 7), (7, 9), (8, 9), (4, 10), (10, 11), (0, 7), (12, 20), (15, 23), (7, 0)]
    probes = [pq for pq in probes if pq in index_of]
    assert len(probes) >= 13
    try:
        ctx.ef_upload_pool(tracks)
        for mode in ("bf16x3", "f16x2"):
            ctx.set_ef_gemm(mode)
            listed = ctx.earlyfusion_pairs(pairs)
            for (i, j) in probes:
                d = ctx.ef_debug_pairs(pairs, index_of[(i, j)])
                one = ctx.ef_debug_pair(i, j)
                assert np.array_equal(d["scores"], listed), (mode, i, j)
                assert d["oti"] == one["oti"]
                assert np.array_equal(d["csm"], one["csm"]), (mode, i, j, "multi-pair rectangle != one-pair rectangle")
                for k, s_ in enumerate(("mfccs", "ssms")):
                    x64, y64 = tracks[i][s_].astype(np.float64), tracks[j][s_].astype(np.float64)
                    scale = float(np.sum(x64 ** 2, 1).max() + np.sum(y64 ** 2, 1).max())
                    true2 = np.maximum(0, np.sum(x64 ** 2, 1)[:, None] + np.sum(y64 ** 2, 1)[None, :] - 2 * x64.dot(y64.T))
                    err = float(np.max(np.abs(d["csm"][k].astype(np.float64) ** 2 - true2))) / scale
                    assert err <= 4e-6, (mode, s_, i, j, err)
                X = tracks[i]["chromas"].astype(np.float64)
                Y = tracks[j]["chromas"].astype(np.float64)
                X = np.roll(X.reshape(len(X), 40, 12), d["oti"], axis=2).reshape(len(X), 480)
                X /= np.linalg.norm(X, axis=1, keepdims=True)
                Y /= np.linalg.norm(Y, axis=1, keepdims=True)
                e2 = float(np.max(np.abs(d["csm"][2] - (1.0 - X @ Y.T))))
                assert e2 <= 2e-6, (mode, "chromas", i, j, e2)
    finally:
        ctx.set_ef_gemm("default")


def test_short_k_loops_of_small_feature_dims(ctx):
    """Feature dimensions far below the reference's 650 / 1225 / 480: k loops of one to four 32-k chunks.  The persistent GEMM
    (ef_gemm_persist_kernels.hpp) resolves its next tile behind the first four chunks of a k loop of at least seven; shorter loops take
    the path that resolves it behind the loop -- no DA-TACOS-shaped set ever runs it.  Cells against f64 products, both 16-bit
    arithmetics, many pairs per rectangle, several tiles per workgroup."""
    rng = np.random.default_rng(77)
    for dims in ((40, 100, 96), (20, 32, 96), (200, 230, 192)):
        G = dims[2] // 12
        nbs = [int(v) for v in rng.integers(20, 300, 14)] + [1, 16, 17]

        def track(nb):
            return dict(mfccs=rng.standard_normal((nb, dims[0])).astype(np.float32), ssms=(2 * rng.random((nb, dims[1]))).astype(np.float32),
                        chromas=(rng.random((nb, dims[2])).astype(np.float32) ** 3 + 1e-3), chroma_med=rng.random(12) ** 2)
        tracks = [track(nb) for nb in nbs]
        n = len(tracks)
        iu, ju = np.triu_indices(n, 1)
        pairs = np.ascontiguousarray(np.stack([iu, ju], 1), np.int32)
        try:
            ctx.ef_upload_pool(tracks)
            ref = None
            for mode in ("f16x2", "bf16x3", "f32"):
                ctx.set_ef_gemm(mode)
                listed = ctx.earlyfusion_pairs(pairs)
                if ref is None:
                    ref = listed
                else:                                           # (threshold ties may move a score between arithmetics)
                    assert np.mean(np.all(listed == ref, axis=1)) >= 0.97 and np.max(np.abs(listed - ref)) <= 3.0, (dims, mode)
                if mode == "f32":
                    continue
                for k in (0, 5, n + 3, len(pairs) - 1, len(pairs) // 2):
                    i, j = (int(v) for v in pairs[k])
                    d = ctx.ef_debug_pairs(pairs, k)
                    assert np.array_equal(d["scores"], listed)
                    for e, s_ in enumerate(("mfccs", "ssms")):
                        x64, y64 = tracks[i][s_].astype(np.float64), tracks[j][s_].astype(np.float64)
                        scale = float(np.sum(x64 ** 2, 1).max() + np.sum(y64 ** 2, 1).max())
                        true2 = np.maximum(0, np.sum(x64 ** 2, 1)[:, None] + np.sum(y64 ** 2, 1)[None, :] - 2 * x64.dot(y64.T))
                        err = float(np.max(np.abs(d["csm"][e].astype(np.float64) ** 2 - true2))) / scale
                        assert err <= 4e-6, (dims, mode, s_, i, j, err)
                    X = tracks[i]["chromas"].astype(np.float64)
                    Y = tracks[j]["chromas"].astype(np.float64)
                    X = np.roll(X.reshape(len(X), G, 12), d["oti"], axis=2).reshape(len(X), dims[2])
                    X /= np.linalg.norm(X, axis=1, keepdims=True)
                    Y /= np.linalg.norm(Y, axis=1, keepdims=True)
                    e2 = float(np.max(np.abs(d["csm"][2] - (1.0 - X @ Y.T))))
                    assert e2 <= 2e-6, (dims, mode, "chromas", i, j, e2)
        finally:
            ctx.set_ef_gemm("default")


def test_batching_is_invisible_on_mixed_row_lengths(ctx):
    """which row-statistics kernel a batch takes is decided by its LONGEST row (<= 512 cells: two rows per wave;
    beyond: the one-row kernels for every pair of the batch) -- a pair's scores must not depend on what shares its batch.  A pool
    that mixes tracks of <= 512 and > 512 blocks: the short pairs alone, the same pairs in a list with long ones, one pair per
    batch, and the grid."""
    from acoss_amd import synth, _lib
    tracks = synth.earlyfusion_set(10, seed=31, nb_range=(200, 500)) + synth.earlyfusion_set(2, seed=32, nb_range=(520, 700))
    rng = np.random.default_rng(8)
    for key in ("mfccs", "ssms", "chromas"):                     # shared structure: alignments, thresholds that matter
        for a, b in ((0, 1), (2, 10), (3, 4)):
            m = 120
            tracks[b][key][40:40 + m] = tracks[a][key][60:60 + m] + 0.02 * rng.standard_normal((m, tracks[a][key].shape[1])).astype(np.float32)
    ctx.ef_upload_pool(tracks)
    n = len(tracks)
    iu, ju = np.triu_indices(n, 1)
    allp = np.ascontiguousarray(np.stack([iu, ju], 1), np.int32)
    short = allp[(allp[:, 0] < 10) & (allp[:, 1] < 10)]
    alone = ctx.earlyfusion_pairs(short)                         # every row <= 512: the two-row kernel
    mixed = ctx.earlyfusion_pairs(allp)                          # one batch with rows of > 512 cells: the one-row kernels
    sel = (allp[:, 0] < 10) & (allp[:, 1] < 10)
    assert np.array_equal(mixed[sel], alone)
    ctx.set_scratch_limit(4 * 700 * 704 * 4 * 2)                 # a pair or two per batch
    try:
        assert np.array_equal(ctx.earlyfusion_pairs(allp), mixed)
    finally:
        ctx.set_scratch_limit(0)
    planes = [np.zeros((n, n), np.float32) for _ in range(4)]
    ctx.pair_grid(_lib.ALGO_EARLYFUSION, True, _lib.EfParams(0.1, 10), planes, mirror=False)
    for e in range(4):
        assert np.array_equal(planes[e][allp[:, 0], allp[:, 1]], mixed[:, e]), e


def test_epilogue_sqrt(ctx):
    """The sqrt of the distance epilogues (ef_sqrt_nonneg: v_sqrt_f32 + one Newton step on the fma residual) against
    numpy's correctly rounded one: never more than 1 ulp apart, identical on all but a sliver of the inputs; 0, tiny and
    huge values pass through."""
    rng = np.random.default_rng(77)
    x = np.concatenate([rng.random(2_000_000).astype(np.float32) * 4000.0,                    # distances^2 of z-scored 650-dim rows
                        np.exp(rng.uniform(-40, 40, 2_000_000)).astype(np.float32),
                        np.arange(0, 70000, dtype=np.float32),
                        np.array([0.0, 1e-45, 1e-38, 1.1754944e-38, 1e-30, 3.4e38, np.inf], np.float32)])
    got = ctx.debug_sqrt(x, ef=True)
    want = np.sqrt(x)
    assert np.all(got[np.isinf(want)] == np.inf) and got[x == 0].max() == 0.0
    fin = np.isfinite(want) & (x >= 1.1754944e-38)
    ulp = np.abs(got[fin].view(np.int32).astype(np.int64) - want[fin].view(np.int32).astype(np.int64))
    assert ulp.max() <= 1, ulp.max()
    frac = float(np.mean(ulp == 0))
    print("ef_sqrt_nonneg: identical to the correctly rounded sqrt on %.6f of %d inputs" % (frac, int(fin.sum())))
    assert frac >= 0.999, frac
    den = (x > 0) & (x < 1.1754944e-38)                      # denormal inputs: the raw instruction's answer (flushed or not), never NaN
    assert not np.any(np.isnan(got[den]))


def test_neighbourhood_sizes(ctx):
    """K (the neighbourhood of getWCSM's kernel widths) up to 16: the column means come from C itself
    (ef_colstat_kernel); beyond: the transposed matrices and the row-selection kernel.  Both against the oracle's
    getWCSM on the device's own matrices, for ragged pairs (the reference's np.partition needs more than K blocks per
    track: the 7-block track only meets the small K)."""
    import oracle
    from acoss_amd import synth
    tracks = synth.earlyfusion_set(5, seed=8, nb_range=(5, 150))
    tracks[4] = {k: (v[:7] if hasattr(v, "shape") and v.ndim == 2 else v) for k, v in tracks[4].items()}      # 7 blocks
    ctx.ef_upload_pool(tracks)
    for K in (1, 3, 10, 11, 16, 17, 25):
        for (i, j) in [(0, 1), (1, 0), (2, 3), (4, 0), (0, 4)]:
            if min(len(tracks[i]["mfccs"]), len(tracks[j]["mfccs"])) <= K:
                continue
            d = ctx.ef_debug_pair(i, j, K=K)
            ws = np.zeros_like(d["csm"][0])
            for k in range(3):
                ws += oracle.get_wcsm(d["csm"][k], K, K)
            np.testing.assert_allclose(d["fused"], np.exp(-ws), rtol=2e-4, atol=1e-6, err_msg="K=%d pair %s" % (K, (i, j)))
            assert round(oracle.sw_constrained(oracle.csm_to_binary(d["fused"], 0.1)) * 10) == round(float(d["scores"][3]) * 10)


def test_fuzz_small(ctx):
    """A few rounds of tests/fuzz_earlyfusion.py: ragged pools, odd feature widths, chroma blocks of 8 / 16 / 40 frames,
    random kappa / K and pair lists -- default arithmetic against the exact-f32 kernels, the oracle and the pair grid."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("fuzz_earlyfusion", os.path.join(os.path.dirname(os.path.abspath(__file__)), "fuzz_earlyfusion.py"))
    fuzz_earlyfusion = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fuzz_earlyfusion)
    assert fuzz_earlyfusion.run(rounds=12, seed=5, ctx=ctx) > 0


def test_pool_slices_must_cover_every_track(ctx):
    """acx_ef_pool_end refuses a pool some track of which was never handed over (the arrays come from hipMalloc: what
    acx_ef_pool_tracks did not write is garbage), names the first missing track and leaves the pool open; the missing
    slice supplied afterwards -- in any order -- gives the pool a whole upload gives."""
    from acoss_amd import _lib, synth
    tracks = synth.earlyfusion_set(5, seed=11, nb_range=(20, 40))
    pairs = np.array([[0, 2], [1, 3], [2, 4], [3, 4]], np.int32)
    ctx.ef_upload_pool(tracks)
    want = ctx.earlyfusion_pairs(pairs)
    nb = [t["mfccs"].shape[0] for t in tracks]

    def hand_over(a, b):
        part = tracks[a:b]
        ctx.ef_pool_tracks(a, b - a, *[np.concatenate([t[k] for t in part]) for k in ("mfccs", "ssms", "chromas")],
                           np.stack([np.asarray(t["chroma_med"], np.float64) for t in part]))
    ctx.ef_pool_begin(nb)
    hand_over(3, 5)
    hand_over(0, 2)
    with pytest.raises(_lib.AcxError, match="first: track 2"):
        ctx.ef_pool_end()
    with pytest.raises(_lib.AcxError):                 # no usable pool in between
        ctx.earlyfusion_pairs(pairs)
    hand_over(2, 3)
    ctx.ef_pool_end()
    assert np.array_equal(ctx.earlyfusion_pairs(pairs), want)


def test_exact_fuse_mode(ctx, golden):
    """acx_set_ef_fuse(EXACT): getWCSM's weights and the fused matrix in the reference's own operation order (IEEE divisions,
    expf) instead of reciprocal + exp2.  The fused matrix then sits closer to numpy's on the DEVICE's cross-similarity
    matrices (what is left is the summation order of the neighbourhood means, which numpy's introselect leaves arbitrary, and
    the last bit of the exponentials); the alignment of the device's fused matrix by the oracle equals the device's score in
    both modes; and the two modes agree on the `early` score of (nearly) every pair -- the approximation only moves ties."""
    import oracle
    from acoss_amd import synth
    g = golden("ef_chain")
    worst = {}
    for mode in ("fast", "exact"):
        ctx.set_ef_fuse(mode)
        try:
            w = 0.0
            for pk in (0, 1):
                f1, f2 = _golden_feats(g, pk)
                ctx.ef_upload_pool([f1, f2])
                d = ctx.ef_debug_pair(0, 1)
                ws = np.zeros_like(d["csm"][0])
                for k in range(3):
                    ws += oracle.get_wcsm(d["csm"][k], 10, 10)
                ref = np.exp(-ws)
                w = max(w, float(np.max(np.abs(d["fused"] - ref) / np.maximum(ref, 1e-30))))
                assert round(oracle.sw_constrained(oracle.csm_to_binary(d["fused"], 0.1)) * 10) == round(float(d["scores"][3]) * 10)
                assert abs(float(d["scores"][3]) - float(g["p%d_scores" % pk][3])) <= 2.0
            worst[mode] = w
        finally:
            ctx.set_ef_fuse("fast")
    assert worst["exact"] <= 5e-5 and worst["fast"] <= 2e-4, worst
    tracks = synth.earlyfusion_set(24, seed=9, nb_range=(60, 140))
    ctx.ef_upload_pool(tracks)
    pairs = oracle.all_pairs(24, True).astype(np.int32)
    fast = ctx.earlyfusion_pairs(pairs)
    ctx.set_ef_fuse("exact")
    try:
        exact = ctx.earlyfusion_pairs(pairs)
    finally:
        ctx.set_ef_fuse("fast")
    assert np.array_equal(fast[:, :3], exact[:, :3])                     # the three feature planes do not depend on the mode
    same = np.mean(fast[:, 3] == exact[:, 3])
    assert same >= 0.97 and np.max(np.abs(fast[:, 3] - exact[:, 3])) <= 2.0, (same, np.max(np.abs(fast[:, 3] - exact[:, 3])))
    with pytest.raises(ValueError):
        ctx.set_ef_fuse(7)


_VARIANT_SNIPPET = r'''
import sys, numpy as np
sys.path.insert(0, %(root)r)
from acoss_amd import _lib, synth
ctx = _lib.Context(0)
tracks = synth.earlyfusion_set(40, seed=5, nb_range=(60, 330)) + synth.earlyfusion_set(2, seed=6, nb_range=(1, 17))
ctx.ef_upload_pool(tracks)
ctx.set_ef_gemm(%(mode)r)
n = len(tracks)
iu, ju = np.triu_indices(n, 1)
pairs = np.ascontiguousarray(np.stack([iu, ju], 1), np.int32)
rng = np.random.default_rng(1)
extra = rng.integers(0, n, (500, 2)).astype(np.int32)
out = np.concatenate([ctx.earlyfusion_pairs(pairs), ctx.earlyfusion_pairs(extra)])
np.save(%(out)r, out)
ctx.close()
'''


@pytest.mark.timeout(900)
def test_gemm_kernel_variants_give_the_same_bits(tmp_path):
    """The rectangle GEMMs exist as several kernels per arithmetic -- one workgroup per tile or a persistent workgroup per CU, operands through
    the staging registers or by LDS-DMA, and (fp16 arithmetic, an experiment) wave-specialised workgroups on 128 x 128 tiles -- chosen per
    process (ACX_EF_PERSIST, ACX_EF_DMA, ACX_EF_WS).  Same
    tiles, same MFMAs in the same order: every variant must return the default's scores bit for bit, on a pair list that fills whole
    grid rectangles, on random pairs (sparse rectangles, one pair per tile) and with tracks of 1-17 blocks (rims everywhere)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    results = {}
    for mode, variants in (("f16x2", ({}, {"ACX_EF_DMA": "0"}, {"ACX_EF_PERSIST": "0"}, {"ACX_EF_PERSIST": "0", "ACX_EF_DMA": "0"}, {"ACX_EF_WS": "1"})),
                           ("bf16x3", ({}, {"ACX_EF_PERSIST": "1"}, {"ACX_EF_DMA": "1"}))):
        for k, extra in enumerate(variants):
            out = str(tmp_path / ("%s_%d.npy" % (mode, k)))
            env = dict(os.environ)
            env.pop("ACX_EF_DMA", None)
            env.pop("ACX_EF_PERSIST", None)
            env.pop("ACX_EF_WS", None)
            env.update(extra)
            subprocess.check_call([sys.executable, "-c", _VARIANT_SNIPPET % {"root": root, "mode": mode, "out": out}], env=env, timeout=300)
            results[(mode, k)] = np.load(out)
        for k in range(1, len(variants)):
            assert np.array_equal(results[(mode, k)], results[(mode, 0)]), (mode, variants[k])
    assert results[("f16x2", 0)].shape == results[("bf16x3", 0)].shape
