"""
GPU tests (-m gpu): non-finite feature values at upload.

The reference zeroes NaN MFCCs itself (earlyfusion_traile.py:105) and hands every other feature to
essentia / numpy unchecked.  libacx scans every pool on its way in (acx_set_nonfinite_policy): by
default an upload with a NaN / Inf value FAILS and names the track; under ACX_NONFINITE_ZERO the
values are replaced by 0 on the device.  What must never happen is the thing round 2's kernels did:
a NaN frame in track k changing the score of a pair that does not involve k (the band kernel reads
a neighbouring track's frames for the cells beyond a matrix's edge).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _pool(n=12, seed=3):
    from acoss_amd import synth
    # lengths that leave every matrix with ragged edge tiles, so that edge tiles do read the neighbours
    rng = np.random.default_rng(seed)
    tracks = [synth._frame_max_normalise(rng.random((int(T), 12))) for T in rng.integers(70, 260, n)]
    return synth.pack(tracks)


def test_rejects_and_names_the_track():
    from acoss_amd import _lib
    frames, offsets = _pool()
    ctx = _lib.Context(0)
    try:
        for bad, k in ((np.nan, 5), (np.inf, 0), (-np.inf, 11)):
            f = frames.copy()
            f[offsets[k] + 7, 3] = bad
            with pytest.raises(ValueError, match=r"track %d holds a non-finite" % k):
                ctx.upload_pool(f, offsets)
            # no pool is left behind
            with pytest.raises(_lib.AcxError, match="not uploaded"):
                ctx.serra09_pairs(np.array([[0, 1]], np.int32))
        f = frames.copy()
        f[offsets[9], 0] = np.nan
        f[offsets[4] + 1, 1] = np.nan
        with pytest.raises(ValueError, match=r"track 4 holds"):          # the FIRST offending track
            ctx.upload_pool(f, offsets)
        ctx.upload_pool(frames, offsets)                                  # a clean pool still goes in
        assert ctx.nonfinite_zeroed() == 0
        # the f64 pool (SiMPle) and the raw pools scan too
        with pytest.raises(ValueError, match=r"track 2 holds"):
            g = frames.astype(np.float64)
            g[offsets[2] + 3, 11] = np.inf
            ctx.upload_pool_f64(g, offsets)
        raw = np.repeat(frames, 3, axis=0)
        raw[3 * offsets[6] + 2, 5] = np.nan
        with pytest.raises(ValueError, match=r"track 6 holds"):
            ctx.upload_raw_pool(raw, 3 * offsets, fac=3)
        with pytest.raises(ValueError, match=r"track 6 holds"):
            ctx.simple_upload_raw_pool(raw, 3 * offsets, win=6, skip=3)
    finally:
        ctx.close()


@pytest.mark.parametrize("bad", [np.nan, np.inf])
def test_zero_policy_isolates_the_track(bad):
    """Policy ZERO: a non-finite frame planted in track k (first, last and a middle frame: the ones edge
    tiles of the neighbours read) -- every pair that does not involve k scores bit for bit what the
    clean pool scores, pairs with k score what the pool with those values set to 0 scores, and the
    oracle agrees on all of them."""
    import oracle
    from acoss_amd import _lib
    frames, offsets = _pool()
    n = len(offsets) - 1
    pairs = oracle.all_pairs(n, True).astype(np.int32)
    ctx = _lib.Context(0)
    try:
        ctx.upload_pool(frames, offsets)
        clean = ctx.serra09_pairs(pairs)
        for k in (0, 5, n - 1):
            f = frames.copy()
            T = int(offsets[k + 1] - offsets[k])
            where = [(0, 2), (T - 1, 7), (T // 2, 0), (T // 2, 11)]
            for t, b in where:
                f[offsets[k] + t, b] = bad
            ctx.set_nonfinite_policy("zero")
            ctx.upload_pool(f, offsets)
            assert ctx.nonfinite_zeroed() == len(where)
            got = ctx.serra09_pairs(pairs)
            touches = (pairs[:, 0] == k) | (pairs[:, 1] == k)
            assert np.array_equal(got[~touches], clean[~touches]), "track %d's values leaked into other pairs" % k
            z = f.copy()
            z[~np.isfinite(z)] = 0.0
            assert np.array_equal(got, oracle.serra09_pairs(z, offsets, pairs))
            ctx.set_nonfinite_policy("raise")
            with pytest.raises(ValueError, match=r"track %d holds" % k):
                ctx.upload_pool(f, offsets)
    finally:
        ctx.close()


def test_zero_policy_other_pools():
    """SiMPle (f64 pool) and EarlyFusion (block-feature pool): zeroed values behave exactly like zeros."""
    import oracle
    from acoss_amd import _lib
    rng = np.random.default_rng(11)
    ctx = _lib.Context(0, nonfinite="zero")
    try:
        # SiMPle
        lens = rng.integers(40, 90, 6)
        offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        F = rng.random((int(offs[-1]), 12))
        F /= np.linalg.norm(F, axis=1, keepdims=True)
        B = F.copy()
        B[offs[3] + 5, 4] = np.nan
        B[offs[3] + 20, 0] = np.inf
        Z = B.copy()
        Z[~np.isfinite(Z)] = 0.0
        pairs = np.array([(i, j) for i in range(6) for j in range(6) if i != j], np.int32)
        ctx.upload_pool_f64(Z, offs)
        want = ctx.simple_pairs(pairs)
        ctx.upload_pool_f64(B, offs)
        assert ctx.nonfinite_zeroed() == 2
        assert np.array_equal(ctx.simple_pairs(pairs), want)
        # EarlyFusion: block features with a NaN block row; the host-side chroma median too
        def track(nb):
            return dict(mfccs=rng.standard_normal((nb, 650)).astype(np.float32), ssms=(2 * rng.random((nb, 1225))).astype(np.float32),
                        chromas=rng.random((nb, 480)).astype(np.float32), chroma_med=rng.random(12))
        tr = [track(int(nb)) for nb in rng.integers(30, 60, 4)]
        bad = [dict((k, np.array(v, copy=True)) for k, v in t.items()) for t in tr]
        bad[2]["ssms"][7, 100] = np.nan
        bad[1]["chroma_med"][3] = np.inf
        zer = [dict((k, np.nan_to_num(v, nan=0.0, posinf=0.0, neginf=0.0)) for k, v in t.items()) for t in bad]
        ep = np.array([(0, 1), (0, 2), (0, 3), (1, 2), (1, 3), (2, 3)], np.int32)
        ctx.ef_upload_pool(zer)
        want = ctx.earlyfusion_pairs(ep)
        ctx.ef_upload_pool(bad)
        assert ctx.nonfinite_zeroed() == 2
        assert np.array_equal(ctx.earlyfusion_pairs(ep), want)
        ctx.set_nonfinite_policy("raise")
        with pytest.raises(ValueError, match=r"track 1 holds"):
            ctx.ef_upload_pool(bad)
    finally:
        ctx.close()


def test_raw_earlyfusion_features():
    """EarlyFusion's raw per-track features: NaN MFCC samples are zeroed under EITHER policy, as the reference does
    (earlyfusion_traile.py:105); an infinite MFCC sample or a NaN chroma frame follows the policy."""
    import oracle
    from acoss_amd import _lib
    rng = np.random.default_rng(5)
    T = 900
    chroma = rng.random((T, 12)).astype(np.float32)
    mfcc = rng.standard_normal((T, 13)).astype(np.float32)
    on = np.sort(rng.choice(T - 10, 40, replace=False)).astype(np.int64)
    ctx = _lib.Context(0)
    try:
        m_nan = mfcc.copy()
        m_nan[300, 4] = np.nan
        got = ctx.ef_block_features(chroma, m_nan, on)
        want = oracle.ef_block_features(chroma, m_nan, on)               # (the oracle zeroes NaN like the reference)
        np.testing.assert_allclose(got["mfccs"], want["mfccs"], atol=2e-6)
        np.testing.assert_allclose(got["chromas"], want["chromas"], atol=2e-6)
        m_inf = mfcc.copy()
        m_inf[300, 4] = np.inf
        with pytest.raises(ValueError, match=r"track 0 holds a non-finite value .* MFCCs"):
            ctx.ef_block_features(chroma, m_inf, on)
        c_nan = chroma.copy()
        c_nan[17, 3] = np.nan
        with pytest.raises(ValueError, match=r"track 0 holds a non-finite value .* raw chroma"):
            ctx.ef_block_features(c_nan, mfcc, on)
        tracks = [dict(chroma=chroma, mfcc=mfcc, onsets=on), dict(chroma=c_nan, mfcc=m_inf, onsets=on)]
        with pytest.raises(ValueError, match=r"track 1 holds"):
            ctx.ef_upload_raw_pool(tracks)
        ctx.set_nonfinite_policy("zero")
        ctx.ef_upload_raw_pool(tracks)
        assert ctx.nonfinite_zeroed() == 2
        z = dict(chroma=np.nan_to_num(c_nan, nan=0.0), mfcc=np.nan_to_num(m_inf, posinf=0.0), onsets=on)
        a = ctx.earlyfusion_pairs(np.array([[0, 1]], np.int32))
        ctx.set_nonfinite_policy("raise")
        ctx.ef_upload_raw_pool([tracks[0], z])
        assert np.array_equal(ctx.earlyfusion_pairs(np.array([[0, 1]], np.int32)), a)
    finally:
        ctx.close()
