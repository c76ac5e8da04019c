"""
GPU parity tests for the on-device feature preparation through the C ABI:
  * Serra09 block-median pooling (rqa_serra09.py:49-52): BIT-EXACT against the oracle's
    librosa.util.sync / np.median restatement, and the scores of a raw upload equal those of a
    host-pooled upload;
  * SiMPle mean pooling + Hann smoothing + L2 (simple_silva.py:34-43, 56-66): window means
    bit-exact in f32 (checked through the features: a 1-ulp f32 difference would show at 1e-8),
    smoothed features within 1e-13 relative (f64, the 6-tap sums may associate differently).
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


def _raw_tracks(rng, lens, quantise=False):
    out = []
    for T0 in lens:
        x = rng.random((T0, 12)).astype(np.float32)
        if quantise:
            x = (np.round(x * 4) / 4).astype(np.float32)          # many equal values inside a block
        out.append(x)
    return out


def _pack(tracks):
    offs = np.concatenate([[0], np.cumsum([len(t) for t in tracks])]).astype(np.int64)
    return np.concatenate(tracks).astype(np.float32), offs


@pytest.mark.parametrize("fac", [40, 1, 2, 7, 39, 64])
def test_block_median_bit_exact(ctx, fac):
    import oracle
    rng = np.random.default_rng(fac)
    lens = [1, 2, 39, 40, 41, 79, 80, 81, 400, 1234, 4001, 12345]
    for quantise in (False, True):
        tracks = _raw_tracks(rng, lens, quantise)
        raw, roff = _pack(tracks)
        poff = ctx.upload_raw_pool(raw, roff, fac)
        want = [oracle.sync_median(t, fac) for t in tracks]
        assert np.array_equal(np.diff(poff), [len(w) for w in want])
        got = ctx.download_pool(poff[-1])
        assert got.dtype == np.float32 and np.array_equal(got, np.concatenate(want))


def test_raw_upload_gives_the_scores_of_a_host_pooled_upload(ctx):
    import oracle
    from acoss_amd import _lib
    rng = np.random.default_rng(5)
    tracks = _raw_tracks(rng, [40 * 60 + 7, 40 * 90, 40 * 75 + 39, 40 * 120 + 1])
    raw, roff = _pack(tracks)
    pairs = oracle.all_pairs(4, False).astype(np.int32)
    ctx.upload_raw_pool(raw, roff, 40)
    a = ctx.serra09_pairs(pairs, _lib.serra09_params())
    pooled = [oracle.sync_median(t, 40) for t in tracks]
    frames, offs = _pack(pooled)
    ctx.upload_pool(frames, offs)
    b = ctx.serra09_pairs(pairs, _lib.serra09_params())
    assert np.array_equal(a, b)
    assert np.array_equal(a, oracle.serra09_pairs(frames, offs, pairs, oracle.serra09_params()))


def test_simple_features_on_device(ctx):
    import oracle
    rng = np.random.default_rng(8)
    lens = [1000, 1099, 1100, 2543, 7777, 20000, 51200]
    tracks = _raw_tracks(rng, lens)
    raw, roff = _pack(tracks)
    poff = ctx.simple_upload_raw_pool(raw, roff)
    want = [oracle.simple_features(t).T for t in tracks]           # (n_i, 12) time-major
    assert np.array_equal(np.diff(poff), [len(w) for w in want]) and np.diff(poff).tolist() == [T // 100 for T in lens]
    got = ctx.download_pool_f64(poff[-1])
    np.testing.assert_allclose(got, np.concatenate(want), rtol=1e-13, atol=1e-16)
    # scores from the device-prepared pool == scores from the oracle's features
    pairs = np.array([[0, 1], [1, 0], [2, 3], [4, 3], [5, 6], [6, 0]], np.int32)
    a = ctx.simple_pairs(pairs, 10)
    ref = np.array([oracle.simple_pair(want[i].T, want[j].T) for i, j in pairs])
    np.testing.assert_allclose(a, ref, rtol=1e-11, atol=1e-13)
    # other window settings (WIN / SKIP / smoothing length)
    poff = ctx.simple_upload_raw_pool(raw[:roff[4]], roff[:5], win=150, skip=50, win_len_smooth=6)
    want = [oracle.simple_smooth(oracle.simple_pool(t, 150, 50), 6).T for t in tracks[:4]]
    np.testing.assert_allclose(ctx.download_pool_f64(poff[-1]), np.concatenate(want), rtol=1e-13, atol=1e-16)


def test_errors(ctx):
    rng = np.random.default_rng(1)
    raw, roff = _pack(_raw_tracks(rng, [100, 200]))
    with pytest.raises(NotImplementedError):
        ctx.upload_raw_pool(raw, roff, 65)
    with pytest.raises(ValueError):
        ctx.upload_raw_pool(raw, roff, 0)
    with pytest.raises(ValueError):
        ctx.upload_raw_pool(raw, roff[::-1].copy(), 40)
    # long tracks: pooled / smoothed in chunks of 480 frames with a halo (no length limit)
    import oracle
    tracks = _raw_tracks(rng, [51300, 97001, 480 * 100, 481 * 100 + 7])
    raw, roff = _pack(tracks)
    poff = ctx.simple_upload_raw_pool(raw, roff)
    want = [oracle.simple_smooth(oracle.simple_pool(t, 200, 100), 4).T for t in tracks]
    assert np.array_equal(np.diff(poff), [513, 970, 480, 481])
    np.testing.assert_allclose(ctx.download_pool_f64(poff[-1]), np.concatenate(want), rtol=1e-13, atol=1e-16)
