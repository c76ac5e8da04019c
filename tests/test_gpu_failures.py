"""
GPU suite: what a failing call leaves behind.  The reference fails a whole chunk of pairs when one of them is bad
(acoss/algorithms/algorithm_template.py:174-177: the exception of a joblib worker ends the run); libacx checks the WHOLE
(K, 2) list -- indices, tracks shorter than the stack, pairs beyond the scratch limit -- before its first launch, drains
every stream it uses when a call fails later than that, and the next call on the same context gives the scores a fresh
context gives.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture()
def ctx():
    from acoss_amd import _lib
    c = _lib.Context(0)
    yield c
    c.close()


def _launches(ctx):
    return sum(v["launches"] for v in ctx.profile().values())


def test_serra09_bad_index_deep_in_a_long_list(ctx):
    from acoss_amd import synth
    d = synth.cover_set(n_works=20, versions=3, seed=11, t_range=(60, 140))
    n = len(d["offsets"]) - 1
    ctx.upload_pool(d["frames"], d["offsets"])
    rng = np.random.default_rng(3)
    good = rng.integers(0, n, (100000, 2)).astype(np.int32)          # two batches (a batch holds at most 65 535 pairs)
    want = ctx.serra09_pairs(good)
    bad = good.copy()
    bad[70000, 1] = n + 5
    ctx.profile_enable(True)
    ctx.profile_reset()
    with pytest.raises(ValueError, match="pair 70000"):
        ctx.serra09_pairs(bad)
    assert _launches(ctx) == 0, "the list is validated before the first launch"
    ctx.profile_enable(False)
    assert np.array_equal(ctx.serra09_pairs(good), want)              # the product path (sweeps on their own streams) after the failure
    both = ctx.chenfusion_pairs(good[:5000])
    assert np.array_equal(both[:, 0], want[:5000])
    bad[70000] = (0, -1)
    with pytest.raises(ValueError, match="pair 70000"):
        ctx.chenfusion_pairs(bad)
    assert np.array_equal(ctx.serra09_pairs(good), want)


def test_serra09_failure_behind_the_first_batches(ctx):
    """A pair that does not fit the scratch limit, behind 70 000 that do: nothing runs, nothing stays in flight, and the same
    context then repeats a good list bit for bit -- also when the failing call was preceded by a call whose sweeps were still
    running on the side streams when it returned its scores."""
    from acoss_amd import synth, _lib
    rng = np.random.default_rng(5)
    tracks = [synth._frame_max_normalise(rng.random((T, 12))) for T in list(rng.integers(60, 140, 40)) + [2600]]
    frames, offsets = synth.pack(tracks)
    ctx.upload_pool(frames, offsets)
    good = rng.integers(0, 40, (100000, 2)).astype(np.int32)
    want = ctx.serra09_pairs(good)
    bad = good.copy()
    bad[70000] = (40, 40)                                             # 2600 x 2600 frames: the streaming kernels' float matrices
    ctx.set_scratch_limit(16 << 20)
    try:
        ctx.profile_enable(True)
        ctx.profile_reset()
        with pytest.raises(MemoryError, match="pair 70000 does not fit"):       # (ACX_ERR_NOMEM)
            ctx.serra09_pairs(bad)
        assert _launches(ctx) == 0
        ctx.profile_enable(False)
        assert np.array_equal(ctx.serra09_pairs(good), want)
    finally:
        ctx.set_scratch_limit(0)
    assert np.array_equal(ctx.serra09_pairs(good), want)


def test_earlyfusion_bad_index_deep_in_a_long_list(ctx):
    from acoss_amd import synth
    tracks = synth.earlyfusion_set(24, seed=2, nb_range=(20, 40))
    ctx.ef_upload_pool(tracks)
    rng = np.random.default_rng(7)
    good = rng.integers(0, 24, (100000, 2)).astype(np.int32)
    want = ctx.earlyfusion_pairs(good)
    bad = good.copy()
    bad[70000, 0] = 24
    ctx.profile_enable(True)
    ctx.profile_reset()
    with pytest.raises(ValueError, match="pair 70000"):
        ctx.earlyfusion_pairs(bad)
    assert _launches(ctx) == 0, "the list is validated before the first launch"
    ctx.profile_enable(False)
    assert np.array_equal(ctx.earlyfusion_pairs(good), want)
