"""
GPU tests of the N x N pair grid (run with -m gpu on a real MI355X): acx_pair_grid / acx_grid_run
against the pair-list entry points, through the real host classes, and on two ranks.
"""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ctx():
    from acoss_amd import _lib
    c = _lib.Context(0)
    yield c
    c.close()


def _from_pairs(n, pairs, scores, mirror):
    D = np.zeros((n, n), np.float32)
    D[pairs[:, 0], pairs[:, 1]] = scores
    if mirror:
        D += D.T
    return D


@pytest.mark.parametrize("tile", [0, 7, 64])
def test_pair_grid_equals_pair_list_serra09(ctx, tile):
    import oracle
    from acoss_amd import synth, _lib
    d = synth.cover_set(clique_sizes=[2] * 9 + [3, 1], seed=31, t_range=(60, 420))
    n = len(d["offsets"]) - 1
    ctx.upload_pool(d["frames"], d["offsets"])
    for sym in (True, False):
        pairs = oracle.all_pairs(n, sym).astype(np.int32)
        want = _from_pairs(n, pairs, ctx.serra09_pairs(pairs), sym)
        D = np.zeros((n, n), np.float32)
        ctx.pair_grid(_lib.ALGO_SERRA09, sym, _lib.serra09_params(), [D], mirror=sym, tile=tile)
        assert np.array_equal(D, want)
    # two planes from one recurrence plot
    pairs = oracle.all_pairs(n, True).astype(np.int32)
    both = ctx.chenfusion_pairs(pairs)
    Q, Dm = np.zeros((n, n), np.float32), np.zeros((n, n), np.float32)
    ctx.pair_grid(_lib.ALGO_CHENFUSION, True, _lib.serra09_params(), [Q, Dm], mirror=True, tile=tile)
    assert np.array_equal(Q, _from_pairs(n, pairs, both[:, 0], True))
    assert np.array_equal(Dm, _from_pairs(n, pairs, both[:, 1], True))


def test_grid_run_ranks_and_slices(ctx):
    """acx_grid_run for every rank of a 3-rank plan into torch device buffers (the all-gather's operands),
    in one call and tile by tile; acx_grid_scatter of the concatenation equals the one-GPU grid."""
    import torch
    from acoss_amd import synth, _lib
    d = synth.cover_set(clique_sizes=[2] * 14, seed=5, t_range=(60, 300))
    n = len(d["offsets"]) - 1
    ctx.upload_pool(d["frames"], d["offsets"])
    want = np.zeros((n, n), np.float32)
    ctx.pair_grid(_lib.ALGO_SERRA09, True, _lib.serra09_params(), [want], mirror=True)
    lengths = ctx.pool_lengths(_lib.ALGO_SERRA09)
    assert np.array_equal(lengths, np.diff(d["offsets"]))
    ws = 3
    plan = _lib.grid_plan(lengths, _lib.ALGO_SERRA09, True, world=ws, tile=5, want_tiles=True)
    stride = int(plan["floats_per_rank"].max())
    for sliced in (False, True):
        bufs = []
        for r in range(ws):
            t = torch.full((stride,), -7.0, dtype=torch.float32, device="cuda:0")
            torch.cuda.synchronize()
            if sliced:
                ntile = sum(1 for tl in plan["tiles"] if tl.rank == r)
                for k in range(ntile):
                    ctx.grid_run(plan["spec"], _lib.serra09_params(), r, t.data_ptr(), first=k, count=1)
            else:
                ctx.grid_run(plan["spec"], _lib.serra09_params(), r, t.data_ptr())
            bufs.append(t.cpu().numpy())
        D = np.zeros((n, n), np.float32)
        _lib.grid_scatter(lengths, plan["spec"], np.concatenate(bufs), stride, [D], mirror=True)
        assert np.array_equal(D, want)


def test_pair_grid_simple_and_earlyfusion(ctx):
    from acoss_amd import synth, _lib
    import oracle
    rng = np.random.default_rng(8)
    feats = [rng.random((int(rng.integers(40, 120)), 12)) for _ in range(9)]
    feats = [f / np.linalg.norm(f, axis=1, keepdims=True) for f in feats]
    offs = np.concatenate([[0], np.cumsum([len(f) for f in feats])]).astype(np.int64)
    ctx.upload_pool_f64(np.concatenate(feats), offs)
    n = len(feats)
    pairs = oracle.all_pairs(n, False).astype(np.int32)
    want = _from_pairs(n, pairs, ctx.simple_pairs(pairs, 10).astype(np.float32), False)
    for tile in (4, 0, 3):                  # device-side pair enumeration: full, ordered-diagonal and ragged edge tiles
        D = np.zeros((n, n), np.float32)
        ctx.pair_grid(_lib.ALGO_SIMPLE, False, _lib.SimpleParams(10, 1), [D], mirror=False, tile=tile)
        assert np.array_equal(D, want)
    # all_pairwise(symmetric=True) on SiMPle: i < j only (triangular diagonal tiles), mirrored
    up = oracle.all_pairs(n, True).astype(np.int32)
    want_sym = _from_pairs(n, up, ctx.simple_pairs(up, 10).astype(np.float32), True)
    for tile in (4, 0):
        D = np.zeros((n, n), np.float32)
        ctx.pair_grid(_lib.ALGO_SIMPLE, True, _lib.SimpleParams(10, 1), [D], mirror=True, tile=tile)
        assert np.array_equal(D, want_sym)
    tracks = synth.earlyfusion_set(6, seed=3, nb_range=(40, 90))
    ctx.ef_upload_pool(tracks)
    pairs = oracle.all_pairs(6, True).astype(np.int32)
    sc = ctx.earlyfusion_pairs(pairs, kappa=0.1, K=10)
    planes = [np.zeros((6, 6), np.float32) for _ in range(4)]
    ctx.pair_grid(_lib.ALGO_EARLYFUSION, True, _lib.EfParams(0.1, 10), planes, mirror=True, tile=4)
    for e in range(4):
        assert np.array_equal(planes[e], _from_pairs(6, pairs, sc[:, e], True))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_ranks_share_one_gpu(tmp_path):
    """The real Serra09 and ChenFusion classes under torch.distributed (gloo, two ranks on GPU 0):
    rank 0's matrices are bit-identical to the single-process run, the statistics agree on all ranks."""
    from acoss_amd.featurestore import save_track
    wd = str(tmp_path)
    with open(os.path.join(wd, "grid.csv"), "w") as f:
        f.write("work_id,track_id\n")
        for k in range(32):
            f.write("w%d,t%d\n" % (k, k))
            save_track(os.path.join(wd, "w%d/t%d.h5" % (k, k)), {"label": "w%d" % k, "track_id": "t%d" % k})
    out = os.path.join(wd, "out")
    os.makedirs(out)
    worker = os.path.join(ROOT, "tests", "_grid_worker.py")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    subprocess.run([sys.executable, worker, wd, out], check=True, env=env, timeout=600)
    subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                    "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), worker, wd, out],
                   check=True, env=env, timeout=900)
    one = np.load(os.path.join(out, "world1_rank0.npz"))
    two0 = np.load(os.path.join(out, "world2_rank0.npz"))
    two1 = np.load(os.path.join(out, "world2_rank1.npz"))
    mats = [k for k in one.files if not k.endswith("_stats")]
    assert len(mats) == 3
    for k in one.files:
        assert np.array_equal(one[k], two0[k], equal_nan=True), k
        if k.endswith("_stats"):
            assert np.array_equal(one[k], two1[k], equal_nan=True), k
    assert float(one["serra09_main"].max()) > 10.0


def test_two_gpus_exchange_over_rccl(tmp_path):
    """Two ranks on TWO GPUs under the "nccl" backend: the tile exchange of the path (torch.distributed.gather, or the
    all-gather the probe of acoss_amd/dist.py falls back to) really crosses xGMI.  Skipped on a box with one GPU --
    every gpurun box of the build rounds; it is here for the first box that has two."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (this box has %d)" % torch.cuda.device_count())
    from acoss_amd.featurestore import save_track
    wd = str(tmp_path)
    with open(os.path.join(wd, "grid.csv"), "w") as f:
        f.write("work_id,track_id\n")
        for k in range(32):
            f.write("w%d,t%d\n" % (k, k))
            save_track(os.path.join(wd, "w%d/t%d.h5" % (k, k)), {"label": "w%d" % k, "track_id": "t%d" % k})
    out = os.path.join(wd, "out")
    os.makedirs(out)
    worker = os.path.join(ROOT, "tests", "_grid_worker.py")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", ACX_TEST_BACKEND="nccl")
    env.pop("WORLD_SIZE", None)
    subprocess.run([sys.executable, worker, wd, out], check=True, env=env, timeout=600)
    subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                    "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), worker, wd, out],
                   check=True, env=env, timeout=900)
    one = np.load(os.path.join(out, "world1_rank0.npz"))
    two0 = np.load(os.path.join(out, "world2_rank0.npz"))
    two1 = np.load(os.path.join(out, "world2_rank1.npz"))
    assert str(two0["exchange"]) in ("gather", "allgather"), two0["exchange"]
    for k in one.files:
        assert np.array_equal(one[k], two0[k], equal_nan=True), k
        if k.endswith("_stats"):
            assert np.array_equal(one[k], two1[k], equal_nan=True), k


def test_libacx_before_torch_in_one_process():
    """Import order must not matter: a process that creates its libacx context FIRST and imports torch later still
    gets a working torch.cuda (acoss_amd._lib preloads the HIP runtime PyTorch bundles, so both end up on the same
    one), and libacx writes into a torch tensor allocated afterwards."""
    code = r"""
import sys
sys.path.insert(0, %r)
import numpy as np
assert "torch" not in sys.modules
from acoss_amd import _lib, synth
ctx = _lib.Context(0)
assert "torch" not in sys.modules, "creating a context must not import torch"
d = synth.cover_set(clique_sizes=[2] * 5, seed=3, t_range=(60, 200))
n = len(d["offsets"]) - 1
ctx.upload_pool(d["frames"], d["offsets"])
want = np.zeros((n, n), np.float32)
ctx.pair_grid(_lib.ALGO_SERRA09, True, _lib.serra09_params(), [want], mirror=False)
import torch
assert torch.cuda.is_available() and torch.cuda.device_count() >= 1
plan = _lib.grid_plan(np.diff(d["offsets"]), _lib.ALGO_SERRA09, True, world=1, tile=n, want_tiles=True)
buf = torch.full((n * n,), -3.0, dtype=torch.float32, device="cuda:0")
torch.cuda.synchronize()
ctx.grid_run(plan["spec"], _lib.serra09_params(), 0, buf.data_ptr())
got = buf.cpu().numpy().reshape(n, n)
assert np.array_equal(np.triu(got, 1), np.triu(want, 1)), "scores through the torch buffer differ"
print("ok", _lib.HIP_VERSIONS)
""" % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, (r.stdout[-1000:], r.stderr[-3000:])


def test_nccl_world_of_one(tmp_path):
    """RCCL on the code path, as far as a 1-GPU box allows: a process group of ONE rank under the "nccl" backend,
    and ACX_GRID_VIA_COLLECTIVE=1 so that the classes take the multi-rank route anyway -- device binding, tile
    buffer on the GPU, torch.distributed.gather over RCCL on that buffer (the exchange of the path), rank-0 scatter, the clique-table and
    statistics broadcasts (broadcast_object_list with an explicit device), barrier(device_ids=...), any_rank's
    all-reduce.  Matrices and statistics must equal the plain single-process run."""
    code = r"""
import os, sys
sys.path.insert(0, %r)
import numpy as np
os.chdir(%r)
from acoss_amd import synth, dist as adist
from acoss_amd.algorithms.rqa_serra09 import Serra09
from acoss_amd.algorithms.latefusion_chen import ChenFusion
d = synth.cover_set(clique_sizes=[2] * 11 + [3], seed=12, t_range=(60, 200))      # 25 tracks: SNF takes K = 20 neighbours
n = len(d["offsets"]) - 1
with open("ds.csv", "w") as f:
    f.write("work_id,track_id\n")
    for i, l in enumerate(d["labels"]):
        f.write("%%s,t%%d\n" %% (l, i))
tracks = [d["frames"][d["offsets"][i]:d["offsets"][i + 1]] for i in range(n)]

def run(cls, tag):
    a = cls("ds.csv", "feat/", shortname=tag)
    a.set_pooled_features(tracks, d["labels"])
    a.all_pairwise(symmetric=True)
    a.normalize_by_length()
    if cls is ChenFusion:
        a.do_late_fusion()
    st = {k: a.getEvalStatistics(k, topsidx=[1, 10]) for k in list(a.Ds.keys())}
    out = {k: np.array(a.Ds[k]) for k in a.Ds}
    a.cleanup_memmap()
    return out, st

plain = {c.__name__: run(c, "plain" + c.__name__) for c in (Serra09, ChenFusion)}
import torch
import torch.distributed as dist
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=%r, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", ACX_GRID_VIA_COLLECTIVE="1")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
dist.init_process_group("nccl")
assert dist.get_backend() == "nccl" and not adist.single()
adist.barrier()
assert adist.any_rank(False) is False and adist.any_rank(True) is True
assert adist.broadcast_object({"x": 3}) == {"x": 3}
coll = {c.__name__: run(c, "coll" + c.__name__) for c in (Serra09, ChenFusion)}
dist.barrier(device_ids=[0])
dist.destroy_process_group()
for name in plain:
    for k in plain[name][0]:
        assert np.array_equal(plain[name][0][k], coll[name][0][k], equal_nan=True), (name, k)
        a, b = plain[name][1][k], coll[name][1][k]
        assert a[:4] == b[:4] and np.array_equal(a[4], b[4]), (name, k)
print("ok nccl world of one")
""" % (ROOT, str(tmp_path), str(_free_port()))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "ok nccl world of one" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


def test_grid_run_ranks_simple_and_earlyfusion(ctx):
    """The per-rank route (acx_grid_run into a device buffer, then acx_grid_scatter of the gathered buffers) for the
    two algorithms whose grid scores stay on the device since round 3: SiMPle (pairs enumerated on the device, ordered
    grid, diagonal tiles with i != j) and EarlyFusion (four planes, rectangles of pairs) on 3-rank plans, whole and
    tile by tile, against the one-GPU grid."""
    import torch
    from acoss_amd import synth, _lib
    rng = np.random.default_rng(9)
    feats = [rng.random((int(rng.integers(30, 90)), 12)) for _ in range(23)]
    feats = [f / np.linalg.norm(f, axis=1, keepdims=True) for f in feats]
    offs = np.concatenate([[0], np.cumsum([len(f) for f in feats])]).astype(np.int64)
    ctx.upload_pool_f64(np.concatenate(feats), offs)
    tracks = synth.earlyfusion_set(11, seed=4, nb_range=(20, 70))
    ctx.ef_upload_pool(tracks)
    cases = [(_lib.ALGO_SIMPLE, False, _lib.SimpleParams(10, 1), 23, 1), (_lib.ALGO_SIMPLE, True, _lib.SimpleParams(10, 1), 23, 1),
             (_lib.ALGO_EARLYFUSION, True, _lib.EfParams(0.1, 10), 11, 4)]
    for algo, sym, params, n, w in cases:
        want = [np.zeros((n, n), np.float32) for _ in range(w)]
        ctx.pair_grid(algo, sym, params, want, mirror=sym)
        lengths = ctx.pool_lengths(algo)
        ws = 3
        plan = _lib.grid_plan(lengths, algo, sym, world=ws, tile=4, want_tiles=True)
        stride = int(plan["floats_per_rank"].max())
        for sliced in (False, True):
            bufs = []
            for r in range(ws):
                t = torch.full((stride,), -7.0, dtype=torch.float32, device="cuda:0")
                torch.cuda.synchronize()
                if sliced:
                    for k in range(sum(1 for tl in plan["tiles"] if tl.rank == r)):
                        ctx.grid_run(plan["spec"], params, r, t.data_ptr(), first=k, count=1)
                else:
                    ctx.grid_run(plan["spec"], params, r, t.data_ptr())
                bufs.append(t.cpu().numpy())
            D = [np.zeros((n, n), np.float32) for _ in range(w)]
            _lib.grid_scatter(lengths, plan["spec"], np.concatenate(bufs), stride, D, mirror=sym)
            for e in range(w):
                assert np.array_equal(D[e], want[e]), (algo, sym, sliced, e)


def _bench_strong(env_extra, nproc, tracks=160, frames=400, tile=32):
    """`bench.py --strong` as a child process (nproc > 1: one torch.distributed.run launch on 127.0.0.1);
    returns the parsed JSON line."""
    import json
    import os
    env = dict(os.environ)
    env.update(env_extra)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    args = ["--gpus", str(nproc), "--strong", "--tracks", str(tracks), "--frames", str(frames), "--tile", str(tile), "--warmup", "1"]
    if nproc > 1:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
               "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py")] + args
    else:
        env.update(MASTER_PORT=str(_free_port()))
        cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + args
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                      # rank 0 prints ONE JSON line
    return json.loads(lines[0])


def test_bench_strong_nccl_world_of_one():
    """bench.py --strong through RCCL as far as one GPU allows (ACX_BENCH_FORCE_COLLECTIVE=1: a one-rank "nccl" group):
    the whole grid into the rank's buffer, all_gather_into_tensor of the REAL buffer, rank-0 scatter + mirror into
    the memmap; the matrix must equal the pair-list path on the sampled pairs."""
    line = _bench_strong({"ACX_BENCH_FORCE_COLLECTIVE": "1"}, 1)
    s = line["strong"]
    assert line["scaling"] == "strong" and line["collectives"] == "nccl" and line["ranks_seen"] == 1
    assert line["config"]["pairs"] == 160 * 159 // 2
    assert s["check"]["matrix_equals_pair_list"] and s["check"]["symmetric"], s["check"]
    assert s["check"]["nonzero_fraction_offdiag"] == 1.0
    assert s["gather_bytes"] >= 4 * line["config"]["pairs"] and s["gather_ms"] > 0
    assert "phases_s" in line and line["value"] > 0
    # the line proves its device: PCI bus id + name of the GPU the rank holds, RCCL's version
    (r0,) = line["ranks"]
    assert r0["rank"] == 0 and len(r0["pci_bus_id"]) >= 7 and "gfx950" in r0["name"] and r0["visible_devices"] >= 1
    assert line["rccl_version"] and "gather to rank 0" in s["exchange"]


def test_bench_strong_gloo_world_of_two():
    """The N > 1 route of bench.py --strong with two REAL ranks (sharing the one GPU of the box, gather through gloo):
    cost-balanced deal, per-rank buffers of the plan's stride, the gathered buffers scattered by rank 0 -- the
    matrix equals the pair-list path, both ranks' kernel times are reported."""
    line = _bench_strong({"ACX_BENCH_BACKEND": "gloo"}, 2)
    s = line["strong"]
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2 and line["collectives"] == "gloo"
    assert len(s["kernels_s_per_rank"]) == 2 and min(s["kernels_s_per_rank"]) > 0
    assert s["check"]["matrix_equals_pair_list"] and s["check"]["symmetric"], s["check"]
    assert s["gather_bytes"] == 2 * s["gather_bytes_per_rank"]
    c = s["plan_cost_per_rank"]
    assert max(c) / (sum(c) / 2) < 1.1                                # the deal balances the modelled cost
    assert [r["rank"] for r in line["ranks"]] == [0, 1] and all(r["pci_bus_id"] for r in line["ranks"])


def _bench_default(env_extra, nproc, extra_args=(), expect_rc=0, with_other=False):
    """The driver's command (`bench.py --gpus N --steps K --warmup W`, weak scaling) as a child process on a small pool."""
    import json
    import os
    env = dict(os.environ)
    env.update(env_extra)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("ACX_BENCH_STRONG_TRACKS", "192")
    args = ["--gpus", str(nproc), "--steps", "2", "--warmup", "1", "--tracks", "640", "--no-cpu"] + ([] if with_other else ["--no-other"]) + list(extra_args)
    if nproc > 1:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
               "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py")] + args
    else:
        env.update(MASTER_PORT=str(_free_port()))
        cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + args
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert (r.returncode == 0) == (expect_rc == 0), (r.returncode, r.stdout[-1500:], r.stderr[-3000:])
    if expect_rc != 0:
        return r
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_default_line_proves_its_ranks_and_carries_the_real_exchange():
    """the DEFAULT (weak) bench line of an N > 1 run names every rank's GPU (PCI bus id, device name, visible
    devices, its kernel seconds) and carries a `strong` object from a sub-grid run as ONE job -- the real plan, ONE exchange of the
    real per-rank buffers, rank-0 scatter, the matrix checked against the pair-list path.  Here: a one-rank RCCL world
    (ACX_BENCH_FORCE_COLLECTIVE=1) and two gloo ranks sharing the GPU."""
    line = _bench_default({"ACX_BENCH_FORCE_COLLECTIVE": "1"}, 1)
    assert line["scaling"] == "weak" and line["collectives"] == "nccl" and line["rccl_version"]
    (r0,) = line["ranks"]
    assert len(r0["pci_bus_id"]) >= 7 and r0["kernels_s"] > 0 and r0["pairs"] == 2 * 8192
    s = line["strong"]
    assert s["tracks"] == 192 and s["pairs"] == 192 * 191 // 2 and s["check"]["matrix_equals_pair_list"] and s["check"]["symmetric"]
    assert s["gather_ms"] > 0 and s["gather_bytes"] >= 4 * s["pairs"] and s["scatter_mirror_s"] >= 0
    line = _bench_default({"ACX_BENCH_BACKEND": "gloo"}, 2)
    assert line["n_gpus"] == 2 and [r["rank"] for r in line["ranks"]] == [0, 1] and line["collectives"] == "gloo"
    assert line["ranks"][0]["pci_bus_id"] == line["ranks"][1]["pci_bus_id"]          # (development mode: the ranks share the box's GPU)
    s = line["strong"]
    assert len(s["kernels_s_per_rank"]) == 2 and s["check"]["matrix_equals_pair_list"], s
    # one rank without a collective: still says which GPU it ran on; the profiler passes' form of the command (--no-other) has no sub-grid leg
    line = _bench_default({}, 1)
    assert line["strong"] is None and len(line["ranks"]) == 1 and line["ranks"][0]["pci_bus_id"]
    # ... the full default run does: the whole (sub-)grid as one job, copy to the host, scatter + mirror, checked against the pair list
    line = _bench_default({}, 1, with_other=True)
    s = line["strong"]
    assert line["collectives"] is None and s["exchange"] is None and s["tracks"] == 192 and s["pairs"] == 192 * 191 // 2
    assert s["check"]["matrix_equals_pair_list"] and s["check"]["symmetric"] and s["value_incl_scatter"] > 0
    assert set(line["other"]) >= {"serra09_covers", "simple", "earlyfusion"}


def test_bench_refuses_ranks_that_share_a_gpu():
    """Two ranks on ONE device are not a two-GPU figure: under RCCL semantics (here claimed for a gloo world by
    ACX_BENCH_REQUIRE_DISTINCT=1 -- a 1-GPU box cannot hold two RCCL ranks) the run ends non-zero with a one-line reason and
    prints no JSON line."""
    r = _bench_default({"ACX_BENCH_BACKEND": "gloo", "ACX_BENCH_REQUIRE_DISTINCT": "1"}, 2, expect_rc=3)
    assert "hold the same GPU" in r.stderr and not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_rccl_inside_libacx_world_of_one():
    """The library's own multi-GPU route (include/acx.h "multi-GPU inside the library"), as far as one GPU allows: a
    process that never imports torch gets a communicator id, joins as rank 0 of 1, runs acx_grid_run into a buffer from
    acx_dev_alloc, all-gathers it over RCCL on the library's stream (acx_grid_allgather), and acx_pair_grid_ranks fills
    the matrices -- equal to the one-GPU acx_pair_grid for Serra09 (one plane) and ChenFusion (two planes)."""
    code = r"""
import sys
sys.path.insert(0, %r)
import numpy as np
from acoss_amd import _lib, synth
d = synth.cover_set(clique_sizes=[2] * 9, seed=5, t_range=(60, 260))
n = len(d["offsets"]) - 1
ctx = _lib.Context(0)
ctx.upload_pool(d["frames"], d["offsets"])
p = _lib.serra09_params()
want = np.zeros((n, n), np.float32)
ctx.pair_grid(_lib.ALGO_SERRA09, True, p, [want], mirror=True)
cid = _lib.comm_id()
assert len(cid) == _lib.COMM_ID_BYTES
ctx.comm_init(cid, 0, 1)
try:
    ctx.comm_init(cid, 0, 1)
    raise SystemExit("a second communicator on one context must be refused")
except _lib.AcxError:
    pass
# the pieces: tiles into a libacx buffer, the all-gather, the host scatter
lengths = ctx.pool_lengths(_lib.ALGO_SERRA09)
plan = _lib.grid_plan(lengths, _lib.ALGO_SERRA09, True, world=1, tile=4, want_tiles=True)
stride = int(plan["floats_per_rank"][0])
loc, allb = ctx.dev_alloc(4 * stride), ctx.dev_alloc(4 * stride)
ctx.grid_run(plan["spec"], p, 0, loc.ptr)
ctx.grid_allgather(loc.ptr, allb.ptr, stride)
got = np.zeros((n, n), np.float32)
_lib.grid_scatter(lengths, plan["spec"], allb.read(np.float32, stride), stride, [got], mirror=True)
assert np.array_equal(got, want) and float(want.max()) > 10.0
loc.free(); allb.free()
# the whole thing in one call, one and two planes
got2 = np.zeros((n, n), np.float32)
ctx.pair_grid_ranks(_lib.ALGO_SERRA09, True, p, [got2], mirror=True, tile=5)
assert np.array_equal(got2, want)
q, dm = np.zeros((n, n), np.float32), np.zeros((n, n), np.float32)
ctx.pair_grid_ranks(_lib.ALGO_CHENFUSION, True, p, [q, dm], mirror=True)
q1, d1 = np.zeros((n, n), np.float32), np.zeros((n, n), np.float32)
ctx.pair_grid(_lib.ALGO_CHENFUSION, True, p, [q1, d1], mirror=True)
assert np.array_equal(q, q1) and np.array_equal(dm, d1) and np.array_equal(q, want)
# (ADVICE r04) a rank whose own checks fail must still take part in the exchanges and come back with the error -- here the one
# rank there is: the planes of an algorithm with two of them handed over as one (a null plane), a leading dimension smaller than
# the pool, and, after them, a good call again (nothing was left half-done inside the communicator)
import ctypes
L, h = ctx._L, ctx._h
spec = _lib.GridSpec(_lib.ALGO_CHENFUSION, 1, 0, 1)
arr = (ctypes.c_void_p * 2)(q.ctypes.data, None)
assert L.acx_pair_grid_ranks(h, ctypes.byref(spec), ctypes.byref(p), arr, n, 1) == -1 and b"null plane" in L.acx_last_error(h)
arr = (ctypes.c_void_p * 2)(q.ctypes.data, dm.ctypes.data)
assert L.acx_pair_grid_ranks(h, ctypes.byref(spec), ctypes.byref(p), arr, n - 1, 1) == -1 and b"leading dimension" in L.acx_last_error(h)
got3 = np.zeros((n, n), np.float32)
ctx.pair_grid_ranks(_lib.ALGO_SERRA09, True, p, [got3], mirror=True)
assert np.array_equal(got3, want)
ctx.comm_destroy()
ctx.close()
assert "torch" not in sys.modules
print("ok rccl inside libacx")
""" % ROOT
    env = dict(__import__("os").environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "ok rccl inside libacx" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
