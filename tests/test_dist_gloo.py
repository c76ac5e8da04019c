"""
CPU suite, part 5: the N > 1 path.  Two processes under torch.distributed (gloo, 127.0.0.1).
  * user subclasses (their own CPU similarity()): the pair list is sharded, the scores gathered;
  * device-backed classes: the tile plan of libacx (acx_grid_plan), every rank fills the score
    buffer of ITS tiles, one all-gather of the buffers, rank 0 scatters (acx_grid_scatter).  The GPU
    context is replaced by a stand-in that writes f(i, j) into the tiles, so that plan, gather and
    scatter -- everything but the kernels -- run here; the real thing is tests/test_gpu_grid.py.
Rank 0 owns the result: it must end with the single-process matrices; the statistics are the same
on every rank.
"""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, ws, port, workdir, sym, out, kind):
    sys.path.insert(0, ROOT)
    os.chdir(workdir)
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=ws)
    try:
        D, stats = {"grid": _run_grid, "toy": _run_toy, "labels": _run_labels}[kind](workdir, sym)
        np.save(os.path.join(out, "D_rank%d.npy" % rank), D)
        np.save(os.path.join(out, "S_rank%d.npy" % rank), np.array(stats[:4]))
    finally:
        dist.destroy_process_group()


def _run_toy(workdir, sym):
    from acoss_amd.algorithms.algorithm_template import CoverAlgorithm
    n = 11
    S = np.random.default_rng(5).random((n, n)).astype(np.float32)

    class Toy(CoverAlgorithm):
        def similarity(self, idxs):
            i, j = idxs[:, 0], idxs[:, 1]
            self.Ds["main"][i, j] = S[i, j] + (S[j, i] if sym else 0)
            self.Ds["aux"][i, j] = 2 * S[i, j]

    toy = Toy(os.path.join(workdir, "toy.csv"), name="Toy", datapath=workdir + "/", shortname="t",
              similarity_types=["main", "aux"])
    toy.all_pairwise(symmetric=sym)
    stats = toy.getEvalStatistics("main", topsidx=[1, 10])
    return np.stack([np.array(toy.Ds["main"]), np.array(toy.Ds["aux"])]), stats


def _run_labels(workdir, sym):
    """Rank-local state must not decide whether a collective runs: rank 1 has its labels injected (its
    `cliques` is not empty), rank 0 has not -- all_pairwise has to take every rank through the clique
    broadcast anyway -- and the owner-only post-processing steps return at once on the other ranks."""
    import torch.distributed as dist
    from acoss_amd.algorithms.algorithm_template import CoverAlgorithm
    n = 11
    S = np.random.default_rng(5).random((n, n)).astype(np.float32)

    class Toy(CoverAlgorithm):
        def similarity(self, idxs):
            i, j = idxs[:, 0], idxs[:, 1]
            self.Ds["main"][i, j] = S[i, j] + (S[j, i] if sym else 0)

        def normalize(self):
            if not self.owns_result():
                return "skipped"
            self.Ds["main"][:] = self.Ds["main"] * 2
            return "done"

    toy = Toy(os.path.join(workdir, "toy.csv"), name="Toy", datapath=workdir + "/", shortname="l")
    if dist.is_initialized() and dist.get_rank() == 1:
        for i in range(n):
            toy._register_label(i, "w%d" % (i % 4))
    toy.all_pairwise(symmetric=sym)
    assert len(toy.cliques) == 4 and sum(len(c) for c in toy.cliques.values()) == n
    rank = dist.get_rank() if dist.is_initialized() else 0
    assert toy.normalize() == ("done" if rank == 0 else "skipped")
    stats = toy.getEvalStatistics("main", topsidx=[1, 10])
    return np.array(toy.Ds["main"])[None], stats


class _FakeContext(object):
    """Stand-in for acoss_amd._lib.Context in the grid path: same methods, CPU memory, and the
    'kernel' is score(i, j, plane) = 1000 plane + 37 i + j (+ 0.5 for ordered pairs with i > j)."""
    device = 0

    def __init__(self, lengths):
        self.lengths = np.asarray(lengths, np.int64)

    def torch_device(self):
        import torch
        return torch.device("cpu")

    def pool_lengths(self, algo):
        return self.lengths

    @staticmethod
    def score(i, j, e):
        return 1000.0 * e + 37.0 * i + j + (0.5 if i > j else 0.0)

    def _fill(self, plan, rank, buf):
        from acoss_amd import _lib
        w = _lib.GRID_PLANES[plan["spec"].algo]
        sym = plan["spec"].symmetric
        for t in plan["tiles"]:
            if t.rank != rank:
                continue
            blk = np.zeros((t.rows, t.cols, w), np.float32)
            for a in range(t.rows):
                for b in range(t.cols):
                    i, j = t.row0 + a, t.col0 + b
                    if t.diagonal and ((sym and not i < j) or (not sym and i == j)):
                        continue
                    blk[a, b] = [self.score(i, j, e) for e in range(w)]
            buf[t.offset:t.offset + blk.size] = blk.ravel()

    def grid_run(self, spec, params, rank, dev_ptr, first=0, count=-1):
        import ctypes
        from acoss_amd import _lib
        plan = _lib.grid_plan(self.lengths, spec.algo, spec.symmetric, world=spec.world, tile=spec.tile, want_tiles=True)
        n = int(plan["floats_per_rank"][rank])
        buf = np.ctypeslib.as_array(ctypes.cast(dev_ptr, ctypes.POINTER(ctypes.c_float)), shape=(n,))
        self._fill(plan, rank, buf)

    def pair_grid(self, algo, symmetric, params, planes, mirror, tile=0):
        from acoss_amd import _lib
        plan = _lib.grid_plan(self.lengths, algo, symmetric, world=1, tile=tile, want_tiles=True)
        buf = np.zeros(int(plan["floats_per_rank"][0]), np.float32)
        self._fill(plan, 0, buf)
        _lib.grid_scatter(self.lengths, plan["spec"], buf, len(buf), planes, mirror)


def _run_grid(workdir, sym):
    from acoss_amd import _lib
    from acoss_amd.algorithms.algorithm_template import CoverAlgorithm
    n = 75
    lengths = np.random.default_rng(3).integers(40, 900, n)

    class Dev(CoverAlgorithm):
        def _grid(self):
            return _FakeContext(lengths), _lib.ALGO_CHENFUSION, _lib.serra09_params(), ["qmax", "dmax"]

    dev = Dev(os.path.join(workdir, "grid.csv"), name="Dev", datapath=workdir + "/", shortname="g",
              similarity_types=["qmax", "dmax"])
    dev.all_pairwise(symmetric=sym)
    stats = dev.getEvalStatistics("qmax", topsidx=[1, 10])
    return np.stack([np.array(dev.Ds["qmax"]), np.array(dev.Ds["dmax"])]), stats


def _expected_grid(n, sym):
    D = np.zeros((2, n, n), np.float32)
    for e in range(2):
        for i in range(n):
            for j in range(n):
                if i == j:
                    continue
                if sym:
                    D[e, i, j] = _FakeContext.score(min(i, j), max(i, j), e)
                else:
                    D[e, i, j] = _FakeContext.score(i, j, e)
    return D


def _dataset(wd, name, n, nworks):
    from acoss_amd.featurestore import save_track
    with open(os.path.join(wd, name), "w") as f:
        f.write("work_id,track_id\n")
        for k in range(n):
            f.write("w%d,t%d\n" % (k % nworks, k))
            save_track(os.path.join(wd, "w%d/t%d.h5" % (k % nworks, k)), {"label": "w%d" % (k % nworks), "track_id": "t%d" % k})


@pytest.mark.timeout(300)
@pytest.mark.parametrize("kind,sym,ws", [("toy", True, 2), ("toy", False, 2), ("grid", True, 2), ("grid", False, 2), ("labels", True, 2),
                                         ("grid", True, 8), ("grid", False, 3), ("toy", True, 3)])
def test_two_ranks_equal_one(tmp_path, kind, sym, ws):
    """(and three, and eight -- the node size the scaling bench runs at: the plan, the gather to rank 0 and the scatter with
    ranks whose tile counts differ)"""
    import torch.multiprocessing as mp
    wd = str(tmp_path)
    _dataset(wd, "toy.csv", 11, 4)
    _dataset(wd, "grid.csv", 75, 20)
    cwd = os.getcwd()
    os.chdir(wd)
    try:
        single, sstats = {"grid": _run_grid, "toy": _run_toy, "labels": _run_labels}[kind](wd, sym)
    finally:
        os.chdir(cwd)
    if kind == "grid":
        assert np.array_equal(single, _expected_grid(75, sym))
    out = str(tmp_path / "out")
    os.makedirs(out)
    for stale in ("results_t_Toy.csv", "results_g_Dev.csv", "results_l_Toy.csv"):
        if os.path.exists(os.path.join(wd, stale)):
            os.remove(os.path.join(wd, stale))
    mp.spawn(_worker, args=(ws, _free_port(), wd, sym, out, kind), nprocs=ws, join=True)
    D0 = np.load(os.path.join(out, "D_rank0.npy"))
    assert np.array_equal(D0, single), "rank 0 differs from the single-process result"
    for r in range(ws):
        assert np.array_equal(np.load(os.path.join(out, "S_rank%d.npy" % r)), np.array(sstats[:4])), "statistics differ on rank %d" % r
    # rank 0 alone wrote the results file: one header, one row
    res = [f for f in os.listdir(wd) if f.startswith("results_")]
    assert len(res) == 1 and len(open(os.path.join(wd, res[0])).read().strip().split("\n")) == 2


@pytest.mark.timeout(600)
def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it starts two ranks itself (a child
    torch.distributed.run, before any GPU call) and the line says n_gpus = ranks_seen = 2; a rank count that
    contradicts --gpus is refused.  --plan-only stops after the rendezvous and the deal: no GPU needed."""
    import json
    import subprocess
    env = dict(os.environ, ACX_BENCH_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    bench = os.path.join(ROOT, "bench.py")
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--plan-only", "--tracks", "640"], env=env, capture_output=True,
                       text=True, timeout=500)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2 and len(line["tiles_per_rank"]) == 2
    assert sum(line["tiles_per_rank"]) == line["tiles"] == 55          # 10 x 10 tile blocks, upper triangle
    c = line["cost_per_rank"]
    assert max(c) / (sum(c) / 2) <= 1.02
    bad = subprocess.run([sys.executable, bench, "--gpus", "2", "--plan-only"], env=dict(env, WORLD_SIZE="3", RANK="0"),
                         capture_output=True, text=True, timeout=120)
    assert bad.returncode != 0 and "WORLD_SIZE=3" in bad.stderr
    one = subprocess.run([sys.executable, bench, "--plan-only", "--tracks", "640"], env=env, capture_output=True, text=True,
                         timeout=120)
    assert one.returncode == 0 and json.loads(one.stdout.strip().splitlines()[-1])["n_gpus"] == 1


def test_bench_rank_identity_rule():
    """bench.shared_device_reason: what ends an N > 1 run before any measurement.  RCCL worlds need one GPU per rank -- the same
    (host, PCI bus id) twice, or a rank that cannot name its device, is refused with one line that names the ranks; different
    hosts may repeat a bus id; the gloo development mode shares GPUs on purpose unless ACX_BENCH_REQUIRE_DISTINCT=1."""
    sys.path.insert(0, ROOT)
    import bench
    def info(rank, host, bus):
        return {"rank": rank, "local_rank": rank, "device_index": 0, "host": host, "pci_bus_id": bus, "name": "AMD Instinct MI355X",
                "visible_devices": 1, "HIP_VISIBLE_DEVICES": str(rank), "error": None if bus else "RuntimeError: no device"}
    eight = [info(r, "node0", "0000:%02x:00.0" % (0x10 + 0x10 * r)) for r in range(8)]
    assert bench.shared_device_reason(eight, "nccl", 8) is None
    two_hosts = [info(0, "node0", "0000:26:00.0"), info(1, "node1", "0000:26:00.0")]
    assert bench.shared_device_reason(two_hosts, "nccl", 2) is None
    shared = [info(0, "node0", "0000:26:00.0"), info(1, "node0", "0000:46:00.0"), info(2, "node0", "0000:26:00.0")]
    why = bench.shared_device_reason(shared, "nccl", 3)
    assert why and "ranks 0 and 2" in why and "0000:26:00.0" in why and "\n" not in why
    assert bench.shared_device_reason(shared, "gloo", 3) is None
    os.environ["ACX_BENCH_REQUIRE_DISTINCT"] = "1"
    try:
        assert "ranks 0 and 2" in bench.shared_device_reason(shared, "gloo", 3)
    finally:
        del os.environ["ACX_BENCH_REQUIRE_DISTINCT"]
    nameless = [info(0, "node0", "0000:26:00.0"), info(1, "node0", None)]
    why = bench.shared_device_reason(nameless, "nccl", 2)
    assert why and "rank 1 cannot name its GPU" in why
    assert bench.shared_device_reason(nameless[:1], "nccl", 1) is None          # a world of one has nothing to share


def test_grid_plan_is_cost_balanced_on_ragged_lengths():
    """acx_grid_plan on ragged track lengths (cost ~ len_i * len_j varies 100 x): the dealt cost of
    the fullest rank is within 2 % of the mean; every pair belongs to exactly one tile."""
    from acoss_amd import _lib
    rng = np.random.default_rng(0)
    for n, ws, sym in [(5000, 8, True), (1200, 8, True), (900, 4, False), (164, 8, True), (333, 3, True)]:
        L = np.concatenate([rng.integers(150, 700, n - n // 10), rng.integers(1500, 5000, n // 10)])
        rng.shuffle(L)
        pl = _lib.grid_plan(L, _lib.ALGO_SERRA09, sym, world=ws, want_tiles=True)
        c = pl["cost_per_rank"]
        assert c.max() / c.mean() <= 1.02, (n, ws, c.max() / c.mean())
        total = sum(t.cost for t in pl["tiles"])
        S, Q = float(L.sum()), float((L.astype(np.float64) ** 2).sum())
        want = (S * S - Q) / 2 if sym else (S * S - Q)
        assert abs(total - want) <= 1e-9 * want
        seen = np.zeros((n, n), np.int32)
        for t in pl["tiles"]:
            seen[t.row0:t.row0 + t.rows, t.col0:t.col0 + t.cols] += 1
        if sym:
            assert np.all(seen[np.triu_indices(n, 1)] == 1)       # every unordered pair in exactly one tile
        else:
            assert np.all(seen == 1)
        # offsets of a rank's tiles are disjoint and dense
        for r in range(ws):
            mine = [t for t in pl["tiles"] if t.rank == r]
            off = 0
            for t in mine:
                assert t.offset == off
                off += t.rows * t.cols
            assert off == pl["floats_per_rank"][r]


def test_shard_bounds_cover_everything():
    from acoss_amd.dist import shard_bounds
    for n in (0, 1, 7, 100, 13366):
        for ws in (1, 2, 3, 8):
            spans = [shard_bounds(n, r, ws) for r in range(ws)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[k][1] == spans[k + 1][0] for k in range(ws - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def _worker_root_fails(rank, ws, port, workdir, out):
    sys.path.insert(0, ROOT)
    os.chdir(workdir)
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=ws)
    try:
        from acoss_amd import dist as adist
        from acoss_amd.algorithms.algorithm_template import CoverAlgorithm
        assert adist.on_root(lambda: {"n": 3}) == {"n": 3}
        # a dataset whose feature files do not exist: rank 0 fails while it builds the clique table -- every rank
        # must see an exception (and none may be left waiting in the broadcast)
        toy = CoverAlgorithm(os.path.join(workdir, "nofiles.csv"), name="Toy", datapath=workdir + "/missing/", shortname="nf")
        try:
            toy.get_all_clique_ids()
            msg = "no exception"
        except Exception as e:                       # noqa: BLE001
            msg = "%s: %s" % (type(e).__name__, e)
        with open(os.path.join(out, "rank%d.txt" % rank), "w") as f:
            f.write(msg)
        dist.barrier()                               # both ranks are still in step after the failure
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_root_failure_reaches_every_rank(tmp_path):
    """An exception in a rank-0-only section (acoss_amd.dist.on_root: clique table, statistics) is re-raised on every
    rank instead of leaving the others blocked in the broadcast."""
    import torch.multiprocessing as mp
    wd = str(tmp_path)
    with open(os.path.join(wd, "nofiles.csv"), "w") as f:
        f.write("work_id,track_id\nw0,t0\nw0,t1\n")
    out = str(tmp_path / "out")
    os.makedirs(out)
    mp.spawn(_worker_root_fails, args=(2, _free_port(), wd, out), nprocs=2, join=True)
    m0 = open(os.path.join(out, "rank0.txt")).read()
    m1 = open(os.path.join(out, "rank1.txt")).read()
    assert m0 != "no exception" and m1 != "no exception", (m0, m1)
    assert m1.startswith("RuntimeError: rank 0 failed in a rank-0-only section"), m1


def _worker_gather_refused(rank, ws, port, out, refuse):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=ws)
    try:
        from acoss_amd import dist as adist
        real = dist.gather
        if refuse == "raises":
            def broken(*a, **k):
                raise RuntimeError("this backend has no gather")
            dist.gather = broken
        elif refuse == "wrong":
            def wrong(t, gather_list=None, dst=0, **k):          # arrives, but every buffer holds rank 0's values
                real(t, gather_list=gather_list, dst=dst, **k)
                if gather_list is not None:
                    for g in gather_list[1:]:
                        g.copy_(gather_list[0])
            dist.gather = wrong
        stride = 37
        local = torch.arange(stride, dtype=torch.float32) + 1000.0 * rank
        got = adist.gather_tiles_device(local, stride)
        kind = adist.exchange_in_use()
        ok = True
        if rank == 0:
            want = torch.cat([torch.arange(stride, dtype=torch.float32) + 1000.0 * r for r in range(ws)])
            ok = torch.equal(got, want)
        else:
            ok = got is None
        with open(os.path.join(out, "rank%d.txt" % rank), "w") as f:
            f.write("%s %s" % (kind, ok))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("refuse,kind", [("no", "gather"), ("raises", "allgather"), ("wrong", "allgather")])
def test_refused_gather_falls_back_to_allgather_on_every_rank(tmp_path, refuse, kind):
    """The default exchange (torch.distributed.gather into views of one tensor) is probed once per process group on a few
    floats; a gather that raises or delivers wrong values makes EVERY rank take the all-gather instead (agreed through an
    all-reduce), and rank 0 still ends with every rank's buffer."""
    import torch.multiprocessing as mp
    out = str(tmp_path)
    mp.spawn(_worker_gather_refused, args=(3, _free_port(), out, refuse), nprocs=3, join=True)
    for r in range(3):
        assert open(os.path.join(out, "rank%d.txt" % r)).read() == "%s True" % kind
