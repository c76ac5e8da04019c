"""
CPU suite, part 5: the N > 1 path.  Two processes under torch.distributed (gloo,
127.0.0.1) shard the pair list, compute their slices with a CPU similarity() (a user
subclass), and all-gather the scores: every rank must end with the single-process matrix.
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


def _worker(rank, ws, port, workdir, sym, out):
    sys.path.insert(0, ROOT)
    os.chdir(workdir)
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=ws)
    try:
        D = _run_toy(workdir, sym)
        np.save(os.path.join(out, "D_rank%d.npy" % rank), D)
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
    return np.stack([np.array(toy.Ds["main"]), np.array(toy.Ds["aux"])])


@pytest.mark.parametrize("sym", [True, False])
def test_two_ranks_equal_one(tmp_path, sym):
    import torch.multiprocessing as mp
    from acoss_amd.featurestore import save_track
    wd = str(tmp_path)
    with open(os.path.join(wd, "toy.csv"), "w") as f:
        f.write("work_id,track_id\n")
        for k in range(11):
            f.write("w%d,t%d\n" % (k % 4, k))
            save_track(os.path.join(wd, "w%d/t%d.h5" % (k % 4, k)), {"label": "w%d" % (k % 4), "track_id": "t%d" % k})
    cwd = os.getcwd()
    os.chdir(wd)
    try:
        single = _run_toy(wd, sym)
    finally:
        os.chdir(cwd)
    out = str(tmp_path / "out")
    os.makedirs(out)
    mp.spawn(_worker, args=(2, _free_port(), wd, sym, out), nprocs=2, join=True)
    for r in range(2):
        D = np.load(os.path.join(out, "D_rank%d.npy" % r))
        assert np.array_equal(D, single), "rank %d differs" % r


def test_shard_bounds_cover_everything():
    from acoss_amd.dist import shard_bounds
    for n in (0, 1, 7, 100, 13366):
        for ws in (1, 2, 3, 8):
            spans = [shard_bounds(n, r, ws) for r in range(ws)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[k][1] == spans[k + 1][0] for k in range(ws - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
