"""
CoverAlgorithm: the harness every cover-id algorithm subclasses.  Mirrors the interface
of acoss/algorithms/algorithm_template.py (ctor :30-62, load_features :71-95,
get_all_clique_ids :97-119, similarity :121-140, all_pairwise :142-192,
cleanup_memmap :194-203, getEvalStatistics :205-290) so that user subclasses written
against acoss keep working, while device-backed subclasses hand the WHOLE pair list to
libacx in a few calls.

Differences from the reference, all deliberate:
  * the pair list is a numpy (K,2) array, not a Python list of tuples (1.1e8 tuples at
    N = 15 000 would need > 10 GB);
  * getEvalStatistics is vectorised (the reference's N x N Python double loop takes
    hours at N = 15 000) and uses a STABLE argsort so that ties have one defined order;
  * cleanup_memmap really removes the memmap files (the reference calls rmtree on a file
    and always fails);
  * the distance matrices are saved as <prefix>_Ds.h5 (libhdf5 / h5py) and, when small or when no HDF5 library exists, <prefix>_Ds.npz;
  * device-backed subclasses (those with a _grid() method) never build the pair list at all:
    libacx enumerates the N x N grid itself in cost-balanced tiles (acx_pair_grid; under
    torch.distributed, one process per GPU, acx_grid_run + ONE gather of the tile scores to rank 0,
    scattered into the matrices by rank 0).  User subclasses that implement similarity()
    themselves keep the reference's chunked pair-list loop (sharded by pair count over the ranks).
  * under torch.distributed rank 0 owns the result: it alone holds the filled matrices, writes
    the cache / results files and computes the statistics, which are broadcast to the others.
"""
import os
import warnings

import numpy as np

from .. import _lib
from .. import dist as _dist
from ..featurestore import load_track, load_matrices_h5, save_matrices_h5
from ..utils import create_dataset_filepaths

__all__ = ["CoverAlgorithm"]


class CoverAlgorithm(object):
    """
    Attributes
    ----------
    filepaths: list(string)   paths of all feature files of the dataset
    cliques: {label: set(int)}  cover cliques, indexing into filepaths
    N: int
    Ds: {similarity type: (N, N) float32 memmap}   pairwise similarity matrices
    """

    # how many chunks all_pairwise cuts the pair list into before calling similarity();
    # 45 is what the reference does (algorithm_template.py:172); device-backed subclasses
    # override this with 1 (whole list per call).
    n_chunks = 45

    def __init__(self, dataset_csv, name="Serra09", datapath="features_benchmark", shortname="full",
                 cachedir="cache", similarity_types=["main"]):
        self.name = name
        self.shortname = shortname
        self.cachedir = cachedir
        self.filepaths = create_dataset_filepaths(dataset_csv, root_audio_dir=datapath, file_format=".h5")
        self.cliques = {}
        self.N = len(self.filepaths)
        if not os.path.exists(cachedir):
            os.makedirs(cachedir, exist_ok=True)
        self.Ds = {}
        # the file names are fixed here: rank 0 owns the reference's names, other ranks (whose matrices
        # stay empty) get a suffix even if the process group is initialised or torn down later
        self._dmat_rank = _dist.world()[0]
        self._dmat_paths = {}
        for s in similarity_types:
            self.Ds[s] = np.memmap(self._dmat_path(s), shape=(self.N, self.N), mode="w+", dtype="float32")
        print("Initialized %s algorithm on %i songs in dataset %s" % (name, self.N, shortname))

    # ------------------------------------------------------------------ files / caches
    def get_cacheprefix(self):
        return "%s/%s_%s" % (self.cachedir, self.name, self.shortname)

    def _dmat_path(self, s):
        if s not in self._dmat_paths:
            rank = self._dmat_rank
            self._dmat_paths[s] = "%s_%s_dmat%s" % (self.get_cacheprefix(), s, "" if rank == 0 else ".rank%d" % rank)
        return self._dmat_paths[s]

    def _bind_collective_device(self, device):
        """Device-backed classes call this from their constructor with the GPU their libacx context will use (None:
        LOCAL_RANK's): under a process group, torch's collectives are bound to THAT device before the first of them
        runs (the clique-table broadcast can precede the first kernel).  A second, different device in one process is
        refused right here, in the constructor, on the rank that asked for it -- not in the middle of all_pairwise
        with the other ranks already waiting in a collective.  Without a process group nothing happens: no device is
        bound, no collective runs (dist.single() does import torch.distributed when torch is installed -- that creates no
        process group and touches no GPU)."""
        dev = int(os.environ.get("LOCAL_RANK", "0")) if device is None else int(device)
        if not _dist.single():
            try:
                import torch.distributed as tdist
                if tdist.get_backend() == "nccl":
                    _dist.bind_device(dev)
            except ImportError:
                pass
        return dev

    def owns_result(self):
        """True on the rank that holds the filled matrices (rank 0, or the only process).  The N x N
        post-processing steps (normalize_by_length, do_late_fusion) return at once everywhere else:
        the other ranks' matrices stay empty by design."""
        return _dist.world()[0] == 0

    def _register_label(self, i, label):
        self.cliques.setdefault(label, set()).add(int(i))

    def load_features(self, i):
        """Feature dict of track i; records its clique (feats['label']) as a side effect."""
        feats = load_track(self.filepaths[i])
        self._register_label(i, feats["label"])
        return feats

    def get_all_clique_ids(self, verbose=False):
        """Clique membership of every track, cached in <prefix>_clique_info.txt ("i,label").
        Under torch.distributed this is a COLLECTIVE (every rank must call it): rank 0 reads the cache
        file -- or builds it from the feature files -- and broadcasts the table; the other ranks touch
        neither the feature files nor the cache (no shared file system is assumed)."""
        path = "%s_clique_info.txt" % self.get_cacheprefix()

        def build():
            table = []
            if os.path.exists(path):
                with open(path) as fin:
                    for line in fin:
                        i, label = line.split(",", 1)
                        table.append((int(i), label.strip()))
            else:
                # written to a temporary name and renamed: nobody ever sees half a file
                tmp = "%s.tmp%d" % (path, os.getpid())
                with open(tmp, "w") as fout:
                    for i in range(len(self.filepaths)):
                        feats = CoverAlgorithm.load_features(self, i)
                        if verbose:
                            print(i)
                        fout.write("%i,%s\n" % (i, feats["label"]))
                        table.append((i, str(feats["label"]).strip()))
                os.replace(tmp, path)
            return table
        table = _dist.on_root(build)          # a failure on rank 0 is re-raised on every rank (no rank is left waiting)
        for i, label in table:
            self._register_label(int(i), label)

    # ------------------------------------------------------------------ pairwise
    def similarity(self, idxs):
        """idxs: (K,2) int array.  Writes Ds[type][i, j] for every row; the return value is
        ignored.  The base class stores zeros."""
        idxs = np.asarray(idxs)
        self.Ds["main"][idxs[:, 0], idxs[:, 1]] = 0.0

    @staticmethod
    def pair_list(N, symmetric):
        """(K,2) int64 array in itertools.combinations / permutations order
        (algorithm_template.py:168-171)."""
        if symmetric:
            i, j = np.triu_indices(N, 1)
        else:
            i, j = np.nonzero(~np.eye(N, dtype=bool))
        return np.stack([i, j], axis=1).astype(np.int64)

    def all_pairwise(self, parallel=0, n_cores=12, symmetric=False, precomputed=False):
        """All pairwise comparisons.  Device-backed classes ignore `parallel` / `n_cores`: their fan-out unit is the GPU
        (one process per GPU under torch.distributed).  User subclasses that bring their own CPU `similarity()` get the
        reference's behaviour: `parallel=1` spreads the 45 chunks over `n_cores` joblib workers (single process group only)."""
        npz = "%s_Ds.npz" % self.get_cacheprefix()
        h5 = "%s_Ds.h5" % self.get_cacheprefix()          # the reference's cache file (algorithm_template.py:163-166,192)
        if precomputed:
            loaded = {}
            if not self.owns_result():
                pass                                           # the cache belongs to rank 0, like the result
            elif os.path.exists(npz):
                with np.load(npz) as z:
                    loaded = {s: z[s] for s in z.files}
            else:
                loaded = load_matrices_h5(h5)                  # a cache written by acoss itself (needs h5py)
            for s, M in loaded.items():
                if s in self.Ds:
                    self.Ds[s][:] = M
                else:
                    self.Ds[s] = M
            self.get_all_clique_ids()
            return
        rank, ws = _dist.world()
        if hasattr(self, "_grid"):
            self._all_pairwise_grid(symmetric)
        else:
            pairs = self.pair_list(self.N, symmetric)
            lo, hi = _dist.shard_bounds(len(pairs), rank, ws)
            mine = pairs[lo:hi]
            chunks = [c for c in np.array_split(mine, max(1, min(self.n_chunks, len(mine)))) if len(c)]
            if parallel and n_cores != 1 and ws == 1 and self._joblib_fanout(chunks, n_cores):
                pass                                           # the reference's joblib fan-out (algorithm_template.py:172-177)
            else:
                for chunk in chunks:
                    self.similarity(chunk)
            if not _dist.single():
                keys = list(self.Ds.keys())
                local = np.stack([np.asarray(self.Ds[s][mine[:, 0], mine[:, 1]]) for s in keys], axis=1)
                full = _dist.gather_scores(local, len(pairs))
                if rank == 0:
                    for c, s in enumerate(keys):
                        self.Ds[s][pairs[:, 0], pairs[:, 1]] = full[:, c]
            if symmetric and rank == 0:
                for s in self.Ds:
                    self.Ds[s] += self.Ds[s].T
        # (a collective decision: a rank whose labels were injected must not skip the broadcast the others wait in)
        if _dist.any_rank(not self.cliques):
            self.get_all_clique_ids()
        if rank == 0:
            self._save_results_cache(npz, h5)

    # a result set up to this size is cached in BOTH formats (the .npz needs no HDF5 library to read back); above it only in
    # the reference's own -- at 15 000 tracks a second copy costs seconds per plane inside all_pairwise
    NPZ_CACHE_BELOW = 256 << 20

    def _save_results_cache(self, npz, h5):
        """The reference's dd.io.save("<prefix>_Ds.h5", self.Ds) (algorithm_template.py:192): <prefix>_Ds.h5 when an HDF5
        backend exists, <prefix>_Ds.npz when none does or the matrices are small; `all_pairwise(precomputed=True)` reads
        whichever is there (a stale .npz of an earlier, smaller run is removed so that it cannot shadow the new .h5)."""
        wrote_h5 = save_matrices_h5(h5, self.Ds)
        total = sum(int(np.asarray(self.Ds[s]).nbytes) for s in self.Ds)
        if not wrote_h5 or total <= self.NPZ_CACHE_BELOW:
            np.savez(npz, **{s: np.asarray(self.Ds[s]) for s in self.Ds})
        elif os.path.exists(npz):
            os.remove(npz)

    def _joblib_fanout(self, chunks, n_cores):
        """User subclasses with their own CPU `similarity()` (README "how to add an algorithm"): `parallel=1` fans the 45
        chunks out over joblib worker processes like the reference does (algorithm_template.py:172-177) -- every worker
        gets a pickled copy of `self` whose `Ds` memmaps joblib re-opens on the same files, so the scores land in the
        owner's matrices.  Returns False (caller runs the chunks serially) when joblib is missing or a matrix is not
        file-backed.  Device-backed classes never come here: their fan-out unit is the GPU."""
        try:
            from joblib import Parallel, delayed
        except ImportError:
            return False
        if not all(isinstance(D, np.memmap) for D in self.Ds.values()):
            return False
        for D in self.Ds.values():
            D.flush()
        Parallel(n_jobs=n_cores, verbose=0)(delayed(self.similarity)(c) for c in chunks)
        return True

    def _all_pairwise_grid(self, symmetric):
        """Device-backed classes: `self._grid()` -> (context with the pool uploaded, ACX_ALGO_*, params
        struct, similarity types in plane order).  One GPU: acx_pair_grid straight into the memmaps.
        N GPUs: every rank runs its tiles into a device buffer, one gather to rank 0, rank 0 scatters."""
        ctx, algo, params, keys = self._grid()
        planes = [self.Ds[k] for k in keys]
        rank, ws = _dist.world()
        if _dist.single():
            ctx.pair_grid(algo, symmetric, params, planes, mirror=symmetric)
            return
        import torch
        # torch's collectives (and its allocator) must work on the GPU libacx works on: one process, one GPU
        if ctx.torch_device().type == "cuda":
            _dist.bind_device(ctx.device)
        lengths = ctx.pool_lengths(algo)
        plan = _lib.grid_plan(lengths, algo, symmetric, world=ws)
        stride = int(max(1, plan["floats_per_rank"].max()))
        # (torch.empty launches nothing: libacx works on its own stream and zeroes what it owns)
        local = torch.empty(stride, dtype=torch.float32, device=ctx.torch_device())
        ctx.grid_run(plan["spec"], params, rank, local.data_ptr())
        gathered = _dist.gather_tiles(local, stride)
        if rank == 0:
            _lib.grid_scatter(lengths, plan["spec"], gathered, stride, planes, mirror=symmetric)

    def cleanup_memmap(self):
        """Remove the memmap files behind the similarity matrices."""
        for s in list(self.Ds.keys()):
            path = self._dmat_path(s)
            if isinstance(self.Ds[s], np.memmap):
                self.Ds[s].flush()
            try:
                if os.path.exists(path):
                    os.remove(path)
            except OSError:
                print("Could not clean-up automatically.")

    # ------------------------------------------------------------------ evaluation
    def getEvalStatistics(self, similarity_type, topsidx=[1, 10, 100, 1000]):
        """MR, MRR, MDR, MAP and Top-k of one similarity matrix; appends a row to
        results_<shortname>_<name>.csv.  Same definitions as the reference (:205-290),
        including MRR's division by ALL N songs and the %.3g CSV format.
        Under torch.distributed this is a COLLECTIVE: rank 0 (the owner of the matrices) evaluates and
        writes the CSV, the tuple is broadcast and returned on every rank -- so every rank must call it
        (`if rank == 0: algo.getEvalStatistics(...)` would leave rank 0 waiting in the broadcast)."""
        rank, ws = _dist.world()

        def evaluate():
            D = np.array(self.Ds[similarity_type], dtype=np.float32)
            return eval_statistics(D, [sorted(self.cliques[s]) for s in self.cliques], topsidx)
        MR, MRR, MDR, MAP, tops = _dist.on_root(evaluate)
        if rank != 0:
            return MR, MRR, MDR, MAP, tops
        print("%s %s STATS\n-------------------------\nMR = %.3g\nMRR = %.3g\nMDR = %.3g\nMAP = %.3g"
              % (self.name, similarity_type, MR, MRR, MDR, MAP))
        for t, v in zip(topsidx, tops):
            print("Top-%i: %i" % (t, v))
        resultsfile = "results_%s_%s.csv" % (self.shortname, self.name)
        if not os.path.exists(resultsfile):
            with open(resultsfile, "w") as fout:
                fout.write("name, MR, MRR, MDR, MAP")
                for t in topsidx:
                    fout.write(",Top-%i" % t)
                fout.write("\n")
        with open(resultsfile, "a") as fout:
            fout.write("%s_%s," % (self.name, similarity_type))
            fout.write("%.3g, %.3g, %.3g, %.3g" % (MR, MRR, MDR, MAP))
            for t in tops:
                fout.write(", %.3g" % t)
            fout.write("\n")
        return MR, MRR, MDR, MAP, tops


def eval_statistics(D, cliques, topsidx=(1, 10, 100, 1000), row_block=1024, count_max_clique=24):
    """Vectorised evaluation.  `cliques`: list of lists of track indices (dict insertion
    order).  Rows are reordered so that cliques are contiguous, largest first; the
    diagonal is -inf; every row is ranked by descending score in STABLE order (ties: the lower
    index first); for every song of a clique of size >= 2 the 1-based ranks of its clique mates are
    collected -- by counting for cliques of up to `count_max_clique` songs, by a stable argsort
    otherwise (and for rows that hold NaN); both give the same numbers."""
    D = np.array(D, dtype=np.float32)
    N = D.shape[0]
    Ks = np.array([len(c) for c in cliques])
    order = np.argsort(-Ks, kind="stable")
    Ks = Ks[order]
    cl = [list(cliques[i]) for i in order]
    idx = np.array([t for c in cl for t in c], dtype=np.int64)
    D = D[idx, :][:, idx]
    np.fill_diagonal(D, -np.inf)
    starts = np.concatenate([[0], np.cumsum(Ks)])[:-1]
    row_start = np.repeat(starts, Ks)            # first column of the row's own clique
    row_K = np.repeat(Ks, Ks)
    n_eval = int(np.sum(Ks[Ks >= 2]))            # cliques are sorted: evaluated rows come first
    ranks = np.full(N, np.nan)
    allmap = np.full(N, np.nan)
    col = np.arange(N, dtype=np.int64)
    for r0 in range(0, n_eval, row_block):
        r1 = min(n_eval, r0 + row_block)
        Db = D[r0:r1]
        Kb = row_K[r0:r1]
        kmax = int(Kb.max())
        if kmax <= count_max_clique and not np.isnan(Db).any():
            # Ranks by COUNTING instead of sorting: only the positions of a row's clique mates are needed, and the
            # position of column c in the stable descending order is 1 + #(cells above D[i, c]) + #(equal cells left of
            # c).  2 (K - 1) passes over the block instead of an N log N sort per row (6 x faster at N = 15 000, K = 5).
            rows_i = np.arange(r0, r1)
            pos = np.full((r1 - r0, kmax), np.inf)
            for m in range(kmax):
                c = row_start[r0:r1] + m
                ok = (m < Kb) & (c != rows_i)
                if not ok.any():
                    continue
                cc = np.where(ok, c, 0)
                v = Db[np.arange(r1 - r0), cc][:, None]
                p = 1.0 + np.count_nonzero(Db > v, axis=1)
                eq = Db == v
                if np.count_nonzero(eq) > r1 - r0:               # ties beyond the cell itself: the left ones come first
                    p = p + np.count_nonzero(eq & (col[None, :] < cc[:, None]), axis=1)
                pos[ok, m] = p[ok]
            pos.sort(axis=1)                                     # ascending; missing mates (+inf) last
            nm = (Kb - 1)[:, None]
            t = np.arange(1, kmax + 1, dtype=np.float64)[None, :]
            contrib = np.where(t <= nm, t / pos, 0.0)
            ranks[r0:r1] = pos[:, 0]
            allmap[r0:r1] = contrib.sum(axis=1) / (Kb - 1)
            continue
        srt = np.argsort(-Db, axis=1, kind="stable")
        member = (srt >= row_start[r0:r1, None]) & (srt < (row_start[r0:r1] + row_K[r0:r1])[:, None])
        rr, kk = np.nonzero(member)              # row-major: per row, ascending rank position
        # the last member of every row is the song itself (-inf sorts last): drop it
        first = np.concatenate([[0], np.cumsum(row_K[r0:r1])])[:-1]
        last = first + row_K[r0:r1] - 1
        keep = np.ones(len(rr), dtype=bool)
        keep[last] = False
        pos = (kk + 1)[keep].astype(np.float64)
        rows = rr[keep]
        within = np.arange(len(rr)) - np.repeat(first, row_K[r0:r1])
        num = (within + 1)[keep].astype(np.float64)
        ranks[r0:r1] = pos[np.searchsorted(rows, np.arange(r1 - r0))]
        sums = np.bincount(rows, weights=num / pos, minlength=r1 - r0)
        allmap[r0:r1] = sums / (row_K[r0:r1] - 1)
    if n_eval == 0:
        warnings.warn("no clique with at least two songs")
    MAP = float(np.nanmean(allmap)) if n_eval else float("nan")
    ranks = ranks[~np.isnan(ranks)]
    MR = float(np.mean(ranks)) if len(ranks) else float("nan")
    MRR = float(1.0 / N * np.sum(1.0 / ranks))
    MDR = float(np.median(ranks)) if len(ranks) else float("nan")
    tops = np.array([np.sum(ranks <= t) for t in topsidx], dtype=np.float64)
    return MR, MRR, MDR, MAP, tops
