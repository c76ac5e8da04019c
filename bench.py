#!/usr/bin/env python3
"""
bench.py -- headline benchmark: track-pairs/sec of the Serra09 chain
(OTI -> embedded CSM -> mutual-kappa thresholds -> Qmax) on BASELINE.json configs[2]: a synthetic
pool of 5 000 tracks x 2000 pooled HPCP frames, the pair grid tiled over the GPUs.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

The 5000 x 5000 grid is cut into 64 x 64 track tiles and dealt to the N ranks by libacx's
cost-balanced plan (acx_grid_plan, the scheduler all_pairwise uses).  A "step" is one pass of the
hot path over the next TILES_PER_STEP tiles of every rank's deal (2 tiles = 8192 pairs per GPU per
step, different tracks every step; weak scaling: per-GPU work is fixed) into a device buffer
(acx_grid_run), followed, for N > 1, by the one collective of the path: an all-gather of the tile
scores over RCCL on the device buffers themselves.  The pool is resident in HBM before the timed
region.  Rank 0 prints ONE JSON line.

Also in the line:
  roofline      the dominant kernel (accumulated HIP-event time on the library's own stream)
                against the byte model SURVEY.md 8d fixes (10 B/cell: f32 distance matrix written
                once + read once, 1-byte recurrence matrix written once + read once) and the
                8 TB/s HBM peak -- a THROUGHPUT PROXY: the production pipeline keeps the distance
                matrix out of HBM, so the kernel is bound by SIMD issue (VALU + f32 MFMA share the
                issue port), which `bound` / `real_bound` say (utilisations from the committed
                rocprofv3 counter run, profiles/real_bound.json); `traffic` = HBM bytes per
                launch from the committed PMC run (profiles/pmc_traffic.json), scaled by cells.
  cpu_baseline  the CPU oracle (a C port of the same chain) on ALL host cores through a process
                fan-out over max(45, cores) chunks (the reference's joblib scheme,
                algorithm_template.py:172-177, has 45), on a bounded sample of the same workload;
                its scores are checked bit-for-bit against the GPU's.  Timed BEFORE the GPU is
                initialised (the worker processes are forks; the pool is closed and joined first).
  phases_s      wall seconds of every untimed phase (pool generation, the two CPU figures, context +
                upload, warm-up, self-check, every `other` leg, total); the same stamps go to stderr as
                they happen.  Every untimed phase has a wall budget; a leg that overruns is reported
                as an error string, never at the cost of the line.
  fence         what brackets the timed region: one GPU -> hipDeviceSynchronize through libacx (the
                process imports neither torch nor pandas: cold shared-library page-in on a fresh box
                was what made round 3's driver run take 27 minutes); N > 1 -> dist.barrier +
                torch.cuda.synchronize.
  other         companion legs outside the timed region (bench_other.py): Serra09 on covers80-shaped
                lengths, the opt-in f16x2 Gram beside the exact one, LateFusionChen's Qmax + Dmax, SiMPle,
                EarlyFusion (its default two-term fp16 GEMMs with the three-term bf16 ones beside them).
`--strong` runs the path as ONE job instead (strong scaling, the whole grid, one final all-gather).
"""
import argparse
import contextlib
import faulthandler
import json
import os
import signal
import subprocess
import sys
import threading
import time

_T_START = time.perf_counter()


def _cpus_allowed():
    """CPUs this process may use (affinity mask capped by the cgroup CPU quota) -- the same rule as
    acoss_amd.utils.effective_cpus, restated here because it has to run BEFORE numpy is imported."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max",):
        try:
            with open(path) as f:
                q, per = f.read().split()[:2]
            if q != "max":
                n = max(1, min(n, int(float(q) / float(per) + 0.999)))
        except (OSError, ValueError):
            try:
                with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                    q = float(f.read())
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                    per = float(f.read())
                if q > 0:
                    n = max(1, min(n, int(q / per + 0.999)))
            except (OSError, ValueError):
                pass
    return n


# Every BLAS / OpenMP pool of this process (numpy's OpenBLAS, torch's CPU pool, MKL of an Anaconda numpy) is sized
# BEFORE the libraries load: a container allowed 16 of a host's 256 hardware threads otherwise starts 256 spinning
# workers per pool, and the untimed host phases (synthetic sets, the numpy oracles of the `other` legs) crawl.
HOST_THREADS = max(1, min(_cpus_allowed(), 16))
for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS", "NUMEXPR_NUM_THREADS", "VECLIB_MAXIMUM_THREADS"):
    os.environ.setdefault(_v, str(HOST_THREADS))

import numpy as np  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# wall budgets of the UNTIMED phases (seconds).  A phase that runs past its budget is abandoned and reported as
# null / {"error": ...}; the headline line is printed regardless (and by a watchdog if an `other` leg hangs).
BUDGET_CPU_1CORE_S = 10.0          # work budget of the one-core oracle figure
BUDGET_CPU_ALL_S = 15.0            # work budget of the all-core oracle figure
HARD_CPU_POOL_S = 120.0            # hard wall limit of one pool round (pilot or timed)
HARD_OTHER_LEG_S = 150.0           # hard wall limit of one `other` leg


class Clock(object):
    """Phase clock: `with clock.phase("name")` records the wall seconds of an untimed phase in `phases` (the
    `phases_s` object of the JSON line) and stamps start / end on stderr as they happen, so that a run that is
    killed from outside still shows where its time went."""

    def __init__(self, verbose=True):
        self.phases = {}
        self.verbose = verbose

    def stamp(self, msg):
        if self.verbose:
            sys.stderr.write("[bench +%7.1fs] %s\n" % (time.perf_counter() - _T_START, msg))
            sys.stderr.flush()

    @contextlib.contextmanager
    def phase(self, name):
        self.stamp("%s ..." % name)
        t0 = time.perf_counter()
        try:
            yield
        finally:
            dt = time.perf_counter() - t0
            self.phases[name] = round(self.phases.get(name, 0.0) + dt, 3)
            self.stamp("%s: %.2f s" % (name, dt))


class BudgetExceeded(Exception):
    pass


@contextlib.contextmanager
def wall_limit(seconds, what):
    """Abandon the enclosed (main-thread) work with BudgetExceeded once `seconds` of wall time are gone: SIGALRM,
    delivered between two bytecodes (a running libacx / BLAS call finishes first -- they are all short)."""
    def on_alarm(signum, frame):
        raise BudgetExceeded("%s ran past its %.0f s wall budget" % (what, seconds))
    old = signal.signal(signal.SIGALRM, on_alarm)
    signal.setitimer(signal.ITIMER_REAL, seconds)
    try:
        yield
    finally:
        signal.setitimer(signal.ITIMER_REAL, 0)
        signal.signal(signal.SIGALRM, old)

T_FRAMES = 2000
N_TRACKS = 5000
TILE = 64
TILES_PER_STEP = 2             # per rank: 2 x 64 x 64 = 8192 pairs per GPU per step
# the sub-grid of the N > 1 run's `strong` leg: 1280 tracks = 210 tiles = 818 560 pairs (~1 s per rank at 8 GPUs, ~8 s at one)
STRONG_LEG_TRACKS = int(os.environ.get("ACX_BENCH_STRONG_TRACKS", "1280"))
# ... and at N = 1 (no collective) a smaller one: 640 tracks = 204 480 pairs, ~2 s -- the whole-grid figure (plan, every tile, the copy
# to the host, scatter + mirror, the matrix checked against the pair list) as something the DRIVER's run observes, not a builder's record
STRONG_LEG_TRACKS_ONE = int(os.environ.get("ACX_BENCH_STRONG_TRACKS", "640"))
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s
F32_MFMA_PEAK_TF = 157.3       # MI355X_MICROARCH.md: f32-input MFMA = f32 vector rate
M_STACK = 9

# algorithmic bytes per matrix cell by kernel (DESIGN.md "Roofline model").
# band_kernel is launched twice per batch (role 1: column thresholds, role 0: row thresholds +
# recurrence bitmap); each launch carries one 4 B/cell pass over the distance matrix of the model
# (write / read-back); qmax_bits_kernel = the DP over the recurrence plot (write + read, 1 B each).
ALGO_BYTES_PER_CELL = {"band_kernel": 4.0, "csm_long_kernel": 4.0, "rowsel_long_kernel": 4.0, "qmax_bits_kernel": 2.0,
                       "oti_kernel": 0.0, "norms_kernel": 0.0}

_CPU = {}


def kernel_source_sha16():
    """Hash of the Serra09 kernel sources (scripts/summarise_profile.py stamps the committed counter records with it)."""
    import hashlib
    h = hashlib.sha256()
    for f in ("serra09_kernels.hpp", "serra09_band2_kernels.hpp", "acx_band.hip", "Makefile"):
        with open(os.path.join(ROOT, "acoss_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def chain_bytes_per_pair(Tq, Tr, m=M_STACK):
    Mq, Mr = Tq - m, Tr - m
    return 10.0 * Mq * Mr + 48.0 * (Tq + Tr) + 4.0


def make_pool(n, T, seed=1234):
    """SURVEY 8d 'rand' set: i.i.d. U[0,1) frames, each divided by its max."""
    rng = np.random.default_rng(seed)
    frames = rng.random((n * T, 12), dtype=np.float32)
    frames /= frames.max(axis=1, keepdims=True)
    return frames, np.arange(n + 1, dtype=np.int64) * T


def _cpu_chunk(chunk):
    import oracle
    return oracle.serra09_pairs(_CPU["frames"], _CPU["offsets"], chunk)


def _cpu_init():
    import oracle
    oracle.lib()


def effective_cpus():
    from acoss_amd.utils import effective_cpus as f
    return f()


def _cpu_triad(_):
    """STREAM triad a = b + s c on 3 x 32 MB f64 per worker for ~1 s: bytes moved per second by this worker."""
    n = 4 << 20
    b, c = np.ones(n), np.full(n, 2.0)
    a = np.empty(n)
    t0 = time.perf_counter()
    reps = 0
    while time.perf_counter() - t0 < 1.0:
        np.multiply(c, 3.0, out=a)
        np.add(a, b, out=a)
        reps += 1
    return reps * 5 * 8 * n / (time.perf_counter() - t0)      # multiply: read c, write a; add: read a, b, write a


def cpu_baseline(frames, offsets, pairs, n_tracks, clock):
    """SURVEY 8d's two CPU figures, both from the oracle (a C port of the chain, gcc -O3) and both timed BEFORE
    the GPU is initialised (the workers are forks):
      (1) one process, one thread, on randomly chosen pairs of the same synthetic pool (seeded), four at a time until
          BUDGET_CPU_1CORE_S seconds of work are done or 256 pairs (the count 8d names; at ~0.2 s per pair that is
          ~50 s, so the time budget is what ends the loop and the count is stated in `sample`) -> value_1core;
      (2) every host core through a process fan-out over max(45, cores) chunks (the reference's joblib scheme,
          algorithm_template.py:172-177, has 45) -> value.  Bounded: a pilot round of one pair per chunk measures
          the rate under full load, the timed round is sized from it to about BUDGET_CPU_ALL_S seconds.
    Every pool round has a hard wall limit (HARD_CPU_POOL_S): past it the pool is terminated and the figure is null.
    The pool is closed and joined before this function returns, i.e. before any GPU call of the process.
    If (2) is less than half of cores x (1), a one-second STREAM triad on the same workers is reported beside it
    (the oracle streams a 16 MB distance matrix per pair three times; hundreds of copies of it share the memory
    controllers)."""
    import multiprocessing as mp
    import oracle
    with clock.phase("oracle_build_load"):
        oracle.lib()
    _CPU["frames"], _CPU["offsets"] = frames, offsets
    cores = effective_cpus()
    with clock.phase("cpu_1core"):
        rng = np.random.default_rng(8)
        rnd = rng.integers(0, n_tracks, (2 * 256, 2))
        rnd = np.ascontiguousarray(rnd[rnd[:, 0] != rnd[:, 1]][:256].astype(np.int32))
        parts, n1, t_1core = [], 0, 0.0
        while n1 < len(rnd) and t_1core < BUDGET_CPU_1CORE_S:
            t0 = time.perf_counter()
            parts.append(oracle.serra09_pairs(frames, offsets, rnd[n1:n1 + 4]))
            t_1core += time.perf_counter() - t0
            n1 += len(parts[-1])
        rnd_scores = np.concatenate(parts)
        v1 = n1 / t_1core
    stream, vall, dt, n, note = None, None, None, 0, ""
    sample, scores = rnd[:0], np.zeros(0, np.float32)
    nchunks = max(45, cores)                                      # the reference's joblib scheme has 45 chunks
    with clock.phase("cpu_all_cores"):
        pool = mp.get_context("fork").Pool(cores, initializer=_cpu_init)
        try:
            pilot = np.ascontiguousarray(pairs[:nchunks])
            t0 = time.perf_counter()
            parts = pool.map_async(_cpu_chunk, [pilot[k:k + 1] for k in range(len(pilot))], chunksize=1).get(HARD_CPU_POOL_S)
            dt = time.perf_counter() - t0
            clock.stamp("cpu_all_cores pilot: %d pairs in %.2f s" % (len(pilot), dt))
            sample, scores = pilot, np.concatenate(parts)
            per_chunk = int(BUDGET_CPU_ALL_S / dt) if dt > 0 else 0
            if per_chunk >= 2:                                     # room for a longer timed round
                n = min(len(pairs), nchunks * per_chunk)
                n -= n % nchunks
                big = np.ascontiguousarray(pairs[:n])
                chunks = [c for c in np.array_split(big, nchunks) if len(c)]
                t0 = time.perf_counter()
                parts = pool.map_async(_cpu_chunk, chunks, chunksize=1).get(HARD_CPU_POOL_S)
                dt = time.perf_counter() - t0
                sample, scores = big, np.concatenate(parts)
            n = len(sample)
            vall = n / dt
            if vall < 0.5 * cores * v1:
                stream = round(sum(pool.map_async(_cpu_triad, range(cores), chunksize=1).get(30.0)) / 1e9, 1)
            pool.close()
        except mp.TimeoutError:
            pool.terminate()
            note = "; a pool round ran past its %.0f s wall limit and was terminated: all-core figure abandoned" % HARD_CPU_POOL_S
            clock.stamp("cpu_all_cores: pool round past its wall limit, terminated")
            if vall is None:
                sample, scores = rnd[:0], np.zeros(0, np.float32)
        finally:
            pool.join()
    model = ""
    try:
        for line in subprocess.run(["lscpu"], capture_output=True, text=True, timeout=10).stdout.splitlines():
            if line.startswith("Model name"):
                model = line.split(":", 1)[1].strip()
    except (OSError, subprocess.SubprocessError):
        pass
    check = np.concatenate([sample, rnd[:n1]]), np.concatenate([scores, rnd_scores])
    return check[0], check[1], {
        "value": None if vall is None else round(vall, 3), "unit": "track-pairs/s", "cores": cores, "kind": "port",
        "sample": "first %d pairs of the first timed step's tiles of the same workload (T=%d), C oracle (gcc -O3), %d worker "
                  "processes over %d chunks (reference scheme: 45 joblib chunks), %s s; value_1core: %d randomly chosen "
                  "pairs of the pool (seed 8) in one process, %.1f s (SURVEY 8d names 256 pairs: ~50 s at this rate; the loop "
                  "stops at a %.0f s work budget); all scores bit-identical to the GPU's%s"
                  % (n, T_FRAMES, cores, nchunks, "%.1f" % dt if dt else "-", n1, t_1core, BUDGET_CPU_1CORE_S, note),
        "value_1core": round(v1, 3),
        "parallel_efficiency": None if vall is None else round(vall / (cores * v1), 3),
        "stream_triad_gbs_all_workers": stream, "cpu_model": model, "host_cpus": os.cpu_count(),
        "cores_note": "cores = CPUs this process may use (affinity mask and cgroup CPU quota), one worker process each"}


def launch_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves -- torch.distributed.run as a
    CHILD process, before this process has made any GPU call (it never does), one rank per GPU on
    127.0.0.1 -- relay what the ranks print (rank 0's ONE JSON line included) and return the child's exit
    code."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env)


def json_only_stdout():
    """Keep the process's stdout for the ONE JSON line: everything else that writes to fd 1 from here on -- RCCL's
    version banner (printf from librccl at communicator creation), Python prints of imported modules -- goes to
    stderr.  Returns the stream the JSON line is written to."""
    sys.stdout.flush()
    keep = os.dup(1)
    os.dup2(2, 1)
    return os.fdopen(keep, "w")


def derive_bound(real_bound):
    """The unit that is busiest in the committed counter record of the kernel: "valu" | "mfma" | "lds" | "hbm".
    None when no record of this build's kernel sources exists (then `bound` falls back to the byte model's "hbm")."""
    if not real_bound:
        return None
    units = {"valu": real_bound.get("valu_busy"), "mfma": real_bound.get("mfma_busy"),
             "lds": real_bound.get("lds_busy"), "hbm": real_bound.get("hbm_util")}
    units = {k: v for k, v in units.items() if isinstance(v, (int, float))}
    return max(units, key=units.get) if units else None


def rank_identity(rank, local_rank, device_index):
    """What one rank says about the GPU it holds (libacx acx_device_info: no torch needed): gathered into the line's
    `ranks` array so that a multi-GPU figure proves which devices produced it."""
    import socket
    from acoss_amd import _lib
    info = {"rank": rank, "local_rank": local_rank, "device_index": device_index, "host": socket.gethostname(),
            "HIP_VISIBLE_DEVICES": os.environ.get("HIP_VISIBLE_DEVICES"), "ROCR_VISIBLE_DEVICES": os.environ.get("ROCR_VISIBLE_DEVICES")}
    try:
        info.update(_lib.device_info(device_index))
    except Exception as e:                      # noqa: BLE001 -- reported, and treated as "cannot prove its device" below
        info.update({"pci_bus_id": None, "name": None, "visible_devices": 0, "error": "%s: %s" % (type(e).__name__, e)})
    return info


def gather_identities(info, collective, dist, backend, local_rank):
    if not collective:
        return [info]
    box = [None] * dist.get_world_size()
    if backend == "nccl":
        import torch
        torch.cuda.set_device(local_rank)
    dist.all_gather_object(box, info)
    return box


def shared_device_reason(infos, backend, world):
    """One line naming the ranks that hold the SAME GPU (same host, same PCI bus id), or a rank that cannot name its GPU;
    None when every rank has a device of its own.  Enforced for RCCL worlds (one communicator rank per device; a figure
    from ranks sharing a GPU is not a multi-GPU figure) and, ACX_BENCH_REQUIRE_DISTINCT=1, for any backend -- the gloo
    development mode deliberately shares devices and says so in `collectives`."""
    if world < 2 or not (backend == "nccl" or os.environ.get("ACX_BENCH_REQUIRE_DISTINCT") == "1"):
        return None
    seen = {}
    for i in infos:
        if not i.get("pci_bus_id"):
            return "bench: rank %d cannot name its GPU (%s): no multi-GPU figure without proof of the devices" % (i["rank"], i.get("error"))
        key = (i["host"], i["pci_bus_id"])
        if key in seen:
            return ("bench: ranks %d and %d hold the same GPU (%s on %s; %d rank(s), rank %d sees %d device(s), HIP_VISIBLE_DEVICES=%s): "
                    "one rank per GPU or no multi-GPU figure" % (seen[key], i["rank"], i["pci_bus_id"], i["host"], world, i["rank"],
                                                                  i["visible_devices"], i["HIP_VISIBLE_DEVICES"]))
        seen[key] = i["rank"]
    return None


def rccl_version(collective, backend):
    if not (collective and backend == "nccl"):
        return None
    try:
        import torch
        return ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception:                           # noqa: BLE001
        return None


def init_torch(clock, local_rank, world, backend, collective):
    # a forced one-rank world must still EXECUTE the exchange (torch.distributed.gather over RCCL), not return its own buffer:
    # acoss_amd.dist treats a world of one as "no collective needed" unless this is set
    if world == 1:
        os.environ["ACX_GRID_VIA_COLLECTIVE"] = "1"
    with clock.phase("import_torch"):
        import torch
        import torch.distributed as dist
        torch.set_num_threads(HOST_THREADS)
    with clock.phase("gpu_init"):
        # (a launcher may hand every rank ONE visible device, HIP_VISIBLE_DEVICES = its own: LOCAL_RANK is then not a device index.
        #  Whether two ranks ended up on one GPU is decided from the PCI bus ids, shared_device_reason, not assumed here)
        local_rank = local_rank % max(1, torch.cuda.device_count())
        torch.cuda.set_device(local_rank)
        if collective:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            dist.init_process_group(backend)
        torch.zeros(1, device=torch.device("cuda", local_rank))
        torch.cuda.synchronize()
    return torch, dist, local_rank


def strong_core(ctx, params, lengths, tile, rank, world, local_rank, collective, backend, torch, dist, clock, warmup, label):
    """One whole pair grid as ONE job over the ranks, the way all_pairwise splits it (algorithm_template.py:168-192 in the
    reference; acoss_amd/algorithms/algorithm_template.py:_all_pairwise_grid here): acx_grid_plan's cost-balanced deal, every
    rank runs ALL its tiles into one device buffer (acx_grid_run), the ONE exchange of the path on the real per-rank buffers
    (acoss_amd.dist.gather_tiles_device: a gather to rank 0 over RCCL; ACX_GRID_EXCHANGE=allgather for the all-gather of
    rounds 1-4), rank 0 copies the result to the host, scatters + mirrors it into an N x N float32 matrix (acx_grid_scatter)
    and checks 256 sampled cells against the pair-LIST path.  The pool must already be in the context; `lengths` may be a
    prefix of it (the sub-grid leg of the default run)."""
    import tempfile
    from acoss_amd import _lib
    from acoss_amd import dist as adist
    n = len(lengths)
    plan = _lib.grid_plan(lengths, _lib.ALGO_SERRA09, True, world=world, tile=tile, want_tiles=True)
    spec = plan["spec"]
    stride = int(max(1, plan["floats_per_rank"].max()))
    if collective:
        dev = torch.device("cuda", local_rank)
        local = torch.zeros(stride, dtype=torch.float32, device=dev)
    else:                                                        # one rank, no collective: no torch in the process
        local = ctx.dev_alloc(4 * stride)
    n_mine = sum(1 for t in plan["tiles"] if t.rank == rank)

    def fence():
        if collective:
            dist.barrier(device_ids=[local_rank]) if backend == "nccl" else dist.barrier()
            torch.cuda.synchronize()
        else:
            ctx.dev_sync()

    if warmup:
        with clock.phase(label + "_warmup"):
            for _ in range(warmup):                              # arena, code objects, clocks: the rank's first tile(s)
                ctx.grid_run(spec, params, rank, local.data_ptr(), first=0, count=min(2, n_mine))
            # (acx_grid_run zeroes the slice it fills: the warm-up's scores are overwritten by the timed pass)
    ctx.profile_enable(False)                                    # the product path: no per-kernel event clocks, sweeps on their own streams
    fence()
    clock.stamp("%s: timed region (whole grid of %d tracks) ..." % (label, n))
    t0 = time.perf_counter()
    ctx.grid_run(spec, params, rank, local.data_ptr())           # all tiles of this rank; returns after its stream drained
    t_compute = time.perf_counter() - t0
    tg0 = time.perf_counter()
    gathered = None
    if collective:
        gathered = adist.gather_tiles_device(local, stride)     # rank 0: (world * stride,) on its GPU (nccl) / the host (gloo)
        if backend == "nccl":
            torch.cuda.synchronize()
    t_gather = time.perf_counter() - tg0
    fence()
    elapsed = time.perf_counter() - t0
    clock.stamp("%s: timed region %.2f s (kernels %.2f s, exchange %.4f s)" % (label, elapsed, t_compute, t_gather))
    # ---- the serial tail: rank 0 brings the gathered buffers to the host and scatters them into the matrix
    t_d2h = t_scatter = None
    check = None
    if rank == 0:
        ts0 = time.perf_counter()
        host_gather = gathered.cpu().numpy() if collective else local.read(np.float32, stride)
        t_d2h = time.perf_counter() - ts0
        tmp = tempfile.mkdtemp(prefix="acx_strong_")
        D = np.lib.format.open_memmap(os.path.join(tmp, "D.npy"), mode="w+", dtype=np.float32, shape=(n, n))
        ts1 = time.perf_counter()
        _lib.grid_scatter(lengths, spec, host_gather, stride, [D], mirror=True)
        t_scatter = time.perf_counter() - ts1
        # the matrix against the pair-LIST path of the library on sampled pairs (and its own transpose)
        rng = np.random.default_rng(5)
        smp = rng.integers(0, n, (512, 2))
        smp = np.sort(smp[smp[:, 0] != smp[:, 1]][:256], axis=1)       # (i < j): the triangle all_pairwise computes
        smp = np.ascontiguousarray(smp.astype(np.int32))
        got = ctx.serra09_pairs(smp, params)
        check = {"sampled_pairs": len(smp),
                 "matrix_equals_pair_list": bool(np.array_equal(D[smp[:, 0], smp[:, 1]], got)),
                 "cells_equal": int(np.sum(D[smp[:, 0], smp[:, 1]] == got)),
                 "symmetric": bool(np.array_equal(D[smp[:, 0], smp[:, 1]], D[smp[:, 1], smp[:, 0]])),
                 "nonzero_fraction_offdiag": float(np.count_nonzero(D) / max(1, n * (n - 1)))}
        del D
        try:
            os.remove(os.path.join(tmp, "D.npy"))
            os.rmdir(tmp)
        except OSError:
            pass
    per_rank = [t_compute]
    if collective:
        cdev = torch.device("cuda", local_rank) if backend == "nccl" else torch.device("cpu")
        tt = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        tc = torch.tensor([t_compute], dtype=torch.float64, device=cdev)
        outs = [torch.zeros_like(tc) for _ in range(world)]
        dist.all_gather(outs, tc)
        per_rank = [float(o.item()) for o in outs]
    if not collective:
        local.free()
    pairs = n * (n - 1) // 2
    exchange = None
    if collective:
        exchange = "%s over %s" % ("all_gather_into_tensor" if adist.exchange_in_use() == "allgather" else "gather to rank 0 (torch.distributed.gather)",
                                   "RCCL, device to device" if backend == "nccl" else "gloo through host memory (development: ranks may share a GPU)")
    res = {"tracks": n, "pairs": pairs, "elapsed_s": elapsed, "spec_tile": int(spec.tile),
           "strong": {"tracks": n, "pairs": pairs, "value": round(pairs / elapsed, 1),
                      "kernels_s_per_rank": [round(t, 3) for t in per_rank],
                      "plan_imbalance_max_over_mean": round(max(per_rank) / (sum(per_rank) / len(per_rank)), 4),
                      "plan_cost_per_rank": [float(c) for c in plan["cost_per_rank"]],
                      "tiles": int(plan["n_tiles"]), "tile": int(spec.tile), "exchange": exchange,
                      "gather_ms": round(1e3 * t_gather, 3), "gather_bytes": int(world * stride * 4),
                      "gather_bytes_per_rank": int(stride * 4),
                      "kernel_clocks": "off (the product path: alignment sweeps on their own streams beside the next band kernels)"}}
    if rank == 0:
        res["strong"].update({"d2h_s": round(t_d2h, 3), "scatter_mirror_s": round(t_scatter, 3),
                              "value_incl_scatter": round(pairs / (elapsed + t_d2h + t_scatter), 1), "check": check})
    return res


def run_strong(args, clock, out, rank, world, local_rank):
    """`--strong`: the WHOLE pair grid of the pool (5 000 tracks: 12 497 500 pairs) once, split over the N ranks
    (strong_core).  Strong scaling: total work fixed.  `value` = pairs / (kernels + exchange), max over ranks;
    `strong.value_incl_scatter` adds the device-to-host copy and rank 0's scatter, the only serial part.  Reports the
    per-rank devices, plan imbalance, exchange ms / bytes, scatter s."""
    from acoss_amd import _lib
    with clock.phase("pool_gen"):
        frames, offsets = make_pool(args.tracks, args.frames)
    backend = os.environ.get("ACX_BENCH_BACKEND", "nccl")
    collective = world > 1 or os.environ.get("ACX_BENCH_FORCE_COLLECTIVE") == "1"
    torch = dist = None
    if collective:
        torch, dist, local_rank = init_torch(clock, local_rank, world, backend, collective)
    infos = gather_identities(rank_identity(rank, int(os.environ.get("LOCAL_RANK", "0")), local_rank), collective, dist, backend, local_rank)
    why = shared_device_reason(infos, backend, world)
    if why:
        if rank == 0:
            print(why, file=sys.stderr, flush=True)
        raise SystemExit(3)
    with clock.phase("context_and_upload"):
        ctx = _lib.Context(local_rank)
        ctx.upload_pool(frames, offsets)
    params = _lib.serra09_params()
    lengths = np.full(args.tracks, args.frames, np.int64)
    res = strong_core(ctx, params, lengths, args.tile, rank, world, local_rank, collective, backend, torch, dist, clock,
                      max(1, args.warmup), "strong")
    if rank == 0:
        pairs, elapsed = res["pairs"], res["elapsed_s"]
        line = {
            "metric": "track-pairs/sec on N x N Serra09 Qmax (HPCP, T=%d)" % args.frames,
            "value": round(pairs / elapsed, 1), "unit": "track-pairs/s", "n_gpus": world,
            "ranks_seen": dist.get_world_size() if collective else 1, "collectives": (backend if collective else None),
            "rccl_version": rccl_version(collective, backend), "ranks": infos,
            "steps": 1, "warmup": max(1, args.warmup), "ms_per_step": round(1e3 * elapsed, 3), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "configs[2] as ONE job: synthetic %d tracks x %d-frame HPCP (seed 1234), Serra09 Qmax, the whole "
                                   "%d x %d upper-triangle grid (%d pairs) in %d x %d tiles dealt to %d rank(s) by cost; every rank "
                                   "runs all its tiles, ONE exchange of the per-rank score buffers, rank 0 scatters"
                                   % (args.tracks, args.frames, args.tracks, args.tracks, pairs, res["spec_tile"], res["spec_tile"], world),
                       "pairs": pairs, "frames_per_track": args.frames, "pool_tracks": args.tracks,
                       "parallelism": "pair-grid tiles over %d GPU(s), strong scaling" % world},
            "strong": res["strong"],
            "phases_s": clock.phases,
        }
        print(json.dumps(line), file=out, flush=True)
    ctx.close()
    if collective:
        dist.barrier(device_ids=[local_rank]) if backend == "nccl" else dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--tracks", type=int, default=N_TRACKS)
    ap.add_argument("--frames", type=int, default=T_FRAMES, help="frames per track (--strong only; the headline is fixed at 2000)")
    ap.add_argument("--tile", type=int, default=TILE, help="grid tile edge in tracks (--strong only)")
    ap.add_argument("--strong", action="store_true",
                    help="strong scaling: the whole pair grid of the pool once, split over the ranks, ONE final all-gather of "
                         "the per-rank score buffers, rank-0 scatter reported beside it (minutes at 5000 tracks on one GPU)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-other", action="store_true", help="skip the SiMPle / EarlyFusion / covers-shaped legs (the `other` object)")
    ap.add_argument("--plan-only", action="store_true",
                    help="rendezvous, deal the tiles, print the plan as one JSON line and stop: no GPU work "
                         "(checks a multi-rank launch on any box; ACX_BENCH_BACKEND=gloo without GPUs)")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("bench: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    out = json_only_stdout()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench: --gpus %d but WORLD_SIZE=%d: start exactly one rank per GPU "
                         "(python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d ...)"
                         % (args.gpus, world, args.gpus, args.gpus))
    clock = Clock(verbose=(rank == 0))
    # a run that hangs anywhere says where: every thread's stack on stderr every 4 minutes
    faulthandler.dump_traceback_later(240, repeat=True, file=sys.stderr)
    clock.stamp("start: rank %d of %d, %d host threads for BLAS / OpenMP pools (of %d hardware threads)"
                % (rank, world, HOST_THREADS, os.cpu_count() or 0))
    clock.phases["interpreter_and_numpy"] = round(time.perf_counter() - _T_START, 3)
    if args.strong:
        return run_strong(args, clock, out, rank, world, local_rank)

    with clock.phase("libacx_load_and_plan"):
        from acoss_amd import _lib
        lengths = np.full(args.tracks, T_FRAMES, np.int64)
        plan = _lib.grid_plan(lengths, _lib.ALGO_SERRA09, True, world=world, tile=TILE, want_tiles=True)
    spec = plan["spec"]
    mine = [t for t in plan["tiles"] if t.rank == rank]
    nslices = len(mine) // TILES_PER_STEP
    assert nslices >= 1

    def slice_of(step):
        k = (step % nslices) * TILES_PER_STEP
        return k, mine[k:k + TILES_PER_STEP]

    def pairs_of(tiles):
        out = []
        for t in tiles:
            i, j = np.meshgrid(np.arange(t.row0, t.row0 + t.rows), np.arange(t.col0, t.col0 + t.cols), indexing="ij")
            keep = (i < j) if t.diagonal else np.ones_like(i, bool)
            out.append(np.stack([i[keep], j[keep]], 1))
        return np.concatenate(out).astype(np.int32)

    if args.plan_only:
        seen = 1
        if world > 1:
            import torch
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group(os.environ.get("ACX_BENCH_BACKEND", "gloo"))
            t = torch.tensor([len(mine)], dtype=torch.int64)
            if dist.get_backend() == "nccl":
                torch.cuda.set_device(local_rank)
                t = t.cuda()
            outs = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(outs, t)
            seen = dist.get_world_size()
            per_rank = [int(o.item()) for o in outs]
            dist.barrier()
            dist.destroy_process_group()
        else:
            per_rank = [len(mine)]
        if rank == 0:
            print(json.dumps({"plan_only": True, "n_gpus": world, "ranks_seen": seen, "tiles_per_rank": per_rank,
                              "tiles": int(plan["n_tiles"]), "tile": TILE, "pool_tracks": args.tracks,
                              "cost_per_rank": [float(c) for c in plan["cost_per_rank"]]}), file=out, flush=True)
        return

    with clock.phase("pool_gen"):
        frames, offsets = make_pool(args.tracks, T_FRAMES)
    # ---- CPU baseline first: worker processes are forked (and joined) before any GPU state exists
    cpu = cpu_sample = cpu_scores = None
    if world == 1 and rank == 0 and not args.no_cpu:
        try:
            cpu_sample, cpu_scores, cpu = cpu_baseline(frames, offsets, pairs_of(slice_of(args.warmup)[1]), args.tracks, clock)
        except Exception as e:                       # the reported baseline must not take the measurement with it
            cpu = {"value": None, "unit": "track-pairs/s", "cores": effective_cpus(), "kind": "port",
                   "sample": "failed: %s: %s" % (type(e).__name__, e)}
            cpu_sample = cpu_scores = None

    # ACX_BENCH_BACKEND=gloo (development): functional run of the N > 1 path on a box with fewer
    # GPUs than ranks -- ranks share the devices and the gather goes through host memory
    backend = os.environ.get("ACX_BENCH_BACKEND", "nccl")
    # ACX_BENCH_FORCE_COLLECTIVE=1: a world of ONE still forms its process group and runs every collective of the
    # N > 1 step (RCCL all_gather_into_tensor, device barrier, all-reduce) -- the multi-GPU code path on a 1-GPU box
    collective = world > 1 or os.environ.get("ACX_BENCH_FORCE_COLLECTIVE") == "1"
    # One GPU, no collective: this process never imports torch.  The score buffer comes from libacx (acx_dev_alloc =
    # hipMalloc) and the fences around the timed region are hipDeviceSynchronize (acx_dev_sync) -- what
    # torch.cuda.synchronize() is.  `import torch` pages in gigabytes of shared libraries; on a box with a cold or slow
    # image store that alone has been seen to take minutes with the GPU idle (round 3's driver run: 1662 s for a 1.7 s
    # timed region).  N > 1 needs torch.distributed and imports it.
    torch = dist = None
    if collective:
        torch, dist, local_rank = init_torch(clock, local_rank, world, backend, collective)
        dev = torch.device("cuda", local_rank)
    # which GPU does every rank hold?  Gathered BEFORE any measurement; two RCCL ranks on one device end the run here
    infos = gather_identities(rank_identity(rank, int(os.environ.get("LOCAL_RANK", "0")), local_rank), collective, dist, backend, local_rank)
    why = shared_device_reason(infos, backend, world)
    if why:
        if rank == 0:
            print(why, file=sys.stderr, flush=True)
        raise SystemExit(3)

    with clock.phase("context_and_upload"):
        ctx = _lib.Context(local_rank)
        ctx.upload_pool(frames, offsets)                 # pool resident in HBM before timing
    params = _lib.serra09_params()
    slice_floats = max(sum(t.rows * t.cols for t in mine[k:k + TILES_PER_STEP]) for k in range(0, nslices * TILES_PER_STEP, TILES_PER_STEP))
    if collective:
        from acoss_amd import dist as adist
        adist.bind_device(local_rank)
        local = torch.zeros(slice_floats, dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
    else:
        local = ctx.dev_alloc(4 * slice_floats)
        ctx.dev_sync()
    pairs_per_step = []

    def step(s, exchange=True):
        k, tiles = slice_of(s)
        # acx_grid_run writes tile t at d_scores + t.offset: rebase so that the slice starts at local[0]
        ctx.grid_run(spec, params, rank, local.data_ptr() - 4 * tiles[0].offset, first=k, count=TILES_PER_STEP)
        if collective and exchange:
            # the one exchange of the path, the library's own (acoss_amd.dist.gather_tiles_device: what
            # CoverAlgorithm.all_pairwise calls -- a gather to rank 0, device to device under RCCL; or the all-gather
            # of rounds 1-4 under ACX_GRID_EXCHANGE=allgather / when the probe of the gather failed)
            adist.gather_tiles_device(local, slice_floats)
            if backend == "nccl":
                # the next step's kernels (libacx's own stream) overwrite `local`: the host waits for the collective
                torch.cuda.current_stream().synchronize()
        return sum((t.rows * (t.rows - 1)) // 2 if t.diagonal else t.rows * t.cols for t in tiles)

    def fence():
        if collective:
            if backend == "nccl":
                dist.barrier(device_ids=[local_rank])
            else:
                dist.barrier()
            torch.cuda.synchronize()
        else:
            ctx.dev_sync()                          # hipDeviceSynchronize

    with clock.phase("warmup"):
        for s in range(args.warmup):
            step(s)
    # The timed region runs the PRODUCT path: the per-kernel event clocks stay off (with them on the library keeps its
    # alignment sweeps on the band kernels' stream, acx.hip run_serra09).  The per-kernel figures of the roofline object
    # come from a second pass over the same steps' tiles behind the timed region, clocks on (as bench_other.py's legs do).
    ctx.profile_enable(False)
    fence()
    clock.stamp("timed region ...")
    t0 = time.perf_counter()
    for s in range(args.warmup, args.warmup + args.steps):
        pairs_per_step.append(step(s))
    fence()
    elapsed = time.perf_counter() - t0
    clock.phases["timed_region"] = round(elapsed, 3)
    clock.stamp("timed region: %.3f s" % elapsed)
    my_pairs = float(sum(pairs_per_step))
    if collective:
        cdev = dev if backend == "nccl" else torch.device("cpu")
        tt = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        pp = torch.tensor([my_pairs], dtype=torch.float64, device=cdev)
        dist.all_reduce(pp, op=dist.ReduceOp.SUM)
        total_pairs = float(pp.item())
    else:
        total_pairs = my_pairs
    ranks_seen = dist.get_world_size() if collective else 1        # what the collective actually spanned
    with clock.phase("kernel_clock_pass"):
        ctx.profile_enable(True)
        ctx.profile_reset()
        fence()
        tk0 = time.perf_counter()
        for s in range(args.warmup, args.warmup + args.steps):
            step(s, exchange=False)
        fence()
        clocked_elapsed = time.perf_counter() - tk0
        prof = ctx.profile()
        ctx.profile_enable(False)
    my_kernels_s = sum(v["ms"] for v in prof.values()) / 1e3
    if collective:
        box = [None] * world
        dist.all_gather_object(box, (rank, round(my_kernels_s, 4), round(my_pairs)))
        for r, ks, pp_ in box:
            infos[r]["kernels_s"] = ks
            infos[r]["pairs"] = int(pp_)
    else:
        infos[0]["kernels_s"] = round(my_kernels_s, 4)
        infos[0]["pairs"] = int(round(my_pairs))
    # ---- N > 1 (or a forced one-rank world): the path as ONE job on a sub-grid of the same pool -- the real plan, the real
    # per-rank buffers through the ONE exchange, rank 0's device-to-host copy + scatter + mirror, the matrix checked against
    # the pair-list path.  Outside the timed region above; reported under `strong` in the same line.
    strong = None
    # (one GPU without a collective: only in the full default run -- `--no-other` is what the profiler passes use, and their
    #  per-kernel means must stay those of the timed steps' launches)
    if (collective or not args.no_other) and not os.environ.get("ACX_BENCH_NO_STRONG_LEG"):
        n_sub = min(args.tracks, STRONG_LEG_TRACKS if collective else STRONG_LEG_TRACKS_ONE)
        with clock.phase("strong_leg"):
            # (acx_grid_run plans over the WHOLE pool of the context: the sub-grid gets a pool of its own, the first n_sub tracks)
            ctx.upload_pool(frames[:int(offsets[n_sub])], offsets[:n_sub + 1])
            strong = strong_core(ctx, params, np.full(n_sub, T_FRAMES, np.int64), TILE, rank, world, local_rank, collective, backend,
                                 torch, dist, clock, 0, "strong_leg")["strong"]
            strong["pool"] = "the first %d tracks of the run's pool, uploaded as a pool of their own" % n_sub
            if world == 1:
                ctx.upload_pool(frames, offsets)        # (the self-check and the companion legs below want the run's pool back)
        ctx.profile_enable(False)

    if rank == 0:
        value = total_pairs / elapsed
        # ---- roofline of the dominant kernel (HIP events on the library's stream)
        kname, kst = max(prof.items(), key=lambda kv: kv[1]["ms"])
        launches = max(1, kst["launches"])
        cells_per_launch = kst["cells"] / launches
        avg_ms = kst["ms"] / launches
        algo_bytes = ALGO_BYTES_PER_CELL[kname] * cells_per_launch
        if kname in ("csm_long_kernel", "band_kernel"):
            algo_bytes += 48.0 * 2 * T_FRAMES * (cells_per_launch / float((T_FRAMES - M_STACK) ** 2))
        achieved = algo_bytes / (avg_ms * 1e-3) / 1e9
        # HBM traffic and pipe utilisations come from rocprofv3 --pmc passes of a SEPARATE run (counters and this run's
        # timing cannot be collected together); the records carry the hash of the kernel sources they were taken
        # on and are reported only when it is the hash of THIS build -- a kernel change without a re-profile
        # yields null, not stale numbers.
        traffic, traffic_source, real_bound = None, None, None
        sha = kernel_source_sha16()
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if tj.get("kernel_source_sha16") != sha:
                    traffic_source = "profiles/pmc_traffic.json was taken on other kernel sources (%s, this build %s): not reported" % (
                        tj.get("kernel_source_sha16"), sha)
                elif kname in tj and tj[kname].get("cells_per_launch"):
                    traffic = tj[kname]["hbm_bytes_per_launch"] * cells_per_launch / tj[kname]["cells_per_launch"]
                    traffic_source = ("profiles/pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in a separate run "
                                      "(%s, same kernel sources %s), scaled by cells per launch" % (tj.get("source"), sha))
            except Exception:
                traffic = None
        rpath = os.path.join(ROOT, "profiles", "real_bound.json")
        if os.path.exists(rpath):
            try:
                rj = json.load(open(rpath))
                real_bound = rj.get(kname) if rj.get("kernel_source_sha16") == sha else None
            except Exception:
                real_bound = None
        bpp = chain_bytes_per_pair(T_FRAMES, T_FRAMES)
        # SURVEY 8d's flop model of the same kernel: 2 x 108 flop per cell (the reference's 108-dim formulation of
        # the cross-similarity matrix), per launch, against the f32-input MFMA peak
        flops_pair = 2.0 * 12 * M_STACK * float((T_FRAMES - M_STACK) ** 2)
        mfma_tf = value / world * flops_pair / 1e12
        bound = derive_bound(real_bound)
        roofline = {"bound": bound or "hbm",
                    "bound_source": ("busiest unit in the counter record of this kernel (real_bound)" if bound else
                                     "no counter record of this build's kernel sources: the byte model's own bound"),
                    "model": "`achieved` / `frac` price the kernel with the HBM byte model of SURVEY 8d (4 B per cell and launch: a "
                             "THROUGHPUT PROXY -- the pipeline keeps the distance matrix out of HBM, see `traffic`); what the SIMDs "
                             "wait for is `bound` / real_bound; the 8d flop model of the same launch is in mfma_model",
                    "kernel": kname, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                    "traffic_source": traffic_source, "real_bound": real_bound,
                    "mfma_model": {"flops_per_pair": flops_pair, "achieved": round(mfma_tf, 2),
                                   "peak": F32_MFMA_PEAK_TF, "unit": "TFLOP/s", "frac": round(mfma_tf / F32_MFMA_PEAK_TF, 4),
                                   "note": "SURVEY 8d: 2 x 108 flop per cell of the pair's matrix, once per pair, over the whole chain's "
                                           "time per pair; the kernels contract K = 12 per frame pair (window sums give the 108-dim "
                                           "product) but sweep the matrix twice (column pass, row pass)"},
                    "avg_launch_ms": round(avg_ms, 3), "algorithmic_bytes_per_launch": algo_bytes,
                    "kernels_ms_per_step": {k: round(v["ms"] / args.steps, 3) for k, v in prof.items()},
                    "kernel_clock_pass": {"what": "avg_launch_ms / kernels_ms_per_step: HIP events around every launch on the library's "
                                                  "stream, taken in a SECOND pass over the same steps' tiles behind the timed region (the "
                                                  "clocks serialise the alignment sweeps onto the band kernels' stream; the timed region "
                                                  "runs without them, sweeps on two side streams: `value` is the product path)",
                                          "ms_per_step": round(1e3 * clocked_elapsed / args.steps, 3),
                                          "timed_region_ms_per_step": round(1e3 * elapsed / args.steps, 3)},
                    "chain": {"algorithmic_bytes_per_pair": bpp,
                              "achieved": round(value / world * bpp / 1e9, 1),
                              "frac": round(value / world * bpp / 1e9 / HBM_PEAK_GBS, 4)}}
        if cpu_sample is not None and len(cpu_sample):
            with clock.phase("self_check"):
                got = ctx.serra09_pairs(cpu_sample, params)
                if not np.array_equal(got, cpu_scores):
                    raise SystemExit("bench: GPU scores differ from the CPU oracle on the sampled pairs")
        line = {
            "metric": "track-pairs/sec on N x N Serra09 Qmax (HPCP, T=2000)",
            "value": round(value, 1), "unit": "track-pairs/s", "n_gpus": world, "ranks_seen": ranks_seen, "collectives": (backend if collective else None),
            "exchange": ((adist.exchange_in_use() + " (acoss_amd.dist.gather_tiles_device)") if collective else None),
            "rccl_version": rccl_version(collective, backend), "ranks": infos, "strong": strong, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "fence": ("dist.barrier + torch.cuda.synchronize on both sides of the timed region" if collective else
                      "hipDeviceSynchronize (libacx acx_dev_sync) on both sides of the timed region; one rank, no torch in the process"),
            "config": {"workload": "configs[2]: synthetic %d tracks x %d-frame HPCP (seed 1234), Serra09 Qmax "
                                   "(m=9, tau=1, kappa=0.095, OTI); the %d x %d pair grid in %d x %d tiles dealt to %d rank(s) "
                                   "by cost (acx_grid_plan); %d tiles = %d pairs per GPU per step, different tiles every "
                                   "step; at N > 1 one exchange of the tile scores per step (the library's: gather to rank 0)"
                                   % (args.tracks, T_FRAMES, args.tracks, args.tracks, TILE, TILE, world, TILES_PER_STEP,
                                      int(pairs_per_step[0])),
                       "pairs_per_step": int(round(total_pairs / args.steps)), "frames_per_track": T_FRAMES,
                       "pool_tracks": args.tracks, "parallelism": "pair-grid tiles over %d GPU(s)" % world},
            "roofline": roofline, "cpu_baseline": cpu, "phases_s": clock.phases,
        }
        printed = threading.Lock()

        def emit():
            if printed.acquire(False):
                line["phases_s"]["total_wall"] = round(time.perf_counter() - _T_START, 3)
                print(json.dumps(line), file=out, flush=True)

        if world == 1 and not args.no_other:
            # the companion legs, outside the timed region above (bench_other.py): Serra09 on covers80-shaped lengths and the
            # other two algorithms at their BASELINE configs[3] / [4] per-track shapes, each with its own roofline and CPU
            # baseline, each under a hard wall limit.  A leg that hangs in native code cannot take the headline with it:
            # the watchdog prints the line as it stands and ends the process.
            import bench_other
            line["other"] = {}
            legs = (("serra09_covers", bench_other.serra09_covers_leg), ("serra09_f16x2", bench_other.serra09_f16x2_leg),
                    ("chenfusion", bench_other.chenfusion_leg),
                    ("simple", bench_other.simple_leg), ("earlyfusion", bench_other.earlyfusion_leg))

            def watchdog():
                clock.stamp("watchdog: the `other` legs hang; printing the headline line without them")
                for key, _ in legs:
                    line["other"].setdefault(key, {"error": "abandoned by the watchdog"})
                emit()
                os._exit(0)

            dog = threading.Timer(len(legs) * HARD_OTHER_LEG_S + 60.0, watchdog)
            dog.daemon = True
            dog.start()
            for key, leg in legs:
                try:
                    with clock.phase("other_" + key), wall_limit(HARD_OTHER_LEG_S, "other." + key):
                        line["other"][key] = leg(ctx)
                except Exception as e:            # a failing companion leg must not take the headline line with it
                    line["other"][key] = {"error": "%s: %s" % (type(e).__name__, e)}
            dog.cancel()
            sc = line["other"].get("serra09_covers", {})
            if "gcells_per_s" in sc:                   # cells per second of the short-track sets against the headline's
                head = value / world * float((T_FRAMES - M_STACK) ** 2) / 1e9
                sc["headline_gcells_per_s"] = round(head, 1)
                sc["cell_rate_vs_T2000"] = round(sc["gcells_per_s"] / head, 3)
        emit()
    with clock.phase("teardown"):
        ctx.close()
        if collective:
            fence()
            dist.destroy_process_group()
    clock.stamp("done")
    faulthandler.cancel_dump_traceback_later()


if __name__ == "__main__":
    main()
