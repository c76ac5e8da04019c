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
                initialised (the worker processes are forks).
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

T_FRAMES = 2000
N_TRACKS = 5000
TILE = 64
TILES_PER_STEP = 2             # per rank: 2 x 64 x 64 = 8192 pairs per GPU per step
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s
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
    for f in ("serra09_kernels.hpp", "acx_band.hip", "Makefile"):
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


def cpu_baseline(frames, offsets, pairs, n_tracks, budget_s=15.0):
    """SURVEY 8d's two CPU figures, both from the oracle (a C port of the chain, gcc -O3) and both timed BEFORE
    the GPU is initialised (the workers are forks):
      (1) one process, one thread, on randomly chosen pairs of the same synthetic pool (seeded; 64 of them -- 8d
          names 256, which at ~0.2 s per pair would be a minute of the bench's budget) -> value_1core;
      (2) every host core through a process fan-out over max(45, cores) chunks (the reference's joblib scheme,
          algorithm_template.py:172-177, has 45) -> value.  Bounded: a pilot round of one pair per worker measures
          the rate under full load, the timed round is sized from it to about `budget_s` seconds.
    If (2) is less than half of cores x (1), a one-second STREAM triad on the same workers is reported beside it
    (the oracle streams a 16 MB distance matrix per pair three times; hundreds of copies of it share the memory
    controllers)."""
    import multiprocessing as mp
    import oracle
    oracle.lib()
    _CPU["frames"], _CPU["offsets"] = frames, offsets
    cores = effective_cpus()
    rng = np.random.default_rng(8)
    rnd = rng.integers(0, n_tracks, (2 * 64, 2))
    rnd = np.ascontiguousarray(rnd[rnd[:, 0] != rnd[:, 1]][:64].astype(np.int32))
    t0 = time.perf_counter()
    rnd_scores = oracle.serra09_pairs(frames, offsets, rnd[:4])
    t4 = time.perf_counter() - t0
    n1 = int(min(64, max(4, 4 * round(0.25 * budget_s / max(t4, 1e-3)))))     # ~budget_s of single-core work
    t0 = time.perf_counter()
    rnd_scores = oracle.serra09_pairs(frames, offsets, rnd[:n1])
    t_1core = time.perf_counter() - t0
    stream = None
    with mp.get_context("fork").Pool(cores, initializer=_cpu_init) as pool:
        nchunks = max(45, cores)                                  # the reference's joblib scheme has 45 chunks
        pilot = np.ascontiguousarray(pairs[:nchunks])
        t0 = time.perf_counter()
        parts = pool.map(_cpu_chunk, [pilot[k:k + 1] for k in range(len(pilot))], chunksize=1)
        dt = time.perf_counter() - t0
        sample, scores = pilot, np.concatenate(parts)
        per_chunk = int(budget_s / dt) if dt > 0 else 0
        if per_chunk >= 2:                                         # room for a longer timed round
            n = min(len(pairs), nchunks * per_chunk)
            n -= n % nchunks
            sample = np.ascontiguousarray(pairs[:n])
            chunks = [c for c in np.array_split(sample, nchunks) if len(c)]
            t0 = time.perf_counter()
            parts = pool.map(_cpu_chunk, chunks, chunksize=1)
            dt = time.perf_counter() - t0
            scores = np.concatenate(parts)
        n = len(sample)
        v1, vall = n1 / t_1core, n / dt
        if vall < 0.5 * cores * v1:
            stream = round(sum(pool.map(_cpu_triad, range(cores), chunksize=1)) / 1e9, 1)
    model = ""
    try:
        for line in subprocess.run(["lscpu"], capture_output=True, text=True).stdout.splitlines():
            if line.startswith("Model name"):
                model = line.split(":", 1)[1].strip()
    except OSError:
        pass
    check = np.concatenate([sample, rnd[:n1]]), np.concatenate([scores, rnd_scores])
    return check[0], check[1], {
        "value": round(vall, 3), "unit": "track-pairs/s", "cores": cores, "kind": "port",
        "sample": "first %d pairs of step 0's tiles of the same workload (T=%d), C oracle (gcc -O3), %d worker "
                  "processes over %d chunks (reference scheme: 45 joblib chunks), %.1f s; value_1core: %d randomly chosen "
                  "pairs of the pool (seed 8) in one process, %.1f s; all scores bit-identical to the GPU's"
                  % (n, T_FRAMES, cores, max(45, cores), dt, n1, t_1core),
        "value_1core": round(v1, 3), "parallel_efficiency": round(vall / (cores * v1), 3),
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--tracks", type=int, default=N_TRACKS)
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-other", action="store_true", help="skip the SiMPle / EarlyFusion legs (the `other` object)")
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
        import torch
        import torch.distributed as dist
        seen = 1
        if world > 1:
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

    frames, offsets = make_pool(args.tracks, T_FRAMES)
    # ---- CPU baseline first: worker processes are forked before any GPU state exists
    cpu = cpu_sample = cpu_scores = None
    if world == 1 and rank == 0 and not args.no_cpu:
        cpu_sample, cpu_scores, cpu = cpu_baseline(frames, offsets, pairs_of(slice_of(args.warmup)[1]), args.tracks)

    import torch
    import torch.distributed as dist
    # ACX_BENCH_BACKEND=gloo (development): functional run of the N > 1 path on a box with fewer
    # GPUs than ranks -- ranks share the devices and the gather goes through host memory
    backend = os.environ.get("ACX_BENCH_BACKEND", "nccl")
    # ACX_BENCH_FORCE_COLLECTIVE=1: a world of ONE still forms its process group and runs every collective of the
    # N > 1 step (RCCL all_gather_into_tensor, device barrier, all-reduce) -- the multi-GPU code path on a 1-GPU box
    collective = world > 1 or os.environ.get("ACX_BENCH_FORCE_COLLECTIVE") == "1"
    if backend != "nccl":
        local_rank = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    if collective:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend)
    dev = torch.device("cuda", local_rank)

    ctx = _lib.Context(local_rank)
    ctx.upload_pool(frames, offsets)                 # pool resident in HBM before timing
    params = _lib.serra09_params()
    slice_floats = max(sum(t.rows * t.cols for t in mine[k:k + TILES_PER_STEP]) for k in range(0, nslices * TILES_PER_STEP, TILES_PER_STEP))
    local = torch.zeros(slice_floats, dtype=torch.float32, device=dev)
    gathered = torch.zeros(world * slice_floats, dtype=torch.float32, device=dev) if collective else None
    torch.cuda.synchronize()
    pairs_per_step = []

    def step(s):
        k, tiles = slice_of(s)
        # acx_grid_run writes tile t at d_scores + t.offset: rebase so that the slice starts at local[0]
        ctx.grid_run(spec, params, rank, local.data_ptr() - 4 * tiles[0].offset, first=k, count=TILES_PER_STEP)
        if collective:                              # the one collective of the path, device to device
            if backend == "nccl":
                dist.all_gather_into_tensor(gathered, local)
                # the next step's kernels (libacx's own stream) overwrite `local`: the host waits for the collective
                torch.cuda.current_stream().synchronize()
            else:
                outs = [torch.empty(slice_floats) for _ in range(world)]
                dist.all_gather(outs, local.cpu())
        return sum((t.rows * (t.rows - 1)) // 2 if t.diagonal else t.rows * t.cols for t in tiles)

    def fence():
        if collective:
            if backend == "nccl":
                dist.barrier(device_ids=[local_rank])
            else:
                dist.barrier()
        torch.cuda.synchronize()

    for s in range(args.warmup):
        step(s)
    ctx.profile_enable(True)
    ctx.profile_reset()
    fence()
    t0 = time.perf_counter()
    for s in range(args.warmup, args.warmup + args.steps):
        pairs_per_step.append(step(s))
    fence()
    elapsed = time.perf_counter() - t0
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
    prof = ctx.profile()

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
        roofline = {"bound": "hbm", "model": "hbm byte model of SURVEY 8d: a throughput proxy -- the pipeline keeps the distance matrix out of "
                                              "HBM, what the SIMDs wait for is in real_bound (valu + f32 mfma issue)",
                    "kernel": kname, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                    "traffic_source": traffic_source, "real_bound": real_bound,
                    "avg_launch_ms": round(avg_ms, 3), "algorithmic_bytes_per_launch": algo_bytes,
                    "kernels_ms_per_step": {k: round(v["ms"] / args.steps, 3) for k, v in prof.items()},
                    "chain": {"algorithmic_bytes_per_pair": bpp,
                              "achieved": round(value / world * bpp / 1e9, 1),
                              "frac": round(value / world * bpp / 1e9 / HBM_PEAK_GBS, 4)}}
        if cpu is not None:
            got = ctx.serra09_pairs(cpu_sample, params)
            if not np.array_equal(got, cpu_scores):
                raise SystemExit("bench: GPU scores differ from the CPU oracle on the sampled pairs")
        line = {
            "metric": "track-pairs/sec on N x N Serra09 Qmax (HPCP, T=2000)",
            "value": round(value, 1), "unit": "track-pairs/s", "n_gpus": world, "ranks_seen": ranks_seen, "collectives": (backend if collective else None), "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "configs[2]: synthetic %d tracks x %d-frame HPCP (seed 1234), Serra09 Qmax "
                                   "(m=9, tau=1, kappa=0.095, OTI); the %d x %d pair grid in %d x %d tiles dealt to %d rank(s) "
                                   "by cost (acx_grid_plan); %d tiles = %d pairs per GPU per step, different tiles every "
                                   "step; one all-gather of the tile scores per step"
                                   % (args.tracks, T_FRAMES, args.tracks, args.tracks, TILE, TILE, world, TILES_PER_STEP,
                                      int(pairs_per_step[0])),
                       "pairs_per_step": int(round(total_pairs / args.steps)), "frames_per_track": T_FRAMES,
                       "pool_tracks": args.tracks, "parallelism": "pair-grid tiles over %d GPU(s)" % world},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        if world == 1 and not args.no_other:
            # the other two algorithms of the path, outside the timed region above: 2 steps each at their
            # BASELINE configs[3] / [4] per-track shapes (bench_other.py), each with its own roofline and CPU baseline
            import bench_other
            line["other"] = {}
            for key, leg in (("simple", bench_other.simple_leg), ("earlyfusion", bench_other.earlyfusion_leg)):
                try:
                    line["other"][key] = leg(ctx, steps=2, warmup=1)
                except Exception as e:            # a failing companion leg must not take the headline line with it
                    line["other"][key] = {"error": "%s: %s" % (type(e).__name__, e)}
        print(json.dumps(line), file=out, flush=True)
    ctx.close()
    if collective:
        fence()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
