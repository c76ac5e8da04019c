#!/usr/bin/env python3
"""
bench.py -- headline benchmark: track-pairs/sec of the Serra09 chain
(OTI -> embedded CSM -> mutual-kappa thresholds -> Qmax) on synthetic HPCP of
T = 2000 pooled frames (BASELINE.json metric; per-track shape of configs[2]).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of PAIRS_PER_RANK track pairs per GPU
(weak scaling: the pair grid grows with N; pairs are independent, the only exchange is one
all-gather of the scores per pass over RCCL).  The feature pool is uploaded to HBM before
the timed region.  Rank 0 prints ONE JSON line.

Also in the line:
  roofline      dominant kernel (by accumulated HIP-event time on the library's own
                stream) against the HBM roofline, with the algorithmic bytes of DESIGN.md
  cpu_baseline  the CPU oracle (a C port of the same chain, 1 thread) timed on the box's
                host cores on a bounded sample of the same pairs -- and checked bit-for-bit
                against the GPU scores of those pairs.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

T_FRAMES = 2000
N_TRACKS = 384                 # C(384, 2) = 73 536 pairs >= 8 ranks x 8192
PAIRS_PER_RANK = 8192
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
M_STACK = 9

# algorithmic bytes per matrix cell by kernel (DESIGN.md "Roofline model"; SURVEY.md 8d:
# 10 B/cell = f32 distance matrix written once + read once, 1-byte recurrence matrix
# written once + read once)
# band_kernel is launched twice per batch (role 1: column thresholds, role 0: row thresholds +
# recurrence bitmap); each launch carries one 4 B/cell pass over the distance matrix of the
# model (write / read-back), together the 8 B/cell of v1's csm_tile_kernel + rowsel_kernel;
# qmax_kernel = the DP over the recurrence plot.
ALGO_BYTES_PER_CELL = {"band_kernel": 4.0, "csm_long_kernel": 4.0, "rowsel_long_kernel": 4.0, "qmax_bits_kernel": 2.0,
                       "oti_kernel": 0.0, "norms_kernel": 0.0}


def chain_bytes_per_pair(Tq, Tr, m=M_STACK):
    Mq, Mr = Tq - m, Tr - m
    return 10.0 * Mq * Mr + 48.0 * (Tq + Tr) + 4.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--pairs-per-rank", type=int, default=PAIRS_PER_RANK)
    ap.add_argument("--cpu-pairs", type=int, default=8, help="pairs timed on ONE thread of the CPU oracle (0 = skip the CPU leg)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from acoss_amd import _lib, synth

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # ACX_BENCH_BACKEND=gloo (development): functional run of the N > 1 path on a box with fewer
    # GPUs than ranks -- ranks share the devices and the score gather goes through host memory
    backend = os.environ.get("ACX_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % max(1, torch.cuda.device_count())
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend)
    else:
        torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank) if backend == "nccl" else torch.device("cpu")

    # ---- synthetic pool (SURVEY 8d "rand" set), resident in HBM before timing
    data = synth.rand_set(N_TRACKS, T=T_FRAMES, seed=1234)
    ctx = _lib.Context(local_rank)
    ctx.upload_pool(data["frames"], data["offsets"])
    iu, ju = np.triu_indices(N_TRACKS, 1)
    all_pairs = np.stack([iu, ju], 1).astype(np.int32)
    ppr = args.pairs_per_rank
    assert world * ppr <= len(all_pairs)
    mine = np.ascontiguousarray(all_pairs[rank * ppr:(rank + 1) * ppr])
    params = _lib.serra09_params()

    def step():
        sc = ctx.serra09_pairs(mine, params)
        if world > 1:
            t = torch.from_numpy(sc).to(dev)
            outs = [torch.empty_like(t) for _ in range(world)]
            dist.all_gather(outs, t)            # the one collective of the path
            return sc, outs
        return sc, None

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    ctx.profile_enable(True)
    ctx.profile_reset()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        scores, _ = step()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    prof = ctx.profile()

    if rank == 0:
        total_pairs = world * ppr * args.steps
        value = total_pairs / elapsed
        # ---- roofline of the dominant kernel (HIP events on the library's stream)
        dom = max(prof.items(), key=lambda kv: kv[1]["ms"])
        kname, kst = dom
        launches = max(1, kst["launches"])
        cells_per_launch = kst["cells"] / launches
        avg_ms = kst["ms"] / launches
        algo_bytes = ALGO_BYTES_PER_CELL[kname] * cells_per_launch
        if kname in ("csm_long_kernel", "band_kernel"):
            algo_bytes += 48.0 * 2 * T_FRAMES * (cells_per_launch / float((T_FRAMES - M_STACK) ** 2))
        achieved = algo_bytes / (avg_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if kname in tj and tj[kname].get("cells_per_launch"):
                    traffic = tj[kname]["hbm_bytes_per_launch"] * cells_per_launch / tj[kname]["cells_per_launch"]
            except Exception:
                traffic = None
        roofline = {"bound": "hbm", "kernel": kname, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                    "avg_launch_ms": round(avg_ms, 3), "algorithmic_bytes_per_launch": algo_bytes,
                    "kernels_ms_per_step": {k: round(v["ms"] / args.steps, 3) for k, v in prof.items()},
                    "chain": {"algorithmic_bytes_per_pair": chain_bytes_per_pair(T_FRAMES, T_FRAMES),
                              "achieved": round(value / world * chain_bytes_per_pair(T_FRAMES, T_FRAMES) / 1e9, 1),
                              "frac": round(value / world * chain_bytes_per_pair(T_FRAMES, T_FRAMES) / 1e9 / HBM_PEAK_GBS, 4)}}
        # ---- CPU baseline: the oracle on a bounded sample of the same pairs (rank 0, N = 1 only)
        cpu = None
        if world == 1 and args.cpu_pairs > 0:
            import oracle
            from concurrent.futures import ThreadPoolExecutor
            # one thread first (per-core figure), then the same C routine on up to 32 host threads
            # (ctypes releases the GIL; the oracle keeps no global state)
            ncpu = min(args.cpu_pairs, ppr)
            sample = mine[:ncpu]
            tc = time.perf_counter()
            ref = oracle.serra09_pairs(data["frames"], data["offsets"], sample)
            tcpu = time.perf_counter() - tc
            if not np.array_equal(ref, scores[:ncpu]):
                raise SystemExit("bench: GPU scores differ from the CPU oracle on the sampled pairs")
            cores = max(1, min(32, os.cpu_count() or 1))
            nmt = min(ppr, 4 * cores)
            chunks = [mine[a:a + 4] for a in range(0, nmt, 4)]
            tc = time.perf_counter()
            with ThreadPoolExecutor(cores) as ex:
                parts = list(ex.map(lambda ch: oracle.serra09_pairs(data["frames"], data["offsets"], ch), chunks))
            tmt = time.perf_counter() - tc
            if not np.array_equal(np.concatenate(parts), scores[:nmt]):
                raise SystemExit("bench: GPU scores differ from the CPU oracle on the sampled pairs (threaded leg)")
            cpu = {"value": round(nmt / tmt, 3), "unit": "track-pairs/s", "cores": cores, "kind": "port",
                   "sample": "first %d pairs of the same workload (T=%d), C oracle -O2 on %d threads; "
                             "scores bit-identical to the GPU's" % (nmt, T_FRAMES, cores),
                   "value_1core": round(ncpu / tcpu, 3), "sample_1core": "first %d pairs, 1 thread" % ncpu,
                   "host_cpus": os.cpu_count()}
        line = {
            "metric": "track-pairs/sec on N x N Serra09 Qmax (HPCP, T=2000)",
            "value": round(value, 1), "unit": "track-pairs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "synthetic %d tracks x %d-frame HPCP (seed 1234), Serra09 Qmax "
                                   "(m=9, tau=1, kappa=0.095, OTI), %d pairs per GPU per step, "
                                   "one all-gather of the scores per step"
                                   % (N_TRACKS, T_FRAMES, ppr),
                       "pairs_per_step": world * ppr, "frames_per_track": T_FRAMES,
                       "parallelism": "pair-grid sharded over %d GPU(s)" % world},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(line))
    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
