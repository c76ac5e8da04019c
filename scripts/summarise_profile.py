#!/usr/bin/env python3
"""Summarise a scripts/profile.sh output directory into profiles/<tag>_summary.md and
profiles/pmc_traffic.json.

HBM bytes: rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB.  Per MI355X_MICROARCH.md
(HBM section) on gfx950 FETCH_SIZE counts 64 B per 128-B request for wide coalesced
streaming reads, i.e. HALF the bytes: the read side is doubled here and both raw and
corrected values are listed.  WRITE_SIZE is taken as is (uncalibrated per that section)."""
import csv
import hashlib
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def kernel_source_sha16():
    """Hash of the Serra09 kernel sources: bench.py reports the committed counters only for the build they were
    taken on (same function in bench.py)."""
    h = hashlib.sha256()
    for f in ("serra09_kernels.hpp", "serra09_band2_kernels.hpp", "acx_band.hip", "Makefile"):
        with open(os.path.join(ROOT, "acoss_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = sys.argv[2] if len(sys.argv) > 2 else os.path.join("gpurun_out", "prof_" + tag)
cells_per_launch = float(sys.argv[3]) if len(sys.argv) > 3 else 8192 * 1991.0 * 1991.0


def short(name):
    return name.split("(")[0].replace("void ", "").replace("acx::", "")


stats = {}
with open(os.path.join(src, "stats", "stats_kernel_stats.csv")) as f:
    for row in csv.DictReader(f):
        stats[short(row["Name"])] = dict(calls=int(row["Calls"]), avg_ms=float(row["AverageNs"]) / 1e6,
                                         pct=float(row["Percentage"]))

pmc = defaultdict(lambda: defaultdict(list))
for sub in sorted(os.listdir(src)):
    p = os.path.join(src, sub, "pmc_counter_collection.csv")
    if not os.path.exists(p):
        continue
    with open(p) as f:
        for row in csv.DictReader(f):
            k = short(row["Kernel_Name"])
            pmc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
            pmc[k]["_vgpr"] = [float(row["VGPR_Count"])]
            pmc[k]["_lds"] = [float(row["LDS_Block_Size"])]

lines = ["# rocprofv3 summary `%s`" % tag, "",
         "Command: `python bench.py --no-cpu --no-other --steps 2 --warmup 1 --tracks 640` "
         "(8192 pairs of 1991 x 1991 cells per launch); one `--kernel-trace --stats` pass and separate "
         "`--pmc` passes (scripts/profile.sh).", "",
         "| kernel | calls | avg ms | % time | VGPR | LDS B | FETCH KiB | WRITE KiB | HBM GB (2 x fetch + write) | GB/s |",
         "|---|---|---|---|---|---|---|---|---|---|"]
traffic = {}
for k, st in sorted(stats.items(), key=lambda kv: -kv[1]["avg_ms"]):
    c = pmc.get(k, {})
    fl, wl = c.get("FETCH_SIZE", [0]), c.get("WRITE_SIZE", [0])
    fetch = sum(fl) / max(1, len(fl))
    write = sum(wl) / max(1, len(wl))
    hbm = (2 * fetch + write) * 1024.0
    gbs = hbm / (st["avg_ms"] * 1e-3) / 1e9 if st["avg_ms"] > 0 else 0
    lines.append("| %s | %d | %.3f | %.2f | %d | %d | %.0f | %.0f | %.3f | %.0f |" % (
        k, st["calls"], st["avg_ms"], st["pct"], c.get("_vgpr", [0])[0], c.get("_lds", [0])[0],
        fetch, write, hbm / 1e9, gbs))
    if "kernel" in k and hbm > 0:
        # template instances of one kernel (band_kernel<.., role>) are averaged over their launches
        base = k.split("<")[0]
        t = traffic.setdefault(base, dict(hbm_bytes_per_launch=0.0, cells_per_launch=cells_per_launch,
                                          fetch_kib_raw=0.0, write_kib_raw=0.0, avg_ms=0.0, _calls=0))
        n0, n1 = t["_calls"], st["calls"]
        for key, val in (("hbm_bytes_per_launch", hbm), ("fetch_kib_raw", fetch), ("write_kib_raw", write),
                         ("avg_ms", st["avg_ms"])):
            t[key] = (t[key] * n0 + val * n1) / (n0 + n1)
        t["_calls"] = n0 + n1
lines += ["", "## SQ counters (per launch, averaged)", ""]
names = sorted({n for k in pmc for n in pmc[k] if not n.startswith("_") and n not in ("FETCH_SIZE", "WRITE_SIZE")})
lines.append("| kernel | " + " | ".join(names) + " |")
lines.append("|---|" + "---|" * len(names))
for k in sorted(stats, key=lambda kk: -stats[kk]["avg_ms"]):
    if k not in pmc:
        continue
    vals = []
    for n in names:
        v = pmc[k].get(n)
        vals.append("%.3g" % (sum(v) / len(v)) if v else "-")
    lines.append("| %s | " % k + " | ".join(vals) + " |")
# ---- what the kernels are really bound by: pipe utilisations from the SQ counters (MI355X_MICROARCH.md:
# SQ_ACTIVE_INST_* count quad-cycles summed over waves, SQ_VALU_MFMA_BUSY_CYCLES cycles summed over SIMDs,
# SQ_LDS_IDX_ACTIVE cycles summed over CUs; GRBM_GUI_ACTIVE = the kernel's own clock ticks)
N_SIMD, N_CU, N_XCD = 1024.0, 256.0, 8.0      # (GRBM_GUI_ACTIVE comes back summed over the 8 XCDs)
real = {}
lines += ["", "## Pipe utilisation (real bound)", "",
          "| kernel | clock GHz (GRBM_GUI_ACTIVE / wall) | VALU busy | f32-MFMA busy | LDS busy | HBM util (corrected traffic / 8 TB/s) |", "|---|---|---|---|---|---|"]
for k in sorted(stats, key=lambda kk: -stats[kk]["avg_ms"]):
    c = pmc.get(k, {})
    if "GRBM_GUI_ACTIVE" not in c:
        continue
    mean = lambda name: sum(c[name]) / len(c[name]) if name in c and len(c[name]) else 0.0
    gui = mean("GRBM_GUI_ACTIVE") / N_XCD
    wall = stats[k]["avg_ms"] * 1e-3
    fl, wl = c.get("FETCH_SIZE", [0]), c.get("WRITE_SIZE", [0])
    hbm = (2 * sum(fl) / max(1, len(fl)) + sum(wl) / max(1, len(wl))) * 1024.0
    r = dict(clock_ghz=gui / wall / 1e9 if wall > 0 else 0.0,
             valu_busy=4.0 * mean("SQ_ACTIVE_INST_VALU") / (gui * N_SIMD) if gui else 0.0,
             mfma_busy=mean("SQ_VALU_MFMA_BUSY_CYCLES") / (gui * N_SIMD) if gui else 0.0,
             lds_busy=mean("SQ_LDS_IDX_ACTIVE") / (gui * N_CU) if gui else 0.0,
             hbm_util=hbm / wall / 8e12 if wall > 0 else 0.0)
    lines.append("| %s | %.2f | %.3f | %.3f | %.3f | %.4f |" % (k, r["clock_ghz"], r["valu_busy"], r["mfma_busy"], r["lds_busy"], r["hbm_util"]))
    base = k.split("<")[0]
    if not base.endswith("_kernel") or "::" in base:
        continue
    acc = real.setdefault(base, dict(n=0))
    for key, val in r.items():
        acc[key] = (acc.get(key, 0.0) * acc["n"] + val) / (acc["n"] + 1)
    acc["n"] += 1
for v in real.values():
    v.pop("n", None)
    for key in list(v):
        v[key] = round(v[key], 4)
    v["source"] = ("profiles/%s_summary.md (rocprofv3 --pmc, separate passes; not measured in the bench run); "
                   "valu_busy = 4 x SQ_ACTIVE_INST_VALU / (clock ticks x 1024 SIMDs) reads 1.32 for the pure-VALU qmax_bits_kernel, "
                   "i.e. it over-counts by up to a third" % tag)
os.makedirs("profiles", exist_ok=True)
open(os.path.join("profiles", tag + "_summary.md"), "w").write("\n".join(lines) + "\n")
real["kernel_source_sha16"] = kernel_source_sha16()
real["source_tag"] = tag
json.dump(real, open(os.path.join("profiles", "real_bound.json"), "w"), indent=1)
for t in traffic.values():
    t.pop("_calls", None)
traffic["source"] = tag
traffic["kernel_source_sha16"] = kernel_source_sha16()
json.dump(traffic, open(os.path.join("profiles", "pmc_traffic.json"), "w"), indent=1)
print("\n".join(lines))
