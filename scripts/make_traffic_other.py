#!/usr/bin/env python3
"""HBM traffic of the companion legs' dominant kernels (bench_other.py) from a `scripts/profile_ef.sh <tag> other` run:
rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in passes of their own (KiB; gfx950: FETCH_SIZE counts 64 B per 128-B request of
wide coalesced reads, so the read side is doubled as the guide prescribes; WRITE_SIZE as reported), summed over the launches of a
kernel family and divided by their number -> profiles/pmc_traffic_other.json, stamped with the hash of the kernel sources.
bench_other.py reports `roofline.traffic` from it only when the hash is that of the build it runs.
    python scripts/make_traffic_other.py <tag>        (reads gpurun_out/prof_ef_<tag>/)"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench_other  # noqa: E402

tag = sys.argv[1]
src = os.path.join(ROOT, "gpurun_out", "prof_ef_" + tag)
# leg -> kernel-name prefixes that make ONE "launch" of the leg's dominant kernel family (bench_other.py's profile slot); the
# first prefix counts the launches.  serra09_covers: the two-rows-per-wave band kernels only (the T = 2000 launches of the f16x2
# leg are band_kernel<9, 8, ...>); earlyfusion: the default two-term fp16 GEMMs, mfcc + ssm and chroma together (the bf16x3
# comparison launches of the same leg are <., 0>).
# (round 6: the default fp16 GEMMs are ef_gemm_rect_persist_dma_kernel<0> (mfcc + ssm) and <1> (chroma))
fam = {"serra09_covers": ["band2_kernel"], "earlyfusion": ["ef_gemm_rect_persist_dma_kernel<0>", "ef_gemm_rect_persist_dma_kernel<1>"],
       "simple": ["simple_kernel"]}
acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
for f in glob.glob(src + "/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] not in ("FETCH_SIZE", "WRITE_SIZE"):
            continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("acx::", "")
        for leg, prefixes in fam.items():
            for i, prefix in enumerate(prefixes):
                if k.startswith(prefix):
                    acc[leg][r["Counter_Name"]] += float(r["Counter_Value"])
                    if i == 0:
                        cnt[leg][r["Counter_Name"]] += 1
out = {"kernel_source_sha16": bench_other.other_source_sha16(),
       "source": "scripts/profile_ef.sh %s other (bench_other.py --steps 2 --warmup 1 under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)" % tag}
for leg in acc:
    nf, nw = max(1, cnt[leg]["FETCH_SIZE"]), max(1, cnt[leg]["WRITE_SIZE"])
    fetch, write = acc[leg]["FETCH_SIZE"] * 1024.0, acc[leg]["WRITE_SIZE"] * 1024.0
    out[leg] = {"kernels": fam[leg], "launches_profiled": nf, "fetch_bytes_raw_per_launch": fetch / nf, "write_bytes_per_launch": write / nw,
                "hbm_bytes_per_launch": 2.0 * fetch / nf + write / nw}
json.dump(out, open(os.path.join(ROOT, "profiles", "pmc_traffic_other.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
