"""From a rocprofv3 --kernel-trace CSV of `python3 bench.py ...`: the launches of the Serra09 kernels that belong to the
bench's warm-up + timed steps (the ones with the step's full grid; the `other` legs and the self-check launch the same
kernels on smaller pair lists, which is why the plain --stats average of a kernel is a mixture) -> a markdown table.
    python scripts/timed_region_stats.py gpurun_out/prof_final/s_kernel_trace.csv > profiles/r04_bench_kernel_stats.md"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
by = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"].split("(")[0].replace("void ", "")
    by[n].append(((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6,
                  int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])))
print("| kernel | launches in the trace | mean ms (all) | launches with the step's grid | mean ms | min | max |")
print("|---|---|---|---|---|---|---|")
tot = 0.0
for n in sorted(by):
    if not (n.startswith("acx::band_kernel") or n.startswith("acx::qmax_bits") or n.startswith("acx::oti_kernel")):
        continue
    v = by[n]
    # the grid the headline's steps launch: the most frequent one (round 6: the strong sub-grid leg launches larger grids, a few
    # times; the steps appear warm-up + timed + once more in the kernel-clock pass behind the timed region)
    g, cnt = collections.Counter(x[1] for x in v).most_common(1)[0]
    big = [x[0] for x in v if x[1] == g]
    if len(big) < 20:          # not a kernel of the headline steps
        continue
    print("| `%s` | %d | %.3f | %d | **%.3f** | %.3f | %.3f |" % (n.replace("acx::", ""), len(v), sum(x[0] for x in v) / len(v), len(big),
                                                               sum(big) / len(big), min(big), max(big)))
    tot += sum(big) / len(big)
print()
print("Sum of the per-step means: %.3f ms (bench.py `ms_per_step` of an unprofiled run of the same build: see `r04_bench.json`)." % tot)
