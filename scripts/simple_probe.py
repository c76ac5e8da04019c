"""Development aid: SiMPle throughput through the grid (bench_other's leg) for the library named by ACX_LIB."""
import json
import sys

sys.path.insert(0, ".")
import bench_other  # noqa: E402
from acoss_amd import _lib  # noqa: E402

ctx = _lib.Context(0)
r = bench_other.simple_leg(ctx, steps=3, warmup=1)
print(_lib.LIB_PATH.split("/")[-1], r["value"], r["roofline"]["kernel_ms_per_step"])
ctx.close()
