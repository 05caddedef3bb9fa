#!/usr/bin/env python3
"""bench.py's compaction leg in a fresh process, then again after the process has churned through physical memory the way the
bench's own variants do (a default cycle, a 48 GiB growth burst, a cycle with one buffer per page). Shows how much of the leg's
rate is a property of where the driver places the library's buffers.

    python benchmarks/probe_compaction_after_churn.py
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("KVCACHED_LOG_LEVEL", "ERROR")
import bench  # noqa: E402
from kvcached_amd import capi  # noqa: E402


def leg(label):
    r = bench.compaction_roofline(capi, "cuda:0")
    print(json.dumps({"when": label, "random_GBps": r["achieved"], "planner_GBps": r["planner_moves_GBps"], "on_torch_buffers_GBps": r["on_torch_buffers_GBps"],
                      "contiguous_copy_GBps": r["copy_ceiling_GBps"]}), flush=True)


leg("fresh process")
r = bench.measure(capi, "cuda:0", 20, 4, "compat", None)
r = bench.measure(capi, "cuda:0", 24, 4, "compat", None, burst=True, prefault=False)
leg("after a default cycle and a 48 GiB growth burst")
os.environ["KVCACHED_PHYS_CHUNK_PAGES"] = "1"
r = bench.measure(capi, "cuda:0", 24, 4, "compat", None, burst=True, prefault=False)
os.environ.pop("KVCACHED_PHYS_CHUNK_PAGES")
leg("after a growth burst with one buffer per page as well")
