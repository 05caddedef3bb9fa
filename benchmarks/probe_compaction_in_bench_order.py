#!/usr/bin/env python3
"""Which of bench.py's earlier legs, if any, costs the compaction leg its rate when it runs behind them in one process."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("KVCACHED_LOG_LEVEL", "ERROR")
import bench  # noqa: E402
from kvcached_amd import capi  # noqa: E402


def leg(label):
    r = bench.compaction_roofline(capi, "cuda:0")
    print(json.dumps({"when": label, "random_GBps": r["achieved"], "planner_GBps": r["planner_moves_GBps"],
                      "on_torch_buffers_GBps": r["on_torch_buffers_GBps"], "contiguous_copy_GBps": r["copy_ceiling_GBps"]}), flush=True)


leg("fresh process")
bench.measure(capi, "cuda:0", 8, 2, "compat", None, backend="hip")
leg("after a cycle on the hip backend")
bench.measure(capi, "cuda:0", 8, 2, "compat", None, backend="hybrid")
leg("after a cycle on the hybrid backend")
bench.engine_geometry(capi, "cuda:0", "compat")
leg("after the engine-geometry leg (compat)")
bench.engine_geometry(capi, "cuda:0", "lazy")
leg("after the engine-geometry leg (lazy)")
bench.measure(capi, "cuda:0", 8, 2, "compat", None, compound_layers=32)
leg("after the contiguous-layout cycle")
