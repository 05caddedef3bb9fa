#!/usr/bin/env python3
"""Strict compat next to relaxed compat (KVCACHED_UNMAP_INVALIDATION_US) on bench.py's own cycle, a few
alternating repetitions in one process, with every timed map call's duration: the relaxed mode's gain depends on the next map
finding a batch of idle pages NEXT TO the parked one (GpuContext::reserve_target_bytes), and this shows whether it does.

    python benchmarks/probe_relaxed_mode.py [--reps 3] [--steps 20] [--us 300]
"""
from __future__ import annotations

import argparse
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--us", default="300")
    args = ap.parse_args()
    os.environ.setdefault("KVCACHED_LOG_LEVEL", "ERROR")
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("needs a GPU")
    import bench
    from kvcached_amd import capi
    device = "cuda:0"
    for rep in range(args.reps):
        for name, env in (("strict", {}), ("relaxed", {"KVCACHED_UNMAP_INVALIDATION_US": args.us})):
            saved = {k: os.environ.get(k) for k in env}
            os.environ.update(env)
            try:
                r = bench.measure(capi, device, args.steps, 4, "compat", None)
                s = bench.summarize(r, args.steps)
            finally:
                for k, v in saved.items():
                    os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
            print(json.dumps({"rep": rep, "mode": name, "GBps": round(s["GBps"]), "map_zero_GBps": round(s["map_zero_GBps"]),
                              "p50_map_ms": round(s["p50_map_batch_ms"], 3), "unmap_us_per_page": round(s["unmap_us_per_page"], 3),
                              "driver_us_per_page": s.get("driver_us_per_page"), "handles_created": s.get("handles_created"),
                              "host_us_per_call": s.get("host_us_per_call"), "tlb_shootdown_us": s.get("tlb_shootdown_us"),
                              "map_ms": [round(t * 1e3, 2) for t in r["per_step"]],
                              "unmap_ms": [round(t * 1e3, 2) for t in r["per_unmap"]]}), flush=True)


if __name__ == "__main__":
    main()
