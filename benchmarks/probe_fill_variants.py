#!/usr/bin/env python3
"""zero_fill_pages placement variants on bench.py's own cycle (pages from the library's pool: 64-page extents, the batch sorted
into runs), alternating in one process: 0 = XCD x owns pages x, x+8, ... (default), 4 = XCD x owns a contiguous eighth of the list,
3 = 1024-thread workgroups. Event-timed fill rate per variant.

    python benchmarks/probe_fill_variants.py [--reps 3] [--steps 20]
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
    args = ap.parse_args()
    os.environ.setdefault("KVCACHED_LOG_LEVEL", "ERROR")
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("needs a GPU")
    import bench
    from kvcached_amd import capi
    real_init = capi.init
    for rep in range(args.reps):
        for variant in (0, 4, 3):
            def init(*a, _v=variant, **k):          # measure() initialises the library itself: set the variant right behind it
                real_init(*a, **k)
                capi.set_option(capi.OPT_FILL_VARIANT, _v)
            capi.init = init
            try:
                r = bench.measure(capi, "cuda:0", args.steps, 4, "compat", None)
                s = bench.summarize(r, args.steps)
                rf = bench.roofline_from(r["stats"])
            finally:
                capi.init = real_init
            print(json.dumps({"rep": rep, "fill_variant": variant, "cycle_GBps": round(s["GBps"]), "fill_GBps": rf and rf["achieved"],
                              "avg_launch_us": rf and rf["avg_launch_us"], "launches": rf and rf["launches"]}), flush=True)


if __name__ == "__main__":
    main()
