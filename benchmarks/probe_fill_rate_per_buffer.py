#!/usr/bin/env python3
"""Is some of this GPU's memory slower than the rest? N buffers of 1 GiB (hipMalloc through torch), zero_fill_pages over each one
separately (512 x 2 MiB, event-timed in the library), rate per buffer; then all of them in launches of 1024 pages that take their
pages (a) from one buffer at a time, (b) round-robin over all buffers."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("KVCACHED_LOG_LEVEL", "ERROR")
import torch  # noqa: E402
from kvcached_amd import capi  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 96
GiB, PAGE = 1 << 30, 2 << 20
capi.init("cuda:0", PAGE, False)
bufs = [torch.empty(GiB, dtype=torch.int8, device="cuda:0") for _ in range(N)]
torch.cuda.synchronize()


def timed_fill(ptrs, reps=3):
    capi.zero_fill_pages(ptrs, PAGE)
    capi.set_option(capi.OPT_PROFILE, 1)
    capi.reset_stats()
    for _ in range(reps):
        capi.zero_fill_pages(ptrs, PAGE)
    st = capi.get_stats()
    capi.set_option(capi.OPT_PROFILE, 0)
    return st["fill_bytes"] / st["fill_ms"] / 1e6


rates = []
for b in bufs:
    base = b.data_ptr()
    rates.append(round(timed_fill([base + i * PAGE for i in range(512)])))
srt = sorted(rates)
print(json.dumps({"buffers": N, "GBps_per_1GiB_buffer": {"min": srt[0], "p10": srt[N // 10], "p50": srt[N // 2], "p90": srt[9 * N // 10], "max": srt[-1]},
                  "in_allocation_order": rates}), flush=True)
allp = [[b.data_ptr() + i * PAGE for i in range(512)] for b in bufs]
one_at_a_time = [p for buf in allp[:8] for p in buf]
round_robin = [allp[j][i] for i in range(512) for j in range(8)]
import random
shuffled = list(one_at_a_time)
random.Random(0).shuffle(shuffled)
runs64 = [one_at_a_time[i:i + 64] for i in range(0, len(one_at_a_time), 64)]
random.Random(1).shuffle(runs64)
runs64 = [p for r in runs64 for p in r]
for v in (0, 5, 4, 0, 5, 4):
    capi.set_option(capi.OPT_FILL_VARIANT, v)
    print(json.dumps({"fill_variant": v, "8 GiB, pages taken buffer by buffer": round(timed_fill(one_at_a_time)),
                      "round-robin over 8 buffers": round(timed_fill(round_robin)), "shuffled page by page": round(timed_fill(shuffled)),
                      "runs of 64 adjacent pages, runs shuffled": round(timed_fill(runs64))}), flush=True)
capi.set_option(capi.OPT_FILL_VARIANT, 0)
capi.shutdown()
