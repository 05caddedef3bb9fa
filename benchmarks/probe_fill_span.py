#!/usr/bin/env python3
"""zero_fill_pages rate against the SPAN of a launch's footprint: always 1024 pages of 2 MiB (2 GiB written per launch), taken as
every k-th page of the first S GiB of one big allocation, S = 2 .. 64 GiB; and against the chunking: contiguous chunks of c pages
spread evenly over 32 GiB. (a compact footprint fills ~12 % slower than a spread one.)"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("KVCACHED_LOG_LEVEL", "ERROR")
import torch  # noqa: E402
from kvcached_amd import capi  # noqa: E402

GiB, PAGE = 1 << 30, 2 << 20
capi.init("cuda:0", PAGE, False)
big = torch.empty(64 * GiB, dtype=torch.int8, device="cuda:0")
base = (big.data_ptr() + PAGE - 1) // PAGE * PAGE
torch.cuda.synchronize()


def timed_fill(ptrs, reps=4):
    capi.zero_fill_pages(ptrs, PAGE)
    capi.set_option(capi.OPT_PROFILE, 1)
    capi.reset_stats()
    for _ in range(reps):
        capi.zero_fill_pages(ptrs, PAGE)
    st = capi.get_stats()
    capi.set_option(capi.OPT_PROFILE, 0)
    return round(st["fill_bytes"] / st["fill_ms"] / 1e6)


for variant in (0, 5):
    capi.set_option(capi.OPT_FILL_VARIANT, variant)
    out = {"fill_variant": variant, "what": "1024 pages = every k-th page of the first S GiB", "GBps_by_span_GiB": {}}
    for span in (2, 3, 4, 6, 8, 16, 32, 63):
        n_avail = span * GiB // PAGE
        ptrs = [base + (i * n_avail // 1024) * PAGE for i in range(1024)]
        out["GBps_by_span_GiB"][span] = timed_fill(ptrs)
    print(json.dumps(out), flush=True)
    out = {"fill_variant": variant, "what": "1024 pages in contiguous chunks of c pages, chunks spread evenly over 32 GiB", "GBps_by_chunk_pages": {}}
    for c in (1, 8, 64, 256, 512, 1024):
        chunks = 1024 // c
        ptrs = [base + ((j * (16384 // chunks)) + i) * PAGE for j in range(chunks) for i in range(c)]
        out["GBps_by_chunk_pages"][c] = timed_fill(ptrs)
    print(json.dumps(out), flush=True)
    out = {"fill_variant": variant, "what": "one contiguous 2 GiB range starting at offset X GiB of the allocation", "GBps_by_start_GiB": {}}
    for start in (0, 1, 2, 3, 5, 8, 13, 21, 34, 55):
        ptrs = [base + (start * 512 + i) * PAGE for i in range(1024)]
        out["GBps_by_start_GiB"][start] = timed_fill(ptrs)
    print(json.dumps(out), flush=True)
import random
rng = random.Random(7)
for variant in (0, 5):
    capi.set_option(capi.OPT_FILL_VARIANT, variant)
    out = {"fill_variant": variant, "what": "1024 pages drawn at RANDOM (no regular stride) from the first S GiB of the one allocation", "GBps_by_span_GiB": {}}
    for span in (2, 4, 8, 16, 63):
        n_avail = span * GiB // PAGE
        ptrs = [base + i * PAGE for i in (rng.sample(range(n_avail), 1024) if n_avail > 1024 else rng.sample(range(1024), 1024))]
        out["GBps_by_span_GiB"][span] = timed_fill(ptrs)
    print(json.dumps(out), flush=True)
del big
torch.cuda.empty_cache()
# the same bytes in SEPARATE allocations: 64 x 1 GiB, pages taken round-robin / at random / buffer by buffer
bufs = [torch.empty(GiB, dtype=torch.int8, device="cuda:0") for _ in range(64)]
torch.cuda.synchronize()
allp = [[b.data_ptr() + i * PAGE for i in range(512)] for b in bufs]
for variant in (0, 5):
    capi.set_option(capi.OPT_FILL_VARIANT, variant)
    out = {"fill_variant": variant, "what": "1024 pages from k separate 1 GiB allocations (1024/k consecutive pages of each, buffer by buffer)", "GBps_by_k": {}}
    for k in (1, 2, 4, 8, 16, 64):
        ptrs = [p for buf in allp[:k] for p in buf[:1024 // k]] if k > 1 else (allp[0] + allp[1])[:1024]
        out["GBps_by_k"][k] = timed_fill(ptrs)
    print(json.dumps(out), flush=True)
capi.shutdown()
