#!/usr/bin/env python3
"""N buffers of 1 GiB in allocation order; every group of 8 consecutive buffers is filled by launches of 1024 pages taken
round-robin from the group's 8 buffers (the regime in which a launch touches 8 allocations: ~7.0 TB/s). Does the rate depend on
WHERE in the device's memory the group lies?"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("KVCACHED_LOG_LEVEL", "ERROR")
import torch  # noqa: E402
from kvcached_amd import capi  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 248
PAD_MIB = int(sys.argv[2]) if len(sys.argv) > 2 else 0          # every buffer that much LARGER than 1 GiB (only the first GiB is used)
GiB, PAGE = 1 << 30, 2 << 20
capi.init("cuda:0", PAGE, False)
bufs = [torch.empty(GiB + (PAD_MIB << 20), dtype=torch.int8, device="cuda:0") for _ in range(N)]
torch.cuda.synchronize()


def timed_fill(ptrs, reps=3):
    capi.zero_fill_pages(ptrs, PAGE)
    capi.set_option(capi.OPT_PROFILE, 1)
    capi.reset_stats()
    for _ in range(reps):
        capi.zero_fill_pages(ptrs, PAGE)
    st = capi.get_stats()
    capi.set_option(capi.OPT_PROFILE, 0)
    return round(st["fill_bytes"] / st["fill_ms"] / 1e6)


rates = []
for g in range(0, N - 7, 8):
    grp = bufs[g:g + 8]
    ptrs = [grp[j].data_ptr() + i * PAGE for i in range(128) for j in range(8)]
    rates.append(timed_fill(ptrs))
print(json.dumps({"buffers": N, "pad_MiB": PAD_MIB, "GBps_per_group_of_8_buffers_in_allocation_order": rates,
                  "first_buffer_addresses_GiB": [round(bufs[g].data_ptr() / GiB, 1) for g in range(0, N - 7, 8)]}), flush=True)
# The same memory with HOLES: free every other buffer, then fill groups of 8 of the remaining ones (each 2 GiB from the next)
del bufs[1::2]
torch.cuda.empty_cache()
torch.cuda.synchronize()
rates = []
for g in range(0, len(bufs) - 7, 8):
    grp = bufs[g:g + 8]
    rates.append(timed_fill([grp[j].data_ptr() + i * PAGE for i in range(128) for j in range(8)]))
print(json.dumps({"every other buffer freed: GBps per group of 8 remaining buffers": rates}), flush=True)
# and pieces of 128 MiB taken from 8 buffers that are 30 buffers apart in allocation order
far = bufs[::15][:8]
print(json.dumps({"8 buffers 30 GiB apart, round-robin": timed_fill([far[j].data_ptr() + i * PAGE for i in range(128) for j in range(8)])}), flush=True)
capi.shutdown()
