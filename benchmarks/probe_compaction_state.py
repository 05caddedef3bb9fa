#!/usr/bin/env python3
"""The compaction of bench.py's leg (64 library-backed KV regions, 2048 random moves), a few launches and the event-timed rate -
small enough to run under `rocprofv3 --pmc ...` in order to compare the fast and the slow placement state by counters
(build/ab_state_counters.sh)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("KVCACHED_LOG_LEVEL", "ERROR")
import numpy as np  # noqa: E402
from kvcached_amd import capi  # noqa: E402

PAGE, block, n_blocks, regions, moves = 2 << 20, 32 * 1024, 4096, 64, 2048
ballast = None
if os.environ.get("KVC_PROBE_BALLAST_GIB"):      # hold that much device memory first: the library's buffers come from elsewhere
    import torch
    ballast = torch.empty(int(os.environ["KVC_PROBE_BALLAST_GIB"]) << 30, dtype=torch.int8, device="cuda:0")
    torch.cuda.synchronize()
capi.init("cuda:0", PAGE, False)
capi.create_kv_tensors(2 * n_blocks * block, 1, "cuda:0", regions // 2, 2, 0, False)
capi.map_to_kv_tensors([p * PAGE for p in range(n_blocks * block // PAGE)])
bases = capi.get_region_bases(0)
ids = np.random.default_rng(0).permutation(n_blocks)[:2 * moves]
src, dst = [int(x) for x in ids[:moves]], [int(x) for x in ids[moves:]]
for _ in range(2):
    capi.compact_blocks(bases, src, dst, block)
capi.set_option(capi.OPT_PROFILE, 1)
capi.reset_stats()
for _ in range(4):
    capi.compact_blocks(bases, src, dst, block, sync=False)
capi.compact_blocks(bases[:1], src[:1], dst[:1], block, sync=True)
st = capi.get_stats()
print(json.dumps({"ballast_GiB": int(os.environ.get("KVC_PROBE_BALLAST_GIB", "0")), "GBps": round(st["compact_bytes"] / st["compact_ms"] / 1e6)}), flush=True)
capi.unmap_from_kv_tensors([p * PAGE for p in range(n_blocks * block // PAGE)])
capi.shutdown()
