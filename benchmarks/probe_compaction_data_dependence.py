#!/usr/bin/env python3
"""Does compact_blocks' rate depend on WHAT it moves? The same 2048 moves on the same 64 library-backed KV regions, first with
the regions all zero (as the library's fill leaves them), then with random bytes in them, then zero again."""
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("KVCACHED_LOG_LEVEL", "ERROR")
import numpy as np  # noqa: E402
import torch  # noqa: E402
from kvcached_amd import capi  # noqa: E402

PAGE, block, n_blocks, regions, moves = 2 << 20, 32 * 1024, 4096, 64, 2048
capi.init("cuda:0", PAGE, False)
capi.create_kv_tensors(2 * n_blocks * block, 1, "cuda:0", regions // 2, 2, 0, False)
capi.map_to_kv_tensors([p * PAGE for p in range(n_blocks * block // PAGE)])
bases = capi.get_region_bases(0)
ids = np.random.default_rng(0).permutation(n_blocks)[:2 * moves]
src, dst = [int(x) for x in ids[:moves]], [int(x) for x in ids[moves:]]
hip = ctypes.CDLL("libamdhip64.so")


def fill(kind):
    if kind == "zeros":
        t = torch.zeros(n_blocks * block, dtype=torch.int8, device="cuda:0")
    elif kind == "random bytes":
        t = torch.randint(-128, 127, (n_blocks * block,), dtype=torch.int8, device="cuda:0")
    else:                                      # bytes 0x55 / 0xAA alternating: every bit flips from byte to byte
        t = torch.tensor([0x55, -86], dtype=torch.int8, device="cuda:0").repeat(n_blocks * block // 2)
    torch.cuda.synchronize()
    for b in bases:
        assert hip.hipMemcpy(ctypes.c_void_p(b), ctypes.c_void_p(t.data_ptr()), ctypes.c_size_t(n_blocks * block), 3) == 0
    del t
    torch.cuda.synchronize()


def rate(src=src, dst=dst):
    for _ in range(2):
        capi.compact_blocks(bases, src, dst, block)
    capi.set_option(capi.OPT_PROFILE, 1)
    capi.reset_stats()
    for _ in range(8):
        capi.compact_blocks(bases, src, dst, block, sync=False)
    capi.compact_blocks(bases[:1], src[:1], dst[:1], block, sync=True)
    st = capi.get_stats()
    capi.set_option(capi.OPT_PROFILE, 0)
    return round(st["compact_bytes"] / st["compact_ms"] / 1e6)


for kind in ("zeros", "random bytes", "zeros"):
    fill(kind)
    print(json.dumps({"contents": kind, "GBps": rate()}), flush=True)
# the same regions, the same bytes, NOT scattered: block i -> i + 2048 in every region (two plain streams per region)
print(json.dumps({"contents": "zeros", "moves": "sequential: i -> i + 2048", "GBps": rate(list(range(2048)), list(range(2048, 4096)))}), flush=True)
# and only the first 8 page ids of every region (one 16 MiB run of lanes per row), random pairing inside them
ids8 = np.random.default_rng(1).permutation(512)
print(json.dumps({"contents": "zeros", "moves": "random pairing inside the first 8 page ids (256 moves)",
                  "GBps": rate([int(x) for x in ids8[:256]], [int(x) for x in ids8[256:]])}), flush=True)
slots = [b + p * PAGE for b in bases for p in range(n_blocks * block // PAGE)]
capi.set_option(capi.OPT_PROFILE, 1)
capi.reset_stats()
for _ in range(4):
    capi.zero_fill_pages(slots, PAGE)
st = capi.get_stats()
capi.set_option(capi.OPT_PROFILE, 0)
print(json.dumps({"zero_fill_pages over the same 4096 slots, GBps": round(st["fill_bytes"] / st["fill_ms"] / 1e6)}), flush=True)
capi.unmap_from_kv_tensors([p * PAGE for p in range(n_blocks * block // PAGE)])
capi.shutdown()
