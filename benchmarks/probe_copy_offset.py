"""How much of a copy's rate depends on the distance between source and destination: the same compact_blocks kernel on ONE
region, moves i -> i + offset (the "contiguous 2 GiB -> 2 GiB copy" is not the ceiling of the scattered case)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("KVCACHED_LOG_LEVEL", "ERROR")
import torch
from kvcached_amd import capi
capi.init("cuda:0", 2 << 20, False)
KiB = 1 << 10
big = torch.randint(0, 127, (5 << 30,), dtype=torch.int8, device="cuda:0")
torch.cuda.synchronize()
def run(block, n, off_blocks, label):
    src = list(range(n)); dst = [i + off_blocks for i in src]
    for _ in range(2): capi.compact_blocks([big.data_ptr()], src, dst, block)
    capi.set_option(capi.OPT_PROFILE, 1); capi.reset_stats()
    for _ in range(5): capi.compact_blocks([big.data_ptr()], src, dst, block, sync=False)
    capi.compact_blocks([big.data_ptr()], src[:1], dst[:1], block, sync=True)
    st = capi.get_stats(); capi.set_option(capi.OPT_PROFILE, 0)
    print(json.dumps({"what": label, "block": block, "n": n, "offset_bytes": off_blocks * block, "GBps": round(st["compact_bytes"] / st["compact_ms"] / 1e6)}), flush=True)
for v in (0, 10):
    capi.set_option(capi.OPT_COMPACT_VARIANT, v)
    run(2 << 20, 1024, 1024, f"variant {v}: 2 GiB apart exactly")
    run(2 << 20, 1024, 1025, f"variant {v}: 2 GiB + 2 MiB apart")
    run(32 * KiB, 65536, 65536, f"variant {v}: 32 KiB blocks, 2 GiB apart")
    run(32 * KiB, 65536, 65536 + 13, f"variant {v}: 32 KiB blocks, 2 GiB + 416 KiB apart")
    run(32 * KiB, 65536, 65536 + 1, f"variant {v}: 32 KiB blocks, 2 GiB + 32 KiB apart")
    run(4096, 262144, 262144 + 3, f"variant {v}: 4 KiB blocks, 1 GiB + 12 KiB apart")
capi.shutdown()
