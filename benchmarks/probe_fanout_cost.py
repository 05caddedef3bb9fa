#!/usr/bin/env python3
"""What one CollectiveFanout call costs on top of the local map/unmap it carries (world size 1 over RCCL: the floor of the
software path - broadcast and all-reduce kernels are launched even for one rank), by phase. bench.py's cycle, 1024 x 2 MiB."""
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29677")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
os.environ.setdefault("KVCACHED_LOG_LEVEL", "ERROR")
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PAGE = 2 << 20


def main():
    torch.cuda.set_device(0)
    backend = sys.argv[1] if len(sys.argv) > 1 else "nccl"
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda:0"))
    else:
        dist.init_process_group("gloo")
    from kvcached_amd import capi
    from kvcached_amd.tp_ipc_util import CMD_MAP, CMD_UNMAP, CollectiveFanout
    capi.init("cuda:0", PAGE, False)
    capi.create_kv_tensors(32 * 1024 * PAGE, 1, "cuda:0", 1, 1, 0, True)
    for deferred in (False, True):
        fan = CollectiveFanout(device="cuda:0" if backend == "nccl" else "cpu", deferred_status=deferred)
        rng = np.random.default_rng(0)
        batches = [np.asarray([(b * 1024 + int(p)) * PAGE for p in rng.permutation(1024)], dtype=np.int64) for b in range(8)]
        ph = {"exchange": 0.0, "apply": 0.0, "status": 0.0}
        for it in range(30):
            offs = batches[it % 8]
            for cmd in (CMD_MAP, CMD_UNMAP):
                t0 = time.perf_counter()
                t1 = t0
                c, g, o = fan._exchange(cmd, offs, 0)
                t2 = time.perf_counter()
                ok = fan._apply_staged(c, len(o), g) if deferred else fan._apply(c, o, g)
                t3 = time.perf_counter()
                fan._finish(ok)
                t4 = time.perf_counter()
                if it >= 5:
                    ph["exchange"] += t2 - t1
                    ph["apply"] += t3 - t2
                    ph["status"] += t4 - t3
        fan.finish()
        capi.flush_unmaps()
        print(json.dumps({"backend": backend, "deferred_status": deferred, "us_per_call": {k: round(v / 50 * 1e6, 1) for k, v in ph.items()}}), flush=True)
    # the same calls without any fan-out
    t = 0.0
    for it in range(30):
        arr = capi.i64_array(batches[it % 8].tolist())
        t0 = time.perf_counter()
        capi.map_to_kv_tensors(arr)
        capi.unmap_from_kv_tensors(arr)
        if it >= 5:
            t += time.perf_counter() - t0
    print(json.dumps({"local_us_per_call": round(t / 50 * 1e6, 1)}))
    capi.shutdown()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
