#!/usr/bin/env python3
"""bench_tp_ipc — BASELINE.json config 4 (the reference's benchmarks/bench_tp_ipc): latency of the
scheduler -> TP-worker fan-out of map/unmap commands.

A scheduler process drives W worker processes (one per TP rank; on a multi-GPU node each owns a GPU,
on the 1-GPU development box they share cuda:0) through `broadcast_map_to_kv_tensors` /
`broadcast_unmap_from_kv_tensors` — the same public API as the reference. Reported per W:
  * rpc_us:   round trip of a command that does no GPU work (kv_tensors_created) — pure transport
  * map_ms / unmap_ms for n page ids in {1, 8, 64} (per-layer layout, `--layers` layers x K/V)
The reference publishes 2.10 ms (contiguous) / 35.96 ms (non-contiguous) for ONE page id at TP=4 on
4 x L40S (benchmarks/bench_tp_ipc/README.md:162,186); its transport alone (connect + pickle +
asyncio.run per call) is ~1.5-2 ms of that.

    python benchmarks/bench_tp_ipc.py [--workers 1,2,4] [--layers 32] [--iters 30]
"""
from __future__ import annotations

import argparse
import json
import multiprocessing as mp
import os
import statistics
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
PAGE = 2 << 20


def worker(rank: int, n_gpus: int, layers: int, per_layer: int, ipc: str, ready, stop):
    os.environ["KVCACHED_IPC_NAME"] = ipc
    os.environ.setdefault("KVCACHED_LOG_LEVEL", "ERROR")
    import torch
    dev = f"cuda:{rank % max(1, n_gpus)}"
    torch.cuda.set_device(dev)
    from kvcached_amd import vmm_ops
    from kvcached_amd.tp_ipc_util import start_worker_listener_thread
    vmm_ops.init_kvcached(dev, PAGE, False)
    vmm_ops.create_kv_tensors(per_layer * 2, 1, dev, layers, 2, 0, False)
    start_worker_listener_thread(rank)
    ready.put(rank)
    stop.wait()
    vmm_ops.shutdown_kvcached()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workers", default="1,2,4")
    ap.add_argument("--layers", type=int, default=32)
    ap.add_argument("--iters", type=int, default=30)
    args = ap.parse_args()
    ipc = f"kvc_tpbench_{os.getpid()}"
    os.environ["KVCACHED_IPC_NAME"] = ipc
    os.environ.setdefault("KVCACHED_LOG_LEVEL", "ERROR")
    import torch
    n_gpus = torch.cuda.device_count()
    ctx = mp.get_context("spawn")
    for W in [int(x) for x in args.workers.split(",")]:
        ready, stop = ctx.Queue(), ctx.Event()
        procs = [ctx.Process(target=worker, args=(r, n_gpus, args.layers, 256 * PAGE, ipc, ready, stop)) for r in range(W)]
        for p in procs:
            p.start()
        for _ in procs:
            ready.get(timeout=180)
        from kvcached_amd import tp_ipc_util as tp
        res = {"workers": W, "gpus_visible": n_gpus, "layers": args.layers, "transport": "unix sockets, persistent"}
        assert tp.broadcast_kv_tensors_created(W)
        t = []
        for _ in range(300):
            t0 = time.perf_counter()
            tp.broadcast_kv_tensors_created(W)
            t.append(time.perf_counter() - t0)
        res["rpc_us"] = {"p50": round(statistics.median(t) * 1e6, 1), "p99": round(sorted(t)[int(.99 * len(t))] * 1e6, 1)}
        for n in (1, 8, 64):
            tm, tu = [], []
            for it in range(args.iters):
                offs = [((it * n + i) % 192) * PAGE for i in range(n)]
                t0 = time.perf_counter()
                tp.broadcast_map_to_kv_tensors(W, offs)
                tm.append(time.perf_counter() - t0)
                t0 = time.perf_counter()
                tp.broadcast_unmap_from_kv_tensors(W, offs)
                tu.append(time.perf_counter() - t0)
            slots = n * args.layers * 2
            res[f"map_{n}_page_ids_ms"] = {"p50": round(statistics.median(tm) * 1e3, 3), "mean": round(statistics.mean(tm) * 1e3, 3),
                                           "slots_2MiB_per_rank": slots}
            res[f"unmap_{n}_page_ids_ms"] = {"p50": round(statistics.median(tu) * 1e3, 3), "mean": round(statistics.mean(tu) * 1e3, 3)}
        print(json.dumps(res), flush=True)
        tp._channels.close()
        stop.set()
        for p in procs:
            p.join(timeout=60)


if __name__ == "__main__":
    main()
