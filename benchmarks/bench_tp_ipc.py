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
    python benchmarks/bench_tp_ipc.py --shared-pool [--workers 1,3]     # the north-star TP sharing mode

--shared-pool: this process is rank 0 AND the scheduler: it backs n page ids itself, exports one dmabuf fd per 2 MiB
slot, ships them with SCM_RIGHTS over the workers' sockets and the W peers (ranks 1..W) import and map the SAME
physical pages at the same offsets (`share_mapped_slots`); unmap goes to all ranks. On the 1-GPU box the peers share
the device (what is measured is the export/ship/import/map machinery, not xGMI).
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


def shared_pool(args):
    ipc = f"kvc_tpshare_{os.getpid()}"
    os.environ["KVCACHED_IPC_NAME"] = ipc
    if args.units == "slots":            # one dmabuf per 2 MiB slot (exportable single pages); "page_ids": one per buffer of lanes
        os.environ["KVCACHED_EXPORTABLE_HANDLES"] = "1"
    os.environ.setdefault("KVCACHED_LOG_LEVEL", "ERROR")
    import torch
    n_gpus = torch.cuda.device_count()
    torch.cuda.set_device(0)
    from kvcached_amd import tp_ipc_util as tp
    from kvcached_amd import vmm_ops
    ctx = mp.get_context("spawn")
    per_layer = 256 * PAGE
    vmm_ops.init_kvcached("cuda:0", PAGE, False)
    vmm_ops.create_kv_tensors(per_layer * 2, 1, "cuda:0", args.layers, 2, 0, False)
    tp.start_worker_listener_thread(0)                 # rank 0's own listener: broadcasts reach this process too
    for W in [int(x) for x in args.workers.split(",")]:
        ready, stop = ctx.Queue(), ctx.Event()
        procs = [ctx.Process(target=worker, args=(r, n_gpus, args.layers, per_layer, ipc, ready, stop)) for r in range(1, W + 1)]
        for p in procs:
            p.start()
        for _ in procs:
            ready.get(timeout=180)
        tp_size = W + 1
        assert tp.broadcast_kv_tensors_created(tp_size)
        res = {"mode": "shared pool: rank 0 backs + exports, peers import + map the same pages", "units": args.units, "peers": W,
               "gpus_visible": n_gpus, "layers": args.layers}
        for n in (1, 8):
            t_back, t_share, t_unmap = [], [], []
            for it in range(args.iters):
                offs = [((it * n + i) % 192) * PAGE for i in range(n)]
                t0 = time.perf_counter()
                assert vmm_ops.map_to_kv_tensors(offs)
                t1 = time.perf_counter()
                tp.share_mapped_slots(tp_size, offs, pp_rank=0, group_id=0, src_rank=0)
                t2 = time.perf_counter()
                tp.broadcast_unmap_from_kv_tensors(tp_size, offs)
                t3 = time.perf_counter()
                t_back.append(t1 - t0), t_share.append(t2 - t1), t_unmap.append(t3 - t2)
            slots = n * args.layers * 2
            res[f"{n}_page_ids"] = {"slots_2MiB": slots,
                                    "rank0_back_ms_p50": round(statistics.median(t_back) * 1e3, 3),
                                    "export_ship_import_map_on_all_peers_ms_p50": round(statistics.median(t_share) * 1e3, 3),
                                    "us_per_slot_per_peer": round(statistics.median(t_share) * 1e6 / slots / W, 2),
                                    "unmap_all_ranks_ms_p50": round(statistics.median(t_unmap) * 1e3, 3)}
        print(json.dumps(res), flush=True)
        tp._channels.close()
        stop.set()
        for p in procs:
            p.join(timeout=60)
    vmm_ops.shutdown_kvcached()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workers", default=None)
    ap.add_argument("--layers", type=int, default=32)
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--shared-pool", action="store_true")
    ap.add_argument("--units", choices=["page_ids", "slots"], default="page_ids", help="--shared-pool: what a descriptor stands for")
    args = ap.parse_args()
    if args.shared_pool:
        args.workers = args.workers or "1,3"
        return shared_pool(args)
    args.workers = args.workers or "1,2,4"
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
