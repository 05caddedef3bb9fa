#!/usr/bin/env python3
"""Map parallelism across processes — the reference's benchmarks/bench_map_parallelism restated for this build
(its numbers, 4 x L40S: 4 processes x 25 page ids are 1.94x faster than 1 process x 100 in the contiguous layout,
1.11x in the non-contiguous one, README.md:92,147).

Serial: 1 process maps N page ids. Parallel: P processes map N/P each, started by a per-iteration barrier.
Single: 1 process maps N/P. Every worker owns its own kvcached (own VA, own handle pool). Workers use GPU
`rank % visible GPUs`: on a 1-GPU box they all share the device, which is the co-located-engines case; on the 8-GPU
node each gets its own GPU, which is the TP case. Llama-3-8B geometry (32 layers x K/V: 64 slots of 2 MiB per page id).
One JSON line per layout."""
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
LAYERS = 32


def worker(rank, n_gpus, contiguous, pages, iters, barrier, out):
    os.environ["KVCACHED_IPC_NAME"] = f"kvc_mappar_{os.getpid()}"
    os.environ.setdefault("KVCACHED_LOG_LEVEL", "ERROR")
    import torch
    dev = f"cuda:{rank % max(1, n_gpus)}"
    torch.cuda.set_device(dev)
    from kvcached_amd import capi
    capi.init(dev, PAGE, contiguous)
    capi.create_kv_tensors(256 * PAGE * 2, 1, dev, LAYERS, 2, 0, False)
    stride = PAGE * LAYERS * 2 if contiguous else PAGE
    offs = [i * stride for i in range(pages)]
    capi.map_to_kv_tensors(offs)           # warm-up: handles end up in the pool, like a running engine
    capi.unmap_from_kv_tensors(offs)
    walls = []
    for _ in range(iters):
        barrier.wait()
        t0 = time.perf_counter()
        capi.map_to_kv_tensors(offs)
        walls.append(time.perf_counter() - t0)
        barrier.wait()
        capi.unmap_from_kv_tensors(offs)
    out.put((rank, walls))
    capi.shutdown()


def run(procs, pages_per_proc, contiguous, iters, n_gpus):
    ctx = mp.get_context("spawn")
    barrier, out = ctx.Barrier(procs), ctx.Queue()
    ps = [ctx.Process(target=worker, args=(r, n_gpus, contiguous, pages_per_proc, iters, barrier, out)) for r in range(procs)]
    for p in ps:
        p.start()
    res = [out.get(timeout=600) for _ in ps]
    for p in ps:
        p.join(60)
    per_iter = [max(w[i] for _, w in res) for i in range(iters)]   # an iteration ends when its slowest worker does
    return {"mean_ms": round(statistics.mean(per_iter) * 1e3, 3), "p95_ms": round(sorted(per_iter)[int(0.95 * (iters - 1))] * 1e3, 3),
            "min_ms": round(min(per_iter) * 1e3, 3)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pages-total", type=int, default=100)
    ap.add_argument("--procs", type=int, default=4)
    ap.add_argument("--iters", type=int, default=20)
    args = ap.parse_args()
    import torch
    n_gpus = torch.cuda.device_count()     # counting devices does not initialise the GPU in this process
    per = args.pages_total // args.procs
    for contiguous in (True, False):
        serial = run(1, args.pages_total, contiguous, args.iters, n_gpus)
        parallel = run(args.procs, per, contiguous, args.iters, n_gpus)
        single = run(1, per, contiguous, args.iters, n_gpus)
        slots = LAYERS * 2
        print(json.dumps({"layout": "contiguous (128 MiB compound pages)" if contiguous else "non-contiguous (64 x 2 MiB per page id)",
                          "gpus_visible": n_gpus, "procs": args.procs, "page_ids_total": args.pages_total,
                          "serial_1proc_N": serial, f"parallel_{args.procs}procs_N_over_P": parallel, "single_1proc_N_over_P": single,
                          "speedup_parallel_vs_serial": round(serial["mean_ms"] / parallel["mean_ms"], 2),
                          "GBps_backed_serial": round(args.pages_total * slots * PAGE / (serial["mean_ms"] * 1e-3) / 1e9, 1),
                          "GBps_backed_parallel": round(args.pages_total * slots * PAGE / (parallel["mean_ms"] * 1e-3) / 1e9, 1)}),
              flush=True)


if __name__ == "__main__":
    main()
