#!/usr/bin/env python3
"""Latency of the SGLang token-index glue on one MI355X: the fused gfx950 kernels (csrc/index_kernels.hip)
next to the torch expressions the reference uses for the same step (kvcached/integration/sglang/patches.py:
alloc :192-196 = torch.tensor(list)+broadcast arithmetic; free :283-286 = torch.unique(idx // ps).cpu().tolist()).
SGLang's Triton alloc_extend/alloc_decode kernels are not in this image, so those two are reported alone.
Host wall time per call including the final synchronisation (what the scheduler thread waits for).
One JSON line per shape."""
from __future__ import annotations

import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))


def timed(fn, reps=200, warm=20):
    import torch
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return {"p50_us": round(ts[len(ts) // 2] * 1e6, 1), "p90_us": round(ts[int(len(ts) * 0.9)] * 1e6, 1)}


def main():
    import numpy as np
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("needs a GPU")
    os.environ.setdefault("KVCACHED_IPC_NAME", f"kvc_bench_glue_{os.getpid()}")
    os.environ.setdefault("KVCACHED_LOG_LEVEL", "ERROR")
    import kvc_glue_cases as G
    from kvcached_amd import vmm_ops
    from kvcached_amd.integration.sglang import allocators as A
    dev = "cuda:0"
    vmm_ops.init_kvcached(dev, 2 << 20, False)
    try:
        rng = np.random.default_rng(0)
        for n_blocks_alloc, tpb in ((1, 16), (64, 16), (512, 16), (1024, 16), (4096, 16)):
            ids = [int(x) for x in rng.permutation(1 << 20)[:n_blocks_alloc]]

            def ref_expr():
                page_ids = torch.tensor(ids, dtype=torch.int64, device=dev)
                return (page_ids[:, None] * tpb + torch.arange(tpb, device=dev)).reshape(-1)
            print(json.dumps({"op": "alloc: block ids -> token indices", "blocks": n_blocks_alloc, "tokens": n_blocks_alloc * tpb,
                              "hip": timed(lambda: A.expand_block_ids(ids, tpb, dev)), "torch_expr": timed(ref_expr)}), flush=True)
        for n_tok, tpb, n_blocks in ((16, 16, 140000), (2048, 16, 140000), (16384, 16, 140000), (131072, 16, 1 << 20)):
            idx = torch.tensor(rng.integers(0, min(n_blocks, 4 * n_tok // tpb + 1) * tpb, size=n_tok), dtype=torch.int64, device=dev)
            print(json.dumps({"op": "free: token indices -> sorted distinct block ids", "tokens": n_tok, "pool_blocks": n_blocks,
                              "hip": timed(lambda: A.unique_block_ids(idx, tpb, n_blocks)),
                              "torch_expr": timed(lambda: torch.unique(idx // tpb).cpu().numpy().tolist())}), flush=True)
        for bs, max_ext in ((1, 8192), (16, 2048), (64, 512), (256, 64)):
            c = G.extend_case(seed=bs, bs=bs, tpb=16, max_prefix=512, max_extend=max_ext)
            pre, seq, loc = (torch.tensor(c[k], device=dev) for k in ("prefix_lens", "seq_lens", "last_loc"))
            fp = [int(x) for x in c["free_pages"]]
            print(json.dumps({"op": "alloc_extend", "bs": bs, "tokens": c["extend_num_tokens"], "new_blocks": len(fp),
                              "hip": timed(lambda: A.alloc_extend_indices(pre, seq, loc, fp, 16, c["extend_num_tokens"]))}), flush=True)
        for bs in (1, 64, 256, 1024, 4096):
            c = G.decode_case(seed=bs, bs=bs, tpb=16, max_len=4000)
            seq, loc = torch.tensor(c["seq_lens"], device=dev), torch.tensor(c["last_loc"], device=dev)
            fp = [int(x) for x in c["free_pages"]]
            print(json.dumps({"op": "alloc_decode", "bs": bs, "new_blocks": len(fp),
                              "hip": timed(lambda: A.alloc_decode_indices(seq, loc, fp, 16))}), flush=True)
    finally:
        vmm_ops.shutdown_kvcached()


if __name__ == "__main__":
    main()
