"""Synthetic workloads for the benchmarks and the parity fixtures.

`poisson_trace` is the Llama-3-8B elastic grow/shrink workload of BASELINE.json config 3: Poisson
arrivals, log-normal prompt lengths, geometric output lengths; a request allocates
ceil(prompt/block) blocks on arrival, one more block every `block_size` decoded tokens, and frees
everything when it completes. The result is a list of trace ops in the tiny language of
tests/kvc_testlib.py (["a", req, n] / ["f", req]), ordered by virtual time; nothing here depends on
what the allocator returns, so the same list drives the reference, the oracle and the product.
"""
from __future__ import annotations

import heapq
from typing import Any, Dict, List

import numpy as np

# Llama-3-8B GQA KV geometry (32 layers, 8 KV heads x 128, bf16): 2048 B per token per layer per K|V
LLAMA3_8B = dict(num_layers=32, block_size=16, cell_size=8 * 128 * 2, num_kv_buffers=2)


def poisson_trace(rate: float, duration_s: float, seed: int = 1, block_size: int = 16, token_time_s: float = 0.02,
                  prompt_mu: float = 6.5, prompt_sigma: float = 1.0, prompt_clip=(16, 8192), out_p: float = 1 / 256,
                  out_clip=(1, 2048)) -> List[List[Any]]:
    rng = np.random.default_rng(seed)
    t, req = 0.0, 0
    # event heap entries: (time, seq, kind, req, payload); kinds: 0 arrival-alloc, 1 decode-alloc, 2 free
    heap: List[Any] = []
    seq = 0
    while True:
        t += rng.exponential(1.0 / rate)
        if t >= duration_s:
            break
        prompt = int(np.clip(rng.lognormal(prompt_mu, prompt_sigma), *prompt_clip))
        out = int(np.clip(rng.geometric(out_p), *out_clip))
        heapq.heappush(heap, (t, seq, 0, req, -(-prompt // block_size)))
        seq += 1
        used_in_last = prompt % block_size or block_size
        room = block_size - used_in_last          # tokens that still fit the last prompt block
        tok = room + 1
        while tok <= out:                         # decoded token `tok` needs a fresh block
            heapq.heappush(heap, (t + tok * token_time_s, seq, 1, req, 1))
            seq += 1
            tok += block_size
        heapq.heappush(heap, (t + out * token_time_s + 1e-9, seq, 2, req, 0))
        seq += 1
        req += 1
    ops: List[List[Any]] = []
    sub: Dict[int, int] = {}
    while heap:
        _, _, kind, r, n = heapq.heappop(heap)
        if kind == 2:
            for k in range(sub.get(r, 0)):        # each allocation of the request has its own id
                ops.append(["f", r * 4096 + k])
        else:
            k = sub.get(r, 0)
            sub[r] = k + 1
            ops.append(["a", r * 4096 + k, int(n)])
    return ops


def trace_stats(ops: List[List[Any]]) -> Dict[str, int]:
    live: Dict[int, int] = {}
    cur = peak = total = 0
    for op in ops:
        if op[0] == "a":
            live[op[1]] = op[2]
            cur += op[2]
            total += op[2]
            peak = max(peak, cur)
        elif op[0] == "f":
            cur -= live.pop(op[1], 0)
    return {"ops": len(ops), "blocks_allocated": total, "peak_live_blocks": peak}
