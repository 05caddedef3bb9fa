"""Inputs of the parity fixtures: the trace ops, geometries and shapes that oracle/gen_golden.py
feeds to the reference and that the tests feed to the oracle and the product. Deterministic
(seeded numpy Generators); the generated op lists are also stored inside the fixtures, so a numpy
upgrade cannot silently change what is being compared.
"""
from __future__ import annotations

import os
import sys
from typing import Any, Dict, List

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

MiB = 1 << 20
PAGE = 2 * MiB

# BASELINE.json configs: cfg1 = tests/test_kvcache_manager.py geometry, cfg3 = Llama-3-8B GQA
CFG1 = dict(num_blocks=65536, block_size=16, cell_size=1024, num_layers=16)
CFG3 = dict(num_blocks=147456, block_size=16, cell_size=2048, num_layers=32)


def shuffled_indices(num_blocks: int, n: int, seed: int) -> List[int]:
    rng = np.random.default_rng(seed)
    return [int(x) for x in rng.choice(num_blocks, size=n, replace=False)]


# ------------------------------------------------------------------ PageAllocator state machine
def _pa_ops_basic():
    ops = [["alloc"]] * 7                                  # slow path x7
    ops += [["frees", [3, 1, 5]], ["free", 0]]             # reserved fills FIFO
    ops += [["alloc"], ["alloc"]]                          # fast path pops reserved front: 3, 1
    ops += [["frees", [2, 4, 6, 3, 1]], ["trim"], ["alloc"], ["reset"], ["alloc"]]
    return [list(o) for o in ops]


def _pa_ops_overflow_reserved():
    ops = [["alloc"]] * 14
    ops += [["frees", list(range(13, -1, -1))]]            # 10 stay reserved, 4 unmapped
    ops += [["alloc"]] * 3 + [["free", 13], ["free", 12]]
    ops += [["trim"], ["alloc"], ["alloc"]]
    return [list(o) for o in ops]


def _pa_ops_resize():
    ops = [["alloc"]] * 4
    ops += [["resize", 10], ["target", 16], ["target", 10]]   # shrink 16 -> 10 straight from the free list
    ops += [["resize", 3]]                                 # refused: 4 in use
    ops += [["resize", 13]]                                # grow: reclaimed ids come back first (FIFO)
    ops += [["alloc"]] * 7
    ops += [["resize", 20]] + [["alloc"]] * 3              # brand-new ids 16..19 after the reused ones
    ops += [["frees", [0, 1, 2, 3, 4, 5]], ["resize", 9]]  # free list too short -> reserved unmapped first
    ops += [["alloc"], ["resize", 16], ["alloc"], ["alloc"]]
    return [list(o) for o in ops]


def _pa_ops_exhaust():
    ops = [["alloc"]] * 5                                  # 4 pages only: 5th raises
    ops += [["free", 2], ["alloc"], ["frees", [0, 1, 2, 3]], ["alloc"]]
    return [list(o) for o in ops]


PAGE_ALLOCATOR_CASES: List[Dict[str, Any]] = [
    dict(name="basic", pages=16, num_layers=2, num_kv_buffers=2, contiguous=False, ops=_pa_ops_basic()),
    dict(name="overflow_reserved", pages=32, num_layers=4, num_kv_buffers=2, contiguous=False,
         ops=_pa_ops_overflow_reserved()),
    dict(name="resize", pages=16, num_layers=2, num_kv_buffers=2, contiguous=False, ops=_pa_ops_resize()),
    dict(name="exhaust", pages=4, num_layers=1, num_kv_buffers=1, contiguous=False, ops=_pa_ops_exhaust()),
    dict(name="contiguous_offsets", pages=16, num_layers=8, num_kv_buffers=2, contiguous=True, ops=_pa_ops_basic()),
]


# ------------------------------------------------------------------ KVCacheManager traces
def _case(name, ops, full=True, **config):
    cfg = dict(num_blocks=4096, block_size=16, cell_size=2048, num_layers=2, world_size=1, reserve_null_block=False,
               num_kv_buffers=2, contiguous=False, phys_pages=1 << 40)
    cfg.update(config)
    return dict(name=name, config=cfg, ops=ops, full=full)


def manager_cases_small() -> List[Dict[str, Any]]:
    cases = []
    # cfg-1: the reference's own test sequence (tests/test_kvcache_manager.py:88-194): alloc 256 / free,
    # over-allocate by one, trim, reserve 512 / free_reserved. Physical memory limits the pool.
    cases.append(_case("cfg1_reference_test_sequence",
                       [["a", 0, 256], ["f", 0], ["ph", 100], ["a", 1, 100 * 128 + 1], ["a", 2, 100 * 128],
                        ["f", 2], ["tr"], ["rs", 512], ["fr"], ["ph", 1 << 40], ["a", 3, 1]],
                       **CFG1))
    cases.append(_case("null_block", [["a", 0, 1], ["a", 1, 62], ["a", 2, 2], ["f", 1], ["a", 3, 70], ["f", 0]],
                       reserve_null_block=True))
    cases.append(_case("contiguous_offsets", [["a", 0, 64 * 3 + 5], ["fp", 0, 0, 64], ["f", 0], ["tr"]],
                       contiguous=True, num_layers=4))
    cases.append(_case("mla_one_buffer", [["a", 0, 200], ["fp", 0, 10, 150], ["a", 1, 64], ["f", 0], ["f", 1], ["tr"]],
                       num_kv_buffers=1, cell_size=1152))
    # most-recently-touched partial page first (dict.popitem) and first-n-free inside a page
    cases.append(_case("partial_page_lifo",
                       [["a", 0, 64 * 4], ["fp", 0, 5, 10], ["fp", 0, 64, 70], ["fp", 0, 120, 135], ["a", 1, 3],
                        ["a", 2, 9], ["a", 3, 14], ["fp", 0, 0, 3], ["a", 4, 2], ["f", 1], ["a", 5, 1], ["f", 0],
                        ["a", 6, 80]]))
    cases.append(_case("reserve_then_alloc",
                       [["rs", 100], ["a", 0, 30], ["a", 1, 90], ["rs", 10], ["fr"], ["f", 0], ["f", 1], ["rs", 5],
                        ["a", 2, 5], ["a", 3, 1]]))
    cases.append(_case("phys_limited",
                       [["ph", 3], ["a", 0, 64 * 3], ["a", 1, 1], ["ph", 0], ["fp", 0, 0, 64], ["a", 2, 64], ["a", 3, 1],
                        ["f", 0], ["a", 4, 64 * 10], ["ph", 20], ["a", 5, 64 * 25], ["a", 6, 64 * 31]]))
    # blocks that do not divide the page: 768 KiB blocks, 2 per page, straddlers dropped
    cases.append(_case("straddling_blocks",
                       [["a", 0, 5], ["a", 1, 4], ["fp", 0, 1, 3], ["a", 2, 3], ["f", 1], ["f", 0], ["f", 2]],
                       num_blocks=96, block_size=16, cell_size=48 * 1024))
    # elastic limit: shrink that fits, grow that reuses reclaimed ids, shrink that must wait (in_shrink)
    P = 64 * 32 * 1024  # bytes of one page worth of blocks == PAGE
    cases.append(_case("resize_shrink_grow",
                       [["a", 0, 64 * 6], ["rz", 40 * P], ["a", 1, 64 * 2], ["rz", 50 * P], ["a", 2, 64 * 40],
                        ["f", 1], ["f", 0], ["rz", 64 * P], ["a", 3, 64 * 20], ["f", 2], ["f", 3], ["tr"]]))
    cases.append(_case("resize_deferred_in_shrink",
                       [["a", 0, 64 * 20], ["a", 1, 64 * 10], ["rz", 12 * P], ["a", 2, 1], ["f", 1], ["fp", 0, 0, 64 * 8],
                        ["a", 3, 64], ["rz", 64 * P], ["a", 4, 64 * 30], ["f", 0], ["f", 3], ["f", 4]]))
    # shrink while the free list is too short: the reserved (mapped, idle) pages are unmapped first
    cases.append(_case("resize_needs_reserved_pages",
                       [["a", 0, 64 * 60], ["f", 0], ["a", 1, 64 * 50], ["fp", 1, 0, 64 * 5], ["rz", 46 * P],
                        ["a", 2, 64], ["a", 3, 1], ["f", 1], ["f", 2]], num_blocks=64 * 64))
    return cases


def random_mix_ops(n_ops: int, seed: int, pool_pages: int, blocks_per_page: int = 64, with_resize: bool = True):
    """Random alloc / partial free / free / reserve / trim / resize / phys ops over a small pool, so that
    None returns, deferred shrinks and reserved-page overflow all occur."""
    rng = np.random.default_rng(seed)
    ops, live, sizes, nxt = [], [], {}, 0
    cap = pool_pages * blocks_per_page
    has_reserved = False
    for _ in range(n_ops):
        u = rng.random()
        if u < 0.42 or not live:
            n = int(rng.integers(1, max(2, cap // 6))) if rng.random() < 0.25 else int(rng.integers(1, 3 * blocks_per_page))
            ops.append(["a", nxt, n])
            live.append(nxt)
            sizes[nxt] = n
            nxt += 1
        elif u < 0.62:
            r = live.pop(int(rng.integers(len(live))))
            ops.append(["f", r])
        elif u < 0.80:
            r = live[int(rng.integers(len(live)))]
            n = sizes[r]
            if n >= 2:
                lo = int(rng.integers(0, n - 1))
                hi = int(rng.integers(lo + 1, n + 1))
                ops.append(["fp", r, lo, hi])
                sizes[r] = n - (hi - lo)
        elif u < 0.86:
            ops.append(["rs", int(rng.integers(1, 2 * blocks_per_page))])
            has_reserved = True
        elif u < 0.90:
            ops.append(["fr"])
            has_reserved = False
        elif u < 0.93:
            ops.append(["tr"])
        elif u < 0.97:
            ops.append(["ph", int(rng.integers(0, pool_pages + 4))] if rng.random() < 0.7 else ["ph", 1 << 40])
        elif with_resize:
            if has_reserved:
                ops.append(["fr"])
                has_reserved = False
            ops.append(["rz", int(rng.integers(max(1, pool_pages // 3), pool_pages + 1)) * PAGE])
    return ops


def manager_cases_large() -> List[Dict[str, Any]]:
    from kvcached_amd.traces import poisson_trace
    cases = []
    cases.append(_case("cfg3_llama3_8b_poisson_l4_60s", poisson_trace(rate=4.0, duration_s=60.0, seed=1), full=False,
                       **CFG3))
    cases.append(_case("cfg3_geometry_tight_pool_poisson", poisson_trace(rate=8.0, duration_s=20.0, seed=7), full=False,
                       num_blocks=64 * 48, cell_size=2048, num_layers=32, phys_pages=40))
    cases.append(_case("random_mix_48_pages", random_mix_ops(2500, seed=11, pool_pages=48), full=False,
                       num_blocks=64 * 48))
    cases.append(_case("random_mix_null_block_contiguous", random_mix_ops(1500, seed=12, pool_pages=32, with_resize=False),
                       full=False, num_blocks=64 * 32, reserve_null_block=True, contiguous=True, num_layers=4))
    cases.append(_case("random_mix_cfg1_geometry", random_mix_ops(1500, seed=13, pool_pages=24, blocks_per_page=128),
                       full=False, num_blocks=128 * 24, block_size=16, cell_size=1024, num_layers=16))
    return cases


# ------------------------------------------------------------------ integration layouts
GiB = 1 << 30
LAYOUT_CASES: List[Dict[str, Any]] = [
    # vLLM FlashAttn (2, N, bs, H, D), Llama-3-8B GQA heads, bf16
    dict(engine="vllm", attention_type="GQA", shape=[2, 1000, 16, 8, 128], block_size=16, dtype="bfloat16",
         num_layers=32, gpu_bytes=8 * GiB, contiguous=False),
    dict(engine="vllm", attention_type="GQA", shape=[2, 1000, 16, 8, 128], block_size=16, dtype="bfloat16",
         num_layers=32, gpu_bytes=8 * GiB, contiguous=True),
    # FlashInfer (N, 2, bs, H, D)
    dict(engine="vllm", attention_type="MHA", shape=[500, 2, 16, 8, 64], block_size=16, dtype="float16",
         num_layers=16, gpu_bytes=4 * GiB, contiguous=False),
    dict(engine="vllm", attention_type="MHA", shape=[500, 2, 16, 8, 64], block_size=16, dtype="float16",
         num_layers=16, gpu_bytes=4 * GiB, contiguous=True),
    # kernel blocks smaller than the virtual block
    dict(engine="vllm", attention_type="MHA", shape=[2, 100, 64, 4, 64], block_size=64, kernel_block_size=16,
         dtype="float16", num_layers=6, gpu_bytes=3 * GiB, contiguous=False),
    # MLA (N, bs, 576)
    dict(engine="vllm", attention_type="MLA", shape=[300, 16, 576], block_size=16, dtype="bfloat16", num_layers=27,
         gpu_bytes=5 * GiB, contiguous=False),
    dict(engine="vllm", attention_type="MLA", shape=[300, 16, 576], block_size=16, dtype="bfloat16", num_layers=27,
         gpu_bytes=5 * GiB, contiguous=True),
    # hybrid linear attention: K/V interleaved per block
    dict(engine="vllm", attention_type="HYBRID_LINEAR", shape=[2, 100, 16, 8, 64], block_size=16, dtype="float16",
         num_layers=4, gpu_bytes=2 * GiB, contiguous=False),
    dict(engine="vllm", attention_type="HYBRID_LINEAR", shape=[100, 2, 32, 8, 64], block_size=32, kernel_block_size=16,
         dtype="float16", num_layers=4, gpu_bytes=2 * GiB, contiguous=False),
    # SGLang (tokens, H, D)
    dict(engine="sglang", attention_type="MHA", shape=[65536, 8, 64], block_size=16, dtype="float16", num_layers=2,
         gpu_bytes=2 * GiB, contiguous=False),
    dict(engine="sglang", attention_type="MHA", shape=[65536, 8, 64], block_size=16, dtype="float16", num_layers=2,
         gpu_bytes=2 * GiB, contiguous=True),
    dict(engine="sglang", attention_type="GQA", shape=[1000, 8, 128], block_size=1, dtype="bfloat16", num_layers=32,
         gpu_bytes=8 * GiB, contiguous=False),
    dict(engine="sglang", attention_type="MLA", shape=[1000, 1, 576], block_size=64, dtype="bfloat16", num_layers=27,
         gpu_bytes=5 * GiB, contiguous=False),
    dict(engine="sglang", attention_type="MLA", shape=[1000, 1, 576], block_size=64, dtype="bfloat16", num_layers=27,
         gpu_bytes=5 * GiB, contiguous=True),
]


# ------------------------------------------------------------------ prefix cache (ElasticBlockPool)
def prefix_cache_ops(n_ops: int, seed: int, num_blocks: int, n_prefixes: int = 12):
    """Requests share prefixes drawn from a small catalogue, so hits, LRU eviction under pressure, capped
    evictable sets and failed allocations all occur."""
    rng = np.random.default_rng(seed)
    catalogue = [[int(p * 1000 + i) for i in range(int(rng.integers(2, 14)))] for p in range(n_prefixes)]
    ops, live, nxt, uniq = [], [], 0, 500000
    for _ in range(n_ops):
        u = rng.random()
        if u < 0.5 or not live:
            base = catalogue[int(rng.integers(n_prefixes))]
            cut = int(rng.integers(1, len(base) + 1))
            tail = int(rng.integers(0, 4))
            hashes = base[:cut] + [uniq + i for i in range(tail)]
            uniq += tail
            ops.append(["req", nxt, hashes, int(rng.integers(0, 2)) if rng.random() < 0.2 else 0])
            live.append(nxt)
            nxt += 1
        elif u < 0.85:
            ops.append(["fin", live.pop(int(rng.integers(len(live))))])
        elif u < 0.92:
            ops.append(["evict", [int(x) for x in rng.choice(num_blocks, size=int(rng.integers(1, 6)), replace=False)]])
        elif u < 0.95:
            ops.append(["reset"])
        else:
            ops.append(["stat"])
    return ops


PREFIX_CACHE_CASES = [
    dict(name="default_cap_1000", num_blocks=64, enable_caching=True, max_cached_blocks=1000, seed=21, n_ops=400),
    dict(name="unlimited", num_blocks=48, enable_caching=True, max_cached_blocks=-1, seed=22, n_ops=400),
    dict(name="cap_5", num_blocks=64, enable_caching=True, max_cached_blocks=5, seed=23, n_ops=400),
    dict(name="cap_0_evict_on_free", num_blocks=64, enable_caching=True, max_cached_blocks=0, seed=24, n_ops=300),
    dict(name="caching_off", num_blocks=64, enable_caching=False, max_cached_blocks=1000, seed=25, n_ops=300),
    dict(name="tiny_pool_pressure", num_blocks=20, enable_caching=True, max_cached_blocks=1000, seed=26, n_ops=400),
]
