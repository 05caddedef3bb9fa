"""Sizing and strided-view arithmetic shared by the vLLM and SGLang interfaces.

Everything here is integer math on shapes (no allocation): how much VA each layer's K (or V) gets,
and which strides make a raw flat tensor look like the engine's KV-cache layout. The results are
pinned against the reference's integration code by tests/golden/alloc_kv_cache_layouts.json.
Reference: kvcached/integration/vllm/interfaces.py:196-298, .../sglang/interfaces.py:103-175.
"""
from __future__ import annotations

import math
from typing import List, Sequence, Tuple

import torch


def per_layer_budget(total_gpu_bytes: int, num_layers: int, num_k_or_v: int, page_size: int, is_mla: bool) -> int:
    """Bytes of VA for one layer's K (or V): an equal share of the whole GPU, rounded DOWN to the
    page size — to 2 pages for MLA, whose single buffer must still split into two page-aligned
    halves for the allocator (vllm/interfaces.py:201-212)."""
    share = total_gpu_bytes // num_layers // num_k_or_v
    unit = 2 * page_size if is_mla else page_size
    return (share // unit) * unit


def row_major_strides(shape: Sequence[int]) -> List[int]:
    strides = [1] * len(shape)
    for i in range(len(shape) - 2, -1, -1):
        strides[i] = strides[i + 1] * shape[i + 1]
    return strides


def split_half_view(raw: torch.Tensor, dtype: torch.dtype, shape: Sequence[int], kv_dim: int, block_dim: int,
                    v_offset_elems: int) -> torch.Tensor:
    """View of one layer's flat tensor in which K lives in the first half and V starts exactly at
    `v_offset_elems` (the allocator's V base), whatever the block count. `shape[kv_dim] == 2`."""
    shape = list(shape)
    inner = row_major_strides(shape[2:])                      # dims after (kv, blocks) in either order
    block_elems = inner[0] * shape[2] if inner else 1
    strides = [0, 0] + inner
    strides[block_dim] = block_elems
    strides[kv_dim] = v_offset_elems
    return torch.as_strided(raw.view(dtype=dtype), shape, strides)


def interleaved_view(raw: torch.Tensor, dtype: torch.dtype, shape: Sequence[int], kv_dim: int,
                     block_dim: int) -> torch.Tensor:
    """Unified-pool view: K and V of a block sit next to each other (block stride = 2 x hidden)."""
    shape = list(shape)
    inner = row_major_strides(shape[2:])
    hidden = inner[0] * shape[2] if inner else 1
    strides = [0, 0] + inner
    strides[block_dim] = 2 * hidden
    strides[kv_dim] = hidden
    return torch.as_strided(raw.view(dtype=dtype), shape, strides)


def flat_prefix_view(raw: torch.Tensor, dtype: torch.dtype, shape: Sequence[int]) -> torch.Tensor:
    n = math.prod(shape)
    return raw.view(dtype=dtype)[:n].view(list(shape))


def packed_state_view(raw: torch.Tensor, dtype: torch.dtype, size: Tuple[int, ...], lead_strides_bytes: Sequence[int],
                      inner_shape: Sequence[int], offset_bytes: int) -> torch.Tensor:
    """Strided view into byte-packed per-slot cells (mamba states): leading strides are given in
    bytes, the inner dims are dense."""
    item = dtype.itemsize
    assert offset_bytes % item == 0 and all(s % item == 0 for s in lead_strides_bytes)
    strides = [s // item for s in lead_strides_bytes] + row_major_strides(list(inner_shape))
    return torch.as_strided(raw.view(dtype=dtype), size=size, stride=strides, storage_offset=offset_bytes // item)


def smallest_divisor_at_least(n: int, lower: int) -> int:
    best, i = n, 1
    while i * i <= n:
        if n % i == 0:
            for d in (i, n // i):
                if d >= lower:
                    best = min(best, d)
        i += 1
    return best
