"""SGLang-facing API of kvcached_amd — unchanged names/arguments/returns w.r.t. the reference
(kvcached/integration/sglang/interfaces.py:28-427):

    init_kvcached · shutdown_kvcached · alloc_kv_cache · alloc_mamba_states · get_kv_cache_manager

SGLang calls a KV block a "page"; here `page_size` (tokens) is the block size and "page" otherwise
means a physical memory page.
"""
from __future__ import annotations

import math
from typing import Any, Dict, List, Optional, Tuple, Union

import torch

from kvcached_amd.integration import _layout
from kvcached_amd.kv_cache_manager import KVCacheManager
from kvcached_amd.tp_ipc_util import start_worker_listener_thread
from kvcached_amd.utils import CONTIGUOUS_LAYOUT, PAGE_SIZE, get_kvcached_logger, normalize_gpu_device
from kvcached_amd.vmm_ops import (
    create_kv_tensors,
    init_kvcached as _init_kvcached_impl,
    shutdown_kvcached as _shutdown_kvcached_impl,
)

logger = get_kvcached_logger()

_kvcached_initialized: bool = False
_kvcached_device = None
_async_sched = False
_contiguous_layout = CONTIGUOUS_LAYOUT
_world_size: int = 1
_pp_rank: int = 0


def init_kvcached(
    tp_rank: int = 0,
    world_size: int = 1,
    pp_rank: int = 0,
    device: Optional[str] = None,
    async_sched: bool = False,
) -> None:
    global _kvcached_initialized, _kvcached_device, _async_sched, _world_size, _pp_rank
    if _kvcached_initialized:
        return
    if device is None:
        device = f"cuda:{torch.cuda.current_device()}"
    device = normalize_gpu_device(device)
    _init_kvcached_impl(device, PAGE_SIZE, _contiguous_layout)
    _kvcached_initialized = True
    _kvcached_device = device
    _async_sched = async_sched
    _world_size = world_size
    _pp_rank = pp_rank
    if world_size > 1:
        start_worker_listener_thread(tp_rank, pp_rank)


def shutdown_kvcached() -> None:
    global _kvcached_initialized, _kvcached_device, _async_sched
    if not _kvcached_initialized:
        return
    _shutdown_kvcached_impl()
    _kvcached_initialized = False
    _kvcached_device = None
    _async_sched = False


def alloc_kv_cache(
    kvcache_shape: Tuple[int, ...],
    dtype: torch.dtype,
    device: str,
    num_layers: int,
    page_size: int = 1,
    attention_type: str = "MHA",  # MHA, GQA, or MLA
    kv_layout: str = "NHD",       # (num_tokens, head_num, head_dim)
    group_id: int = 0,
) -> Union[Tuple[List[torch.Tensor], List[torch.Tensor]], List[torch.Tensor]]:
    """Token-major KV buffers over reserved VA: (k_tensors, v_tensors) for MHA/GQA, one list of
    (num_tokens, 1, kv_cache_dim) buffers for MLA."""
    if not _kvcached_initialized:
        raise RuntimeError("kvcached is not initialized. Please call init_kvcached() first.")
    if attention_type not in ["MHA", "GQA", "MLA"]:
        raise ValueError(f"Attention type {attention_type} is not supported.")
    is_mla = attention_type == "MLA"
    if not is_mla and kv_layout != "NHD":
        raise ValueError(f"KV layout {kv_layout} is not supported.")
    num_k_or_v = 1 if is_mla else 2
    requested_num_tokens = kvcache_shape[0]
    if len(kvcache_shape) <= 2:
        raise ValueError(f"Unsupported kv cache shape: {kvcache_shape}")

    assert torch.cuda.is_available(), "GPU backend is not available via torch.cuda."
    device = normalize_gpu_device(device)

    block_size = page_size
    token_elems = math.prod(kvcache_shape[1:])
    block_mem_size = block_size * token_elems * dtype.itemsize

    total = torch.cuda.get_device_properties(device).total_memory
    per_layer = _layout.per_layer_budget(total, num_layers, num_k_or_v, PAGE_SIZE, is_mla)

    raw = create_kv_tensors(per_layer * num_k_or_v, dtype.itemsize, device, num_layers,
                            num_kv_buffers=num_k_or_v, group_id=group_id)

    num_blocks = per_layer // block_mem_size
    num_tokens = num_blocks * block_size
    if requested_num_tokens > num_tokens:
        logger.warning(f"Requested {requested_num_tokens} tokens, but only {num_tokens} tokens are available.")
    per_token = list(kvcache_shape[1:])
    actual = [num_tokens] + per_token

    if is_mla:
        if not _contiguous_layout:
            return [_layout.flat_prefix_view(t, dtype, actual) for t in raw]
        whole = _layout.flat_prefix_view(raw[0], dtype, [num_tokens, num_layers] + per_token)
        return [whole[:, i, :, :] for i in range(num_layers)]

    k_tensors: List[torch.Tensor] = []
    v_tensors: List[torch.Tensor] = []
    if not _contiguous_layout:
        # V begins at the allocator's V base (= per_layer bytes), not right after the last K token
        v_off = per_layer // dtype.itemsize
        for t in raw:
            kv = _layout.split_half_view(t, dtype, [2] + actual, kv_dim=0, block_dim=1, v_offset_elems=v_off)
            k_tensors.append(kv[0])
            v_tensors.append(kv[1])
    else:
        whole = _layout.flat_prefix_view(raw[0], dtype, [num_tokens, num_layers, 2] + per_token)
        for i in range(num_layers):
            k_tensors.append(whole[:, i, 0, :, :])
            v_tensors.append(whole[:, i, 1, :, :])
    return k_tensors, v_tensors


def alloc_mamba_states(
    *,
    num_slots: int,
    num_mamba_layers: int,
    cache_params: Any,
    device: str,
    group_id: int = 0,
) -> Tuple[Any, Any, Dict[str, Any]]:
    """Mamba conv + temporal (SSM) states over reserved VA, one "super-cell" per (slot, layer):

        [ conv[0] bytes | conv[1] bytes | ... | temporal bytes ]   (each kind aligned to its dtype)

    One slot is one kvcached block, so one map call backs every state kind of a slot in all
    layers. Contiguous layout: conv_state is a list of (layers, slots, *shape) tensors and
    temporal_state one such tensor (same shapes as SGLang's MambaPool). Non-contiguous: every
    layer has its own reservation, so conv_state is [kind][layer] -> (slots, *shape) and
    temporal_state is [layer] -> (slots, *temporal_shape). layout_info["is_contiguous"] tells which.
    """
    if not _kvcached_initialized:
        raise RuntimeError("kvcached is not initialized. Please call init_kvcached() first.")
    assert torch.cuda.is_available(), "GPU backend is not available via torch.cuda."
    device = normalize_gpu_device(device)

    conv_shapes = [tuple(s) for s in cache_params.shape.conv]
    temporal_shape = tuple(cache_params.shape.temporal)
    conv_dtype = cache_params.dtype.conv
    ssm_dtype = cache_params.dtype.temporal

    def up(x: int, a: int) -> int:
        return (x + a - 1) // a * a

    conv_offsets: List[int] = []
    cursor = 0
    for shape in conv_shapes:
        cursor = up(cursor, conv_dtype.itemsize)
        conv_offsets.append(cursor)
        cursor += int(math.prod(shape)) * conv_dtype.itemsize
    cursor = up(cursor, ssm_dtype.itemsize)
    temporal_offset = cursor
    cursor += int(math.prod(temporal_shape)) * ssm_dtype.itemsize
    raw_cell_size = up(cursor, max(conv_dtype.itemsize, ssm_dtype.itemsize))

    if raw_cell_size > PAGE_SIZE:
        raise RuntimeError(f"Mamba per-slot super-cell ({raw_cell_size} bytes) exceeds kvcached PAGE_SIZE "
                           f"({PAGE_SIZE} bytes). Raise KVCACHED_PAGE_SIZE_MB so a single physical page can back "
                           "at least one slot.")

    # A cell size that divides the page exactly: otherwise blocks straddling a page edge are
    # dropped by the page allocator and fewer than num_slots slots would be deliverable.
    cell_size = _layout.smallest_divisor_at_least(PAGE_SIZE, raw_cell_size)
    if cell_size != raw_cell_size:
        overhead = (cell_size - raw_cell_size) * num_mamba_layers * num_slots
        logger.info(f"[kvcached] Elastic mamba cell padded: raw={raw_cell_size}B -> {cell_size}B (divisor of "
                    f"PAGE_SIZE={PAGE_SIZE}B). Virtual overhead: {overhead / (1024**3):.2f} GB. Raise "
                    "KVCACHED_PAGE_SIZE_MB for finer divisors if needed.")

    per_layer_bytes = up(num_slots * cell_size, PAGE_SIZE)
    # single-buffer states: one page per offset per layer (unified pool) unless the compound-page
    # layout is active, where the flag is ignored
    raw = create_kv_tensors(per_layer_bytes, torch.int8.itemsize, device, num_mamba_layers, num_kv_buffers=1,
                            group_id=group_id, unified_pool=not _contiguous_layout)

    layout_info: Dict[str, Any] = {
        "cell_size": cell_size,
        "num_slots": num_slots,
        "num_mamba_layers": num_mamba_layers,
        "conv_offsets": conv_offsets,
        "temporal_offset": temporal_offset,
        "is_contiguous": _contiguous_layout,
    }

    if _contiguous_layout:  # bytes are [slot][layer][cell]
        flat = raw[0]

        def view(shape, dtype, off):
            assert cell_size % dtype.itemsize == 0
            return _layout.packed_state_view(flat, dtype, (num_mamba_layers, num_slots, *shape),
                                             [cell_size, num_mamba_layers * cell_size], shape, off)

        conv_state: Any = [view(s, conv_dtype, conv_offsets[i]) for i, s in enumerate(conv_shapes)]
        return conv_state, view(temporal_shape, ssm_dtype, temporal_offset), layout_info

    def layer_view(t, shape, dtype, off):  # bytes are [slot][cell] inside one layer's reservation
        assert cell_size % dtype.itemsize == 0
        return _layout.packed_state_view(t, dtype, (num_slots, *shape), [cell_size], shape, off)

    conv_per_layer = [[layer_view(raw[l], s, conv_dtype, conv_offsets[k]) for l in range(num_mamba_layers)]
                      for k, s in enumerate(conv_shapes)]
    temporal_per_layer = [layer_view(raw[l], temporal_shape, ssm_dtype, temporal_offset)
                          for l in range(num_mamba_layers)]
    return conv_per_layer, temporal_per_layer, layout_info


def get_kv_cache_manager(
    num_blocks: int,
    block_size: int,
    cell_size: int,
    num_layers: int,
    reserve_null_block: bool = True,
    num_kv_buffers: int = 2,
    group_id: int = 0,
) -> KVCacheManager:
    if not _kvcached_initialized:
        raise RuntimeError("kvcached is not initialized. Please call init_kvcached() first.")
    return KVCacheManager(num_blocks, block_size, cell_size, num_layers, world_size=_world_size, pp_rank=_pp_rank,
                          async_sched=_async_sched, reserve_null_block=reserve_null_block,
                          num_kv_buffers=num_kv_buffers, group_id=group_id)
