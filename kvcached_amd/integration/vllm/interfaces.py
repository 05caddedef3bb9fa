"""vLLM-facing API of kvcached_amd — unchanged names/arguments/returns w.r.t. the reference
(kvcached/integration/vllm/interfaces.py:29-338), so the existing vLLM patches
(ElasticBlockPool, GPUModelRunner hooks, ...) run on top of it as they are:

    init_kvcached · shutdown_kvcached · alloc_kv_cache · get_kv_cache_manager · should_use_worker_ipc
"""
from __future__ import annotations

import math
from typing import List, Optional, Tuple

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
_world_size: int = 1
_pp_rank: int = 0
_contiguous_layout: bool = CONTIGUOUS_LAYOUT
_is_worker: bool = False


def should_use_worker_ipc() -> bool:
    """True in a process that schedules but does not own the KV tensors (vLLM V1 EngineCore):
    its map/unmap requests must travel to the workers even when world_size == 1."""
    return _kvcached_initialized and not _is_worker


def init_kvcached(
    tp_rank: int = 0,
    world_size: int = 1,
    pp_rank: int = 0,
    is_worker: bool = False,
    device: Optional[str] = None,
    async_sched: bool = False,
) -> None:
    global _kvcached_initialized, _kvcached_device, _world_size, _async_sched, _pp_rank, _is_worker
    if _kvcached_initialized:
        # Second call in the same process: at TP=1 EngineCore initialises first (is_worker=False)
        # and the model runner follows (is_worker=True). Promote to worker and start the listener,
        # otherwise the manager would try to reach a socket nobody serves.
        if is_worker and not _is_worker:
            _is_worker = True
            start_worker_listener_thread(tp_rank, pp_rank)
        if async_sched and not _async_sched:
            _async_sched = True
            logger.info("kvcached async scheduler enabled")
        _pp_rank = pp_rank
        _world_size = world_size
        return

    if device is None:
        device = f"cuda:{torch.cuda.current_device()}"
    device = normalize_gpu_device(device)

    _init_kvcached_impl(device, PAGE_SIZE, _contiguous_layout)
    _kvcached_initialized = True
    _kvcached_device = device
    _world_size = world_size
    _pp_rank = pp_rank
    _async_sched = async_sched
    _is_worker = is_worker
    if _async_sched:
        logger.info("kvcached async scheduler enabled")
    if is_worker:
        # always listen: with PP > 1 the EngineCore reaches this worker even at TP = 1
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
    block_size: int,
    dtype: torch.dtype,
    device: str,
    num_layers: int,
    attention_type: str = "MHA",  # MHA, GQA, MLA, or HYBRID_LINEAR
    kv_layout: str = "NHD",
    group_id: int = 0,
    kernel_block_size: Optional[int] = None,
) -> List[torch.Tensor]:
    """Reserve VA for the whole KV cache and return per-layer views shaped like vLLM expects.

    kvcache_shape: FlashAttn (2, num_blocks, block_size, heads, head_dim), FlashInfer
    (num_blocks, 2, block_size, heads, head_dim) or MLA (num_blocks, block_size, head_size).
    HYBRID_LINEAR (full + linear attention sharing a pool) interleaves K and V per block in one
    buffer per pool and additionally returns a dict of raw int8 buffers + geometry; pass the
    group size as num_layers. kernel_block_size < block_size exposes each block as
    block_size/kernel_block_size kernel-sized blocks. Nothing is physically backed here.
    """
    if not _kvcached_initialized:
        raise RuntimeError("kvcached is not initialized. Please call init_kvcached() first.")
    if attention_type not in ["MHA", "GQA", "MLA", "HYBRID_LINEAR"]:
        raise ValueError(f"Attention type {attention_type} is not supported.")
    if kv_layout != "NHD":
        raise ValueError(f"KV layout {kv_layout} is not supported.")

    is_mla = attention_type == "MLA"
    unified_pool = attention_type == "HYBRID_LINEAR"
    if unified_pool and _contiguous_layout:
        raise ValueError(
            "kvcached detected a hybrid linear-attention model (e.g. Jamba/Bamba/NemotronH/Zamba2/Plamo2), which "
            "requires the non-contiguous KV layout. Re-launch with KVCACHED_CONTIGUOUS_LAYOUT=false. Also do NOT "
            "pass --disable-hybrid-kv-cache-manager to vLLM for these models.")
    num_k_or_v = 1 if is_mla else 2

    if kernel_block_size is None:
        kernel_block_size = block_size
    if block_size % kernel_block_size != 0:
        raise ValueError(f"block_size ({block_size}) must be a multiple of kernel_block_size ({kernel_block_size})")
    ratio = block_size // kernel_block_size

    # which dim counts blocks, which one is K/V
    if is_mla:
        if len(kvcache_shape) <= 2:
            raise ValueError(f"Unsupported MLA kv cache shape: {kvcache_shape}")
        if kvcache_shape[1] != block_size:
            raise ValueError(f"block_size mismatch: kvcache_shape[1]={kvcache_shape[1]} != block_size={block_size}")
        blocks_dim, kv_dim, token_dim = 0, None, 1
        permute_order = list(range(len(kvcache_shape)))
        block_mem_bytes = math.prod(kvcache_shape[1:]) * dtype.itemsize
    else:
        if (len(kvcache_shape) <= 3 or (kvcache_shape[0] != 2 and kvcache_shape[1] != 2)
                or kvcache_shape[2] != block_size):
            raise ValueError(f"Unsupported kv cache shape: {kvcache_shape}")
        if kvcache_shape[0] == 2:      # FlashAttn
            blocks_dim, kv_dim = 1, 0
            permute_order = [1, 0] + list(range(2, len(kvcache_shape)))
        else:                          # FlashInfer
            blocks_dim, kv_dim = 0, 1
            permute_order = list(range(len(kvcache_shape)))
        token_dim = 2
        block_mem_bytes = math.prod(kvcache_shape[2:]) * dtype.itemsize
    requested_num_blocks = kvcache_shape[blocks_dim]

    assert torch.cuda.is_available(), "GPU backend is not available via torch.cuda."
    device = normalize_gpu_device(device)

    total = torch.cuda.get_device_properties(device).total_memory
    per_layer = _layout.per_layer_budget(total, num_layers, num_k_or_v, PAGE_SIZE, is_mla)
    num_blocks = per_layer // block_mem_bytes
    if requested_num_blocks > num_blocks:
        logger.warning(f"Requested {requested_num_blocks} blocks, but only {num_blocks} blocks are available.")

    raw = create_kv_tensors(per_layer * num_k_or_v, dtype.itemsize, device, num_layers,
                            num_kv_buffers=num_k_or_v, group_id=group_id, unified_pool=unified_pool)

    actual = list(kvcache_shape)
    actual[blocks_dim] = num_blocks
    kernel_shape = list(actual)      # the same memory indexed in kernel-sized blocks
    if ratio > 1:
        kernel_shape[blocks_dim] = num_blocks * ratio
        kernel_shape[token_dim] = kernel_block_size

    if not _contiguous_layout:
        if is_mla:
            kv_tensors = [_layout.flat_prefix_view(t, dtype, kernel_shape) for t in raw]
        elif unified_pool:
            kv_tensors = [_layout.interleaved_view(t, dtype, kernel_shape, kv_dim, blocks_dim) for t in raw]
        else:
            v_off = per_layer // dtype.itemsize
            kv_tensors = [_layout.split_half_view(t, dtype, kernel_shape, kv_dim, blocks_dim, v_off) for t in raw]
    else:
        # one buffer [block][layer][per-layer element]; layer i is a strided slice of it
        per_block = actual[:blocks_dim] + actual[blocks_dim + 1:]
        whole = _layout.flat_prefix_view(raw[0], dtype, [num_blocks, num_layers] + per_block)
        kv_tensors = [whole[:, i].permute(*permute_order) for i in range(num_layers)]

    if not unified_pool:
        return kv_tensors

    page_size_bytes = math.prod(actual[:blocks_dim] + actual[blocks_dim + 1:]) * dtype.itemsize
    pool_bytes = num_blocks * page_size_bytes
    raw_info = {
        "buffers": [t.view(torch.int8)[:pool_bytes] for t in raw],
        "num_blocks": num_blocks,
        "page_size_bytes": page_size_bytes,
        "block_stride_bytes": page_size_bytes,
        "num_pools": num_layers,
    }
    return kv_tensors, raw_info  # type: ignore[return-value]


def get_kv_cache_manager(
    num_blocks: int,
    block_size: int,
    cell_size: int,
    num_layers: int,
    num_kv_buffers: int = 2,
    group_id: int = 0,
) -> KVCacheManager:
    if not _kvcached_initialized:
        raise RuntimeError("kvcached is not initialized. Please call init_kvcached() first.")
    return KVCacheManager(num_blocks, block_size, cell_size, num_layers, _world_size, pp_rank=_pp_rank,
                          async_sched=_async_sched, num_kv_buffers=num_kv_buffers, group_id=group_id)
