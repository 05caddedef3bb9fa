"""Elastic token-pool allocators for SGLang on top of KVCacheManager.

Behaviourally the two classes the reference injects into `sglang.srt.mem_cache.allocator`
(kvcached/integration/sglang/patches.py:60-139 `ElasticTokenToKVPoolAllocator`, :142-318
`ElasticPagedTokenToKVPoolAllocator`): `alloc*` turn the block ids KVCacheManager hands out into the token
slot indices SGLang writes KV to, `free` turns freed token indices back into block ids. They are built by a
factory because their base class belongs to SGLang:

    Elastic, ElasticPaged = build_elastic_allocators(alloc_mod.BaseTokenToKVPoolAllocator)

What differs from the reference is only HOW the index tensors are produced. The reference builds them with
`torch.tensor(list)` (a pageable host->device copy) + broadcasting arithmetic + SGLang's Triton
`alloc_extend_kernel`/`alloc_decode_kernel`, and frees through `torch.unique(...).cpu()` (a device sort). Here
each call is ONE launch of a gfx950 kernel (kvcached_amd/csrc/index_kernels.hip) that takes the block-id list
straight from the host in its kernarg segment, on torch's current stream; free() is a bitmap mark + ordered
sweep. No Triton, no torch arithmetic, and no fallback: without the HIP library these classes do not import.
"""
from __future__ import annotations

from typing import List, Optional

import torch

from kvcached_amd import vmm_ops as _ops
from kvcached_amd.utils import get_kvcached_logger

logger = get_kvcached_logger()


def _is_supported_gpu_device(device) -> bool:
    s = str(device).lower()
    return s.startswith("cuda") or s.startswith("hip")


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _i64(t: torch.Tensor) -> torch.Tensor:
    return t if (t.dtype == torch.int64 and t.is_contiguous()) else t.to(torch.int64).contiguous()


def num_new_pages(seq_lens_cpu, page_size: int, prefix_lens_cpu=None, decode: bool = False) -> int:
    """Blocks a batch needs on top of what its sequences already hold (sglang.srt.utils.get_num_new_pages
    as called at patches.py:215-219,254-258). CPU tensors in, int out."""
    seq = torch.as_tensor(seq_lens_cpu, dtype=torch.int64)
    pre = seq - 1 if decode else torch.as_tensor(prefix_lens_cpu, dtype=torch.int64)
    after = (seq + page_size - 1) // page_size
    before = (pre + page_size - 1) // page_size
    return int((after - before).sum().item())


# ---- the four index operations (thin, typed wrappers over vmm_ops -> C ABI; also used directly by tests/benchmarks)
def expand_block_ids(block_ids: List[int], tokens_per_block: int, device) -> torch.Tensor:
    out = torch.empty((len(block_ids) * tokens_per_block,), dtype=torch.int64, device=device)
    if block_ids:
        _ops.expand_block_ids(block_ids, tokens_per_block, out.data_ptr(), _stream())
    return out


def alloc_extend_indices(prefix_lens: torch.Tensor, seq_lens: torch.Tensor, last_loc: torch.Tensor,
                         new_block_ids: List[int], tokens_per_block: int, extend_num_tokens: int) -> torch.Tensor:
    prefix_lens, seq_lens, last_loc = _i64(prefix_lens), _i64(seq_lens), _i64(last_loc)
    out = torch.empty((extend_num_tokens,), dtype=torch.int64, device=seq_lens.device)
    if extend_num_tokens:
        _ops.alloc_extend_indices(prefix_lens.data_ptr(), seq_lens.data_ptr(), last_loc.data_ptr(), seq_lens.numel(),
                                  new_block_ids, tokens_per_block, out.data_ptr(), extend_num_tokens, _stream())
    return out


def alloc_decode_indices(seq_lens: torch.Tensor, last_loc: torch.Tensor, new_block_ids: List[int],
                         tokens_per_block: int) -> torch.Tensor:
    seq_lens, last_loc = _i64(seq_lens), _i64(last_loc)
    out = torch.empty((seq_lens.numel(),), dtype=torch.int64, device=seq_lens.device)
    if seq_lens.numel():
        _ops.alloc_decode_indices(seq_lens.data_ptr(), last_loc.data_ptr(), seq_lens.numel(), new_block_ids,
                                  tokens_per_block, out.data_ptr(), _stream())
    return out


def unique_block_ids(token_indices: torch.Tensor, tokens_per_block: int, num_blocks: int) -> List[int]:
    """Sorted distinct block ids of the given device token indices (blocks until the result is on the host)."""
    token_indices = _i64(token_indices.reshape(-1))
    if token_indices.numel() == 0:
        return []
    return _ops.unique_block_ids(token_indices.data_ptr(), token_indices.numel(), tokens_per_block, num_blocks, _stream())


def build_elastic_allocators(base_cls: type):
    """-> (ElasticTokenToKVPoolAllocator, ElasticPagedTokenToKVPoolAllocator) deriving from SGLang's
    BaseTokenToKVPoolAllocator (which provides size/page_size/device/free_group bookkeeping)."""

    class ElasticTokenToKVPoolAllocator(base_cls):  # type: ignore[misc, valid-type]
        """page_size == 1: a block is a token."""

        def __init__(self, size: int, dtype, device: str, kvcache, *args, **kwargs) -> None:
            super().__init__(size, 1, dtype, device, kvcache, *args, **kwargs)
            if not hasattr(kvcache, "kvcached_allocator"):
                raise ValueError("ElasticTokenToKVPoolAllocator requires elastic MHA pool")
            if not _is_supported_gpu_device(device):
                raise ValueError("ElasticTokenToKVPoolAllocator only supports GPU devices (cuda/hip)")
            self.kvcached_allocator = kvcache.kvcached_allocator
            logger.info(f"[kvcached] ElasticTokenToKVPoolAllocator in use: size={size} (page_size=1 path)")

        def available_size(self):
            # the manager holds size+1 blocks (the null block) and follows physical memory, so it can report a
            # little more than the pool's declared capacity; SGLang asserts available <= size
            return min(self.kvcached_allocator.available_size(), self.size)

        def alloc(self, need_size: int):
            indices = self.kvcached_allocator.alloc(need_size)
            if indices is None:
                return None
            return expand_block_ids(indices, 1, self.device)

        def free(self, free_index):
            if self.is_not_in_free_group:
                # order and multiplicity are kept as given (patches.py:104-110): they decide the order in which the
                # manager's pages get their blocks back, i.e. later block tables
                return self.kvcached_allocator.free(free_index.cpu().numpy().tolist())
            self.free_group.append(free_index)

        def clear(self):
            if hasattr(self, "kvcached_allocator"):
                self.kvcached_allocator.clear()

    class ElasticPagedTokenToKVPoolAllocator(base_cls):  # type: ignore[misc, valid-type]
        """page_size > 1: SGLang "pages" are kvcached blocks of page_size tokens."""

        def __init__(self, size: int, page_size: int, dtype, device: str, kvcache, *args, **kwargs) -> None:
            super().__init__(size, page_size, dtype, device, kvcache, *args, **kwargs)
            if not hasattr(kvcache, "kvcached_allocator"):
                raise ValueError("ElasticPagedTokenToKVPoolAllocator requires elastic MHA pool")
            if not _is_supported_gpu_device(device):
                raise ValueError("ElasticPagedTokenToKVPoolAllocator only supports GPU devices (cuda/hip)")
            self.kvcached_allocator = kvcache.kvcached_allocator
            self.num_pages = size // page_size
            self.seen_max_num_extend_tokens_next_power_of_2 = 1   # kept for SGLang code that reads it
            logger.info(f"[kvcached] ElasticPagedTokenToKVPoolAllocator in use: size={size}, page_size={page_size}")
            # the base class expects these tensors for backup_state / free_group_end
            self.free_pages = torch.empty((0,), dtype=torch.int64, device=self.device)
            self.release_pages = torch.empty((0,), dtype=torch.int64, device=self.device)

        def available_size(self):
            return self.kvcached_allocator.available_size() * self.page_size

        def alloc(self, need_size: int):
            block_ids = self.kvcached_allocator.alloc(need_size // self.page_size)
            if block_ids is None:
                return None
            return expand_block_ids(block_ids, self.page_size, self.device)

        def alloc_extend(self, prefix_lens, prefix_lens_cpu, seq_lens, seq_lens_cpu, last_loc, extend_num_tokens: int):
            n_new = num_new_pages(seq_lens_cpu, self.page_size, prefix_lens_cpu)
            block_ids: Optional[List[int]] = []
            if n_new > 0:
                block_ids = self.kvcached_allocator.alloc(n_new)
                if block_ids is None:
                    return None
            return alloc_extend_indices(prefix_lens, seq_lens, last_loc, block_ids, self.page_size, extend_num_tokens)

        def alloc_decode(self, seq_lens, seq_lens_cpu, last_loc):
            n_new = num_new_pages(seq_lens_cpu, self.page_size, decode=True)
            block_ids: Optional[List[int]] = []
            if n_new > 0:
                block_ids = self.kvcached_allocator.alloc(n_new)
                if block_ids is None:
                    return None
            return alloc_decode_indices(seq_lens, last_loc, block_ids, self.page_size)

        def free(self, free_index):
            if free_index.numel() == 0:
                return
            if self.is_not_in_free_group:
                ids = unique_block_ids(free_index, self.page_size, self._num_blocks())
                return self.kvcached_allocator.free(ids)
            self.free_group.append(free_index)

        def clear(self):
            if hasattr(self, "kvcached_allocator"):
                self.kvcached_allocator.clear()
            self.free_pages = torch.empty((0,), dtype=torch.int64, device=self.device)
            self.release_pages = torch.empty((0,), dtype=torch.int64, device=self.device)
            self.is_not_in_free_group = True
            self.free_group = []

        def merge_and_sort_free(self):
            pass  # kvcached owns the free list

        def _num_blocks(self) -> int:
            return int(getattr(self.kvcached_allocator, "num_blocks", self.num_pages + 1))

    return ElasticTokenToKVPoolAllocator, ElasticPagedTokenToKVPoolAllocator
