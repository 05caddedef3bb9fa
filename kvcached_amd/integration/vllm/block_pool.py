"""ElasticBlockPool — vLLM's BlockPool interface on top of KVCacheManager, with a lazy-eviction prefix cache.

Behaviourally identical to the class the reference injects into `vllm.v1.core.block_pool`
(kvcached/integration/vllm/patches.py:308-614); tests/golden/prefix_cache.json holds traces recorded
from that class. It is built by a factory because its base class and block type belong to vLLM:

    ElasticBlockPool = build_elastic_block_pool(block_pool_mod.BlockPool, block_pool_mod.KVCacheBlock)

State:
  _cached_blocks     (block_hash, group) key -> block      prefix-cache index
  _block_id_to_key   block id -> key                        reverse index, O(1) eviction
  _evictable_blocks  OrderedDict id -> block                ref_cnt == 0 but still cached; LRU = insertion order
Rules the goldens pin: a block whose ref count drops to 0 is kept (evictable) iff it is cached, otherwise
freed at once; evictable blocks count as free; allocation evicts the oldest evictable blocks only when the
manager cannot satisfy the request; `max_cached_blocks` (>= 0) caps the evictable set after every free
(-1 = unlimited, 0 = evict immediately); lookups over several groups are all-or-nothing.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Any, Iterable, List, Optional

from kvcached_amd.utils import get_kvcached_logger


def make_cache_key(block_hash: Any, group_id: int) -> bytes:
    """Prefix-cache key: the block hash's bytes followed by the 4-byte big-endian KV-cache group id
    (hybrid models keep several groups in one pool). Reference: patches.py:259-274."""
    if isinstance(block_hash, str):
        block_hash = block_hash.encode()
    return bytes(block_hash) + group_id.to_bytes(4, "big", signed=False)


def build_elastic_block_pool(block_pool_cls: type, kv_cache_block_cls: type, logger=None) -> type:
    log = logger or get_kvcached_logger()

    class ElasticBlockPool(block_pool_cls):  # type: ignore[misc, valid-type]
        """BlockPool whose blocks are backed on demand by kvcached."""

        def __init__(self, num_gpu_blocks: int, block_size: int, cell_size: int, num_layers: int, enable_caching: bool,
                     enable_kv_cache_events: bool = False, num_kv_buffers: int = 2, max_cached_blocks: int = 1000) -> None:
            assert isinstance(num_gpu_blocks, int) and num_gpu_blocks > 0
            assert not enable_kv_cache_events, "KV cache events are not supported in ElasticBlockPool"
            self.enable_prefix_cache = enable_caching
            self.max_cached_blocks = max_cached_blocks          # -1 unlimited, 0 disabled, > 0 cap
            if enable_caching:
                log.info("Prefix caching enabled for ElasticBlockPool")
            self.num_gpu_blocks = num_gpu_blocks
            self.enable_kv_cache_events = enable_kv_cache_events
            self.kv_event_queue: list = []
            self.kv_block_pool = [kv_cache_block_cls(i) for i in range(num_gpu_blocks)]

            from kvcached_amd.integration.vllm.interfaces import get_kv_cache_manager
            self.kv_cache_manager = get_kv_cache_manager(num_gpu_blocks, block_size, cell_size, num_layers,
                                                         num_kv_buffers=num_kv_buffers)
            # vLLM's pool reserves block 0 as the "null" block for skipped positions; here it is a real,
            # backed block so that kernels may read it (their results are masked)
            null_ids = self.kv_cache_manager.alloc(1)
            assert null_ids is not None and len(null_ids) == 1
            self.null_block = self.kv_block_pool[null_ids[0]]
            self.null_block.is_null = True

            self._cached_blocks: dict = {}
            self._block_id_to_key: dict = {}
            self._evictable_blocks: "OrderedDict[int, Any]" = OrderedDict()

        # ---- lookup / registration
        def get_cached_block(self, block_hash: Any, kv_cache_group_ids: Optional[Iterable[int]] = None):
            if not self.enable_prefix_cache:
                return None
            if kv_cache_group_ids is None:       # old call form: one block, group 0
                return self._cached_blocks.get(make_cache_key(block_hash, 0))
            if isinstance(kv_cache_group_ids, int):
                kv_cache_group_ids = [int(kv_cache_group_ids)]
            hits = []
            for gid in kv_cache_group_ids:
                block = self._cached_blocks.get(make_cache_key(block_hash, int(gid)))
                if block is None:
                    return None                   # every group must hit
                hits.append(block)
            return hits or None

        def cache_full_blocks(self, request, blocks, *args: Any, **kwargs: Any) -> None:
            """Accepts every call form vLLM has used: (request, blocks, [block_hashes,] num_cached_blocks,
            num_full_blocks, block_size[, kv_cache_group_id][, hash_fn]) positionally or by keyword."""
            if not self.enable_prefix_cache:
                return
            block_hashes = kwargs.pop("block_hashes", None)
            num_cached = kwargs.pop("num_cached_blocks", None)
            num_full = kwargs.pop("num_full_blocks", None)
            block_size = kwargs.pop("block_size", None)
            group_id = kwargs.pop("kv_cache_group_id", 0)
            kwargs.pop("hash_fn", None)
            rest = list(args)
            if block_hashes is None and rest and isinstance(rest[0], (list, tuple)):
                block_hashes = rest.pop(0)
            if num_cached is None and rest:
                num_cached = rest.pop(0)
            if num_full is None and rest:
                num_full = rest.pop(0)
            if block_size is None and rest:
                block_size = rest.pop(0)
            if rest and isinstance(rest[0], int):
                group_id = rest.pop(0)
            if num_cached is None or num_full is None:
                raise TypeError("cache_full_blocks requires num_cached_blocks and num_full_blocks")
            num_cached, num_full, group_id = int(num_cached), int(num_full), int(group_id)
            if num_cached >= num_full:
                return
            if block_hashes is None:
                assert hasattr(request, "block_hashes"), "Request missing block_hashes attribute"
                block_hashes = request.block_hashes
            assert len(block_hashes) >= num_full, f"Request has {len(block_hashes)} hashes but need {num_full}"
            for i, block in enumerate(blocks[num_cached:num_full]):
                if getattr(block, "is_null", False):
                    continue
                key = make_cache_key(block_hashes[num_cached + i], group_id)
                if key in self._cached_blocks:
                    continue                       # idempotent: first registration wins
                self._cached_blocks[key] = block
                self._block_id_to_key[block.block_id] = key

        # ---- allocation / release
        def _evict_blocks_from_pool(self, num_to_evict: int) -> int:
            victims: List[int] = []
            for _ in range(min(num_to_evict, len(self._evictable_blocks))):
                bid, _ = self._evictable_blocks.popitem(last=False)        # oldest first
                key = self._block_id_to_key.pop(bid, None)
                if key is not None:
                    self._cached_blocks.pop(key, None)
                victims.append(bid)
            if victims:
                self.kv_cache_manager.free(victims)
            return len(victims)

        def get_new_blocks(self, num_blocks: int):
            if num_blocks > self.get_num_free_blocks():
                raise ValueError(f"Cannot get {num_blocks} free blocks from the pool")
            ids = None
            for _ in range(2):
                if self.enable_prefix_cache:
                    have = self.kv_cache_manager.available_size()
                    if have < num_blocks and self._evictable_blocks:
                        self._evict_blocks_from_pool(num_blocks - have)
                ids = self.kv_cache_manager.alloc(num_blocks)
                if ids is not None:
                    break
            if ids is None:
                raise ValueError("Unable to allocate KV cache blocks from physical pool; "
                                 f"requested={num_blocks}, available={self.kv_cache_manager.available_size()}")
            assert len(ids) == num_blocks, f"alloc returned {len(ids)} blocks, expected {num_blocks}"
            out = []
            for bid in ids:
                block = self.kv_block_pool[bid]
                block.ref_cnt = 1
                out.append(block)
            return out

        def touch(self, blocks) -> None:
            if not self.enable_prefix_cache:
                return
            groups = blocks if isinstance(blocks, tuple) else (blocks,)
            for group in groups:
                for block in group:
                    block.ref_cnt += 1
                    self._evictable_blocks.pop(block.block_id, None)     # in use again

        def free_blocks(self, ordered_blocks: Iterable[Any]) -> None:
            if not self.enable_prefix_cache:
                ids = [b.block_id for b in ordered_blocks if b is not None and not getattr(b, "is_null", False)]
                if ids:
                    self.kv_cache_manager.free(ids)
                return
            to_free: List[int] = []
            for block in ordered_blocks:
                if block is None or getattr(block, "is_null", False):
                    continue
                block.ref_cnt -= 1
                if block.ref_cnt == 0:
                    if block.block_id in self._block_id_to_key:
                        self._evictable_blocks[block.block_id] = block    # keep for later requests
                    else:
                        to_free.append(block.block_id)                    # never cached (e.g. partial block)
            if to_free:
                self.kv_cache_manager.free(to_free)
            if self.max_cached_blocks >= 0 and len(self._evictable_blocks) > self.max_cached_blocks:
                self._evict_blocks_from_pool(len(self._evictable_blocks) - self.max_cached_blocks)

        def evict_blocks(self, block_ids) -> None:
            if not self.enable_prefix_cache:
                return
            removed, to_free = 0, []
            for bid in block_ids:
                key = self._block_id_to_key.pop(bid, None)
                if key is not None:
                    self._cached_blocks.pop(key, None)
                    removed += 1
                if bid in self._evictable_blocks:
                    self._evictable_blocks.pop(bid)
                    to_free.append(bid)
            if to_free:
                self.kv_cache_manager.free(to_free)
            if removed:
                log.debug(f"Evicted {removed} blocks from prefix cache")

        def reset_prefix_cache(self) -> bool:
            if not self.enable_prefix_cache:
                return True
            if self._evictable_blocks:
                ids = list(self._evictable_blocks.keys())
                self._evictable_blocks.clear()
                self.kv_cache_manager.free(ids)
            self._cached_blocks.clear()
            self._block_id_to_key.clear()
            log.info("Prefix cache reset")
            return True

        # ---- accounting
        def get_num_free_blocks(self) -> int:
            free = self.kv_cache_manager.available_size()
            return free + len(self._evictable_blocks) if self.enable_prefix_cache else free

        def get_usage(self) -> float:
            return 1.0 - (self.get_num_free_blocks() / self.num_gpu_blocks)

        def take_events(self) -> list:
            return []

    return ElasticBlockPool
