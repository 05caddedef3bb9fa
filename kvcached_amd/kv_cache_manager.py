"""KVCacheManager — block-granular bookkeeping on top of the native PageAllocator.

Public surface and observable behaviour follow the reference class of the same name
(kvcached/kv_cache_manager.py:58-506): pages (PAGE_SIZE, backed/unbacked as a unit) are cut
into blocks (block_size tokens x cell_size bytes); `alloc` hands out block ids, `free` takes them
back and releases pages that became empty. On identical call traces the returned block ids, the
page ids handed to map/unmap and every counter are bit-identical to the reference
(tests/golden/*, generated from the reference itself).

Order-defining rules that the golden traces pin (reference line numbers):
  * reserved blocks are consumed first, oldest first                               (:272-277)
  * partially used pages are reused most-recently-touched first (dict.popitem)     (:292)
  * a new page comes from PageAllocator.alloc_page(); its blocks are taken in
    ascending order                                                                (:281-283)
  * free() visits pages in the iteration order of PageAllocator.group_indices_by_page (:319-321)
  * a shrink that cannot complete is finished inside the free() that makes room    (:354-360)

Additions (not in the reference, off unless called): `compact()` — plans and executes, with the
compact_blocks HIP kernel, the block moves that empty sparsely used pages so they can be unmapped.
"""
from __future__ import annotations

import functools
import threading
import time
from typing import Any, Dict, List, Optional, Tuple

from kvcached_amd import vmm_ops as _ops
from kvcached_amd.locks import NoOpLock
from kvcached_amd.tp_ipc_util import broadcast_kv_tensors_created
from kvcached_amd.utils import (
    BATCH_PAGE_ALLOC,
    CONTIGUOUS_LAYOUT,
    DEFAULT_IPC_NAME,
    PAGE_PREALLOC_ENABLED,
    PAGE_SIZE,
    SANITY_CHECK,
    KVCachedConfigError,
    get_kvcached_logger,
)
from kvcached_amd.vmm_ops import kv_tensors_created

PageAllocator = _ops.PageAllocator
InternalPage: Any = _ops.InternalPage

logger = get_kvcached_logger()

KV_TENSOR_WAIT_TIMEOUT: float = 10.0  # seconds
PREALLOC_THREAD_TIMEOUT: float = 2.0  # seconds


def synchronized(method):
    """Run the method under self._lock (an RLock with async scheduling, a no-op otherwise)."""

    @functools.wraps(method)
    def locked(self, *args, **kwargs):
        with self._lock:
            return method(self, *args, **kwargs)

    return locked


def _worker_ipc_hook():
    """integration.vllm.interfaces.should_use_worker_ipc, if that integration is importable."""
    try:
        from kvcached_amd.integration.vllm.interfaces import should_use_worker_ipc
        return should_use_worker_ipc
    except ImportError:
        return None


class KVCacheManager:

    def __init__(
        self,
        num_blocks: int,
        block_size: int,
        cell_size: int,
        num_layers: int,
        world_size: int = 1,
        pp_rank: int = 0,
        async_sched: bool = False,
        reserve_null_block: bool = False,
        num_kv_buffers: int = 2,
        group_id: int = 0,
    ):
        """
        num_blocks: blocks in the pool; block_size: tokens per block; cell_size: bytes of one
        token in one layer's K (or V); num_layers: layers sharing the page ids; world_size: TP size
        inside one PP stage; reserve_null_block: keep block 0 out of circulation (SGLang pads with
        it); num_kv_buffers: 2 for K+V, 1 for MLA; group_id: KV-cache group of hybrid models.
        """
        self.num_blocks = num_blocks
        self.block_mem_size = block_size * cell_size
        self.num_layers = num_layers
        self.num_kv_buffers = num_kv_buffers
        self.reserve_null_block = reserve_null_block
        self.group_id = group_id
        self.page_size = PAGE_SIZE
        # uniform blocks per page (0 when blocks straddle page edges: then pages differ and nothing is batched)
        self._blocks_per_page = PAGE_SIZE // self.block_mem_size if PAGE_SIZE % self.block_mem_size == 0 else 0
        if self.block_mem_size > self.page_size:
            # no block would fit a page: the pool would stay empty and the engine would hang in
            # warm-up (hybrid linear-attention states can be this large) — refuse with the fix
            need_mb = -(-self.block_mem_size // (2 * 1024 * 1024)) * 2
            raise KVCachedConfigError(
                f"kvcached KV block size ({self.block_mem_size} bytes, "
                f"{self.block_mem_size / (1024 * 1024):.2f} MiB) is larger than the page size "
                f"({self.page_size} bytes, {self.page_size // (1024 * 1024)} MiB), so no block fits in a page "
                f"and the KV pool would be empty. Re-launch with KVCACHED_PAGE_SIZE_MB={need_mb} "
                f"(or larger; must be a multiple of 2).")
        # bytes of the K (or V) tensor of one layer
        self.mem_size = self.num_blocks * self.block_mem_size
        self.world_size = world_size
        self.pp_rank = pp_rank
        self.page_allocator = PageAllocator(
            self.num_layers,
            self.mem_size,
            self.page_size,
            self.world_size,
            pp_rank=self.pp_rank,
            async_sched=async_sched,
            contiguous_layout=CONTIGUOUS_LAYOUT,
            enable_page_prealloc=PAGE_PREALLOC_ENABLED,
            num_kv_buffers=self.num_kv_buffers,
            group_id=self.group_id,
            ipc_name=DEFAULT_IPC_NAME,
        )

        # The scheduler may live in another process than the workers even at world_size == 1
        # (vLLM V1 EngineCore); the allocator asks this hook before every map/unmap.
        hook = _worker_ipc_hook()
        remote = False
        if hook is not None:
            self.page_allocator.set_should_use_worker_ipc_callback(hook)
            remote = hook()
        if self.world_size > 1 or remote:
            self._install_broadcast_callbacks(remote)

        self.num_avail_blocks = 0  # free blocks inside avail_pages only
        self.avail_pages: Dict[int, InternalPage] = {}
        self.full_pages: Dict[int, InternalPage] = {}
        self.reserved_blocks: List[int] = []
        self.null_block: Optional[List[int]] = None
        self.in_shrink: bool = False
        self.target_num_blocks: Optional[int] = None
        self._lock = threading.RLock() if async_sched else NoOpLock()

        # Finishing touches need the KV tensors, which the worker creates later: do them from a
        # helper thread and let the public methods wait on this event.
        self._post_init_done = threading.Event()
        threading.Thread(target=self._post_init, daemon=True).start()

    def _install_broadcast_callbacks(self, remote: bool) -> None:
        try:
            from kvcached_amd.tp_ipc_util import broadcast_map_to_kv_tensors, broadcast_unmap_from_kv_tensors
            pp_rank, group_id = self.pp_rank, self.group_id

            def on_map(world_size: int, offsets: List[int]) -> None:
                broadcast_map_to_kv_tensors(world_size, offsets, pp_rank, group_id)

            def on_unmap(world_size: int, offsets: List[int]) -> None:
                broadcast_unmap_from_kv_tensors(world_size, offsets, pp_rank, group_id)

            self.page_allocator.set_broadcast_map_callback(on_map)
            self.page_allocator.set_broadcast_unmap_callback(on_unmap)
            logger.info("Set up broadcast callbacks for multi-process (world_size=%d, use_worker_ipc=%s)",
                        self.world_size, remote)
        except Exception as e:  # same tolerance as the reference (:166-169)
            logger.warning("Failed to set up broadcast callbacks: %s. Falling back to single-process mode.", e)

    # ------------------------------------------------------------------ deferred initialisation
    def _kv_tensors_ready(self) -> bool:
        hook = _worker_ipc_hook()
        if self.world_size > 1 or (hook is not None and hook()):
            return broadcast_kv_tensors_created(self.world_size, self.pp_rank, group_id=self.group_id)
        return kv_tensors_created(group_id=self.group_id)

    def _post_init(self):
        if self.null_block is not None:
            return
        try:
            waited = 0.0
            while not self._kv_tensors_ready():
                if waited >= KV_TENSOR_WAIT_TIMEOUT:
                    raise TimeoutError(f"KV tensors not created after {KV_TENSOR_WAIT_TIMEOUT} seconds")
                time.sleep(0.001)
                waited += 0.001
            self._reserve_null_block()
            self.page_allocator.start_prealloc_thread()
        except Exception as e:
            logger.error(f"Error during KVCacheManager post-initialization: {e}")
            raise
        finally:
            self._post_init_done.set()  # also on error, so callers do not hang

    def _wait_post_init(self):
        if not self._post_init_done.is_set():
            self._post_init_done.wait()

    def _reserve_null_block(self) -> None:
        if not self.reserve_null_block:
            self.null_block = None
            return
        self.null_block = self._alloc(1, _skip_wait=True)
        if self.null_block != [0]:
            logger.error(f"Failed to reserve null block, got {self.null_block}")
            raise RuntimeError("Failed to reserve null block at index 0")

    # ------------------------------------------------------------------ alloc / free
    def alloc(self, need_size: int) -> Optional[List[int]]:
        return self._alloc(need_size)

    @synchronized
    def _alloc(self, need_size: int, _skip_wait: bool = False) -> Optional[List[int]]:
        if not _skip_wait:
            self._wait_post_init()

        pending_limit = self.page_allocator.get_resize_target()  # set by the shm watcher (kvctl limit)
        if pending_limit > 0:
            self.resize(pending_limit)

        if self.available_size() < need_size:
            logger.warning(f"available_size()={self.available_size()} < need_size={need_size}")
            return None

        out: List[int] = []
        missing = need_size
        from_reserved = 0
        if self.reserved_blocks:
            take = min(len(self.reserved_blocks), missing)
            out = self.reserved_blocks[:take]
            self.reserved_blocks = self.reserved_blocks[take:]
            missing -= take
            from_reserved = take

        bms = self.block_mem_size
        fresh: List[Any] = []  # pages backed ahead for this call, in the order the loop below consumes them
        try:
            if BATCH_PAGE_ALLOC:
                # how many new pages the loop below will ask for: what the partially used pages cannot cover
                bpp = self._blocks_per_page
                beyond = missing - self.num_avail_blocks
                if bpp > 0 and beyond > bpp:  # two or more: back them with one map call
                    fresh = self.page_allocator.alloc_pages(-(-beyond // bpp))
                    fresh.reverse()  # consumed with pop()
            while missing > 0:
                if self.avail_pages:
                    _, page = self.avail_pages.popitem()  # most recently touched partial page
                else:
                    page = fresh.pop() if fresh else self.page_allocator.alloc_page()
                    page.init(bms)
                    if page.num_free_blocks() == 0:
                        # every aligned block of this page straddles its edge: park it where free()
                        # can still find it and take another page
                        self.full_pages[page.page_id] = page
                        continue
                    self.num_avail_blocks += page.num_free_blocks()
                take = min(page.num_free_blocks(), missing)
                out.extend(page.alloc(take))
                (self.full_pages if page.full() else self.avail_pages)[page.page_id] = page
                self.num_avail_blocks -= take
                missing -= take
        except Exception:
            # Backing a page failed (driver error, worker down). The reference never gets here with a GPU
            # error (it aborts the process) and would leak what this call had already taken; give it back.
            taken_from_pages = out[from_reserved:]
            self.reserved_blocks = out[:from_reserved] + self.reserved_blocks
            if taken_from_pages:
                self.free(taken_from_pages)
            if fresh:
                self.page_allocator.free_pages([p.page_id for p in reversed(fresh)])
            raise
        return out

    @synchronized
    def free(self, indices: List[int]):
        self._wait_post_init()
        if len(indices) == 0:
            return

        if SANITY_CHECK:
            for idx in indices:
                if idx in self.reserved_blocks:
                    raise ValueError(f"Freed index {idx} is in  reserved_blocks, which is not allowed.")

        by_page = self.page_allocator.group_indices_by_page(indices, self.block_mem_size)
        emptied: List[int] = []
        for page_id, blocks in by_page.items():
            page = self.full_pages.pop(page_id, None)
            if page is None:
                page = self.avail_pages.pop(page_id, None)
            if page is None:
                if SANITY_CHECK:
                    raise ValueError(f"Page {page_id} not found in avail_pages or full_pages. "
                                     f"This indicates a serious state inconsistency.")
                logger.error(f"Page {page_id} not found in avail_pages or full_pages. "
                             f"Skipping to avoid crash, but this indicates a serious bug.")
                continue
            self.num_avail_blocks += len(blocks)
            page.free_batch(blocks)
            if page.empty():
                emptied.append(page.page_id)
                self.num_avail_blocks -= page.num_free_blocks()
            else:
                self.avail_pages[page_id] = page

        if emptied:
            self.page_allocator.free_pages(emptied)

        if self.in_shrink:
            assert self.target_num_blocks is not None
            if self._get_num_alloced_blocks() <= self.target_num_blocks:
                self.page_allocator.resize(self.target_num_blocks * self.block_mem_size)
                self.in_shrink = False
                self.target_num_blocks = None

    @synchronized
    def try_to_reserve(self, need_size: int) -> bool:
        self._wait_post_init()
        if self.available_size() < need_size:
            return False
        got = self.alloc(need_size)
        if got is None:
            logger.warning("Failed to reserve blocks.")
            return False
        self.reserved_blocks.extend(got)
        return True

    @synchronized
    def free_reserved(self):
        if self.reserved_blocks:
            self.free(self.reserved_blocks)
            self.reserved_blocks.clear()

    # ------------------------------------------------------------------ elasticity
    @synchronized
    def resize(self, new_mem_size: int):
        """Set the limit of the K (or V) tensor of one layer, in bytes. Returns True when the new
        limit is in force, False when it has to wait for blocks to be freed (finished in free())."""
        self._wait_post_init()
        assert new_mem_size > 0, "new_mem_size must be positive"
        if self.page_allocator.resize(new_mem_size):
            if self.in_shrink:
                self.in_shrink = False
                self.target_num_blocks = None
            return True
        assert len(self.reserved_blocks) == 0, "Reserved blocks must be freed before resizing."
        self.in_shrink = True
        self.target_num_blocks = new_mem_size // self.block_mem_size
        self.free_reserved()
        return False

    @synchronized
    def trim(self) -> None:
        """Unmap the idle (reserved) pages and drop pooled physical handles."""
        self._wait_post_init()
        self.page_allocator.trim()

    @synchronized
    def available_size(self) -> int:
        blocks = self.num_avail_blocks + len(self.reserved_blocks)
        if self.in_shrink:
            return blocks
        pa = self.page_allocator
        virtual_free = pa.get_num_free_pages()
        physical_free = pa.get_avail_physical_pages() + pa.get_num_reserved_pages()
        return blocks + min(virtual_free, physical_free) * InternalPage.get_num_blocks(self.page_size, self.block_mem_size)

    @synchronized
    def get_mapped_memory_size(self, unit='bytes') -> float:
        """Physical memory behind in-use pages, in bytes / kb / mb / gb."""
        nbytes = self.page_allocator.get_num_inuse_pages() * self.num_layers * self.page_size * self.num_kv_buffers
        scale = {'bytes': 1, 'kb': 1024, 'mb': 1024**2, 'gb': 1024**3}
        if unit not in scale:
            raise ValueError(f"Unknown unit: {unit}")
        return nbytes if unit == 'bytes' else nbytes / scale[unit]

    @synchronized
    def clear(self):
        """Give every block back and return to the just-constructed state."""
        self._wait_post_init()
        # Stop the prealloc thread first: it could grab pages between the steps below and the
        # null block would then not be block 0. (The reference calls a method name its binding
        # does not export here and raises AttributeError; this is the evident intent.)
        self.page_allocator.stop_prealloc_thread()
        self.free_reserved()
        pages = [p.page_id for p in self.avail_pages.values()] + [p.page_id for p in self.full_pages.values()]
        if pages:
            self.page_allocator.free_pages(pages)
        self.avail_pages.clear()
        self.full_pages.clear()
        self.trim()
        # freed pages were appended to the free list; page 0 must come first again for the null block
        self.page_allocator.reset_free_page_order()
        self.target_num_blocks = None
        self.in_shrink = False
        self.num_avail_blocks = 0
        self._reserve_null_block()
        self.page_allocator.start_prealloc_thread()

    @synchronized
    def _get_num_alloced_blocks(self) -> int:
        per_page = InternalPage.get_num_blocks(self.page_size, self.block_mem_size)
        in_full = len(self.full_pages) * per_page
        in_partial = len(self.avail_pages) * per_page - self.num_avail_blocks
        return in_full + in_partial + len(self.reserved_blocks)

    # ------------------------------------------------------------------ compaction (addition)
    @synchronized
    def plan_compaction(self, max_moves: Optional[int] = None) -> List[Tuple[int, int]]:
        """Block moves (src, dst) that completely empty the sparsest partially used pages by
        filling the free blocks of the fullest ones. Pure planning; no state changes."""
        per_page = InternalPage.get_num_blocks(self.page_size, self.block_mem_size)
        pages = []
        for pid, page in self.avail_pages.items():
            free = page.get_free_blocks()
            start, end = InternalPage.get_block_range(pid, self.page_size, self.block_mem_size)
            free_set = set(free)
            used = [b for b in range(start, end) if b not in free_set]
            if self.null_block and self.null_block[0] in used:
                continue  # the null block must stay block 0
            pages.append((pid, used, free))
        if len(pages) < 2 or per_page <= 1:
            return []
        donors = sorted(pages, key=lambda t: (len(t[1]), t[0]))                # fewest live blocks first
        receivers = sorted(pages, key=lambda t: (-len(t[1]), t[0]))            # fullest first
        moves: List[Tuple[int, int]] = []
        taken: Dict[int, int] = {}   # receiver pid -> free blocks already promised
        gone = set()
        ri = 0
        for pid, used, _ in donors:
            if pid in gone or pid in taken:
                continue
            # can the remaining receivers absorb this whole page?
            plan, need, j = [], len(used), ri
            while need > 0 and j < len(receivers):
                rpid, _, rfree = receivers[j]
                if rpid == pid or rpid in gone:
                    j += 1
                    continue
                room = len(rfree) - taken.get(rpid, 0)
                if room <= 0:
                    j += 1
                    continue
                k = min(room, need)
                plan.append((rpid, k))
                need -= k
                if k == room:
                    j += 1
            if need > 0:
                break
            if max_moves is not None and len(moves) + len(used) > max_moves:
                break
            it = iter(used)
            for rpid, k in plan:
                rfree = next(f for p, _, f in receivers if p == rpid)
                base = taken.get(rpid, 0)
                for d in rfree[base:base + k]:   # InternalPage.alloc takes the first k free blocks
                    moves.append((next(it), d))
                taken[rpid] = base + k
            gone.add(pid)
        return moves

    @synchronized
    def compact(self, max_moves: Optional[int] = None) -> Dict[int, int]:
        """Move live blocks out of sparse pages (compact_blocks HIP kernel), free the emptied
        pages, and return {old_block_id: new_block_id}; the caller rewrites its block tables.
        Only meaningful when the KV tensors live in this process (world_size == 1, local map)."""
        self._wait_post_init()
        moves = self.plan_compaction(max_moves)
        if not moves:
            return {}
        from kvcached_amd import capi
        bases = capi.get_region_bases(self.group_id)
        if CONTIGUOUS_LAYOUT:   # one region; a block of all layers and K/V is one contiguous run
            block_bytes = self.block_mem_size * self.num_layers * self.num_kv_buffers
        else:
            block_bytes = self.block_mem_size
        src = [s for s, _ in moves]
        dst = [d for _, d in moves]
        capi.compact_blocks(bases, src, dst, block_bytes, sync=True)
        # bookkeeping: destinations become allocated (first-k order), sources become free
        by_dst = self.page_allocator.group_indices_by_page(dst, self.block_mem_size)
        for pid, blocks in by_dst.items():
            page = self.avail_pages.pop(pid)
            got = page.alloc(len(blocks))
            assert got == blocks, "compaction plan and page free-list order diverged"
            self.num_avail_blocks -= len(blocks)
            (self.full_pages if page.full() else self.avail_pages)[pid] = page
        self.free(src)
        return dict(moves)
