"""Host-side behaviour that has no golden vector: the shm wire format and its concurrency, the
prealloc thread (invariants only: with it running, traces are timing-dependent in the reference
too), the resize watcher, clear(), the compaction planner, and the oracle's deterministic
prealloc statement against the product's thread."""
import os
import struct
import threading
import time

import numpy as np
import pytest

import kvc_testlib as T

MiB = 1 << 20
PAGE = 2 * MiB


@pytest.fixture()
def cpu_ops():
    from kvcached_amd import capi, vmm_ops
    vmm_ops.init_kvcached("cpu", PAGE, False)
    T.set_product_phys_pages(1 << 30, PAGE, 2, 2)
    yield vmm_ops, capi
    vmm_ops.shutdown_kvcached()
    capi.set_mem_info_override(0, 0)


def _pa(ops, name, pages=64, layers=2, kv=2, prealloc=False):
    ops.create_kv_tensors(pages * PAGE * kv, 1, "cpu", layers, kv, 0, False)
    return ops.PageAllocator(layers, pages * PAGE, PAGE, 1, 0, False, False, prealloc, kv, 0,
                             os.environ["KVCACHED_IPC_NAME"] + name)


def test_shm_segment_is_the_reference_wire_format(cpu_ops):
    """24 bytes, int64 little-endian {total, used, prealloc}; readable by anything that reads the
    reference's segment (kvcached/cli/utils.py: numpy int64 view, flock'd)."""
    ops, capi = cpu_ops
    pa = _pa(ops, "_shm")
    path = "/dev/shm/" + pa._ipc_name()
    assert os.path.getsize(path) == 24 and oct(os.stat(path).st_mode & 0o777) in ("0o666", "0o644", "0o664")
    total = 64 * PAGE * 2 * 2
    assert struct.unpack("<3q", open(path, "rb").read()) == (total, 0, 0)
    a = pa.alloc_page()
    assert struct.unpack("<3q", open(path, "rb").read()) == (total, PAGE * 2 * 2, 0)
    pa.free_page(a.page_id)       # stays mapped as a reserved page
    assert struct.unpack("<3q", open(path, "rb").read()) == (total, 0, PAGE * 2 * 2)
    # an external controller lowers the limit under flock(LOCK_EX), like kvctl does
    import fcntl
    with open(path, "r+b") as f:
        fcntl.flock(f, fcntl.LOCK_EX)
        f.write(struct.pack("<q", total // 2))
        fcntl.flock(f, fcntl.LOCK_UN)
    assert pa.check_and_get_resize_target(64 * PAGE) == 32 * PAGE
    assert pa.check_and_get_resize_target(32 * PAGE) == -1
    pa.trim()
    assert struct.unpack("<3q", open(path, "rb").read()) == (total // 2, 0, 0)   # our writes never touch field 0
    del pa
    assert not os.path.exists(path)


def test_shm_concurrent_readers_never_see_torn_fields(cpu_ops):
    ops, capi = cpu_ops
    pa = _pa(ops, "_shm2", pages=256)
    path = "/dev/shm/" + pa._ipc_name()
    unit = PAGE * 2 * 2
    stop, bad = threading.Event(), []

    def reader():
        seg = np.memmap(path, dtype=np.int64, mode="r", shape=(3,))
        while not stop.is_set():
            t, u, p = int(seg[0]), int(seg[1]), int(seg[2])
            if t != 256 * unit or u % unit or p % unit or not (0 <= u <= t and 0 <= p <= t):
                bad.append((t, u, p))

    threads = [threading.Thread(target=reader) for _ in range(4)]
    for t in threads:
        t.start()
    for _ in range(300):
        ids = [pa.alloc_page().page_id for _ in range(20)]
        pa.free_pages(ids)
    stop.set()
    for t in threads:
        t.join()
    assert not bad, bad[:3]
    del pa


def test_prealloc_thread_invariants(cpu_ops):
    """Thread on: reserved pool is refilled to MIN_RESERVED (5), free+inuse==total always, no page id is
    handed out twice, stop/start is idempotent, destruction joins the thread."""
    ops, capi = cpu_ops
    pa = _pa(ops, "_pre", pages=64, prealloc=True)
    pa.start_prealloc_thread()
    pa.start_prealloc_thread()
    t0 = time.time()
    while pa.get_num_reserved_pages() < 5 and time.time() - t0 < 5:
        time.sleep(0.005)
    assert pa.get_num_reserved_pages() == 5 and pa.get_num_free_pages() == 64   # reserved pages still count as free
    seen = set()
    held = []
    for _ in range(40):
        p = pa.alloc_page().page_id
        assert p not in seen
        seen.add(p)
        held.append(p)
        assert pa.get_num_free_pages() + pa.get_num_inuse_pages() == pa.get_num_total_pages()
    pa.free_pages(held[:25])
    t0 = time.time()
    while pa.get_num_reserved_pages() < 5 and time.time() - t0 < 5:
        time.sleep(0.005)
    # (free_pages fills the reserved list up to MAX_RESERVED (10) while the thread may have a refill of up to MIN_RESERVED (5)
    # pages in flight - decided before the free, appended after it: the same in the reference, whose list is not capped there)
    assert 5 <= pa.get_num_reserved_pages() <= 15
    pa.stop_prealloc_thread()   # (a page the thread is moving from the free list to the reserved list is in neither for a moment)
    pa.stop_prealloc_thread()
    assert sorted(pa._page_list(0) + pa._page_list(1) + held[25:]) == list(range(64))
    pa.free_pages(held[25:])
    assert pa.get_num_inuse_pages() == 0
    pa.start_prealloc_thread()
    del pa   # must not hang


def test_prealloc_refill_matches_the_oracle_statement(cpu_ops, oracle_lib):
    """With the caller idle while the thread works, the thread's refill is deterministic: free-front ->
    reserved-back, min(MIN_RESERVED - reserved, free, physical) pages — the oracle's prealloc_step()."""
    ops, capi = cpu_ops
    pa = _pa(ops, "_pre2", pages=16, prealloc=True)
    orc = T.OraclePA.create(oracle_lib, 2, 16 * PAGE, PAGE, prealloc=True)

    def settle(n):
        t0 = time.time()
        while pa.get_num_reserved_pages() != n and time.time() - t0 < 5:
            time.sleep(0.002)
        time.sleep(0.02)

    pa.start_prealloc_thread()
    orc.set_prealloc_needed(True)
    orc.prealloc_step()
    settle(5)
    assert pa._page_list(1) == orc.lst(1) == [0, 1, 2, 3, 4]
    for _ in range(3):
        assert pa.alloc_page().page_id == orc.alloc_page()
        orc.prealloc_step()
        settle(5)
        assert pa._page_list(1) == orc.lst(1) and pa._page_list(0) == orc.lst(0)
    # physical memory runs out: the refill is capped by what hipMemGetInfo would allow
    for phys, want_reserved in ((0, 4), (0, 3), (2, 4)):
        T.set_product_phys_pages(phys, PAGE, 2, 2)
        orc.set_phys(phys)
        assert pa.alloc_page().page_id == orc.alloc_page()
        orc.prealloc_step()
        settle(want_reserved)
        assert pa._page_list(1) == orc.lst(1) and len(orc.lst(1)) == want_reserved
        assert pa._page_list(0) == orc.lst(0)
    del pa
    orc.close()


def test_resize_watcher_publishes_the_limit(cpu_ops, monkeypatch):
    """kvctl-style limit change -> watcher (10 Hz) -> get_resize_target() -> next alloc() applies it."""
    ops, capi = cpu_ops
    import kvcached_amd.kv_cache_manager as kcm
    monkeypatch.setattr(kcm, "PAGE_PREALLOC_ENABLED", True)
    ops.create_kv_tensors(64 * PAGE * 2, 1, "cpu", 2, 2, 0, False)
    m = kcm.KVCacheManager(num_blocks=64 * 64, block_size=16, cell_size=2048, num_layers=2)
    assert m._post_init_done.wait(10)
    pa = m.page_allocator
    assert pa.get_resize_target() == -1
    path = "/dev/shm/" + pa._ipc_name()
    seg = np.memmap(path, dtype=np.int64, mode="r+", shape=(3,))
    seg[0] = 32 * PAGE * 2 * 2
    seg.flush()
    t0 = time.time()
    while pa.get_resize_target() == -1 and time.time() - t0 < 3:
        time.sleep(0.02)
    assert pa.get_resize_target() == 32 * PAGE
    assert m.alloc(10) is not None               # applies the pending limit first
    assert pa.get_num_total_pages() == 32
    # the segment is deleted under us (kvctl delete): the watcher re-creates it
    del seg
    os.unlink(path)
    t0 = time.time()
    while not os.path.exists(path) and time.time() - t0 < 3:
        time.sleep(0.02)
    assert os.path.exists(path) and os.path.getsize(path) == 24
    del m


def test_clear_and_null_block(cpu_ops):
    ops, capi = cpu_ops
    import kvcached_amd.kv_cache_manager as kcm
    ops.create_kv_tensors(64 * PAGE * 2, 1, "cpu", 2, 2, 0, False)
    m = kcm.KVCacheManager(num_blocks=64 * 64, block_size=16, cell_size=2048, num_layers=2, reserve_null_block=True)
    assert m._post_init_done.wait(10) and m.null_block == [0]
    a = m.alloc(500)
    m.try_to_reserve(40)
    m.free(a[100:300])
    m.clear()
    assert m.null_block == [0] and m.reserved_blocks == [] and not m.in_shrink
    assert m.page_allocator.get_num_inuse_pages() == 1 and m.page_allocator.get_num_reserved_pages() == 0
    assert m.alloc(3) == [1, 2, 3]
    assert m.get_mapped_memory_size() == 1 * PAGE * 2 * 2 and m.get_mapped_memory_size("mb") == 8.0
    with pytest.raises(ValueError):
        m.get_mapped_memory_size("tb")
    del m


def test_alloc_is_exception_safe_when_backing_a_page_fails(cpu_ops):
    """A worker (or the driver) fails while the 2nd page of an alloc() is being backed: alloc() raises, and the
    blocks it had already taken from the partially used page and from the reservation are given back — the same
    call succeeds with the same ids once the fault is gone. (The reference leaks them: its loop at
    kvcached/kv_cache_manager.py:279-304 has no unwinding; with GPU errors it aborts before that matters.)
    The GPU twin with an injected hipMemCreate failure is tests/test_gpu_manager.py."""
    ops, capi = cpu_ops
    import kvcached_amd.kv_cache_manager as kcm
    kcm.BATCH_PAGE_ALLOC = True                                # the product default (tests before may have left it off)
    ops.create_kv_tensors(16 * PAGE * 2, 1, "cpu", 2, 2, 0, False)
    m = kcm.KVCacheManager(num_blocks=16 * 64, block_size=16, cell_size=2048, num_layers=2)
    assert m._post_init_done.wait(10)
    first = m.alloc(10)
    assert m.try_to_reserve(4) and m.reserved_blocks == [10, 11, 12, 13]
    before = (m.available_size(), list(m.reserved_blocks), m.page_allocator.get_num_free_pages(),
              m.page_allocator._page_list(0), m.num_avail_blocks)
    pa = m.page_allocator
    pa.set_should_use_worker_ipc_callback(lambda: True)

    def boom(ws, offs):
        raise ValueError("worker down")
    pa.set_broadcast_map_callback(boom)
    with pytest.raises(RuntimeError, match="Failed to map page 1"):
        m.alloc(100)
    after = (m.available_size(), list(m.reserved_blocks), m.page_allocator.get_num_free_pages(),
             m.page_allocator._page_list(0), m.num_avail_blocks)
    assert after == before
    # the batched path (several new pages, ONE map call): all-or-nothing as well
    with pytest.raises(RuntimeError, match="Failed to map page 1"):
        m.alloc(300)
    after = (m.available_size(), list(m.reserved_blocks), m.page_allocator.get_num_free_pages(),
             m.page_allocator._page_list(0), m.num_avail_blocks)
    assert after == before
    pa.set_broadcast_map_callback(None)
    pa.set_should_use_worker_ipc_callback(None)
    got = m.alloc(100)
    assert got == list(range(10, 110))
    m.free(first + got)
    assert m.page_allocator.get_num_inuse_pages() == 0
    del m


def test_config_error_for_oversized_blocks(cpu_ops):
    import kvcached_amd.kv_cache_manager as kcm
    from kvcached_amd.utils import KVCachedConfigError
    with pytest.raises(KVCachedConfigError, match="KVCACHED_PAGE_SIZE_MB=4"):
        kcm.KVCacheManager(num_blocks=8, block_size=16, cell_size=200 * 1024, num_layers=1)


def test_compaction_planner(cpu_ops):
    """plan_compaction empties the sparsest pages into the free blocks of the fullest ones; moves are
    disjoint, destinations follow InternalPage's first-free order, emptied pages really become free."""
    ops, capi = cpu_ops
    import kvcached_amd.kv_cache_manager as kcm
    ops.create_kv_tensors(64 * PAGE * 2, 1, "cpu", 2, 2, 0, False)
    m = kcm.KVCacheManager(num_blocks=64 * 64, block_size=16, cell_size=2048, num_layers=2)
    assert m._post_init_done.wait(10)
    a = m.alloc(64 * 6)                      # pages 0..5 full
    keep = set(a[0:60]) | set(a[64:64 + 50]) | set(a[128:128 + 6]) | set(a[192:192 + 3]) | set(a[256:256 + 64]) \
        | set(a[320:320 + 1])                # live blocks per page: 60, 50, 6, 3, 64, 1
    m.free([b for b in a if b not in keep])
    before_inuse = m.page_allocator.get_num_inuse_pages()
    moves = m.plan_compaction()
    src, dst = [s for s, _ in moves], [d for _, d in moves]
    assert len(set(src)) == len(src) and len(set(dst)) == len(dst) and not set(src) & set(dst)
    assert set(src) <= keep and not set(dst) & keep
    assert {s // 64 for s in src} == {2, 3, 5}            # the three sparsest pages are emptied
    assert {d // 64 for d in dst} <= {0, 1}                # into the fullest partial pages
    assert len(moves) == 6 + 3 + 1
    # max_moves caps whole pages, never half a page
    assert len(m.plan_compaction(max_moves=5)) == 4        # pages 5 (1 block) and 3 (3 blocks)
    assert m.plan_compaction(max_moves=0) == []
    # executing the plan needs the GPU kernel: on the cpu device it must refuse, not fake it
    with pytest.raises(capi.KvcError):
        m.compact()
    assert m.page_allocator.get_num_inuse_pages() == before_inuse
    del m


def test_the_compaction_benchmarks_moves_are_the_planners(cpu_ops):
    """bench.py / benchmarks/bench_compact.py time compact_blocks on `planned_moves` (pages 30 % full at random)
    without a manager in the loop: the list must be what KVCacheManager.plan_compaction makes of that occupancy."""
    import sys
    import numpy as np
    ops, capi = cpu_ops
    import kvcached_amd.kv_cache_manager as kcm
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "benchmarks"))
    from bench_compact import planned_moves
    n_blocks, per_page = 64 * 64, 64
    ops.create_kv_tensors(64 * PAGE * 2, 1, "cpu", 2, 2, 0, False)
    m = kcm.KVCacheManager(num_blocks=n_blocks, block_size=16, cell_size=2048, num_layers=2)
    assert m._post_init_done.wait(10)
    a = m.alloc(n_blocks)
    assert sorted(a) == list(range(n_blocks))
    live = np.random.default_rng(2).random(n_blocks) < 0.3
    m.free([b for b in a if not live[b]])
    moves = m.plan_compaction()
    src, dst = planned_moves(n_blocks, per_page)
    assert len(moves) > 500 and moves == list(zip(src, dst))
    del m


def test_async_unmap_queue_on_the_cpu_device(cpu_ops):
    """KVC_OPT_ASYNC_UNMAP's queue/reclaimer/re-backing logic without a GPU (the cpu device has no driver calls, the
    state machine and the threads are the same): several threads map and unmap disjoint slot ranges while the
    reclaimer drains; after a flush the ledger balances. Run under TSAN by tools_sanitize_cpu.sh."""
    ops, capi = cpu_ops
    ops.create_kv_tensors(512 * PAGE * 2, 1, "cpu", 2, 2, 0, False)
    capi.set_option(capi.OPT_ASYNC_UNMAP, 1)
    try:
        capi.reset_stats()
        errors = []

        def churn(base):
            try:
                offs = [(base + i) * PAGE for i in range(64)]
                for r in range(40):
                    assert ops.map_to_kv_tensors(offs)
                    assert ops.unmap_from_kv_tensors(offs[r % 7:])
                    assert ops.unmap_from_kv_tensors(offs[:r % 7])
            except Exception as e:  # surfaced below
                errors.append(e)
        ts = [threading.Thread(target=churn, args=(b,)) for b in (0, 64, 128, 192)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        assert not errors, errors
        capi.flush_unmaps()
        st = capi.get_stats()
        slots = 4 * 40 * 64 * 4                                   # threads x rounds x offsets x (2 layers x K/V)
        assert st["pages_mapped"] == slots and st["unmaps_queued"] == slots
        assert st["pages_unmapped"] + st["unmaps_cancelled"] == slots
        # everything is unbacked now: mapping again works, and switching the option off flushes by itself
        assert ops.map_to_kv_tensors([0, PAGE]) and ops.unmap_from_kv_tensors([0, PAGE])
        capi.set_option(capi.OPT_ASYNC_UNMAP, 0)
        assert capi.get_stats()["pages_unmapped"] + capi.get_stats()["unmaps_cancelled"] == slots + 8
    finally:
        capi.set_option(capi.OPT_ASYNC_UNMAP, 0)
