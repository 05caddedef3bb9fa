"""BASELINE.json configs[4] at the level of the hot path: two engines (processes) with their own kvcached share
one MI355X. What the reference guarantees there (examples/01_simple_two_models): an engine's `available_size()`
follows the memory the OTHER engine maps and gives back, because every unmap releases physical memory at once
(csrc/page.cpp:17). Here handles are recycled through a pool, so the same guarantee needs the pool's idle decay
(ExtentPool::decay, KVCACHED_POOL_IDLE_MS) and pressure drain: this test pins both, with the device made artificially
small by a ballast allocation so that only a few GiB are in play."""
import multiprocessing as mp
import os
import time

import pytest
import torch

pytestmark = pytest.mark.gpu

GiB = 1 << 30
PAGE = 2 << 20
LAYERS, BLOCK_TOKENS, CELL = 4, 16, 2048                 # 32 KiB blocks, 64 per page; a page id = 8 slots = 16 MiB
BLOCKS_PER_GIB = GiB // (PAGE * LAYERS * 2) * 64
RESERVE_MB = 1024                                        # KVCACHED_PHYS_RESERVE_MB: idle pages the pool never gives back (and pre-creates)


def _engine(name, conn, unmap_invalidation_us=0):
    os.environ["KVCACHED_IPC_NAME"] = name
    os.environ["KVCACHED_UNMAP_INVALIDATION_US"] = str(unmap_invalidation_us)   # 0: strict compat (the default); > 0: relaxed
    os.environ["KVCACHED_PAGE_PREALLOC_ENABLED"] = "true"          # the watcher thread (10 Hz housekeeping) runs with it
    os.environ["KVCACHED_LOG_LEVEL"] = "ERROR"
    os.environ["KVCACHED_POOL_IDLE_MS"] = "500"
    os.environ["KVCACHED_PHYS_RESERVE_MB"] = str(RESERVE_MB)         # what the engine keeps ready for itself (the default)
    import kvcached_amd.kv_cache_manager as kcm
    from kvcached_amd import capi, vmm_ops
    torch.cuda.set_device(0)
    vmm_ops.init_kvcached("cuda:0", PAGE, False)
    num_blocks = 64 * BLOCKS_PER_GIB                                # 64 GiB of virtual pool per engine
    vmm_ops.create_kv_tensors(num_blocks * BLOCK_TOKENS * CELL * 2, 1, "cuda:0", LAYERS, 2, 0, False)
    m = kcm.KVCacheManager(num_blocks=num_blocks, block_size=BLOCK_TOKENS, cell_size=CELL, num_layers=LAYERS)
    assert m._post_init_done.wait(20)
    held = []
    conn.send("ready")
    while True:
        cmd, arg = conn.recv()
        if cmd == "avail":
            conn.send(m.available_size())
        elif cmd == "alloc":
            ids = m.alloc(arg)
            if ids is not None:
                held.append(ids)
            conn.send(None if ids is None else len(ids))
        elif cmd == "free":
            for ids in held:
                m.free(ids)
            held.clear()
            conn.send(True)
        elif cmd == "stats":
            st = capi.get_stats()
            conn.send({"created": st["handles_created"], "released": st["handles_released"], "reused": st["handles_reused"],
                       "inuse_pages": m.page_allocator.get_num_inuse_pages(),
                       "reserved_pages": m.page_allocator.get_num_reserved_pages()})
        elif cmd == "quit":
            del m
            vmm_ops.shutdown_kvcached()
            conn.send(True)
            return


def _ask(conn, cmd, arg=None, timeout=60):
    conn.send((cmd, arg))
    assert conn.poll(timeout), f"engine did not answer {cmd}"
    return conn.recv()


def _wait_for(fn, pred, timeout):
    t0 = time.perf_counter()
    while True:
        v = fn()
        if pred(v):
            return v, time.perf_counter() - t0
        if time.perf_counter() - t0 > timeout:
            return v, None
        time.sleep(0.05)


@pytest.mark.parametrize("unmap_invalidation_us", [0, 300])
def test_two_engines_share_one_gpu(unmap_invalidation_us):
    ctx = mp.get_context("spawn")
    pipes, procs = [], []
    ballast = None
    for name in ("kvc_colo_a", "kvc_colo_b"):
        parent, child = ctx.Pipe()
        p = ctx.Process(target=_engine, args=(f"{name}_{os.getpid()}", child, unmap_invalidation_us), daemon=True)
        p.start()
        pipes.append(parent)
        procs.append(p)
    try:
        for c in pipes:
            assert c.poll(120) and c.recv() == "ready"
        a, b = pipes
        # shrink the device: leave ~8 GiB above the allocator's own 5 % headroom
        free, total = torch.cuda.mem_get_info(0)
        ballast_bytes = int(free - 0.05 * total - 8 * GiB)
        assert ballast_bytes > 0
        ballast = torch.empty(ballast_bytes, dtype=torch.uint8, device="cuda:0")
        time.sleep(0.5)                                             # both prealloc threads have done their first refill
        a0, b0 = _ask(a, "avail"), _ask(b, "avail")
        # (8 GiB the device has left + the engine's own reserve, which it counts as free: it is its to use - complete from the
        # first map call on)
        assert 5 * BLOCKS_PER_GIB < a0 < (9 + RESERVE_MB / 1024) * BLOCKS_PER_GIB and abs(a0 - b0) <= BLOCKS_PER_GIB, (a0, b0)

        # A grows by 6 GiB: B sees it at once (hipMemGetInfo), and refuses what no longer fits
        got = _ask(a, "alloc", 6 * BLOCKS_PER_GIB)
        assert got == 6 * BLOCKS_PER_GIB, (got, a0, b0)
        b1 = _ask(b, "avail")
        # (up to RESERVE_MB of A's growth comes out of the reserve A was already holding - B had seen that as used - and
        # A's housekeeping refills it a tick later)
        assert b1 <= b0 - (5 * BLOCKS_PER_GIB - RESERVE_MB * BLOCKS_PER_GIB // 1024), (b0, b1)
        # (B: what the device has left - 8 GiB minus the 5-6 that A took from it - plus its own reserve)
        got = _ask(b, "alloc", 5 * BLOCKS_PER_GIB)
        assert got is None, (got, b0, b1)

        # A finishes: its pages are unmapped and the handles parked in A's pool; within the idle window + a few
        # watcher ticks they are back with the driver and B can have them
        assert _ask(a, "free")
        # ... and are A's own to take back at once: its available_size counts what sits in its pool
        a1 = _ask(a, "avail")
        assert a1 >= a0 - BLOCKS_PER_GIB, (a0, a1, _ask(a, "stats"))
        got = _ask(a, "alloc", 6 * BLOCKS_PER_GIB)
        assert got == 6 * BLOCKS_PER_GIB, (got, a0, a1)
        assert _ask(a, "free")
        b2, took = _wait_for(lambda: _ask(b, "avail"), lambda v: v >= b0 - BLOCKS_PER_GIB, timeout=20)
        assert took is not None, f"B still sees {b2} blocks (started with {b0}) 10 s after A freed 6 GiB"
        # ... because the handles really went back to the driver (all but the reserved pages, a few ticks later)
        # (created - released = what the engine still holds: the physical reserve, the reserved page ids - still mapped,
        # 8 slots each - and, per region, at most one extent that such a mapped page keeps from going back whole)
        keeps = RESERVE_MB * (1 << 20) // PAGE + 10 * LAYERS * 2 + LAYERS * 2 * 64
        sa, took_all = _wait_for(lambda: _ask(a, "stats"), lambda st: st["created"] - st["released"] <= keeps, timeout=20)
        assert took_all is not None, sa
        got = _ask(b, "alloc", 4 * BLOCKS_PER_GIB)
        assert got == 4 * BLOCKS_PER_GIB, (got, b0, b2)

        # now B holds 4 GiB (3-4 of them from the device, depending on whether its housekeeping has refilled the reserve it
        # dipped into): A sees 4-5 GiB of the device + its own reserve + a few straggler extents, at most 6.2 GiB. 7 does not
        # fit -> None, allocator state untouched; 3 GiB does fit
        a2 = _ask(a, "avail")
        got = _ask(a, "alloc", 7 * BLOCKS_PER_GIB)
        assert got is None, (got, a0, a2)
        got = _ask(a, "alloc", 3 * BLOCKS_PER_GIB)
        assert got == 3 * BLOCKS_PER_GIB, (got, a0, a2)
        assert _ask(a, "free") and _ask(b, "free")
        both, took2 = _wait_for(lambda: min(_ask(a, "avail"), _ask(b, "avail")), lambda v: v >= a0 - BLOCKS_PER_GIB, timeout=10)
        assert took2 is not None, both
        print(f"[colocation] reclaim after free: {took:.2f} s (idle window 0.5 s), second round {took2:.2f} s")
        for c in pipes:
            assert _ask(c, "quit")
    finally:
        ballast = None
        torch.cuda.empty_cache()          # give the ballast back to the driver: later tests need the memory
        for p in procs:
            p.join(10)
            if p.is_alive():
                p.kill()
                p.join(10)
        # whatever happened above, the next test starts with the memory back (the kernel wipes what a killed engine held
        # at ~30 GB/s before it is free again)
        t0 = time.time()
        while time.time() - t0 < 30:
            free, total = torch.cuda.mem_get_info(0)
            if free > 0.9 * total:
                break
            time.sleep(0.5)
