"""Page ids of the geometry engines use on ROCm - one region per layer, K half and V half (the reference forces the
non-contiguous layout there: kvcached/utils.py:150-171, csrc/allocator.cpp:189-206) - are backed as units: a LANE is the
`rows` pages behind one page id, one buffer holds k lanes row-major, k consecutive page ids are one ioctl per row
(DESIGN.md §4.11). What must hold is what held slot by slot: every page of every page id reads zero when it is handed
out, keeps what was written to it and nothing else, goes back as a whole, and the ledger balances."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

MiB = 1 << 20
PAGE = 2 * MiB
DEV = "cuda:0"
EPP = PAGE // 2   # int16 elements per page


@pytest.fixture()
def lanes():
    from kvcached_amd import capi, vmm_ops
    saved = {k: os.environ.get(k) for k in ("KVCACHED_PHYS_RESERVE_MB", "KVCACHED_ZERO_BACKFILL", "KVCACHED_LANE_EXTENT_MB")}
    os.environ["KVCACHED_PHYS_RESERVE_MB"] = "0"
    yield {"ops": vmm_ops, "capi": capi}
    vmm_ops.shutdown_kvcached()
    for k, v in saved.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    capi.set_option(capi.OPT_ZERO_BACKFILL, 1)


def _engine(state, layers, ids_per_half, compat, lane_mb=None, unified=False):
    ops, capi = state["ops"], state["capi"]
    os.environ["KVCACHED_ZERO_BACKFILL"] = "true" if compat else "false"
    if lane_mb is not None:
        os.environ["KVCACHED_LANE_EXTENT_MB"] = str(lane_mb)
    ops.init_kvcached(DEV, PAGE, False)
    if unified:                                    # one slot per layer per page id (MLA / unified pool: csrc/allocator.cpp:176-187)
        ts = ops.create_kv_tensors(ids_per_half * PAGE, 2, DEV, layers, 1, 0, True)
        return ops, capi, [t.view(ids_per_half, EPP) for t in ts]
    ts = ops.create_kv_tensors(2 * ids_per_half * PAGE, 2, DEV, layers, 2, 0, False)
    views = []
    for t in ts:                                   # row order = the order map_to_kv_tensors walks: layer-major, K then V
        views.append(t[:ids_per_half * EPP].view(ids_per_half, EPP))
        views.append(t[ids_per_half * EPP:].view(ids_per_half, EPP))
    return ops, capi, views


def _stamp(views, ids, base):
    for r, v in enumerate(views):
        for p in ids:
            v[p].fill_(base + 16 * r + (p % 16))
    torch.cuda.synchronize()


def _check(views, ids, base):
    for r, v in enumerate(views):
        for p in ids:
            want = base + 16 * r + (p % 16)
            assert bool((v[p][::4096] == want).all()), (r, p)


@pytest.mark.parametrize("compat,unified", [(True, False), (False, False), (True, True)])
def test_page_ids_are_backed_and_given_back_as_units(lanes, compat, unified):
    ops, capi, views = _engine(lanes, layers=8 if unified else 4, ids_per_half=64, compat=compat, unified=unified)
    R = len(views)
    assert R == 8 and capi.get_option(129) == 8          # page ids are backed by lanes, 8 to a buffer at most
    capi.reset_stats()                                   # once: the ledger at the end counts from here
    snap = lambda: (capi.get_stats()["pages_mapped"], capi.get_option(130 + 21), capi.get_option(130 + 23))   # noqa: E731  pages, ioctls, runs
    delta = lambda a, b: tuple(y - x for x, y in zip(a, b))                                                     # noqa: E731
    # one page id: R slots, R ioctls, one lane
    s0 = snap()
    assert ops.map_to_kv_tensors([5 * PAGE])
    assert delta(s0, snap()) == (R, R, 1)
    for v in views:
        assert int(torch.count_nonzero(v[5])) == 0
    # eight consecutive page ids in one call: still R ioctls (one run per row), one buffer
    s0 = snap()
    ids = list(range(16, 24))
    assert ops.map_to_kv_tensors([p * PAGE for p in ids])
    assert delta(s0, snap()) == (8 * R, R, 1)
    for v in views:
        assert int(torch.count_nonzero(v[16:24])) == 0
    _stamp(views, [5] + ids, 1000)
    _check(views, [5] + ids, 1000)
    # a page id out of the middle of the run goes back alone: its slots are unbacked again (zeros in compat mode), its
    # neighbours - pages of the SAME buffer, mapped by the same ioctl - keep every word
    assert ops.unmap_from_kv_tensors([19 * PAGE])
    if compat:
        for v in views:
            assert int(torch.count_nonzero(v[19][::64])) == 0
    _check(views, [5, 16, 17, 18, 20, 21, 22, 23], 1000)
    # ... and the lane that came back serves the next single page id (a free lane of a partly used buffer goes first):
    # nothing is created, the pages read zero, nothing of the old contents shines through
    created = capi.get_stats()["handles_created"]
    assert ops.map_to_kv_tensors([40 * PAGE])
    assert capi.get_stats()["handles_created"] == created
    for v in views:
        assert int(torch.count_nonzero(v[40])) == 0
    _stamp(views, [40], 3000)
    _check(views, [5, 16, 17, 18, 20, 21, 22, 23], 1000)
    _check(views, [40], 3000)
    # scattered page ids, named in any order, some of them neighbours: runs are found whatever the order
    s0 = snap()
    more = [50, 2, 51, 30, 1, 52]
    assert ops.map_to_kv_tensors([p * PAGE for p in more])
    assert delta(s0, snap())[0] == len(more) * R and delta(s0, snap())[2] == 3     # {1,2} {30} {50,51,52}
    _stamp(views, more, 5000)
    _check(views, more, 5000)
    _check(views, [40], 3000)
    everything = [5, 16, 17, 18, 20, 21, 22, 23, 40] + more
    u0 = capi.get_stats()["pages_unmapped"]
    assert ops.unmap_from_kv_tensors([p * PAGE for p in everything])
    assert capi.get_stats()["pages_unmapped"] - u0 == len(everything) * R
    if compat:
        for v in views:
            assert int(torch.count_nonzero(v[:, ::4096])) == 0
    # recycled lanes come back clean
    assert ops.map_to_kv_tensors([p * PAGE for p in (7, 8, 9)])
    for v in views:
        assert int(torch.count_nonzero(v[7:10])) == 0
    assert ops.unmap_from_kv_tensors([p * PAGE for p in (7, 8, 9)])
    assert capi.get_option(124) == 0                     # no release the pool did not know
    ops.shutdown_kvcached()
    st = capi.get_stats()
    assert st["handles_created"] == st["handles_released"] and st["handles_created"] % R == 0


def test_every_byte_is_zeroed_once_per_use_and_the_fill_has_left_the_map_call(lanes):
    """§4.9 holds lane by lane: pages are zeroed on their way back (through the alias mapping of their buffer, page (row, lane)
    at row x k + lane), so a map call that gets recycled lanes launches nothing and still hands out zeros."""
    ops, capi, views = _engine(lanes, layers=3, ids_per_half=32, compat=True)
    R = len(views)
    capi.reset_stats()
    ids = [3, 4, 5, 6, 20]
    assert ops.map_to_kv_tensors([p * PAGE for p in ids])            # fresh memory: filled by the map call
    st = capi.get_stats()
    assert st["fill_bytes"] == len(ids) * R * PAGE and capi.get_option(capi.OPT_PAGES_PRESCRUBBED) == 0
    _stamp(views, ids, 700)
    assert ops.unmap_from_kv_tensors([p * PAGE for p in ids])        # scrubbed behind the unmap
    capi.flush_unmaps()
    assert capi.get_option(capi.OPT_PAGES_SCRUBBED) == len(ids) * R
    fills = capi.get_stats()["fill_launches"]
    assert ops.map_to_kv_tensors([p * PAGE for p in (10, 11, 12, 13, 30)])
    assert capi.get_stats()["fill_launches"] == fills                # nothing launched by the map call ...
    assert capi.get_option(capi.OPT_PAGES_PRESCRUBBED) == len(ids) * R
    for v in views:                                                  # ... and every word is zero
        assert int(torch.count_nonzero(v[10:14])) == 0 and int(torch.count_nonzero(v[30])) == 0
    st = capi.get_stats()
    assert st["fill_bytes"] == (st["pages_mapped"] - capi.get_option(capi.OPT_PAGES_PRESCRUBBED) + capi.get_option(capi.OPT_PAGES_SCRUBBED)) * PAGE


def test_a_call_the_lanes_cannot_take_goes_the_per_slot_way(lanes):
    """A page id named twice, or one that is backed already: the reference logs and goes on (csrc/ftensor.cpp:104-107). Such a
    call takes the per-slot path as a whole; pages backed either way live side by side and both kinds go back."""
    ops, capi, views = _engine(lanes, layers=2, ids_per_half=32, compat=True)
    R = len(views)
    capi.reset_stats()
    assert ops.map_to_kv_tensors([4 * PAGE])                         # a lane
    _stamp(views, [4], 100)
    m0 = capi.get_stats()["pages_mapped"]
    assert ops.map_to_kv_tensors([4 * PAGE, 9 * PAGE])               # 4 is backed: logged and skipped; 9 is backed slot by slot
    assert capi.get_stats()["pages_mapped"] - m0 == R
    _check(views, [4], 100)
    for v in views:
        assert int(torch.count_nonzero(v[9])) == 0
    _stamp(views, [9], 200)
    assert ops.map_to_kv_tensors([6 * PAGE, 6 * PAGE])               # named twice
    _check(views, [9], 200)
    assert ops.unmap_from_kv_tensors([p * PAGE for p in (9, 4, 6)])  # both kinds in one call
    for v in views:
        assert int(torch.count_nonzero(v[:, ::4096])) == 0
    assert ops.unmap_from_kv_tensors([4 * PAGE])                     # not mapped: logged, the call succeeds
    ops.shutdown_kvcached()
    st = capi.get_stats()
    assert st["handles_created"] == st["handles_released"]


def test_a_driver_failure_rolls_a_lane_batch_back(lanes):
    ops, capi, views = _engine(lanes, layers=2, ids_per_half=32, compat=True, lane_mb=16)    # 16 MiB buffers: 2 lanes each (4 rows x 2 MiB)
    assert capi.get_option(129) == 2
    capi.reset_stats()
    assert ops.map_to_kv_tensors([0])
    _stamp(views, [0], 50)
    capi.set_option(104, 1)                                          # the second creation from now fails
    with pytest.raises(RuntimeError):
        ops.map_to_kv_tensors([p * PAGE for p in range(10, 16)])     # 3 buffers of 2 lanes
    capi.set_option(104, -1)
    for v in views:                                                  # nothing of the batch is left behind
        assert int(torch.count_nonzero(v[10:16, ::4096])) == 0
    _check(views, [0], 50)
    assert ops.map_to_kv_tensors([p * PAGE for p in range(10, 16)])  # and the same call works afterwards
    for v in views:
        assert int(torch.count_nonzero(v[10:16])) == 0
    assert ops.unmap_from_kv_tensors([p * PAGE for p in [0] + list(range(10, 16))])
    ops.shutdown_kvcached()
    st = capi.get_stats()
    assert st["handles_created"] == st["handles_released"]


def test_lanes_can_be_switched_off(lanes, monkeypatch):
    monkeypatch.setenv("KVCACHED_LANE_EXTENTS", "false")
    ops, capi, views = _engine(lanes, layers=2, ids_per_half=16, compat=True)
    assert capi.get_option(129) == 0
    capi.reset_stats()
    assert ops.map_to_kv_tensors([3 * PAGE])
    assert capi.get_stats()["pages_mapped"] == len(views)
    assert ops.unmap_from_kv_tensors([3 * PAGE])


@pytest.mark.parametrize("layers", [1, 3])      # 1 layer, unified: the per-slot path with run-sized extents; 3 layers x K/V: lanes
def test_the_unmap_side_invalidation_may_trail_the_call_but_pages_wait_for_it(lanes, monkeypatch, layers):
    """KVCACHED_UNMAP_INVALIDATION_US=T (compat, relaxed): kvc_unmap_from_kv_tensors returns without its TLB
    invalidation; the freed pages are parked - not zeroed, not on offer - until an invalidation that covers them has
    happened: the library's own thread performs it within T microseconds, or the next map batch's own invalidation absorbs it
    (ONE invalidation per free+alloc cycle instead of two). Strictly: page tables -> invalidation -> zero fill -> pool."""
    import time
    from kvcached_amd import capi, vmm_ops as ops
    monkeypatch.setenv("KVCACHED_UNMAP_INVALIDATION_US", "400")
    monkeypatch.setenv("KVCACHED_ZERO_BACKFILL", "true")
    monkeypatch.setenv("KVCACHED_PHYS_RESERVE_MB", "256")   # a second set of pages to alternate with (the default reserve is 2 GiB); without
    #                                                         one a map call that finds the pool empty has the parked pages' invalidation
    #                                                         done at once rather than create memory - two per cycle again
    ops.init_kvcached(DEV, PAGE, False)
    if layers == 1:
        ts = ops.create_kv_tensors(64 * PAGE, 2, DEV, 1, 1, 0, True)
        views, R = [ts[0].view(64, EPP)], 1
    else:
        ts = ops.create_kv_tensors(2 * 64 * PAGE, 2, DEV, layers, 2, 0, False)
        views, R = [], 2 * layers
        for t in ts:
            views += [t[:64 * EPP].view(64, EPP), t[64 * EPP:].view(64, EPP)]
    assert capi.get_option(capi.OPT_UNMAP_INVALIDATION_US) == 400 and (capi.get_option(129) > 0) == (layers > 1)
    a, b = list(range(0, 8)), list(range(20, 28))
    capi.reset_stats()                                                       # once: the ledger at the end counts from here
    assert ops.map_to_kv_tensors([p * PAGE for p in a])
    _stamp(views, a, 900)
    capi.flush_unmaps()
    s0, z0 = capi.get_stats(), capi.get_option(capi.OPT_PAGES_SCRUBBED)
    # 1. a lone unmap: returns with the invalidation owed and the pages parked ...
    t0 = time.perf_counter()
    assert ops.unmap_from_kv_tensors([p * PAGE for p in a])
    took = time.perf_counter() - t0
    st = capi.get_stats()
    assert st["tlb_shootdowns"] == s0["tlb_shootdowns"] and capi.get_option(capi.OPT_PAGES_SCRUBBED) == z0, (st, took)
    assert st["pages_unmapped"] - s0["pages_unmapped"] == 8 * R
    # ... and the library's thread has it done a moment later, by itself: invalidation first, zero fill after, then the pool
    time.sleep(0.02)
    st = capi.get_stats()
    assert st["tlb_shootdowns"] == s0["tlb_shootdowns"] + 1 and capi.get_option(capi.OPT_PAGES_SCRUBBED) == z0 + 8 * R
    for v in views:
        assert int(torch.count_nonzero(v[0:8, ::512])) == 0                 # the freed addresses read zero
    # 2. a cycle: the map batch that follows an unmap at once absorbs its invalidation - one per cycle
    assert ops.map_to_kv_tensors([p * PAGE for p in a])
    _stamp(views, a, 1700)
    capi.flush_unmaps()
    before = capi.get_stats()["tlb_shootdowns"]
    for rnd in range(6):
        cur, nxt = (a, b) if rnd % 2 == 0 else (b, a)
        assert ops.unmap_from_kv_tensors([p * PAGE for p in cur])
        assert ops.map_to_kv_tensors([p * PAGE for p in nxt])              # pages from the pool (or parked ones, after their invalidation)
        for v in views:
            assert int(torch.count_nonzero(v[nxt[0]:nxt[-1] + 1])) == 0, rnd   # zero, every word: nothing of the other set's stamps
        _stamp(views, nxt, 2000 + 100 * rnd)
        _check(views, nxt, 2000 + 100 * rnd)
    cycles = capi.get_stats()["tlb_shootdowns"] - before
    assert cycles <= 6 + 2, cycles                                           # (strict compat: 12)
    assert ops.unmap_from_kv_tensors([p * PAGE for p in (b if rnd % 2 == 0 else a)])
    capi.flush_unmaps()
    assert capi.get_option(124) == 0
    ops.shutdown_kvcached()
    st = capi.get_stats()
    assert st["handles_created"] == st["handles_released"]


def test_the_reserve_covers_what_is_parked(lanes, monkeypatch):
    """Relaxed compat with a reserve of exactly ONE batch: the reserve target grows by what unmaps have parked
    (GpuContext::reserve_target_bytes), so that the map batch behind an unmap finds idle pages next to the parked ones instead of
    having their invalidation performed inside its own acquire - one invalidation per cycle once the reserve stands, and no memory
    created from then on."""
    from kvcached_amd import capi, vmm_ops as ops
    monkeypatch.setenv("KVCACHED_UNMAP_INVALIDATION_US", "100000")          # (long: only map batches invalidate during the cycles)
    monkeypatch.setenv("KVCACHED_ZERO_BACKFILL", "true")
    monkeypatch.setenv("KVCACHED_PHYS_RESERVE_MB", "32")                    # = one batch of 16 pages
    ops.init_kvcached(DEV, PAGE, False)
    ts = ops.create_kv_tensors(64 * PAGE, 2, DEV, 1, 1, 0, True)
    views = [ts[0].view(64, EPP)]
    a, b = list(range(0, 16)), list(range(30, 46))
    capi.reset_stats()
    assert ops.map_to_kv_tensors([p * PAGE for p in a])
    for rnd in range(6):                                                     # the reserve is brought along by these calls
        cur, nxt = (a, b) if rnd % 2 == 0 else (b, a)
        assert ops.unmap_from_kv_tensors([p * PAGE for p in cur])
        assert ops.map_to_kv_tensors([p * PAGE for p in nxt])
    s0 = capi.get_stats()
    for rnd in range(6):
        cur, nxt = (a, b) if rnd % 2 == 0 else (b, a)
        assert ops.unmap_from_kv_tensors([p * PAGE for p in cur])
        assert ops.map_to_kv_tensors([p * PAGE for p in nxt])
        assert int(torch.count_nonzero(views[0][nxt[0]:nxt[-1] + 1])) == 0, rnd
        _stamp(views, nxt, 3000 + 100 * rnd)
        _check(views, nxt, 3000 + 100 * rnd)
    s1 = capi.get_stats()
    assert s1["tlb_shootdowns"] - s0["tlb_shootdowns"] <= 6 + 2, (s0, s1)   # the map batch's own (+ the library's thread, should this
    #                                                                          process stall for 150 us between an unmap and its map); 12 = every acquire had to invalidate
    assert s1["handles_created"] == s0["handles_created"], (s0, s1)
    assert s1["handles_created"] >= 48, s1                                   # a mapped batch + a parked one + an idle one
    assert ops.unmap_from_kv_tensors([p * PAGE for p in a])
    capi.flush_unmaps()
    ops.shutdown_kvcached()
    st = capi.get_stats()
    assert st["handles_created"] == st["handles_released"]


def test_whole_tensors_can_be_read_next_to_background_mapping(lanes, monkeypatch):
    """"Unbacked VA reads as zeros, never faults" holds for slots at rest; a slot in transition - inside the one ioctl that backs
    it or gives it up - has invalid entries for ~2 us, and an access that lands there is a GPU fault
    (profiles/r02_soak_touch_unbacked_fault.log; no DRM sequence avoids it: tools/engine_ioctl_probe.cpp part H).
    kvc_quiesce_begin/_end hold every page-table update meanwhile: a thread hammers map/unmap while this one sums EVERY word
    of every region, backed or not, under the fence - no fault, the sums only ever show what the other thread stamped."""
    import threading
    import time
    ops, capi, views = _engine(lanes, layers=2, ids_per_half=64, compat=True)
    stop, errors, rounds = threading.Event(), [], [0]

    def churn():
        torch.cuda.set_device(0)
        try:
            k = 0
            while not stop.is_set():
                ids = [(k * 5 + j) % 64 for j in range(1 + k % 6)]
                ids = sorted(set(ids))
                assert ops.map_to_kv_tensors([p * PAGE for p in ids])
                assert ops.unmap_from_kv_tensors([p * PAGE for p in ids])
                k += 1
                rounds[0] = k
        except Exception as e:   # noqa: BLE001
            errors.append(repr(e))

    t = threading.Thread(target=churn)
    t.start()
    try:
        t0, sweeps = time.time(), 0
        while time.time() - t0 < 6:
            with capi.quiesced():
                total = sum(int(v[:, ::256].to(torch.int64).sum()) for v in views)   # a word of every 512 B of every slot
            assert total == 0                                                       # nobody stamps anything here: zeros everywhere
            sweeps += 1
            time.sleep(0.002)                                                       # (the fence is a plain mutex: give the other thread its turn)
    finally:
        stop.set()
        t.join(30)
    assert not errors, errors
    assert sweeps > 20 and rounds[0] > 100, (sweeps, rounds)
    with pytest.raises(RuntimeError):
        capi.check(capi.lib.kvc_quiesce_end())                                     # an end without a begin is refused


@pytest.mark.parametrize("compat", [True, False])
def test_page_ids_are_shared_as_units(lanes, compat):
    """The shared pool with page ids as units (north star: rank 0 creates, peers map; no reference counterpart): ONE dmabuf per page
    id - the buffer its lane lives in - plus (lanes in the buffer, lane index); the importer maps row r at (r x lanes + lane) x page.
    Page ids that are neighbours in one buffer travel as ONE descriptor and are one ioctl per row on the importing side.
    Here the "peer" is a second group of the same process: both groups see the same bytes in every row; an unmap on the importing
    side drops the import (and, compat, reads zero again) without touching the owner's page; the owner's memory goes home once."""
    from kvcached_amd import capi, vmm_ops as ops
    os.environ["KVCACHED_ZERO_BACKFILL"] = "true" if compat else "false"
    ops.init_kvcached(DEV, PAGE, False)
    layers, half = 3, 32
    a = ops.create_kv_tensors(2 * half * PAGE, 2, DEV, layers, 2, 0, False)
    b = ops.create_kv_tensors(2 * half * PAGE, 2, DEV, layers, 2, 1, False)        # group 1 = the "peer"
    rows = lambda ts: [v for t in ts for v in (t[:half * EPP].view(half, EPP), t[half * EPP:].view(half, EPP))]   # noqa: E731
    va, vb = rows(a), rows(b)
    R = len(va)
    capi.reset_stats()
    ids = [4, 5, 6, 20]                                                # a run of three (one buffer, three lanes) and a single one
    assert ops.map_to_kv_tensors([p * PAGE for p in ids], 0)
    _stamp(va, ids, 4000)
    fds, meta = capi.export_page_ids([p * PAGE for p in ids], 0)
    # two buffers: the run of three (three lanes, in order) and the single one
    assert len(fds) == 2 and meta == [0, 3, 0, 0, 3, 1, 0, 3, 2, 1, 1, 0], meta
    i0 = capi.get_option(130 + 21)
    m0 = capi.get_stats()["pages_mapped"]
    capi.map_imported_page_ids([p * PAGE for p in ids], fds, meta, 1)
    for fd in fds:
        os.close(fd)
    assert capi.get_stats()["pages_mapped"] - m0 == 4 * R and capi.get_option(130 + 21) - i0 == 2 * R   # one ioctl per row per run of neighbours
    _check(vb, ids, 4000)                                              # the peer sees the owner's bytes, row by row, page id by page id
    vb[R - 1][5][100:116] = 77                                         # ... and writes into the same memory
    torch.cuda.synchronize()
    assert bool((va[R - 1][5][100:116] == 77).all())
    with pytest.raises(RuntimeError):                                  # a page id that is backed already cannot take an import
        f2, m2 = capi.export_page_ids([4 * PAGE], 0)
        try:
            capi.map_imported_page_ids([4 * PAGE], f2, m2, 1)
        finally:
            for fd in f2:
                os.close(fd)
    assert ops.unmap_from_kv_tensors([p * PAGE for p in (5, 20)], 1)   # the peer lets two of them go
    if compat:
        for v in vb:
            assert int(torch.count_nonzero(v[5][::64])) == 0 and int(torch.count_nonzero(v[20][::64])) == 0
    _check(va, [4, 6, 20], 4000)                                       # the owner's pages are untouched
    _check(vb, [4, 6], 4000)
    assert ops.unmap_from_kv_tensors([p * PAGE for p in (4, 6)], 1)
    assert ops.unmap_from_kv_tensors([p * PAGE for p in ids], 0)
    with pytest.raises(RuntimeError):                                  # nothing to export any more
        capi.export_page_ids([4 * PAGE], 0)
    ops.shutdown_kvcached()
    st = capi.get_stats()
    assert st["handles_created"] == st["handles_released"]
