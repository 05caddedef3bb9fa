"""The reserve-then-back allocator on a real MI355X, driven through `vmm_ops` / the C ABI.

Mirrors what the reference's GPU tests check (tests/test_paged_allocator_aliasing.py: unbacked VA
aliases one zero page, backed pages are private) and adds what the north star adds: freshly backed
pages read as zeros even when the physical handle is recycled and dirty.
"""
import os
import time

import pytest
import torch

pytestmark = pytest.mark.gpu

MiB = 1 << 20
PAGE = 2 * MiB
DEV = "cuda:0"


@pytest.fixture()
def vmm():
    """Fresh allocator per test (compat defaults: zero backfill on, zero fill on)."""
    from kvcached_amd import capi, vmm_ops
    state = {"ops": vmm_ops, "capi": capi}
    # no pre-created reserve unless a test asks for one: these tests count handles and want RECYCLED pages early (with the
    # default 2 GiB of never-used pages on offer, the oldest idle first, a small test would not see a recycled page at all)
    had = os.environ.get("KVCACHED_PHYS_RESERVE_MB")
    os.environ.setdefault("KVCACHED_PHYS_RESERVE_MB", "0")
    yield state
    if had is None:
        os.environ.pop("KVCACHED_PHYS_RESERVE_MB", None)
    vmm_ops.shutdown_kvcached()
    capi.set_option(capi.OPT_ZERO_BACKFILL, 1)
    capi.set_option(capi.OPT_ZERO_FILL, 1)
    capi.set_option(capi.OPT_ASYNC_UNMAP, 0)


def _setup(vmm, layers=2, per_layer=64 * MiB, contiguous=False, backfill=True, kv=2, unified=False):
    ops, capi = vmm["ops"], vmm["capi"]
    os.environ["KVCACHED_ZERO_BACKFILL"] = "true" if backfill else "false"
    try:
        ops.init_kvcached(DEV, PAGE, contiguous)
    finally:
        os.environ.pop("KVCACHED_ZERO_BACKFILL", None)
    assert capi.get_option(capi.OPT_ZERO_BACKFILL) == int(backfill)
    ts = ops.create_kv_tensors(per_layer, 2, DEV, layers, kv, 0, unified)
    return ops, capi, ts


def test_tensor_surface(vmm):
    ops, capi, ts = _setup(vmm)
    assert len(ts) == 2 and ops.kv_tensors_created()
    for t in ts:
        assert t.dtype == torch.int16 and t.device == torch.device(DEV) and t.numel() == 64 * MiB // 2
        assert t.data_ptr() % PAGE == 0 and not t.requires_grad
    assert ts[1].data_ptr() - ts[0].data_ptr() == 64 * MiB       # consecutive reservations behind the hint
    assert capi.get_region_bases(0) == [ts[0].data_ptr(), ts[0].data_ptr() + 32 * MiB,
                                        ts[1].data_ptr(), ts[1].data_ptr() + 32 * MiB]


def test_unbacked_va_aliases_the_zero_page(vmm):
    """Reference semantics (csrc/ftensor.cpp:160-176): reads of unbacked VA return zeros and all
    unbacked pages are one physical page."""
    ops, capi, ts = _setup(vmm)
    k = ts[0].view(torch.int16)
    epp = PAGE // 2
    assert int(torch.count_nonzero(k[:4 * epp])) == 0
    k[5] = 1234                       # write through page 0's alias ...
    torch.cuda.synchronize()
    z = capi.get_option(capi.OPT_ZERO_EXTENT_PAGES)
    if capi.get_option(capi.OPT_PRT):   # drm backend: no page at all behind unbacked VA - the write was dropped, reads stay 0
        assert int(k[5]) == 0 and int(k[3 * epp + 5]) == 0 and int(ts[1].view(torch.int16)[5]) == 0
    elif z:  # zero extent: slot i of every region shows page i % z of ONE buffer - slot 0 of the next layer is the same memory
        assert int(ts[1].view(torch.int16)[5]) == 1234 and int(k[3 * epp + 5]) == 0
    else:   # sharded zero pages through ROCr: the slots of one shard are one physical page, like the reference's single one
        assert int(k[3 * epp + 5]) == 1234  # ... is visible through page 3's alias
    k[5] = 0
    torch.cuda.synchronize()


def test_map_gives_private_zeroed_pages_even_when_recycled(vmm):
    ops, capi, ts = _setup(vmm)
    epp = PAGE // 2
    offs = [0, 3 * PAGE, 7 * PAGE]
    capi.reset_stats()
    assert ops.map_to_kv_tensors(offs)
    st = capi.get_stats()
    assert st["pages_mapped"] == len(offs) * 2 * 2 and st["fill_bytes"] == st["pages_mapped"] * PAGE
    # K slot and V slot (second half of the tensor) of every layer are backed, zero, and private
    for li, t in enumerate(ts):
        for half in (0, t.numel() // 2):
            for j, o in enumerate(offs):
                page = t[half + o // 2: half + o // 2 + epp]
                assert int(torch.count_nonzero(page)) == 0
                page.fill_(100 * li + 10 * j + (1 if half else 0) + 1)
    torch.cuda.synchronize()
    for li, t in enumerate(ts):
        for half in (0, t.numel() // 2):
            for j, o in enumerate(offs):
                page = t[half + o // 2: half + o // 2 + epp]
                assert bool((page == 100 * li + 10 * j + (1 if half else 0) + 1).all())
    # neighbours are still the shared zero page
    assert int(torch.count_nonzero(ts[0][epp:2 * epp])) == 0
    # unmap (handles go to the pool, dirty), remap elsewhere: recycled and zero again
    assert ops.unmap_from_kv_tensors(offs)
    assert int(torch.count_nonzero(ts[0][:epp])) == 0               # back on the zero page
    capi.reset_stats()
    assert ops.map_to_kv_tensors([PAGE, 2 * PAGE, 4 * PAGE])
    st = capi.get_stats()
    assert st["handles_reused"] == 12 and st["handles_created"] == 0
    for t in ts:
        for half in (0, t.numel() // 2):
            for o in (PAGE, 2 * PAGE, 4 * PAGE):
                assert int(torch.count_nonzero(t[half + o // 2: half + o // 2 + epp])) == 0
    assert ops.unmap_from_kv_tensors([PAGE, 2 * PAGE, 4 * PAGE])


_CONFIGS = [(b, mode, au, ck, "") for b in ("drm", "hybrid", "hip") for mode in ("lazy", "compat") for au in (0, 1)
            for ck in ((1, 64) if b == "drm" else (1,)) if not (mode == "compat" and au)]   # async unmap is ignored in compat mode
# ... and the default backend with each of round 2's mechanisms switched off in turn (every switch that ships is a
# correctness surface for stale translations and recycled pages)
_CONFIGS += [("drm", mode, 0, 64, off) for mode in ("lazy", "compat")
             for off in ("KVCACHED_SCRUB_ON_RELEASE=false", "KVCACHED_KFD_TLB_FLUSH=false", "KVCACHED_MAP_WAITS_FOR_ALL_FLUSHES=true",
                         "KVCACHED_ASYNC_SHOOTDOWN=false")]
_CONFIGS += [("drm", mode, 0, ck, "KVCACHED_PHYS_RESERVE_MB=48") for mode in ("lazy", "compat") for ck in (1, 64)]   # never-used and recycled pages mixed
_CONFIGS += [("drm", "lazy", 0, 64, "KVCACHED_DEFER_UNMAP_SHOOTDOWN=true"), ("drm", "lazy", 1, 64, "KVCACHED_PRT=true"),
             ("drm", "lazy", 0, 64, "KVCACHED_PRT=true"), ("drm", "lazy", 0, 1, "KVCACHED_PRT=true"),      # lazy with PRT (opt-in)
             ("drm", "compat", 0, 64, "KVCACHED_PRT=false"),                                              # zero extent
             ("drm", "compat", 0, 1, "KVCACHED_PRT=false")]
# compat, relaxed: the invalidation an unmap owes trails the call by at most 300 us; the pages wait for it un-scrubbed, un-offered
_CONFIGS += [(b, "compat", 0, ck, "KVCACHED_UNMAP_INVALIDATION_US=300") for b, ck in (("drm", 64), ("drm", 1), ("hybrid", 1), ("hip", 1))]
# the two layers of this geometry make every page id a lane of two pages in the drm rows above with 64-page extents;
# here: lanes switched off (slot-by-slot backing with run-sized extents), and two lanes to a buffer (partly used buffers all the time)
_CONFIGS += [("drm", mode, 0, 64, sw) for mode in ("lazy", "compat") for sw in ("KVCACHED_LANE_EXTENTS=false", "KVCACHED_LANES_PER_BUFFER=2")]


@pytest.mark.parametrize("backend,mode,async_unmap,extent_pages,switch", _CONFIGS)
def test_pages_are_private_and_zeroed_in_every_shipped_configuration(vmm, monkeypatch, backend, mode, async_unmap, extent_pages,
                                                                     switch):
    """The stale-translation / recycled-page property over every combination that ships: VMM backend x lazy|compat x
    synchronous|queued unmaps x single-page|run-sized extents. Each round backs a random set of slots (they must read
    zero - never the page's or the slot's previous contents), stamps them, takes a random subset away again (in compat
    mode those must read zero at once: back on a zero alias) and checks every survivor, i.e. pages whose neighbours were
    just unmapped, re-aliased or re-backed."""
    import random
    monkeypatch.setenv("KVCACHED_VMM_BACKEND", backend)
    monkeypatch.setenv("KVCACHED_ASYNC_UNMAP", "true" if async_unmap else "false")
    monkeypatch.setenv("KVCACHED_PHYS_CHUNK_PAGES", str(extent_pages))
    if switch:
        monkeypatch.setenv(*switch.split("="))
    ops, capi, ts = _setup(vmm, layers=2, per_layer=64 * PAGE, backfill=(mode == "compat"), kv=1, unified=False)
    lanes_expected = backend == "drm" and extent_pages > 1 and not async_unmap and switch != "KVCACHED_LANE_EXTENTS=false"
    assert (capi.get_option(129) > 0) == lanes_expected, capi.get_option(129)
    # PRT behind unbacked VA: the compat default on the drm backend; lazy mode leaves unbacked VA unmapped unless asked
    assert capi.get_option(capi.OPT_PRT) == int(backend == "drm" and (switch == "KVCACHED_PRT=true" or (mode == "compat" and switch != "KVCACHED_PRT=false")))
    want_backend = {"drm": 3, "hybrid": 2, "hip": 0}[backend]
    assert capi.get_option(108) == want_backend
    assert capi.get_option(capi.OPT_ASYNC_UNMAP) == async_unmap
    capi.reset_stats()
    n, epp = 64, PAGE // 2
    rng = random.Random(sum(map(ord, backend + mode + switch)) + 7 * async_unmap + extent_pages)   # (str hashes differ per process)
    stamp, serial = {}, 0
    for rnd in range(8):
        free = [i for i in range(n) if i not in stamp]
        want = rng.sample(free, rng.randint(1, len(free)))
        assert ops.map_to_kv_tensors([i * PAGE for i in want])
        for i in want:
            serial += 1
            for t in ts:
                assert int(torch.count_nonzero(t[i * epp:(i + 1) * epp])) == 0, (rnd, i)
                t[i * epp:(i + 1) * epp] = serial
            stamp[i] = serial
        torch.cuda.synchronize()
        victims = rng.sample(sorted(stamp), rng.randint(1, max(1, len(stamp) * 2 // 3)))
        assert ops.unmap_from_kv_tensors([i * PAGE for i in victims])
        if capi.get_option(capi.OPT_PRT) and mode == "lazy":
            capi.flush_unmaps()                                        # lazy: the invalidation runs behind the call
        if switch.startswith("KVCACHED_UNMAP_INVALIDATION_US"):
            if rnd % 2:
                time.sleep(0.003)                                      # relaxed compat: "reads as zeros" holds 300 us after the call at the latest, by itself
            else:
                capi.flush_unmaps()                                    # ... or at once for whoever asks
        for i in victims:
            stamp.pop(i)
            if mode == "compat" or capi.get_option(capi.OPT_PRT):      # (PRT: unbacked VA reads as zeros in lazy mode too)
                for t in ts:
                    assert int(torch.count_nonzero(t[i * epp:(i + 1) * epp])) == 0, (rnd, i)
        for i, v in stamp.items():
            for t in ts:
                assert bool((t[i * epp:(i + 1) * epp] == v).all()), (rnd, i, v)
    assert ops.unmap_from_kv_tensors([i * PAGE for i in sorted(stamp)])
    capi.flush_unmaps()
    ops.shutdown_kvcached()
    st = capi.get_stats()
    assert st["handles_created"] == st["handles_released"]


def test_double_map_and_unmapped_unmap_are_tolerated_like_the_reference(vmm):
    ops, capi, ts = _setup(vmm)
    assert ops.map_to_kv_tensors([0])
    ts[0][:8] = 7
    torch.cuda.synchronize()
    assert ops.map_to_kv_tensors([0])             # logs "already mapped", still True (ftensor.cpp:104-107)
    assert bool((ts[0][:8] == 7).all())           # and the existing page was not replaced
    assert ops.unmap_from_kv_tensors([0])
    assert ops.unmap_from_kv_tensors([0])         # logs "not mapped", still True


def test_lazy_mode_without_backfill(vmm):
    """KVCACHED_ZERO_BACKFILL=false: unbacked VA stays unmapped; backed pages behave the same."""
    ops, capi, ts = _setup(vmm, backfill=False)
    epp = PAGE // 2
    assert ops.map_to_kv_tensors([2 * PAGE, 3 * PAGE])
    page = ts[1][2 * epp:4 * epp]
    assert int(torch.count_nonzero(page)) == 0
    page.fill_(9)
    torch.cuda.synchronize()
    assert bool((ts[1][2 * epp:4 * epp] == 9).all())
    assert ops.unmap_from_kv_tensors([2 * PAGE, 3 * PAGE])
    assert ops.map_to_kv_tensors([2 * PAGE])
    assert int(torch.count_nonzero(ts[1][2 * epp:3 * epp])) == 0
    assert ops.unmap_from_kv_tensors([2 * PAGE])


def test_deferred_unmap_shootdown_keeps_pages_private(vmm):
    """KVC_OPT_DEFER_UNMAP_SHOOTDOWN in lazy mode: an unmap batch whose handles all return to the pool performs no
    TLB invalidation of its own; a later map batch invalidates - before anything touches its pages - exactly when it
    re-backs a slot whose last unmap is still owed one (a stale translation of some OTHER address cannot shadow a
    mapping made here: the per-slot epochs of DESIGN.md §4.3), and handles that leave the pool for the driver are
    preceded by one. Whatever the slot a recycled (dirty) handle lands on, reads see zeros first and every page stays
    private. A map batch with nothing owed (the very first one here) invalidates nothing: a translation that was
    invalid is never cached."""
    ops, capi, ts = _setup(vmm, layers=1, per_layer=64 * MiB, backfill=False, kv=1, unified=True)
    assert capi.get_option(capi.OPT_DEFER_UNMAP_SHOOTDOWN) == 0          # off by default
    capi.set_option(capi.OPT_DEFER_UNMAP_SHOOTDOWN, 1)
    t = ts[0]
    epp = PAGE // 2
    import random
    rng = random.Random(0)
    live = {}
    capi.reset_stats()
    owed = set()                                                         # slots unmapped since the last invalidation
    seen_both = set()
    for r in range(12):
        slots = rng.sample([s for s in range(32) if s not in live], rng.randint(3, 8))   # recycled handles, other slots
        n0 = capi.get_stats()["tlb_shootdowns"]
        assert ops.map_to_kv_tensors([s * PAGE for s in slots])
        needs = bool(owed & set(slots))
        assert capi.get_stats()["tlb_shootdowns"] == n0 + int(needs), (r, sorted(owed), slots)
        seen_both.add(needs)
        if needs:
            owed.clear()                                                 # one invalidation covers every unmap before it
        for s in slots:
            page = t[s * epp:(s + 1) * epp]
            assert int(torch.count_nonzero(page)) == 0, (r, s)          # recycled handles were dirty
            page.fill_(100 * r + s + 1)
            live[s] = 100 * r + s + 1
        torch.cuda.synchronize()
        for s, v in live.items():
            assert bool((t[s * epp:(s + 1) * epp] == v).all()), (r, s)    # nobody else's write landed here
        victims = slots[:len(slots) // 2 + 1]
        n1 = capi.get_stats()["tlb_shootdowns"]
        assert ops.unmap_from_kv_tensors([s * PAGE for s in victims])
        assert capi.get_stats()["tlb_shootdowns"] == n1                  # deferred
        owed |= set(victims)
        for s in victims:
            live.pop(s)
    assert seen_both == {True, False}                                    # the trace exercised both kinds of map batch
    # the pool shrinks to nothing: the owed invalidation happens before the first handle goes to the driver
    st0 = capi.get_stats()
    capi.set_option(capi.OPT_POOL_BYTES, 0)
    last = sorted(live)
    assert ops.unmap_from_kv_tensors([s * PAGE for s in last])
    st1 = capi.get_stats()
    assert st1["handles_released"] > st0["handles_released"] and st1["tlb_shootdowns"] == st0["tlb_shootdowns"] + 1
    assert capi.get_option(capi.OPT_POOL_HELD_PAGES) == 0
    # switched off, every unmap invalidates by itself
    capi.set_option(capi.OPT_POOL_BYTES, 1 << 30)
    capi.set_option(capi.OPT_DEFER_UNMAP_SHOOTDOWN, 0)
    assert ops.map_to_kv_tensors([5 * PAGE])
    n = capi.get_stats()["tlb_shootdowns"]
    assert ops.unmap_from_kv_tensors([5 * PAGE])
    capi.flush_unmaps()                                                  # wait for the library's own thread
    assert capi.get_stats()["tlb_shootdowns"] == n + 1


def test_map_batches_invalidate_only_when_a_stale_translation_can_exist(vmm, monkeypatch):
    """Unmaps invalidate the TLBs (the VMM calls do not); maps need to only where something may still be cached for the
    address: a zero alias or a PRT entry being replaced (compat mode: a PRT entry is cached like a valid one once anything has
    looked at the address, tools/prt_tlb_probe.cpp) or an invalidation somebody deferred. Mapping an UNMAPPED slot (lazy
    mode) whose last unmap was invalidated needs nothing - an invalid translation is never cached on GFX9+ (KFD itself
    flushes after unmap only on this GPU family; tools/drm_vmm_probe.cpp mode 3), and nothing can have looked at it
    without faulting. Data check: pages recycled through the pool land on other slots between live neighbours and read
    zeros, neighbours keep their contents."""
    epp = PAGE // 2
    for mode, per_map in (("lazy", 0), ("compat", 1), ("compat-zero-extent", 1), ("always", 1)):
        if mode == "always":
            monkeypatch.setenv("KVCACHED_MAP_SHOOTDOWN", "always")
        if mode == "compat-zero-extent":
            monkeypatch.setenv("KVCACHED_PRT", "false")
            mode = "compat"
        else:
            monkeypatch.delenv("KVCACHED_PRT", raising=False)
        ops, capi, ts = _setup(vmm, layers=1, per_layer=64 * MiB, backfill=(mode == "compat"), kv=1, unified=True)
        assert capi.get_option(capi.OPT_PRT) == int(mode == "compat" and os.environ.get("KVCACHED_PRT") != "false" and capi.get_option(108) == 3)
        t = ts[0]
        capi.reset_stats()
        even = [s * PAGE for s in range(0, 32, 2)]
        assert ops.map_to_kv_tensors(even)
        assert capi.get_stats()["tlb_shootdowns"] == per_map, mode
        for s in range(0, 32, 2):
            t[s * epp:(s + 1) * epp].fill_(s + 1)
        for r in range(4):
            odd = [s * PAGE for s in range(1, 32, 2)]
            n0 = capi.get_stats()["tlb_shootdowns"]
            assert ops.map_to_kv_tensors(odd[r:] + odd[:r])                # recycled pages, rotated over the slots
            assert capi.get_stats()["tlb_shootdowns"] == n0 + per_map, (mode, r)
            for s in range(1, 32, 2):
                assert int(torch.count_nonzero(t[s * epp:(s + 1) * epp])) == 0, (mode, r, s)
                t[s * epp:(s + 1) * epp].fill_(1000 * (r + 1) + s)
            torch.cuda.synchronize()
            for s in range(0, 32, 2):
                assert bool((t[s * epp:(s + 1) * epp] == s + 1).all()), (mode, r, s)
            n1 = capi.get_stats()["tlb_shootdowns"]
            assert ops.unmap_from_kv_tensors(odd)
            capi.flush_unmaps()                                            # lazy mode: the library's own thread does it
            assert capi.get_stats()["tlb_shootdowns"] == n1 + 1, (mode, r)   # every unmap batch invalidates
        assert ops.unmap_from_kv_tensors(even)
        ops.shutdown_kvcached()


def test_slots_that_were_looked_at_while_unbacked_are_backed_correctly(vmm):
    """A PRT translation is cached by the TLBs once anything has looked at the address (tools/prt_tlb_probe.cpp,
    profiles/r02_prt_tlb_probe.log: after a chip-wide read of 512 PRT slots, backing them with no invalidation left every
    read stale and lost 0.8 % of the writes). An engine does look at unbacked KV addresses - padded block tables are why the
    reference keeps a zero page there (csrc/ftensor.cpp:160-176) - so in compat mode a map batch invalidates before its
    pages are used. Here: chip-wide reads of every unbacked slot, then back them all, write every word, invalidate (an
    unmap elsewhere), read every word back."""
    ops, capi, ts = _setup(vmm, layers=1, per_layer=260 * PAGE, backfill=True, kv=1, unified=True)
    t = ts[0]                                             # int16 elements
    epp = PAGE // 2
    n = 256
    for rnd in range(3):
        assert int(t[:n * epp:64].to(torch.int64).sum()) == 0          # every unbacked slot, one word per 128 B, read chip-wide
        n0 = capi.get_stats()["tlb_shootdowns"]
        assert ops.map_to_kv_tensors([i * PAGE for i in range(n)])
        if capi.get_option(capi.OPT_PRT):
            assert capi.get_stats()["tlb_shootdowns"] == n0 + 1        # before anything uses the new pages
        t[:n * epp] = 0x1357 + rnd                                      # every word
        torch.cuda.synchronize()
        assert ops.map_to_kv_tensors([(n + 1) * PAGE]) and ops.unmap_from_kv_tensors([(n + 1) * PAGE])   # an invalidation
        assert int((t[:n * epp] != 0x1357 + rnd).sum()) == 0, rnd       # no write was dropped, no read is stale
        assert ops.unmap_from_kv_tensors([i * PAGE for i in range(n)])
        assert int(t[:n * epp:64].to(torch.int64).sum()) == 0          # zeros again from the moment unmap returns


_NEIGHBOUR_CHILD = r"""
import os, sys
sys.path.insert(0, %r)
import torch
from kvcached_amd import capi, vmm_ops
PAGE = 2 << 20
hooked = bool(os.environ.get("KVCACHED_TEST_SKIP_PRT_REMAINDER_REFRESH"))
vmm_ops.init_kvcached("cuda:0", PAGE, False)
ta = vmm_ops.create_kv_tensors(512 * PAGE, 2, "cuda:0", 1, 1, 0, True)[0]     # 512 slots (8 groups of 64), int16 elements
epp = PAGE // 2
a = ta.view(512, epp)
bad, attempts = 0, 0
for attempt in range(12):
    for rnd, step in enumerate((2, 3, 5)):
        attempts += 1
        stamp = 0x100 + 8 * attempt + rnd
        mine = torch.tensor([i for i in range(512) if i %% step == 0], device="cuda:0")
        rest = torch.tensor([i for i in range(512) if i %% step != 0], device="cuda:0")
        assert vmm_ops.map_to_kv_tensors([int(i) * PAGE for i in mine])      # the call ends with its invalidation
        # 1. the unbacked neighbours are looked at, every word of them, chip-wide: PRT entries enter the TLBs
        assert int(a[rest].to(torch.int64).sum()) == 0
        # 2. the new pages are written through their own addresses ...
        a[mine] = stamp
        torch.cuda.synchronize()
        # 3. ... and read back the same way: neighbours first, then neighbours and new pages interleaved in ONE kernel
        assert int(a[rest].to(torch.int64).sum()) == 0
        bad += int((a[mine] != stamp).sum())
        col = a[:, 4096].to(torch.int64)                                      # one word of every slot, backed or not, in address order
        want = torch.zeros(512, dtype=torch.int64, device="cuda:0")
        want[mine] = stamp
        bad += int((col != want).sum())
        assert vmm_ops.unmap_from_kv_tensors([int(i) * PAGE for i in mine])
        assert int(a[:, ::256].to(torch.int64).sum()) == 0                    # zeros everywhere from the moment unmap returns
    if hooked and bad:
        break                                                                 # the hazard has shown: that is all the hooked run is for
print("WRONG WORDS", bad, "in", attempts, "rounds")
vmm_ops.shutdown_kvcached()
"""


def test_backed_slots_are_not_shadowed_by_their_unbacked_neighbours():
    """A PRT mapping that a REPLACE has split keeps page-table entries with the fragment size of the ORIGINAL mapping; a TLB
    that caches one of them (PRT entries are cached once looked at) answers for the whole fragment, backed slots included:
    they read as zeros and swallow writes, for good (tools/prt_tlb_probe.cpp "neighbours": 255 of 256 backed slots). The
    kernel rewrites those remainders at the next PRT operation of the process; a map batch makes one before its
    invalidation (DrmVm::refresh_prt_remainders). Here: every 2nd / 3rd / 5th slot of a region is backed, the unbacked ones
    are read chip-wide FIRST (their entries enter the TLBs), then the new pages are written and read back through their
    own addresses - in a child, and once more with the rewrite switched off by a hook, where those writes and reads go
    to the neighbours' PRT entries and come back wrong (the test has teeth)."""
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = dict(os.environ, KVCACHED_LOG_LEVEL="ERROR", KVCACHED_PHYS_RESERVE_MB="0")
    base.pop("KVCACHED_ZERO_BACKFILL", None)
    ok = subprocess.run([sys.executable, "-c", _NEIGHBOUR_CHILD % repo], env=base, capture_output=True, text=True, timeout=300)
    assert ok.returncode == 0 and "WRONG WORDS 0" in ok.stdout, (ok.stdout[-500:], ok.stderr[-1500:])
    from conftest import hooks_env   # the hook exists only in the tests' own build of the library
    broken = subprocess.run([sys.executable, "-c", _NEIGHBOUR_CHILD % repo], env=hooks_env(dict(base, KVCACHED_TEST_SKIP_PRT_REMAINDER_REFRESH="1")),
                            capture_output=True, text=True, timeout=300)
    assert broken.returncode == 0, broken.stderr[-1500:]
    wrong = int(broken.stdout.split("WRONG WORDS")[1].split()[0])
    assert wrong > 0, "without the rewrite of the split PRT mappings nothing was shadowed: the hook or the hazard is gone"


def test_unmap_batches_with_runs_of_neighbours(vmm):
    """drm backend: an unmap batch is sorted and every run of adjacent slots goes in one ranged ioctl (<= 16 slots per
    call). Whatever the listing order, the run lengths or a slot listed twice: exactly the listed slots
    lose their pages, their neighbours keep theirs (data intact), and every freed slot can be backed again."""
    import random
    ops, capi, ts = _setup(vmm, layers=1, per_layer=160 * MiB, backfill=False, kv=1, unified=True)
    t, epp = ts[0], PAGE // 2
    n = 80
    capi.reset_stats()
    assert ops.map_to_kv_tensors([i * PAGE for i in range(n)])
    for i in range(n):
        t[i * epp:(i + 1) * epp].fill_(i + 1)
    torch.cuda.synchronize()
    rng = random.Random(5)
    # runs of 1, 2, 3, 16, 17 (split 16 + 1) and 20 slots, with gaps of live neighbours in between
    victims = [0] + [2, 3] + [5, 6, 7] + list(range(9, 25)) + list(range(26, 43)) + list(range(44, 64))
    listed = victims[:]
    rng.shuffle(listed)
    assert ops.unmap_from_kv_tensors([i * PAGE for i in listed])
    capi.flush_unmaps()
    assert capi.get_stats()["pages_unmapped"] == len(victims)
    live = [i for i in range(n) if i not in set(victims)]
    for i in live:
        assert bool((t[i * epp:(i + 1) * epp] == i + 1).all()), i          # neighbours of every run untouched
    # a slot listed twice is tolerated like in the reference (logged, skipped), the rest of the batch goes through
    dup = [70, 71, 72, 71, 73]
    assert ops.unmap_from_kv_tensors([i * PAGE for i in dup])
    capi.flush_unmaps()
    assert capi.get_stats()["pages_unmapped"] == len(victims) + 4
    for i in (69, 74):
        assert bool((t[i * epp:(i + 1) * epp] == i + 1).all())
    # everything freed can be backed again (the ranges really are empty) and comes back zeroed
    again = victims + [70, 71, 72, 73]
    rng.shuffle(again)
    assert ops.map_to_kv_tensors([i * PAGE for i in again])
    for i in again:
        assert int(torch.count_nonzero(t[i * epp:(i + 1) * epp])) == 0, i
    for i in live:
        if i not in (70, 71, 72, 73):
            assert bool((t[i * epp:(i + 1) * epp] == i + 1).all()), i
    assert ops.unmap_from_kv_tensors([i * PAGE for i in range(n)])
    ops.shutdown_kvcached()
    st = capi.get_stats()
    assert st["handles_created"] == st["handles_released"]


_EXIT_CHILD = """
import os, sys
sys.path.insert(0, %r)
os.environ["KVCACHED_IPC_NAME"] = "kvc_exit_%%d" %% os.getpid()
import torch
from kvcached_amd import vmm_ops
vmm_ops.init_kvcached("cuda:0", 2 << 20, False)
ts = vmm_ops.create_kv_tensors(64 << 20, 2, "cuda:0", 2, 2, 0, False)
assert vmm_ops.map_to_kv_tensors([0, 2 << 20, 6 << 20])
ts[0][:100].fill_(3)
assert vmm_ops.unmap_from_kv_tensors([2 << 20])
torch.cuda.synchronize()
print("leaving without shutdown_kvcached", flush=True)
"""


_EXIT_CHILD_MANAGER = """
import os, sys, time
sys.path.insert(0, %r)
os.environ["KVCACHED_IPC_NAME"] = "kvc_exitm_%%d" %% os.getpid()
os.environ["KVCACHED_PAGE_PREALLOC_ENABLED"] = "true"
import torch
import kvcached_amd.kv_cache_manager as kcm
from kvcached_amd import vmm_ops
vmm_ops.init_kvcached("cuda:0", 2 << 20, False)
vmm_ops.create_kv_tensors(64 * (2 << 20) * 2, 1, "cuda:0", 2, 2, 0, False)
m = kcm.KVCacheManager(num_blocks=64 * 64, block_size=16, cell_size=2048, num_layers=2)
assert m._post_init_done.wait(20)
ids = m.alloc(20 * 64)
m.free(ids[:600])
time.sleep(0.25)          # prealloc + watcher threads are running, pages are mapped, an invalidation may be in flight
print("leaving with a live manager", flush=True)
"""


def test_process_exit_with_a_live_manager_is_clean():
    """... and neither must its threads: an engine exits with a KVCacheManager alive (prealloc thread, 10 Hz watcher,
    the context's invalidation thread), pages mapped and blocks allocated."""
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, KVCACHED_LOG_LEVEL="ERROR")
    env.pop("KVCACHED_VMM_BACKEND", None)
    r = subprocess.run([sys.executable, "-c", _EXIT_CHILD_MANAGER % repo], env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, (r.returncode, r.stderr[-800:])
    assert "leaving with a live manager" in r.stdout


@pytest.mark.parametrize("backend", ["drm", "hybrid", "hip"])
def test_process_exit_without_shutdown_is_clean(backend):
    """An engine that simply exits (no shutdown_kvcached) must not crash on the way out: nothing of the library tears
    GPU state down from static destructors - the HIP runtime, ROCr and libdrm are going away in an unspecified order
    then (that used to end in a segmentation fault); the kernel reclaims mappings and memory with the process."""
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, KVCACHED_VMM_BACKEND=backend, KVCACHED_LOG_LEVEL="ERROR")
    r = subprocess.run([sys.executable, "-c", _EXIT_CHILD % repo], env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, (r.returncode, r.stderr[-800:])
    assert "leaving without shutdown_kvcached" in r.stdout


@pytest.mark.parametrize("max_pages", [16, 64])
def test_run_sized_extents_keep_every_page_with_its_slot(vmm, monkeypatch, max_pages):
    """Run-sized physical extents (the default with the drm backend): a run of adjacent slots is backed
    by ONE buffer and mapped with one ioctl; single pages are mapped at an offset into it. The hazard this has to
    handle: pages taken out of the middle of a multi-page mapping leave neighbours whose page-table entries still claim
    the old extent - a slot backed afresh would show its previous page. Seeded churn over two regions with every slot
    stamped and checked, neighbours read in between; then the ledger: nothing leaked, and the pool's own accounting
    (held = handed out + idle + free pieces) agrees with the driver-side counters."""
    import random
    monkeypatch.setenv("KVCACHED_PHYS_CHUNK_PAGES", str(max_pages))
    ops, capi, ts = _setup(vmm, layers=2, per_layer=192 * MiB, backfill=False, kv=1, unified=False)
    if capi.get_option(108) != 3 or capi.get_option(110) != 1:
        pytest.skip("needs the drm backend with pages straight from KFD")
    assert capi.get_option(capi.OPT_MAX_EXTENT_PAGES) == max_pages
    capi.reset_stats()
    n, epp = 96, PAGE // 2
    rng = random.Random(11)
    stamp = {}                                                        # slot -> value both layers carry
    serial = 0

    def check_all():
        torch.cuda.synchronize()
        for i, v in stamp.items():
            for t in ts:
                assert bool((t[i * epp:(i + 1) * epp] == v).all()), (i, v)

    for rnd in range(12):
        free = [i for i in range(n) if i not in stamp]
        want = rng.sample(free, rng.randint(1, len(free)))
        want.sort(key=lambda _: rng.random())
        assert ops.map_to_kv_tensors([i * PAGE for i in want])         # one offset = the slot in BOTH layers
        for i in want:
            serial += 1
            for t in ts:
                assert int(torch.count_nonzero(t[i * epp:(i + 1) * epp])) == 0, (rnd, i)   # never a previous page
                t[i * epp:(i + 1) * epp].fill_(serial)
            stamp[i] = serial
        check_all()
        victims = rng.sample(sorted(stamp), rng.randint(1, max(1, len(stamp) * 2 // 3)))   # holes inside the runs
        assert ops.unmap_from_kv_tensors([i * PAGE for i in victims])
        for i in victims:
            stamp.pop(i)
        check_all()                                                    # the survivors, right after their mappings were split
        held, out, waste = (capi.get_option(k) for k in (capi.OPT_POOL_HELD_PAGES, capi.OPT_POOL_OUT_PAGES, capi.OPT_POOL_FREE_PIECES))
        st = capi.get_stats()
        assert out == 2 * len(stamp) and held == st["handles_created"] - st["handles_released"] and held >= out + waste
    st = capi.get_stats()
    assert st["pages_mapped"] <= st["handles_created"] + st["handles_reused"]
    assert st["handles_created"] < st["pages_mapped"]                  # pieces and whole extents were reused
    assert ops.unmap_from_kv_tensors([i * PAGE for i in sorted(stamp)])
    ops.shutdown_kvcached()
    st = capi.get_stats()
    assert st["handles_created"] == st["handles_released"]


def test_a_batch_of_adjacent_slots_is_a_handful_of_ioctls(vmm):
    """The point of extents: 1024 adjacent 2 MiB slots are backed by 1024 / 32 buffers, mapped with as many ioctls and
    dropped with as many - against 1024 + 1024 with one buffer per page - and the same extents serve the next batch."""
    ops, capi, ts = _setup(vmm, layers=1, per_layer=4096 * PAGE, backfill=False, kv=1, unified=True)
    if capi.get_option(108) != 3 or capi.get_option(110) != 1:
        pytest.skip("needs the drm backend with pages straight from KFD")
    k = capi.get_option(capi.OPT_MAX_EXTENT_PAGES)
    assert k > 1
    import random
    offs = [i * PAGE for i in range(1024)]
    random.Random(0).shuffle(offs)                                      # the order the caller lists them in does not matter
    capi.reset_stats()
    assert ops.map_to_kv_tensors(offs)
    st = capi.get_stats()
    assert st["handles_created"] == 1024 and capi.get_option(capi.OPT_POOL_FREE_PIECES) == 0
    creates = capi.get_option(115)
    t = ts[0]
    assert int(torch.count_nonzero(t[:1024 * PAGE // 2])) == 0
    t[:1024 * PAGE // 2] = 7
    torch.cuda.synchronize()
    assert ops.unmap_from_kv_tensors(offs)
    capi.reset_stats()
    offs2 = [(2048 + i) * PAGE for i in range(1024)]                   # other slots, the same physical extents
    assert ops.map_to_kv_tensors(offs2)
    st = capi.get_stats()
    assert st["handles_created"] == 0 and st["handles_reused"] == 1024
    assert capi.get_option(115) == creates                             # not one driver allocation
    assert int(torch.count_nonzero(t[2048 * PAGE // 2:3072 * PAGE // 2])) == 0    # recycled, dirty, zero again
    assert ops.unmap_from_kv_tensors(offs2)


def test_pages_are_zeroed_on_their_way_back_not_on_their_way_out(vmm, monkeypatch):
    """DESIGN.md §4.9. With the drm backend every physical buffer keeps a second, private mapping; when its pages are given
    back the fill kernel is queued through that alias on a stream of the library's own, and the map call that hands them
    out again launches nothing - it waits for that scrub (usually long finished) and returns: the zero fill has left the
    allocation path. Fresh memory from the driver is still filled by the map call that first uses it."""
    epp = PAGE // 2
    for scrub in (True, False):
        monkeypatch.setenv("KVCACHED_SCRUB_ON_RELEASE", "true" if scrub else "false")
        ops, capi, ts = _setup(vmm, layers=1, per_layer=512 * PAGE, backfill=False, kv=1, unified=True)
        if capi.get_option(108) != 3 or capi.get_option(110) != 1:
            pytest.skip("needs the drm backend with pages straight from KFD")
        t = ts[0]
        offs = [i * PAGE for i in range(0, 96)] + [i * PAGE for i in (200, 202, 204)]
        capi.reset_stats()
        assert ops.map_to_kv_tensors(offs)
        st = capi.get_stats()
        assert st["fill_bytes"] == len(offs) * PAGE and st["handles_created"] == len(offs)     # fresh memory: filled by the map
        for o in offs:
            assert int(torch.count_nonzero(t[o // 2:o // 2 + epp])) == 0
            t[o // 2:o // 2 + epp] = 0x5a5a
        torch.cuda.synchronize()
        before_unmap = capi.get_stats()
        assert ops.unmap_from_kv_tensors(offs)
        capi.flush_unmaps()
        after_unmap = capi.get_stats()
        assert after_unmap["fill_bytes"] - before_unmap["fill_bytes"] == (len(offs) * PAGE if scrub else 0)   # the scrub ran behind the unmap
        offs2 = [(256 + i) * PAGE for i in range(0, 96)] + [(400 + 2 * i) * PAGE for i in range(3)]   # other slots, same pages
        assert ops.map_to_kv_tensors(offs2)
        st = capi.get_stats()
        assert st["handles_reused"] == len(offs2) and st["handles_created"] == len(offs)       # not one new page
        assert st["fill_bytes"] - after_unmap["fill_bytes"] == (0 if scrub else len(offs2) * PAGE)   # nothing launched by this map
        for o in offs2:
            assert int(torch.count_nonzero(t[o // 2:o // 2 + epp])) == 0                      # and still: zero, not 0x5a5a
        # a page that comes back while zero fill is switched off is not trusted later
        capi.set_option(capi.OPT_ZERO_FILL, 0)
        t[offs2[0] // 2:offs2[0] // 2 + epp] = 0x1111
        torch.cuda.synchronize()
        assert ops.unmap_from_kv_tensors(offs2[:1])
        capi.set_option(capi.OPT_ZERO_FILL, 1)
        assert ops.map_to_kv_tensors([500 * PAGE])
        assert int(torch.count_nonzero(t[500 * epp:501 * epp])) == 0
        assert ops.unmap_from_kv_tensors(offs2[1:] + [500 * PAGE])
        ops.shutdown_kvcached()
        st = capi.get_stats()
        assert st["handles_created"] == st["handles_released"]


def test_the_first_creation_brings_the_reserve_along_and_a_cycle_never_waits_for_its_own_scrub(vmm, monkeypatch):
    """DESIGN.md §4.9. Pages are zeroed on their way back; a free()+alloc() cycle that had to take the pages it has just given
    back would wait for that fill. The first extents a pool creates therefore bring KVCACHED_PHYS_RESERVE_MB of idle memory
    along (sized like themselves, created in that same call - a bare C-ABI caller has no watcher thread to do it later), the
    pool hands out what has been idle longest, and the cycle alternates between two sets of pages."""
    monkeypatch.setenv("KVCACHED_PHYS_RESERVE_MB", "128")          # 64 pages
    ops, capi, ts = _setup(vmm, layers=1, per_layer=512 * PAGE, backfill=True, kv=1, unified=True)
    if capi.get_option(108) != 3 or capi.get_option(110) != 1:
        pytest.skip("needs the drm backend with pages straight from KFD")
    t, epp = ts[0], PAGE // 2
    offs = [i * PAGE for i in range(64)]
    capi.reset_stats()
    assert ops.map_to_kv_tensors(offs)
    st = capi.get_stats()
    assert st["handles_created"] == 64 + 64                          # its own pages + the reserve
    assert capi.get_option(120) == 128 and capi.get_option(121) == 64   # held / handed out
    first = set()
    for cycle in range(6):
        for o in offs:
            assert int(torch.count_nonzero(t[o // 2:o // 2 + epp])) == 0, cycle
        t[:64 * epp] = 0x1234 + cycle
        torch.cuda.synchronize()
        assert ops.unmap_from_kv_tensors(offs)
        assert int(torch.count_nonzero(t[:64 * epp])) == 0           # compat: zeros from the moment unmap returns
        assert ops.map_to_kv_tensors(offs)
    st = capi.get_stats()
    assert st["handles_created"] == 128                              # nothing created after the first call
    assert capi.get_option(capi.OPT_PAGES_PRESCRUBBED) >= 4 * 64     # from the third cycle on every page arrives zeroed
    assert ops.unmap_from_kv_tensors(offs)
    ops.shutdown_kvcached()
    st = capi.get_stats()
    assert st["handles_created"] == st["handles_released"]


_TLB_CHILD = r"""
import os, sys
sys.path.insert(0, %r)
from kvcached_amd import capi, vmm_ops
try:
    vmm_ops.init_kvcached("cuda:0", 2 << 20, False)
except RuntimeError as e:
    print("INIT REFUSED:", e)
    sys.exit(0)
print("INIT OK backend", capi.get_option(108), "kfd flush", capi.get_option(118))
vmm_ops.shutdown_kvcached()
"""


@pytest.mark.parametrize("backend", ["drm", "hybrid", "hip"])
def test_init_refuses_to_start_when_tlb_invalidation_is_ineffective(backend):
    """Page privacy rests on the TLB invalidation after unmaps. init proves on a scratch slot - two pages
    swapped under one VA, written and read through the GPU's own translation - that the invalidation in effect works,
    with whatever backend is left standing after the fallback chain; when it does not (here: a hook turns the flush into
    a no-op, as a runtime that stopped flushing would) the library refuses to serve a single page."""
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = dict(os.environ, KVCACHED_VMM_BACKEND=backend, KVCACHED_LOG_LEVEL="ERROR")
    ok = subprocess.run([sys.executable, "-c", _TLB_CHILD % repo], env=base, capture_output=True, text=True, timeout=240)
    assert ok.returncode == 0 and "INIT OK" in ok.stdout, (ok.stdout, ok.stderr[-800:])
    from conftest import hooks_env   # the hook exists only in the tests' own build of the library
    broken = subprocess.run([sys.executable, "-c", _TLB_CHILD % repo], env=hooks_env(dict(base, KVCACHED_TEST_BREAK_TLB_FLUSH="1")),
                            capture_output=True, text=True, timeout=240)
    assert broken.returncode == 0, broken.stderr[-800:]
    assert "INIT REFUSED" in broken.stdout and "TLB invalidation is ineffective" in broken.stdout, broken.stdout


def test_tlb_invalidation_goes_straight_to_kfd(vmm, monkeypatch):
    """The default trigger is the KFD map/unmap ioctl pair on the library's own buffer (no user-space cache can answer
    it); KVCACHED_KFD_TLB_FLUSH=false keeps the hipMalloc+hipFree path, and privacy holds with either."""
    epp = PAGE // 2
    for use_kfd in (True, False):
        monkeypatch.setenv("KVCACHED_KFD_TLB_FLUSH", "true" if use_kfd else "false")
        # the flush object belongs to the per-device context, which shutdown destroys: a fresh init re-reads the switch
        ops, capi, ts = _setup(vmm, layers=1, per_layer=64 * PAGE, backfill=False, kv=1, unified=True)
        assert capi.get_option(capi.OPT_KFD_TLB_FLUSH_ACTIVE) == int(use_kfd)
        t = ts[0]
        for rnd in range(4):
            offs = [((rnd * 5 + i * 3) % 64) * PAGE for i in range(16)]
            offs = sorted(set(offs))
            assert ops.map_to_kv_tensors(offs)
            for o in offs:
                assert int(torch.count_nonzero(t[o // 2:o // 2 + epp])) == 0
                t[o // 2:o // 2 + epp] = rnd + 1
            torch.cuda.synchronize()
            assert ops.unmap_from_kv_tensors(offs)
        st = capi.get_stats()
        assert st["tlb_shootdowns"] >= 1
        ops.shutdown_kvcached()


def test_a_second_init_for_another_device_is_refused_while_kv_tensors_exist(vmm):
    """One device per process, as in the reference (one FTensorAllocator, csrc/allocator.cpp:18-22): pooled pages and the
    driver handles behind them belong to the device of the first init."""
    ops, capi, ts = _setup(vmm, layers=1, per_layer=8 * PAGE, backfill=False, kv=1, unified=True)
    assert ops.map_to_kv_tensors([0])
    ts[0][:16] = 3
    with pytest.raises(RuntimeError, match="one device per process"):
        ops.init_kvcached("cpu", PAGE, False)
    if torch.cuda.device_count() > 1:
        with pytest.raises(RuntimeError, match="one device per process"):
            ops.init_kvcached("cuda:1", PAGE, False)
    assert bool((ts[0][:16] == 3).all())                                # nothing was torn down
    assert ops.unmap_from_kv_tensors([0])
    ops.init_kvcached(DEV, PAGE, False)                                 # the same device: re-created like the reference
    ops.shutdown_kvcached()
    ops.init_kvcached("cpu", PAGE, False)                               # after a shutdown any device goes
    ops.shutdown_kvcached()


def test_reinit_never_writes_through_a_translation_of_a_previous_life(vmm):
    """Map batches no longer invalidate by themselves, so every path that removes a LIVE translation must: region
    teardown (pages and zero aliases), the init self tests, rollbacks. Otherwise: a region is torn down, its physical
    pages go back to the driver and are handed to somebody else (here: torch canaries), the next init reserves the
    same VA, and the zero fill of its first map batch lands - through the stale translation - in the canaries."""
    epp = PAGE // 2
    canaries = []
    n_pages = 96
    for life in range(6):
        backfill = life % 2 == 1
        ops, capi, ts = _setup(vmm, layers=1, per_layer=n_pages * PAGE, backfill=backfill, kv=1, unified=True)
        offs = [i * PAGE for i in range(n_pages) if (i * 7 + life) % 3]
        assert ops.map_to_kv_tensors(offs)                           # zero fill of (probably) the same VAs as last life
        for c, want in canaries:
            assert bool((c == want).all()), f"life {life}: a canary that owns recycled physical pages was overwritten"
        t = ts[0]
        for o in offs:
            i = o // PAGE
            assert int(torch.count_nonzero(t[i * epp:(i + 1) * epp])) == 0
            t[i * epp:(i + 1) * epp].fill_(life + 1)                    # make the pages dirty and their translations hot
        torch.cuda.synchronize()
        if life % 3 == 0:
            assert ops.unmap_from_kv_tensors(offs[::2])                # some go through the pool, the rest die mapped
        ops.shutdown_kvcached()
        # whoever gets those physical pages next. No canary is ever freed and the cache is never emptied: a hipFree
        # that reaches the driver invalidates the TLBs on its own and would hide a missing invalidation of ours.
        for k in range(3):
            c = torch.full((n_pages * PAGE // 8,), 0x5A5A0000 + life * 16 + k, dtype=torch.int32, device=DEV)
            canaries.append((c, 0x5A5A0000 + life * 16 + k))
        torch.cuda.synchronize()
    for c, want in canaries:
        assert bool((c == want).all())


def test_async_unmap_queue_reclaimer_and_rebacking(vmm):
    """KVC_OPT_ASYNC_UNMAP: unmap_from_kv_tensors returns after queueing; the reclaimer thread does the driver calls.
    Queued bytes count as free; a slot mapped again before its turn is kept as it is (no driver call) and reads
    zero again; everything else is unmapped for real (flush) and can be backed afresh; the handle ledger balances."""
    import time
    ops, capi, ts = _setup(vmm, layers=1, per_layer=8192 * MiB, backfill=False, kv=1, unified=True)
    capi.set_option(capi.OPT_ASYNC_UNMAP, 1)
    t = ts[0]
    epp = PAGE // 2
    n = 2048                                                     # 4 GiB: ~30 ms of driver unmaps
    offs = [i * PAGE for i in range(n)]
    assert ops.map_to_kv_tensors(offs)
    v = t[:n * epp].view(n, epp)
    v.copy_((torch.arange(n, device=DEV) % 200 + 1).to(torch.int16).unsqueeze(1).expand_as(v))
    torch.cuda.synchronize()
    free_before, _ = capi.mem_get_info()
    capi.reset_stats()
    t0 = time.perf_counter()
    assert ops.unmap_from_kv_tensors(offs)
    dt_queue = time.perf_counter() - t0
    free_after, _ = capi.mem_get_info()
    st = capi.get_stats()
    assert st["unmaps_queued"] == n and st["unmap_calls"] == 1
    assert dt_queue < 0.010, dt_queue                            # the synchronous path needs ~30 ms for this batch
    assert free_after >= free_before + (n - 300) * PAGE          # queued or already pooled: ours either way
    # the tail of the queue is mapped again at once: kept, not re-created, zero-filled
    tail = offs[-64:]
    assert ops.map_to_kv_tensors(tail)
    st = capi.get_stats()
    assert st["unmaps_cancelled"] == 64 and st["handles_created"] == 0 and st["handles_reused"] == 0
    assert int(torch.count_nonzero(v[-64:])) == 0
    v[-64:].fill_(7)
    capi.flush_unmaps()
    st = capi.get_stats()
    assert st["pages_unmapped"] == n - 64
    torch.cuda.synchronize()
    assert bool((v[-64:] == 7).all())                            # the kept pages were not touched by the reclaimer
    # a double unmap of a queued/unmapped slot is tolerated like the reference's (log + skip)
    assert ops.unmap_from_kv_tensors(offs[:4])
    # back everything again: recycled handles, zeros, private
    assert ops.map_to_kv_tensors(offs[:-64])
    assert all(int(torch.count_nonzero(v[i:min(i + 512, n - 64)])) == 0 for i in range(0, n - 64, 512))
    assert ops.unmap_from_kv_tensors(offs)
    capi.flush_unmaps()
    st = capi.get_stats()
    assert st["pages_unmapped"] == 2 * n - 64 and st["pages_mapped"] == n
    capi.set_option(capi.OPT_ASYNC_UNMAP, 0)


@pytest.mark.hooks_build
def test_default_backend_and_its_fallback_chain(vmm, monkeypatch):
    """Default: drm (own pages mapped with one GEM_VA ioctl) on top of hybrid (slots registered with
    HIP once, everything else through ROCr, §4.6); plain HIP copies keep working on such memory. Each layer is checked
    by a self test at init: drm falls back to hybrid, hybrid to plain HIP (forced here) - everything still works,
    only slower."""
    monkeypatch.delenv("KVCACHED_VMM_BACKEND", raising=False)
    epp = PAGE // 2
    for hook, want in ((None, 3), ("KVCACHED_TEST_FAIL_DRM_SELFTEST", 2), ("KVCACHED_TEST_FAIL_HYBRID_SELFTEST", 0)):
        if hook:
            monkeypatch.setenv(hook, "1")
        ops, capi, ts = _setup(vmm, layers=1, per_layer=16 * MiB, backfill=False, kv=1, unified=True)
        assert capi.get_option(108) == want
        assert ops.map_to_kv_tensors([0, 3 * PAGE])
        assert int(torch.count_nonzero(ts[0][3 * epp:4 * epp])) == 0
        ts[0][:epp].fill_(321)
        torch.cuda.synchronize()
        assert ts[0][:4].cpu().tolist() == [321] * 4                        # hipMemcpy D2H straight from the KV tensor
        ts[0][3 * epp:3 * epp + 4] = torch.tensor([1, 2, 3, 4], dtype=torch.int16)   # H2D into it
        assert ts[0][3 * epp:3 * epp + 4].clone().cpu().tolist() == [1, 2, 3, 4]     # D2D out of it
        assert ops.unmap_from_kv_tensors([0, 3 * PAGE])
        ops.shutdown_kvcached()


@pytest.mark.hooks_build
def test_drm_backend_one_ioctl_per_map_and_its_fallback(vmm, monkeypatch):
    """KVCACHED_VMM_BACKEND=drm: own pages are mapped with one DRM_AMDGPU_GEM_VA ioctl on a buffer
    object imported once per handle. Same observable behaviour as hybrid: zero-filled pages, HIP copies in and out,
    contents follow the physical page when slots are re-backed from the pool, handle ledger balanced; in compat mode
    the aliases (ROCr's) and the pages (DRM's) take turns at the same VA. A failing self test means hybrid."""
    monkeypatch.setenv("KVCACHED_VMM_BACKEND", "drm")
    for backfill, kfd_create in ((False, True), (True, True), (False, False), (False, "selftest fails")):
        monkeypatch.setenv("KVCACHED_DRM_KFD_CREATE", "false" if kfd_create is False else "true")
        if kfd_create == "selftest fails":
            monkeypatch.setenv("KVCACHED_TEST_FAIL_KFD_SELFTEST", "1")
        ops, capi, ts = _setup(vmm, layers=1, per_layer=64 * MiB, backfill=backfill, kv=1, unified=True)
        assert capi.get_option(108) == 3
        assert capi.get_option(110) == int(kfd_create is True)         # physical pages straight from KFD, or via ROCr
        capi.reset_stats()
        epp = PAGE // 2
        t = ts[0]
        if backfill:
            assert int(torch.count_nonzero(t[5 * epp:6 * epp])) == 0   # an alias of a zero page
        offs = [i * PAGE for i in (0, 3, 4, 9, 31)]
        assert ops.map_to_kv_tensors(offs)
        for k, i in enumerate((0, 3, 4, 9, 31)):
            assert int(torch.count_nonzero(t[i * epp:(i + 1) * epp])) == 0
            t[i * epp:(i + 1) * epp].fill_(100 + k)
        torch.cuda.synchronize()
        assert t[3 * epp:3 * epp + 4].cpu().tolist() == [101] * 4       # hipMemcpy D2H straight from the KV tensor
        t[9 * epp:9 * epp + 4] = torch.tensor([1, 2, 3, 4], dtype=torch.int16)
        assert t[9 * epp:9 * epp + 4].clone().cpu().tolist() == [1, 2, 3, 4]
        ops.map_to_kv_tensors([3 * PAGE])                               # double map: logged and skipped, like the reference
        assert t[3 * epp:3 * epp + 4].cpu().tolist() == [101] * 4
        assert ops.unmap_from_kv_tensors(offs)
        if backfill:
            assert int(torch.count_nonzero(t[3 * epp:4 * epp])) == 0   # the alias is back
        # the same five handles come back from the pool under other VAs: zero-filled, nothing stale shines through
        offs2 = [i * PAGE for i in (7, 8, 20, 21, 22, 23)]
        assert ops.map_to_kv_tensors(offs2)
        for i in (7, 8, 20, 21, 22, 23):
            assert int(torch.count_nonzero(t[i * epp:(i + 1) * epp])) == 0
        st = capi.get_stats()
        assert st["handles_reused"] >= 5 and st["pages_mapped"] == 11, (backfill, kfd_create, st, capi.get_option(124))
        assert ops.unmap_from_kv_tensors(offs2)
        ops.shutdown_kvcached()
        st = capi.get_stats()
        assert st["handles_created"] == st["handles_released"]
    monkeypatch.setenv("KVCACHED_TEST_FAIL_DRM_SELFTEST", "1")
    ops, capi, ts = _setup(vmm, layers=1, per_layer=16 * MiB, backfill=False, kv=1, unified=True)
    assert capi.get_option(108) == 2                                   # hybrid
    assert ops.map_to_kv_tensors([PAGE]) and int(torch.count_nonzero(ts[0][PAGE // 2:PAGE])) == 0
    assert ops.unmap_from_kv_tensors([PAGE])


def test_larger_page_size_8MiB(vmm):
    """KVCACHED_PAGE_SIZE_MB=8 (a multiple of 2 MiB, utils.py:95-124): slots, placeholders, zero fill and the handle
    ledger all follow the page size; fewer, larger mappings are what the driver charges least for."""
    ops, capi = vmm["ops"], vmm["capi"]
    P8 = 8 * MiB
    os.environ["KVCACHED_ZERO_BACKFILL"] = "false"
    try:
        ops.init_kvcached(DEV, P8, False)
    finally:
        os.environ.pop("KVCACHED_ZERO_BACKFILL", None)
    ts = ops.create_kv_tensors(64 * P8 * 2, 2, DEV, 2, 2, 0, False)
    epp = P8 // 2
    capi.reset_stats()
    assert ops.map_to_kv_tensors([5 * P8, 2 * P8, 9 * P8])
    st = capi.get_stats()
    assert st["pages_mapped"] == 3 * 2 * 2 and st["fill_bytes"] == 12 * P8
    half = ts[0].numel() // 2
    for t in ts:
        for base in (0, half):
            for p in (5, 2, 9):
                page = t[base + p * epp: base + (p + 1) * epp]
                assert int(torch.count_nonzero(page)) == 0
                page.fill_(p)
    torch.cuda.synchronize()
    assert ts[1][half + 9 * epp: half + 9 * epp + 4].cpu().tolist() == [9] * 4
    with pytest.raises(RuntimeError):
        ops.map_to_kv_tensors([P8 + MiB])                       # offsets must be page-size multiples
    assert ops.unmap_from_kv_tensors([5 * P8, 2 * P8, 9 * P8])
    assert ops.map_to_kv_tensors([2 * P8])                        # recycled 8 MiB handles read zero again
    assert int(torch.count_nonzero(ts[0][2 * epp:3 * epp])) == 0
    assert ops.unmap_from_kv_tensors([2 * P8])
    assert capi.get_stats()["pages_unmapped"] == 16


def test_contiguous_layout_compound_pages(vmm):
    """One region for all layers; an offset backs page x layers x kv bytes in ONE mapping."""
    layers = 4
    ops, capi, ts = _setup(vmm, layers=layers, per_layer=32 * MiB, contiguous=True)
    assert len(ts) == 1 and ts[0].numel() * 2 == 32 * MiB * layers
    compound = PAGE * layers * 2
    capi.reset_stats()
    assert ops.map_to_kv_tensors([0, 2 * compound])
    st = capi.get_stats()
    assert st["pages_mapped"] == 2 and st["fill_bytes"] == 2 * compound
    e = compound // 2
    assert int(torch.count_nonzero(ts[0][:e])) == 0
    ts[0][:e] = 3
    ts[0][2 * e:3 * e] = 4
    torch.cuda.synchronize()
    assert bool((ts[0][:e] == 3).all()) and bool((ts[0][2 * e:3 * e] == 4).all())
    assert int(torch.count_nonzero(ts[0][e:2 * e])) == 0
    assert ops.unmap_from_kv_tensors([0, 2 * compound])


def test_unified_pool_and_single_buffer_back_one_slot_per_layer(vmm):
    ops, capi, ts = _setup(vmm, layers=3, per_layer=16 * MiB, kv=1, unified=True)
    capi.reset_stats()
    assert ops.map_to_kv_tensors([0, PAGE])
    assert capi.get_stats()["pages_mapped"] == 2 * 3
    assert ops.unmap_from_kv_tensors([0, PAGE])


def test_large_batch_random_order(vmm):
    """512 page ids x 2 layers x K/V = 2048 slots (4 GiB) in one call, shuffled: all zero, all private."""
    import numpy as np
    ops, capi, ts = _setup(vmm, layers=2, per_layer=2048 * MiB)
    offs = [int(i) * PAGE for i in np.random.default_rng(0).permutation(512)]
    capi.reset_stats()
    assert ops.map_to_kv_tensors(offs)
    st = capi.get_stats()
    assert st["pages_mapped"] == 2048 and st["fill_bytes"] == 2048 * PAGE
    for t in ts:
        assert int(torch.count_nonzero(t)) == 0
    # a checksum of checksums: page p of tensor i holds the value 1 + (p mod 250) everywhere
    epp = PAGE // 2
    for t in ts:
        v = t.view(-1, epp)
        v.copy_((torch.arange(v.shape[0], device=DEV) % 250 + 1).to(torch.int16).unsqueeze(1).expand_as(v))
    torch.cuda.synchronize()
    for t in ts:
        v = t.view(-1, epp)
        want = (torch.arange(v.shape[0], device=DEV) % 250 + 1).to(torch.int64) * epp
        assert torch.equal(v.to(torch.int64).sum(dim=1), want)
    assert ops.unmap_from_kv_tensors(offs)
    assert capi.get_stats()["pages_unmapped"] == 2048


def test_full_bench_window_64GiB_size_independent_properties(vmm):
    """BASELINE configs[1] at its full size: a 64 GiB window, all 32 768 x 2 MiB slots backed in shuffled 1024-page
    batches (lazy mode, like the bench). Properties that do not depend on the size: (1) every freshly backed byte reads
    zero; (2) a per-page stamp survives until that page is unmapped and no stamp leaks into another page — checked
    with per-page sums; (3) re-backing half of the slots in a different order (recycled, dirty handles) gives zeros
    again and leaves the other half's stamps intact; (4) the driver-handle ledger balances."""
    import numpy as np
    n_pages = 32768
    ops, capi, ts = _setup(vmm, layers=1, per_layer=n_pages * PAGE, backfill=False, kv=1, unified=True)
    t = ts[0]                                                   # int16 elements
    epp = PAGE // 2
    rng = np.random.default_rng(0)
    order = rng.permutation(n_pages)
    capi.reset_stats()
    for b in range(0, n_pages, 1024):
        assert ops.map_to_kv_tensors([int(p) * PAGE for p in order[b:b + 1024]])
    st = capi.get_stats()
    assert st["pages_mapped"] == n_pages and st["fill_bytes"] == n_pages * PAGE
    v = t.view(n_pages, epp)
    assert all(int(torch.count_nonzero(v[i:i + 1024])) == 0 for i in range(0, n_pages, 1024))   # (1), in 2 GiB pieces
    stamp = (torch.arange(n_pages, device=DEV) % 251 + 1).to(torch.int16)
    v.copy_(stamp.unsqueeze(1).expand_as(v))
    torch.cuda.synchronize()

    def page_sums():
        return torch.cat([v[i:i + 1024].sum(dim=1, dtype=torch.int32) for i in range(0, n_pages, 1024)])
    assert torch.equal(page_sums(), stamp.to(torch.int32) * epp)                              # (2)
    victims = rng.permutation(n_pages)[: n_pages // 2]
    for b in range(0, len(victims), 1024):
        assert ops.unmap_from_kv_tensors([int(p) * PAGE for p in victims[b:b + 1024]])
    back = rng.permutation(victims)
    for b in range(0, len(back), 1024):
        assert ops.map_to_kv_tensors([int(p) * PAGE for p in back[b:b + 1024]])
    want = stamp.to(torch.int32) * epp
    want[torch.as_tensor(victims, device=DEV)] = 0
    assert torch.equal(page_sums(), want)                                                     # (3)
    for b in range(0, n_pages, 4096):
        assert ops.unmap_from_kv_tensors([int(p) * PAGE for p in range(b, b + 4096)])
    st = capi.get_stats()
    assert st["pages_mapped"] == n_pages + len(victims) == st["pages_unmapped"]
    assert st["handles_created"] + st["handles_reused"] == st["pages_mapped"]                 # (4)
    assert st["handles_reused"] >= len(victims) // 2


def test_mem_get_info_and_avail_physical_pages(vmm):
    ops, capi, ts = _setup(vmm)
    free, total = capi.mem_get_info()
    f2, t2 = torch.cuda.mem_get_info(0)
    assert total == t2 and abs(free - f2) < 2048 * MiB
    pa = ops.PageAllocator(2, 64 * MiB // 2, PAGE, 1, 0, False, False, False, 2, 0,
                           os.environ["KVCACHED_IPC_NAME"] + "_mi")
    want = int((free - int(total * (1.0 - 0.95))) // PAGE) // 2 // 2
    assert abs(pa.get_avail_physical_pages() - want) <= 64
    del pa


def test_export_import_same_process(vmm):
    """Shared-pool plumbing: slots backed with exportable handles are exported as POSIX fds and
    imported + mapped into a second group's VA; both views see the same bytes."""
    ops, capi = vmm["ops"], vmm["capi"]
    os.environ["KVCACHED_EXPORTABLE_HANDLES"] = "1"
    try:
        ops.init_kvcached(DEV, PAGE, False)
        a = ops.create_kv_tensors(16 * MiB, 2, DEV, 1, 2, 0, False)
        b = ops.create_kv_tensors(16 * MiB, 2, DEV, 1, 2, 1, False)     # group 1 = the "peer"
    finally:
        os.environ.pop("KVCACHED_EXPORTABLE_HANDLES", None)
    assert ops.map_to_kv_tensors([PAGE], 0)
    fds = capi.export_mapped_slots([PAGE], 0)
    assert len(fds) == 2 and all(fd > 2 for fd in fds)
    capi.map_imported_slots([PAGE], fds, 1)
    for fd in fds:
        os.close(fd)
    epp = PAGE // 2
    a[0][epp:epp + 16] = 4321
    torch.cuda.synchronize()
    assert bool((b[0][epp:epp + 16] == 4321).all())
    assert ops.unmap_from_kv_tensors([PAGE], 1)
    assert ops.unmap_from_kv_tensors([PAGE], 0)


@pytest.mark.hooks_build
@pytest.mark.parametrize("mode", ["lazy", "compat"])
def test_import_falls_back_to_rocr_when_the_direct_import_is_refused(vmm, monkeypatch, mode):
    """A peer's page is taken straight into KFD + DRM (AMDKFD_IOC_IMPORT_DMABUF, one GEM_VA map). A buffer that lives on
    ANOTHER GPU may be refused by that shortcut; ROCr's import - which sets up peer access for the local agent - is the
    fallback. One GPU cannot show the cross-GPU case itself: a hook makes the direct import fail, and the
    page must arrive through ROCr all the same, be shared byte for byte, and leave no imported handle behind."""
    ops, capi = vmm["ops"], vmm["capi"]
    monkeypatch.setenv("KVCACHED_EXPORTABLE_HANDLES", "1")
    monkeypatch.setenv("KVCACHED_TEST_FAIL_KFD_IMPORT", "1")
    monkeypatch.setenv("KVCACHED_ZERO_BACKFILL", "true" if mode == "compat" else "false")
    ops.init_kvcached(DEV, PAGE, False)
    a = ops.create_kv_tensors(16 * MiB, 2, DEV, 1, 2, 0, False)
    b = ops.create_kv_tensors(16 * MiB, 2, DEV, 1, 2, 1, False)         # group 1 = the "peer"
    epp = PAGE // 2
    for rnd in range(3):
        assert ops.map_to_kv_tensors([PAGE, 3 * PAGE], 0)
        fds = capi.export_mapped_slots([PAGE, 3 * PAGE], 0)
        assert len(fds) == 4
        capi.map_imported_slots([PAGE, 3 * PAGE], fds, 1)
        for fd in fds:
            os.close(fd)
        a[0][epp:epp + 16] = 4321 + rnd
        a[0][3 * epp:3 * epp + 16] = 77 + rnd
        torch.cuda.synchronize()
        assert bool((b[0][epp:epp + 16] == 4321 + rnd).all()) and bool((b[0][3 * epp:3 * epp + 16] == 77 + rnd).all())
        assert ops.unmap_from_kv_tensors([PAGE, 3 * PAGE], 1)              # the imports are dropped at once
        if mode == "compat":
            assert int(torch.count_nonzero(b[0][epp:2 * epp])) == 0           # zeros behind the unmap
        assert ops.unmap_from_kv_tensors([PAGE, 3 * PAGE], 0)
    # an import into a slot that is already backed is skipped like a double map - and its handle must not stay imported
    assert ops.map_to_kv_tensors([PAGE], 0) and ops.map_to_kv_tensors([PAGE], 1)
    fds = capi.export_mapped_slots([PAGE], 0)
    capi.map_imported_slots([PAGE], fds, 1)                               # logged "already mapped", nothing consumed
    for fd in fds:
        os.close(fd)
    b[0][epp:epp + 8] = 5
    a[0][epp:epp + 8] = 6
    torch.cuda.synchronize()
    assert bool((b[0][epp:epp + 8] == 5).all())                           # group 1 kept its own page
    assert ops.unmap_from_kv_tensors([PAGE], 1) and ops.unmap_from_kv_tensors([PAGE], 0)
    ops.shutdown_kvcached()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs: a page of cuda:0 mapped by a process on cuda:1 over xGMI "
                    "(BASELINE configs[3]); never ran on hardware - the driver's 8-GPU node is the only place it can")
def test_a_peer_on_another_gpu_maps_rank0_pages():
    """Cross-GPU shared pool: rank 0 (cuda:0) backs and exports, a peer process on cuda:1 imports and maps."""
    import multiprocessing as mp
    import subprocess
    import sys
    child = r"""
import os, sys, socket, array
sys.path.insert(0, %r)
import torch
torch.cuda.set_device(1)
from kvcached_amd import capi, vmm_ops
PAGE = 2 << 20
vmm_ops.init_kvcached("cuda:1", PAGE, False)
ts = vmm_ops.create_kv_tensors(16 << 20, 2, "cuda:1", 1, 2, 0, False)
s = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM); s.connect(sys.argv[1])
msg, anc, _, _ = s.recvmsg(16, socket.CMSG_LEN(8 * 4))
fds = array.array("i"); fds.frombytes(anc[0][2][:8])
capi.map_imported_slots([PAGE], list(fds), 0)
epp = PAGE // 2
print("PEER SEES", int(ts[0][epp + 7]), int(ts[0][ts[0].numel() // 2 + epp + 7]), flush=True)
ts[0][epp + 9] = 777
torch.cuda.synchronize()
s.sendall(b"k")
s.recv(1)
vmm_ops.unmap_from_kv_tensors([PAGE]); vmm_ops.shutdown_kvcached()
"""
    import socket, array, tempfile
    from kvcached_amd import capi, vmm_ops
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    os.environ["KVCACHED_EXPORTABLE_HANDLES"] = "1"
    try:
        vmm_ops.init_kvcached(DEV, PAGE, False)
        ts = vmm_ops.create_kv_tensors(16 * MiB, 2, DEV, 1, 2, 0, False)
        assert vmm_ops.map_to_kv_tensors([PAGE])
        epp = PAGE // 2
        ts[0][epp + 7] = 1001
        ts[0][ts[0].numel() // 2 + epp + 7] = 1002
        torch.cuda.synchronize()
        fds = capi.export_mapped_slots([PAGE], 0)
        path = os.path.join(tempfile.mkdtemp(), "x.sock")
        srv = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
        srv.bind(path)
        srv.listen()
        p = subprocess.Popen([sys.executable, "-c", child % repo, path], stdout=subprocess.PIPE, text=True)
        conn, _ = srv.accept()
        conn.sendmsg([b"fds"], [(socket.SOL_SOCKET, socket.SCM_RIGHTS, array.array("i", fds))])
        assert conn.recv(1) == b"k"
        torch.cuda.synchronize()
        assert int(ts[0][epp + 9]) == 777                                 # the peer's write over xGMI landed in rank 0's page
        conn.sendall(b"k")
        out, _ = p.communicate(timeout=120)
        assert "PEER SEES 1001 1002" in out, out
        for fd in fds:
            os.close(fd)
        assert vmm_ops.unmap_from_kv_tensors([PAGE])
    finally:
        os.environ.pop("KVCACHED_EXPORTABLE_HANDLES", None)
        vmm_ops.shutdown_kvcached()


# ------------------------------------------------------------------ shared pool across processes
def _peer_process(sock_path, ipc, q):
    """The 'peer TP rank': owns its own VA reservation, receives fds over the Unix socket and maps them."""
    try:
        os.environ.update(KVCACHED_IPC_NAME=ipc, KVCACHED_LOG_LEVEL="ERROR")
        import sys
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import torch
        torch.cuda.set_device(0)
        from kvcached_amd import vmm_ops
        from kvcached_amd import tp_ipc_util as tp
        vmm_ops.init_kvcached(DEV, PAGE, False)
        ts = vmm_ops.create_kv_tensors(16 * MiB, 2, DEV, 2, 2, 0, False)
        srv = tp.start_worker_listener_thread(1)
        q.put("ready")
        assert q.get(timeout=120) == "mapped"          # rank 0 has shared its slots with us
        epp = PAGE // 2
        vals = [int(t[half + epp + 7]) for t in ts for half in (0, t.numel() // 2)]
        ts[1][epp + 9] = 777                             # write back through the shared page
        torch.cuda.synchronize()
        q.put(("peer_sees", vals))
        assert q.get(timeout=120) == "done"
        vmm_ops.unmap_from_kv_tensors([PAGE])
        vmm_ops.shutdown_kvcached()
        q.put("bye")
    except Exception as e:
        q.put(("ERR", repr(e)))


@pytest.mark.parametrize("units", ["slots", "page_ids"])
def test_shared_pool_between_two_processes(vmm, units):
    """north-star TP sharing: rank 0 backs a page id, exports it - slot by slot (exportable handles, one POSIX fd per slot:
    hipMemExportToShareableHandle / AMDKFD_IOC_EXPORT_DMABUF) or, where page ids are backed as lanes, ONE dmabuf for the buffer the
    page id lives in - and ships the descriptors with SCM_RIGHTS over the worker's Unix socket; the peer imports and maps at the
    same offsets. Both see the same bytes."""
    import multiprocessing as mp
    ops, capi = vmm["ops"], vmm["capi"]
    ipc = os.environ["KVCACHED_IPC_NAME"]
    if units == "slots":
        os.environ["KVCACHED_EXPORTABLE_HANDLES"] = "1"
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    from kvcached_amd import tp_ipc_util as tp
    peer = ctx.Process(target=_peer_process, args=(tp.get_worker_socket_path(1), ipc, q))
    try:
        peer.start()
        ops.init_kvcached(DEV, PAGE, False)
        ts = ops.create_kv_tensors(16 * MiB, 2, DEV, 2, 2, 0, False)
        assert (capi.get_option(129) > 0) == (units == "page_ids")
        msg = q.get(timeout=180)
        assert msg == "ready", msg
        i0 = capi.get_option(163)
        assert ops.map_to_kv_tensors([PAGE])
        epp = PAGE // 2
        k = 0
        for t in ts:
            for half in (0, t.numel() // 2):
                k += 1
                t[half + epp + 7] = 1000 + k
        torch.cuda.synchronize()
        tp.share_mapped_slots(2, [PAGE], pp_rank=0, group_id=0, src_rank=0)   # export + SCM_RIGHTS + peer maps
        q.put("mapped")
        msg = q.get(timeout=120)
        assert msg == ("peer_sees", [1001, 1002, 1003, 1004]), msg
        torch.cuda.synchronize()
        assert int(ts[1][epp + 9]) == 777
        q.put("done")
        assert q.get(timeout=120) == "bye"
        assert ops.unmap_from_kv_tensors([PAGE])
    finally:
        os.environ.pop("KVCACHED_EXPORTABLE_HANDLES", None)
        tp._channels.close()
        peer.join(timeout=60)
        if peer.is_alive():
            peer.kill()
