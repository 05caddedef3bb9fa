"""The product's host logic (C++ PageAllocator/InternalPage through the C ABI and the pybind11
`vmm_ops` layer, Python KVCacheManager on top) replayed against the golden vectors of the REAL
reference, on the explicit "cpu" device (the reference's own host device path: bookkeeping only).
The same traces run on the GPU in tests/test_gpu_manager.py."""
import ctypes
import json
import os

import pytest

import kvc_testlib as T
import kvc_traces

G = T.GOLDEN_DIR


def load(name):
    with open(os.path.join(G, name)) as f:
        return json.load(f)


@pytest.fixture(scope="module")
def ops():
    from kvcached_amd import vmm_ops
    return vmm_ops


def test_block_range_table(ops):
    from kvcached_amd import capi
    for pid, P, B, start, end, nb in load("block_range.json")["rows"]:
        assert tuple(ops.InternalPage.get_block_range(pid, P, B)) == (start, end)
        assert ops.InternalPage.get_num_blocks(P, B) == nb
        s, e = ctypes.c_int64(), ctypes.c_int64()
        capi.lib.kvc_page_get_block_range(pid, P, B, ctypes.byref(s), ctypes.byref(e))
        assert (s.value, e.value) == (start, end)


def test_internal_page_sequences(ops):
    for c in load("internal_page.json")["cases"]:
        p = ops.InternalPage(c["page_id"], c["page_size"])
        assert (p.page_id, p.page_size) == (c["page_id"], c["page_size"])
        for step in c["steps"]:
            kind = step[0]
            if kind == "init":
                p.init(c["block_mem_size"])
                assert p.get_free_blocks() == step[1] and p.empty() == step[2] and p.full() == step[3]
            elif kind == "alloc":
                assert p.alloc(step[1]) == step[2] and p.get_free_blocks() == step[3]
            elif kind == "free":
                p.free(step[1])
                assert p.get_free_blocks() == step[2]
            elif kind == "alloc_all":
                assert p.alloc(p.num_free_blocks()) == step[1] and p.full() == step[2] and p.empty() == step[3]
            elif kind == "free_batch_reversed":
                p.free_batch(step[1])
                assert p.get_free_blocks() == step[1] and p.empty() == step[2]
            elif kind == "over_alloc":
                with pytest.raises(RuntimeError, match=step[1]):
                    p.alloc(p.num_free_blocks() + 1)


def test_group_indices_iteration_order(ops):
    d = load("group_indices.json")
    ops.init_kvcached("cpu", d["page_size"], False)
    try:
        for c in d["cases"]:
            idx = kvc_traces.shuffled_indices(c["num_blocks"], c["n"], c["seed"])
            assert T.h64(idx) == c["indices_sha"]
            pa = ops.PageAllocator(2, c["num_blocks"] * c["block_mem_size"], d["page_size"], 1, 0, False, False, False,
                                   2, 0, os.environ["KVCACHED_IPC_NAME"] + "_g")
            got = pa.group_indices_by_page(idx, c["block_mem_size"])
            assert list(got.keys()) == c["keys"]            # dict order == reference's unordered_map order
            assert [len(v) for v in got.values()] == c["counts"]
            assert T.h64([v for vs in got.values() for v in vs]) == c["values_sha"]
            del pa
    finally:
        ops.shutdown_kvcached()


def test_page_allocator_state_machine(ops):
    import numpy as np
    d = load("page_allocator.json")
    P = d["page_size"]
    ops.init_kvcached("cpu", P, False)
    try:
        for case in d["cases"]:
            cfg = case["config"]
            ipc = os.environ["KVCACHED_IPC_NAME"] + "_pa"
            events = []
            pa = ops.PageAllocator(cfg["num_layers"], cfg["pages"] * P, P, 1, 0, False, cfg["contiguous"], False,
                                   cfg["num_kv_buffers"], 0, ipc)
            pa.set_should_use_worker_ipc_callback(lambda: True)
            pa.set_broadcast_map_callback(lambda ws, offs: events.append([0, list(offs)]))
            pa.set_broadcast_unmap_callback(lambda ws, offs: events.append([1, list(offs)]))
            for i, (op, want) in enumerate(zip(case["ops"], case["records"])):
                r = None
                try:
                    if op[0] == "alloc":
                        r = pa.alloc_page().page_id
                    elif op[0] == "free":
                        pa.free_page(op[1])
                    elif op[0] == "frees":
                        pa.free_pages(op[1])
                    elif op[0] == "resize":
                        r = pa.resize(op[1] * P)
                    elif op[0] == "trim":
                        pa.trim()
                    elif op[0] == "reset":
                        pa.reset_free_page_order()
                    elif op[0] == "target":
                        r = pa.check_and_get_resize_target(op[1] * P)
                except RuntimeError as e:
                    r = "RuntimeError: " + str(e)
                shm = [int(x) for x in np.fromfile("/dev/shm/" + ipc, dtype=np.int64)[:3]]
                got = {"r": r, "s": [pa.get_num_free_pages(), pa.get_num_inuse_pages(), pa.get_num_total_pages(),
                                     pa.get_num_reserved_pages()] + shm, "e": events[:]}
                events.clear()
                assert got == want, f"{cfg['name']} op {i} {op}: {got} != {want}"
            del pa
            assert not os.path.exists("/dev/shm/" + ipc), "shm segment must be unlinked with the allocator"
    finally:
        ops.shutdown_kvcached()


def _adapter(cfg, **kw):
    return T.ProductAdapter(cfg["num_blocks"], cfg["block_size"], cfg["cell_size"], cfg["num_layers"],
                            world_size=cfg["world_size"], reserve_null_block=cfg["reserve_null_block"],
                            num_kv_buffers=cfg["num_kv_buffers"], contiguous=cfg["contiguous"],
                            phys_pages=cfg["phys_pages"], **kw)


@pytest.mark.parametrize("idx", range(11))
def test_manager_small_traces(idx):
    case = load("manager_small.json")["cases"][idx]
    ad = _adapter(case["config"])
    try:
        init = {"s": ad.snapshot(), "e": ad.drain_events()}
        assert init == case["init"], case["name"]
        got = T.replay(ad, case["ops"], full=True)
        for i, (g, want) in enumerate(zip(got, case["records"])):
            assert g == want, f"{case['name']} op {i} {case['ops'][i]}"
    finally:
        ad.close()


@pytest.mark.parametrize("idx", range(5))
def test_manager_large_traces(idx):
    case = load("manager_large.json")["cases"][idx]
    ad = _adapter(case["config"])
    try:
        init = {"s": ad.snapshot(), "e": ad.drain_events()}
        assert init == case["init"], case["name"]
        got = T.replay(ad, case["ops"], full=False)
        for i, (g, want) in enumerate(zip(got[:40], case["records_head"])):
            assert g == want, f"{case['name']} op {i} {case['ops'][i]}"
        chain = T.chain_hash(got)
        assert chain["checkpoints"] == case["chain"]["checkpoints"]
        assert chain["final"] == case["chain"]["final"]
        assert got[-5:] == case["records_tail"]
    finally:
        ad.close()


# ------------------------------------------------------------------ KVCACHED_BATCH_PAGE_ALLOC (the product default)
def _merged(records):
    return [{"r": r["r"], "s": r["s"], "e": T.merge_events(r["e"])} for r in records]


@pytest.mark.parametrize("idx", range(11))
def test_small_traces_with_batched_page_alloc(idx):
    """An alloc() that needs several new pages backs them with ONE map call. Against the reference's recorded traces
    everything is identical - results, every counter, the shm triple, the offsets and their order - except the call
    boundaries: the golden's consecutive map calls of one alloc() arrive folded into one."""
    case = load("manager_small.json")["cases"][idx]
    ad = _adapter(case["config"], batch_page_alloc=True)
    try:
        init = {"s": ad.snapshot(), "e": ad.drain_events()}
        assert init == case["init"], case["name"]
        got = T.replay(ad, case["ops"], full=True)
        for i, (g, want) in enumerate(zip(_merged(got), _merged(case["records"]))):
            assert g == want, f"{case['name']} op {i} {case['ops'][i]}"
    finally:
        ad.close()


@pytest.mark.parametrize("idx", range(5))
def test_large_traces_with_batched_page_alloc(idx):
    """The large goldens only hold hashes, so the batched run is compared with the page-by-page run of the product
    (which test_manager_large_traces pins to the reference): equal after folding consecutive map calls."""
    case = load("manager_large.json")["cases"][idx]
    runs = []
    for batch in (False, True):
        ad = _adapter(case["config"], batch_page_alloc=batch)
        try:
            ad.drain_events()
            runs.append(T.replay(ad, case["ops"], full=True))
        finally:
            ad.close()
    assert _merged(runs[0]) == _merged(runs[1])
    calls = [sum(len(r["e"]) for r in run) for run in runs]
    assert calls[1] <= calls[0]
    if any(len([e for e in r["e"] if e[0] == 0]) > 1 for r in runs[0]):
        assert calls[1] < calls[0]                               # some alloc needed several pages: fewer calls now
