#!/usr/bin/env python3
"""oracle/gen_golden.py — produce tests/golden/*.json by running the REAL reference.

Runs only in the build container (it needs /root/reference). What it executes:
  * the reference's native code, compiled by oracle/Makefile from the sources where they lie
    into oracle/_ref/vmm_ops.so (nothing of the reference is copied into this repo);
  * the reference's own Python (kvcached/kv_cache_manager.py, kvcached/integration/*/interfaces.py)
    imported from /root/reference, with `kvcached.vmm_ops` resolved to that .so.
Both run on the reference's own `cpu` device path (csrc/ftensor.cpp:40-44, csrc/page.cpp:28-37).
The one thing that needs a GPU there — PageAllocator.get_avail_physical_pages (hipMemGetInfo) —
is overridden in a Python subclass so that the free-memory reading is an explicit trace input.
Prealloc is disabled (KVCACHED_PAGE_PREALLOC_ENABLED=false): with the thread running the
reference's traces are timing-dependent.

Only DATA is written: inputs (trace ops, shapes) and the outputs the reference produced.
Usage:  python oracle/gen_golden.py            (rewrites every fixture)
"""
from __future__ import annotations

import importlib
import importlib.machinery
import importlib.util
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("KVC_REFERENCE", "/root/reference")
REF_SO = os.path.join(REPO, "oracle", "_ref", "vmm_ops.so")
OUT = os.path.join(REPO, "tests", "golden")

os.environ["KVCACHED_PAGE_PREALLOC_ENABLED"] = "false"
os.environ["KVCACHED_IPC_NAME"] = f"kvc_golden_{os.getpid()}"
os.environ["KVCACHED_LOG_LEVEL"] = "ERROR"
os.environ.setdefault("KVCACHED_CONTIGUOUS_LAYOUT", "false")

sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import kvc_testlib as T  # noqa: E402
import kvc_traces  # noqa: E402

# ---- load the reference: its Python from where it lies, its extension from oracle/_ref
sys.path.insert(0, REF)
import kvcached  # noqa: E402  (the reference package)

assert os.path.realpath(kvcached.__file__).startswith(os.path.realpath(REF)), kvcached.__file__
_loader = importlib.machinery.ExtensionFileLoader("kvcached.vmm_ops", REF_SO)
_spec = importlib.util.spec_from_loader("kvcached.vmm_ops", _loader)
ref_ops = importlib.util.module_from_spec(_spec)
_loader.exec_module(ref_ops)
sys.modules["kvcached.vmm_ops"] = ref_ops
kvcached.vmm_ops = ref_ops
import kvcached.kv_cache_manager as ref_kcm  # noqa: E402
import kvcached.utils as ref_utils  # noqa: E402

PAGE = 2 << 20
_shm = "/dev/shm/" + os.environ["KVCACHED_IPC_NAME"]


class _RefPA(ref_ops.PageAllocator):
    """Reference PageAllocator with the free-memory reading made an input."""
    phys_pages = 1 << 40

    def get_avail_physical_pages(self):
        return self.phys_pages


ref_kcm.PageAllocator = _RefPA


def _shm_triple():
    try:
        return [int(x) for x in np.fromfile(_shm, dtype=np.int64)[:3]]
    except Exception:
        return [0, 0, 0]


class RefAdapter(T.Adapter):
    """Drives the reference KVCacheManager; map/unmap requests are captured through the
    reference's own broadcast-callback hook instead of being executed."""

    def __init__(self, num_blocks, block_size, cell_size, num_layers, world_size=1, reserve_null_block=False,
                 num_kv_buffers=2, contiguous=False, phys_pages=1 << 40, group_id=0):
        ref_kcm.CONTIGUOUS_LAYOUT = contiguous
        ref_ops.init_kvcached("cpu", PAGE, contiguous)
        _RefPA.phys_pages = phys_pages
        self.events = []
        self.m = ref_kcm.KVCacheManager(num_blocks, block_size, cell_size, num_layers, world_size=world_size,
                                        reserve_null_block=reserve_null_block, num_kv_buffers=num_kv_buffers,
                                        group_id=group_id)
        pa = self.m.page_allocator
        pa.set_should_use_worker_ipc_callback(lambda: True)
        pa.set_broadcast_map_callback(lambda ws, offs: self.events.append([0, [int(o) for o in offs]]))
        pa.set_broadcast_unmap_callback(lambda ws, offs: self.events.append([1, [int(o) for o in offs]]))
        # only now let _post_init proceed (it polls kv_tensors_created): its null-block alloc must
        # already see the capture callbacks
        mem = num_blocks * block_size * cell_size
        mem = (mem + 2 * PAGE - 1) // (2 * PAGE) * (2 * PAGE)
        ref_ops.create_kv_tensors(mem * num_kv_buffers, 1, "cpu", num_layers, num_kv_buffers, group_id, False)
        assert self.m._post_init_done.wait(20), "reference _post_init did not finish"

    def alloc(self, n): return self.m.alloc(n)
    def free(self, ids): self.m.free(ids)
    def try_to_reserve(self, n): return self.m.try_to_reserve(n)
    def free_reserved(self): self.m.free_reserved()
    def resize(self, mem): return self.m.resize(mem)
    def trim(self): self.m.trim()
    def set_phys(self, pages): self.m.page_allocator.phys_pages = pages
    def available_size(self): return self.m.available_size()

    def snapshot(self):
        pa = self.m.page_allocator
        t, u, p = _shm_triple()
        return [self.m.available_size(), pa.get_num_free_pages(), pa.get_num_inuse_pages(), pa.get_num_total_pages(),
                pa.get_num_reserved_pages(), len(self.m.reserved_blocks), int(self.m.in_shrink), t, u, p]

    def drain_events(self):
        ev, self.events = self.events, []
        return ev

    def close(self):
        self.m = None
        ref_ops.shutdown_kvcached()


def dump(name, obj):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name)
    with open(path, "w") as f:
        json.dump(obj, f, separators=(",", ":"))
        f.write("\n")
    print(f"  wrote {name}  ({os.path.getsize(path)} bytes)")


META = {"generator": "oracle/gen_golden.py", "reference": "ztang2370/kvcached v0.1.5 @ /root/reference",
        "prealloc": False, "torch": torch.__version__}


# ------------------------------------------------------------------ 1. static integer functions
def gen_block_range():
    IP = ref_ops.InternalPage
    rows = []
    for P in (2 << 20, 4 << 20, 6 << 20):
        for B in (16 << 10, 32 << 10, 48 << 10, 768 << 10, 1536 << 10, 1 << 20, 2 << 20, 3 << 19, 5000, 4 << 20):
            for pid in (0, 1, 2, 3, 7, 100, 4095):
                s, e = IP.get_block_range(pid, P, B)
                rows.append([pid, P, B, int(s), int(e), int(IP.get_num_blocks(P, B))])
    dump("block_range.json", {"meta": META, "columns": ["page_id", "page_size", "block_mem_size", "start", "end",
                                                        "num_blocks"], "rows": rows})


def gen_internal_page():
    IP = ref_ops.InternalPage
    cases = []
    for (pid, P, B) in ((0, PAGE, 32 << 10), (3, PAGE, 16 << 10), (1, PAGE, 768 << 10), (5, 4 << 20, 48 << 10)):
        p = IP(pid, P)
        p.init(B)
        steps = [["init", list(p.get_free_blocks()), p.empty(), p.full()]]
        n0 = p.num_free_blocks()
        a = p.alloc(min(3, n0))
        steps.append(["alloc", min(3, n0), list(a), list(p.get_free_blocks())])
        if a:
            p.free(a[0])
            steps.append(["free", a[0], list(p.get_free_blocks())])
        rest = p.alloc(p.num_free_blocks())
        steps.append(["alloc_all", list(rest), p.full(), p.empty()])
        p.free_batch(list(reversed(rest)))
        steps.append(["free_batch_reversed", list(p.get_free_blocks()), p.empty()])
        try:
            p.alloc(p.num_free_blocks() + 1)
            steps.append(["over_alloc", "no error"])
        except RuntimeError as e:
            steps.append(["over_alloc", str(e)])
        cases.append({"page_id": pid, "page_size": P, "block_mem_size": B, "steps": steps})
    dump("internal_page.json", {"meta": META, "cases": cases})


def gen_group_indices():
    ref_ops.init_kvcached("cpu", PAGE, False)
    cases = []
    for (B, nblocks, n, seed) in ((32 << 10, 4096, 64, 0), (32 << 10, 147456, 1024, 1), (16 << 10, 65536, 16384, 2),
                                  (768 << 10, 4096, 200, 3), (32 << 10, 4096, 1, 4), (2 << 20, 512, 300, 5)):
        pa = ref_ops.PageAllocator(2, nblocks * B, PAGE, 1, 0, False, False, False, 2, 0,
                                   os.environ["KVCACHED_IPC_NAME"] + "_g")
        idx = kvc_traces.shuffled_indices(nblocks, n, seed)
        d = pa.group_indices_by_page(idx, B)
        keys = [int(k) for k in d.keys()]
        flat = [int(v) for k in d for v in d[k]]
        cases.append({"block_mem_size": B, "num_blocks": nblocks, "n": n, "seed": seed, "indices_sha": T.h64(idx),
                      "keys": keys, "counts": [len(d[k]) for k in d], "values_sha": T.h64(flat)})
        del pa
    ref_ops.shutdown_kvcached()
    dump("group_indices.json", {"meta": META, "page_size": PAGE, "cases": cases})


# ------------------------------------------------------------------ 2. PageAllocator state machine
def gen_page_allocator():
    ref_ops.init_kvcached("cpu", PAGE, False)
    cases = []
    for cfg in kvc_traces.PAGE_ALLOCATOR_CASES:
        events = []
        pa = ref_ops.PageAllocator(cfg["num_layers"], cfg["pages"] * PAGE, PAGE, 1, 0, False, cfg["contiguous"], False,
                                   cfg["num_kv_buffers"], 0, os.environ["KVCACHED_IPC_NAME"] + "_pa")
        pa.set_should_use_worker_ipc_callback(lambda: True)
        pa.set_broadcast_map_callback(lambda ws, offs: events.append([0, [int(o) for o in offs]]))
        pa.set_broadcast_unmap_callback(lambda ws, offs: events.append([1, [int(o) for o in offs]]))
        recs = []
        for op in cfg["ops"]:
            r = None
            try:
                if op[0] == "alloc":
                    r = int(pa.alloc_page().page_id)
                elif op[0] == "free":
                    pa.free_page(op[1])
                elif op[0] == "frees":
                    pa.free_pages(op[1])
                elif op[0] == "resize":
                    r = bool(pa.resize(op[1] * PAGE))
                elif op[0] == "trim":
                    pa.trim()
                elif op[0] == "reset":
                    pa.reset_free_page_order()
                elif op[0] == "target":
                    r = int(pa.check_and_get_resize_target(op[1] * PAGE))
                else:
                    raise ValueError(op)
            except RuntimeError as e:
                r = "RuntimeError: " + str(e)
            shm = [int(x) for x in np.fromfile("/dev/shm/" + os.environ["KVCACHED_IPC_NAME"] + "_pa", dtype=np.int64)[:3]]
            recs.append({"r": r, "s": [pa.get_num_free_pages(), pa.get_num_inuse_pages(), pa.get_num_total_pages(),
                                       pa.get_num_reserved_pages()] + shm, "e": events[:]})
            events.clear()
        cases.append({"config": {k: v for k, v in cfg.items() if k != "ops"}, "ops": cfg["ops"], "records": recs})
        del pa
    ref_ops.shutdown_kvcached()
    dump("page_allocator.json", {"meta": META, "page_size": PAGE, "cases": cases})


# ------------------------------------------------------------------ 3. KVCacheManager traces
def run_manager_case(case):
    ad = RefAdapter(**case["config"])
    try:
        init = {"s": ad.snapshot(), "e": ad.drain_events()}
        recs = T.replay(ad, case["ops"], full=case.get("full", False))
    finally:
        ad.close()
    out = {"name": case["name"], "config": case["config"], "full": case.get("full", False), "init": init,
           "ops": case["ops"]}
    if case.get("full", False):
        out["records"] = recs
    else:
        out["chain"] = T.chain_hash(recs)
        out["records_head"] = recs[:40]
        out["records_tail"] = recs[-5:]
    return out


def gen_manager():
    small = [run_manager_case(c) for c in kvc_traces.manager_cases_small()]
    dump("manager_small.json", {"meta": META, "snapshot_columns": ["available_size", "free_pages", "inuse_pages",
                                                                   "total_pages", "reserved_pages", "n_reserved_blocks",
                                                                   "in_shrink", "shm_total", "shm_used", "shm_prealloc"],
                                "cases": small})
    big = [run_manager_case(c) for c in kvc_traces.manager_cases_large()]
    dump("manager_large.json", {"meta": META, "cases": big})


# ------------------------------------------------------------------ 4. integration layouts
class _Props:
    def __init__(self, total):
        self.total_memory = total


def _tensor_desc(t, base):
    return {"shape": list(t.shape), "stride": list(t.stride()), "dtype": str(t.dtype),
            "offset_bytes": int(t.data_ptr() - base), "storage_offset": int(t.storage_offset())}


def gen_layouts():
    import kvcached.integration.sglang.interfaces as ref_sg
    import kvcached.integration.vllm.interfaces as ref_vl
    cases = []
    orig_props, orig_avail = torch.cuda.get_device_properties, torch.cuda.is_available
    torch.cuda.is_available = lambda: True
    try:
        for c in kvc_traces.LAYOUT_CASES:
            torch.cuda.get_device_properties = lambda dev=None, _t=c["gpu_bytes"]: _Props(_t)
            contiguous = c["contiguous"]
            ref_ops.init_kvcached("cpu", PAGE, contiguous)
            dtype = getattr(torch, c["dtype"])
            captured = {}
            real_create = ref_ops.create_kv_tensors

            def spy(size, dtype_size, dev, num_layers, num_kv_buffers=2, group_id=0, unified_pool=False):
                captured.update(size=int(size), dtype_size=int(dtype_size), num_layers=int(num_layers),
                                num_kv_buffers=int(num_kv_buffers), unified_pool=bool(unified_pool))
                ts = real_create(size, dtype_size, dev, num_layers, num_kv_buffers, group_id, unified_pool)
                captured["bases"] = [int(t.data_ptr()) for t in ts]
                return ts

            mod = ref_vl if c["engine"] == "vllm" else ref_sg
            mod._kvcached_initialized = True
            mod._contiguous_layout = contiguous
            mod.create_kv_tensors = spy
            try:
                if c["engine"] == "vllm":
                    out = mod.alloc_kv_cache(tuple(c["shape"]), c["block_size"], dtype, "cpu", c["num_layers"],
                                             attention_type=c["attention_type"],
                                             kernel_block_size=c.get("kernel_block_size"))
                else:
                    out = mod.alloc_kv_cache(tuple(c["shape"]), dtype, "cpu", c["num_layers"],
                                             page_size=c["block_size"], attention_type=c["attention_type"])
            finally:
                mod.create_kv_tensors = real_create
                mod._kvcached_initialized = False
            bases = captured.pop("bases")
            rec = {"case": c, "create_kv_tensors": captured}
            extra = None
            if c["attention_type"] == "HYBRID_LINEAR":
                out, extra = out

            def desc_list(tensors):
                res = []
                for i, t in enumerate(tensors):
                    b = bases[0] if contiguous else bases[i]
                    res.append(_tensor_desc(t, b))
                return res

            if isinstance(out, tuple):
                rec["k"] = desc_list(out[0])[:3]
                rec["v"] = desc_list(out[1])[:3]
            else:
                rec["kv"] = desc_list(out)[:3]
            if extra is not None:
                rec["raw_info"] = {k: (v if not isinstance(v, list) else [list(b.shape) for b in v][:2])
                                   for k, v in extra.items()}
            cases.append(rec)
            ref_ops.shutdown_kvcached()
    finally:
        torch.cuda.get_device_properties, torch.cuda.is_available = orig_props, orig_avail
    dump("alloc_kv_cache_layouts.json", {"meta": META, "page_size": PAGE, "cases": cases})


# ------------------------------------------------------------------ 5. ElasticBlockPool (prefix cache layer)
def gen_prefix_cache():
    """Builds the reference's ElasticBlockPool exactly like its own tests/test_prefix_cache.py does (its patch's
    inject method on a fake block_pool module, get_kv_cache_manager patched to hand out a fake manager) and records
    what it does on seeded request traces."""
    import types
    from unittest import mock
    import kvcached.integration.vllm.interfaces  # noqa: F401  (resolvable target for mock.patch)
    from kvcached.integration.vllm.patches import ElasticBlockPoolPatch, _make_cache_key
    cases = []
    for c in kvc_traces.PREFIX_CACHE_CASES:
        manager = T.FifoBlockManager(c["num_blocks"])
        mod = types.ModuleType("fake_block_pool")
        mod.BlockPool, mod.KVCacheBlock = T.FakeBlockPool, T.FakeKVCacheBlock
        with mock.patch("kvcached.integration.vllm.interfaces.get_kv_cache_manager", return_value=manager):
            ElasticBlockPoolPatch().inject_elastic_block_pool(mod)
            pool = mod.ElasticBlockPool(num_gpu_blocks=c["num_blocks"], block_size=16, cell_size=1024, num_layers=1,
                                        enable_caching=c["enable_caching"], max_cached_blocks=c["max_cached_blocks"])
        ops = kvc_traces.prefix_cache_ops(c["n_ops"], c["seed"], c["num_blocks"])
        recs = T.replay_prefix_cache(pool, manager, ops)
        cases.append({"config": c, "null_block": pool.null_block.block_id, "ops": ops, "records": recs})
    keys = [[h, g, _make_cache_key(h, g).hex()] for h, g in (("abc", 0), ("abc", 7), ("", 1))]
    keys += [[h.hex(), g, _make_cache_key(h, g).hex()] for h, g in ((b"\x00\x01", 0), (b"hash", 65536))]
    dump("prefix_cache.json", {"meta": META, "cache_keys": keys, "cases": cases})


def gen_prefix_cache_real_manager():
    """The same request traces through the reference's ElasticBlockPool over the reference's REAL KVCacheManager (on its cpu
    device, like the manager traces above; the free-memory reading is the _RefPA input, far above what the traces need): block
    ids, free counts, cached keys and the eviction order now include what a page-granular manager does below the pool
    (VERDICT r02 #6). The product is compared with this recording op for op, on its cpu device and on cuda:0."""
    import types
    import kvcached.integration.vllm.interfaces as ref_vi
    from kvcached.integration.vllm.patches import ElasticBlockPoolPatch
    g = T.PREFIX_REAL_GEOMETRY
    block_bytes = g["block_tokens"] * g["cell"]
    cases = []
    for c in kvc_traces.PREFIX_CACHE_CASES:
        if c["name"] not in ("default_cap_1000", "cap_5", "tiny_pool_pressure"):
            continue
        n = c["num_blocks"]
        pages = -(-n * block_bytes // PAGE)
        ref_kcm.CONTIGUOUS_LAYOUT = False
        ref_ops.init_kvcached("cpu", PAGE, False)
        _RefPA.phys_pages = 1 << 40
        ref_vi._kvcached_initialized, ref_vi._world_size, ref_vi._pp_rank, ref_vi._async_sched = True, 1, 0, False
        ref_vi._is_worker = True      # (worker and scheduler in one process, as with TP=1: pages are mapped locally, vllm/interfaces.py:29-30,42-48)
        try:
            ref_ops.create_kv_tensors(pages * PAGE * 2, 1, "cpu", g["layers"], 2, 0, False)
            mod = types.ModuleType("fake_block_pool")
            mod.BlockPool, mod.KVCacheBlock = T.FakeBlockPool, T.FakeKVCacheBlock
            ElasticBlockPoolPatch().inject_elastic_block_pool(mod)
            pool = mod.ElasticBlockPool(num_gpu_blocks=n, block_size=g["block_tokens"], cell_size=g["cell"], num_layers=g["layers"],
                                        enable_caching=c["enable_caching"], max_cached_blocks=c["max_cached_blocks"])
            assert isinstance(pool.kv_cache_manager, ref_kcm.KVCacheManager)          # the real one, not a stand-in
            assert pool.kv_cache_manager._post_init_done.wait(20)
            ops = kvc_traces.prefix_cache_ops(c["n_ops"], c["seed"], n)
            recs = T.replay_prefix_over_real_manager(pool, ops)
            pa = pool.kv_cache_manager.page_allocator
            cases.append({"config": c, "geometry": g, "null_block": pool.null_block.block_id, "ops": ops, "records": recs,
                          "pages_at_end": [pa.get_num_inuse_pages(), pa.get_num_reserved_pages(), pa.get_num_free_pages()]})
            del pool
        finally:
            ref_vi._kvcached_initialized = ref_vi._is_worker = False
            ref_ops.shutdown_kvcached()
    dump("prefix_cache_real_manager.json", {"meta": META, "cases": cases})


# ------------------------------------------------------------------ 8. get_avail_physical_pages on fed readings
_AVAIL_CHILD = r"""
import importlib.machinery, importlib.util, json, os, sys
import torch
so, rows_in = sys.argv[1], json.loads(sys.argv[2])
loader = importlib.machinery.ExtensionFileLoader("vmm_ops", so)
spec = importlib.util.spec_from_loader("vmm_ops", loader)
ref = importlib.util.module_from_spec(spec); loader.exec_module(ref)
PAGE = 2 << 20
ref.init_kvcached("cpu", PAGE, False)
out, pas = [], {}
for free, total, L, kv in rows_in:
    if (L, kv) not in pas:
        pas[(L, kv)] = ref.PageAllocator(L, 64 * PAGE, PAGE, 1, 0, False, False, False, kv, 0, os.environ["KVCACHED_IPC_NAME"] + f"_av{L}_{kv}")
    os.environ["KVC_FEED_FREE"], os.environ["KVC_FEED_TOTAL"] = str(free), str(total)
    out.append([free, total, L, kv, int(pas[(L, kv)].get_avail_physical_pages())])
pas.clear()
ref.shutdown_kvcached()
print(json.dumps(out))
"""


def gen_avail_physical():
    """The reference's OWN arithmetic (csrc/page_allocator.cpp:442-455) on (free, total) readings that are inputs: the
    real reference .so in a child process with oracle/_ref/libmemfeed.so preloaded (it answers hipMemGetInfo - and only
    that - with the pair in the environment). Includes readings below the headroom, where the reference's unsigned
    subtraction wraps."""
    import subprocess
    feed = os.path.join(os.path.dirname(REF_SO), "libmemfeed.so")
    GiB = 1 << 30
    total = 288 * GiB
    frees = [0, 1, PAGE, 5 * GiB, int(total * 0.05) - 1, int(total * 0.05), int(total * 0.05) + PAGE - 1, int(total * 0.05) + PAGE,
             int(total * 0.05) + 64 * PAGE, 20 * GiB, 100 * GiB + 12345, 287 * GiB, total]
    rows_in = [[f, total, L, kv] for f in frees for (L, kv) in ((1, 1), (2, 2), (32, 2), (61, 1), (80, 2))]
    rows_in += [[f, 80 * GiB, 32, 2] for f in (0, 3 * GiB, 4 * GiB, 4 * GiB + 64 * PAGE * 64, 79 * GiB)]
    cases = []
    for util in (None, "0.5", "1.0"):
        env = dict(os.environ, LD_PRELOAD=feed)
        env.pop("KVCACHED_GPU_UTILIZATION", None)
        if util is not None:
            env["KVCACHED_GPU_UTILIZATION"] = util
        out = subprocess.run([sys.executable, "-c", _AVAIL_CHILD, REF_SO, json.dumps(rows_in)], env=env, capture_output=True,
                             text=True, timeout=300)
        line = [l for l in out.stdout.splitlines() if l.startswith("[[")]
        assert out.returncode == 0 and line, out.stderr[-2000:]
        cases.append({"gpu_utilization": 0.95 if util is None else float(util), "rows": json.loads(line[-1])})
    dump("avail_physical_pages.json", {"meta": META, "page_size": PAGE, "columns": ["free_bytes", "total_bytes", "num_layers",
                                                                                  "num_kv_buffers", "avail_pages_per_layer"],
                                      "cases": cases})


if __name__ == "__main__":
    t0 = time.time()
    print("reference module:", ref_ops.__file__ if hasattr(ref_ops, "__file__") else REF_SO)
    gen_block_range()
    gen_internal_page()
    gen_group_indices()
    gen_page_allocator()
    gen_manager()
    gen_layouts()
    gen_prefix_cache()
    gen_prefix_cache_real_manager()
    gen_avail_physical()
    print(f"done in {time.time() - t0:.1f}s")
