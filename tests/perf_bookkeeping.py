#!/usr/bin/env python3
"""Bookkeeping micro-benchmark ("µs per alloc(k)+free, available_size(), group_indices_by_page").
Not a pytest file. Times, on the host CPU and on the `cpu` device (no GPU work at all):
  product    kvcached_amd.KVCacheManager over the C ABI
  reference  the REAL reference (oracle/_ref/vmm_ops.so + /root/reference Python) — build container only
  oracle     oracle/libkvc_oracle.so (C++ restatement, no Python in the loop)
Same geometry as BASELINE.md §2: 32 layers, block 16 tok, cell 2048 B, 2 MiB pages, per-layer layout.
    python tests/perf_bookkeeping.py [--prealloc]
"""
import importlib.machinery
import importlib.util
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)
PREALLOC = "--prealloc" in sys.argv
os.environ["KVCACHED_PAGE_PREALLOC_ENABLED"] = "true" if PREALLOC else "false"
os.environ.setdefault("KVCACHED_IPC_NAME", f"kvc_perf_{os.getpid()}")
os.environ["KVCACHED_LOG_LEVEL"] = "ERROR"
os.environ["KVCACHED_CONTIGUOUS_LAYOUT"] = "false"

GEOM = dict(num_blocks=147456, block_size=16, cell_size=2048, num_layers=32)
PAGE = 2 << 20


def timeit(fn, n):
    fn()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    return (time.perf_counter() - t0) / n * 1e6


def bench_manager(m, group_fn):
    out = {}
    for k in (1, 16, 64, 256):
        def cycle():
            b = m.alloc(k)
            m.free(b)
        out[f"alloc({k})+free_us"] = round(timeit(cycle, 3000), 2)
    out["alloc(16)+free_Kops"] = round(1e3 / out["alloc(16)+free_us"], 1)
    out["available_size_us"] = round(timeit(m.available_size, 20000), 3)
    import numpy as np
    idx = [int(x) for x in np.random.default_rng(0).choice(GEOM["num_blocks"], 1024, replace=False)]
    out["group_indices_by_page_N1024_us"] = round(timeit(lambda: group_fn(idx), 3000), 2)
    return out


def product():
    import kvc_testlib as T
    from kvcached_amd import capi, vmm_ops
    import kvcached_amd.kv_cache_manager as kcm
    vmm_ops.init_kvcached("cpu", PAGE, False)
    T.set_product_phys_pages(1 << 30, PAGE, 32, 2)
    mem = GEOM["num_blocks"] * GEOM["block_size"] * GEOM["cell_size"]
    vmm_ops.create_kv_tensors(mem * 2, 1, "cpu", 32, 2, 0, False)
    m = kcm.KVCacheManager(**GEOM)
    assert m._post_init_done.wait(10)
    time.sleep(0.2)
    r = bench_manager(m, lambda idx: m.page_allocator.group_indices_by_page(idx, m.block_mem_size))
    del m
    vmm_ops.shutdown_kvcached()
    capi.set_mem_info_override(0, 0)
    return r


def reference():
    so = os.path.join(REPO, "oracle", "_ref", "vmm_ops.so")
    if not (os.path.exists(so) and os.path.isdir("/root/reference/kvcached")):
        return None
    code = r'''
import sys, os, json, time, importlib.machinery, importlib.util
sys.path.insert(0, %r); sys.path.insert(0, %r); sys.path.insert(0, "/root/reference")
import torch, kvcached
ld = importlib.machinery.ExtensionFileLoader("kvcached.vmm_ops", %r)
mod = importlib.util.module_from_spec(importlib.util.spec_from_loader("kvcached.vmm_ops", ld)); ld.exec_module(mod)
sys.modules["kvcached.vmm_ops"] = mod; kvcached.vmm_ops = mod
import kvcached.kv_cache_manager as kcm
class PA(mod.PageAllocator):
    def get_avail_physical_pages(self): return 1 << 30
kcm.PageAllocator = PA
import perf_bookkeeping as pb
mod.init_kvcached("cpu", pb.PAGE, False)
mem = pb.GEOM["num_blocks"] * pb.GEOM["block_size"] * pb.GEOM["cell_size"]
mod.create_kv_tensors(mem * 2, 1, "cpu", 32, 2, 0, False)
m = kcm.KVCacheManager(**pb.GEOM)
assert m._post_init_done.wait(10); time.sleep(0.2)
print("RESULT" + json.dumps(pb.bench_manager(m, lambda idx: m.page_allocator.group_indices_by_page(idx, m.block_mem_size))), flush=True)
os._exit(0)
''' % (REPO, HERE, so)
    import subprocess
    env = dict(os.environ, KVCACHED_IPC_NAME=os.environ["KVCACHED_IPC_NAME"] + "_ref")
    out = subprocess.run([sys.executable, "-c", code] + (["--prealloc"] if PREALLOC else []), capture_output=True, text=True,
                         timeout=600, env=env)
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT")]
    return json.loads(line[-1][6:]) if line else {"error": (out.stderr or out.stdout)[-300:]}


def oracle():
    import kvc_testlib as T
    lib = T.load_oracle()
    ad = T.OracleAdapter(lib, **GEOM)
    r = bench_manager(ad, lambda idx: ad.pa.group_indices_by_page(idx, GEOM["block_size"] * GEOM["cell_size"]))
    ad.close()
    return r


if __name__ == "__main__":
    res = {"prealloc_thread": PREALLOC, "cpu": os.popen("grep -m1 'model name' /proc/cpuinfo").read().split(":")[-1].strip(),
           "nproc": os.cpu_count(), "threads_used": 1, "product": product(), "oracle_via_ctypes": oracle(), "reference": reference()}
    if res["reference"] and "error" not in res["reference"]:
        res["speedup_vs_reference"] = {k: round(res["reference"][k] / res["product"][k], 2)
                                       for k in res["product"] if k.endswith("_us")}
    print(json.dumps(res, indent=1))
