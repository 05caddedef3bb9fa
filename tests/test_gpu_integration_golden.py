"""The two integration-level goldens on the real device:

a19  alloc_kv_cache layouts (kvcached/integration/{vllm,sglang}/interfaces.py:196-298): the reference's shapes, strides and
     byte offsets checked on views of REAL reserved VA on cuda:0 - and a page mapped behind them is reachable through every
     layer's view at exactly the bytes the layout says.
f1   ElasticBlockPool (kvcached/integration/vllm/patches.py:308-614): the reference's recorded request traces replayed over a
     GPU-backed KVCacheManager; every block is signed in device memory when it is filled and a prefix-cache HIT must find
     the signature intact - i.e. cached blocks keep their pages (and their contents) while they sit evictable."""
import json
import os

import pytest
import torch

import kvc_testlib as T

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
LAYOUTS = json.load(open(os.path.join(T.GOLDEN_DIR, "alloc_kv_cache_layouts.json")))["cases"]
PREFIX_REAL = json.load(open(os.path.join(T.GOLDEN_DIR, "prefix_cache_real_manager.json")))


class _Props:
    def __init__(self, total):
        self.total_memory = total


def _desc(t, base):
    return {"shape": list(t.shape), "stride": list(t.stride()), "dtype": str(t.dtype),
            "offset_bytes": int(t.data_ptr() - base), "storage_offset": int(t.storage_offset())}


@pytest.mark.parametrize("idx", range(len(LAYOUTS)))
def test_layout_on_real_va_matches_reference(monkeypatch, idx):
    rec = LAYOUTS[idx]
    c = rec["case"]
    import kvcached_amd.integration.sglang.interfaces as sg
    import kvcached_amd.integration.vllm.interfaces as vl
    from kvcached_amd import vmm_ops
    mod = vl if c["engine"] == "vllm" else sg
    # the golden was recorded for a device of c["gpu_bytes"]: the size of the reservation follows from it
    monkeypatch.setattr(torch.cuda, "get_device_properties", lambda dev=None: _Props(c["gpu_bytes"]))
    monkeypatch.setattr(mod, "_kvcached_initialized", True)
    monkeypatch.setattr(mod, "_contiguous_layout", c["contiguous"])
    captured = {}
    real = vmm_ops.create_kv_tensors

    def spy(size, dtype_size, dev, num_layers, num_kv_buffers=2, group_id=0, unified_pool=False):
        captured.update(size=int(size), dtype_size=int(dtype_size), num_layers=int(num_layers),
                        num_kv_buffers=int(num_kv_buffers), unified_pool=bool(unified_pool))
        ts = real(size, dtype_size, dev, num_layers, num_kv_buffers, group_id, unified_pool)
        captured["bases"] = [int(t.data_ptr()) for t in ts]
        captured["raw"] = ts
        return ts

    monkeypatch.setattr(mod, "create_kv_tensors", spy)
    vmm_ops.init_kvcached(DEV, 2 << 20, c["contiguous"])
    try:
        dtype = getattr(torch, c["dtype"])
        if c["engine"] == "vllm":
            out = mod.alloc_kv_cache(tuple(c["shape"]), c["block_size"], dtype, DEV, c["num_layers"],
                                     attention_type=c["attention_type"], kernel_block_size=c.get("kernel_block_size"))
        else:
            out = mod.alloc_kv_cache(tuple(c["shape"]), dtype, DEV, c["num_layers"], page_size=c["block_size"],
                                     attention_type=c["attention_type"])
        bases, raw = captured.pop("bases"), captured.pop("raw")
        assert captured == rec["create_kv_tensors"]
        assert all(b % (2 << 20) == 0 for b in bases)                      # real reservations, page aligned
        extra = None
        if c["attention_type"] == "HYBRID_LINEAR":
            out, extra = out

        def descs(tensors):
            assert all(t.device == torch.device(DEV) for t in tensors)
            return [_desc(t, bases[0] if c["contiguous"] else bases[i]) for i, t in enumerate(tensors)][:3]

        views = []
        if isinstance(out, tuple):
            assert descs(out[0]) == rec["k"] and descs(out[1]) == rec["v"]
            assert len(out[0]) == len(out[1]) == c["num_layers"]
            views = [out[0][0], out[1][0], out[0][-1], out[1][-1]]
        else:
            assert descs(out) == rec["kv"] and len(out) == c["num_layers"]
            views = [out[0], out[-1]]
        if extra is not None:
            got = {k: (v if not isinstance(v, list) else [list(b.shape) for b in v][:2]) for k, v in extra.items()}
            assert got == rec["raw_info"]
        # back page id 0 everywhere and reach it through the views: the first element of every view lies in that page
        # (K at offset 0; V either in the same compound page or at size/2 - both are slots of page id 0)
        assert vmm_ops.map_to_kv_tensors([0])
        firsts = {}
        for v in views:
            firsts.setdefault(v.data_ptr(), v[(0,) * v.dim()])              # a 0-d view of the element the layout puts first
        for n, f in enumerate(firsts.values()):
            assert float(f) == 0                                            # zero-filled, and not a fault
            f.fill_(n + 1)
        torch.cuda.synchronize()
        for n, f in enumerate(firsts.values()):
            assert float(f) == n + 1                                        # distinct places, no aliasing between them
        assert vmm_ops.unmap_from_kv_tensors([0])
        del raw, views, out
    finally:
        vmm_ops.shutdown_kvcached()


def _blk_view(tensors, block_id, block_bytes):
    """The int64 words of block `block_id` in K and V of every layer (raw per-layer tensors: K half, V half)."""
    words = block_bytes // 8
    out = []
    for t in tensors:
        w = t.view(torch.int64)
        half = w.numel() // 2
        out += [w[block_id * words:(block_id + 1) * words], w[half + block_id * words:half + (block_id + 1) * words]]
    return out


def _run_prefix_trace(monkeypatch, case, device):
    """The recorded request trace through ElasticBlockPool over a real KVCacheManager on `device`. On the GPU every new block is
    signed in device memory and every hit is checked against the signature of its hash. Returns one record per op:
    [hits + block ids | error, free blocks, cached keys, evictable ids] (kvc_testlib.replay_prefix_over_real_manager)."""
    cfg = case["config"]
    import kvcached_amd.integration.vllm.interfaces as vi
    from kvcached_amd import capi, vmm_ops
    from kvcached_amd.integration.vllm.block_pool import build_elastic_block_pool
    g = case["geometry"]
    layers, block_tokens, cell = g["layers"], g["block_tokens"], g["cell"]     # 256 KiB blocks: 8 per 2 MiB page
    block_bytes = block_tokens * cell
    n = cfg["num_blocks"]
    pages = -(-n * block_bytes // T.PAGE)
    on_gpu = device != "cpu"
    vmm_ops.init_kvcached(device, T.PAGE, False)
    if not on_gpu:
        T.set_product_phys_pages(1 << 30, T.PAGE, layers, 2)
    monkeypatch.setattr(vi, "_kvcached_initialized", True)
    monkeypatch.setattr(vi, "_is_worker", True)
    try:
        raw = vmm_ops.create_kv_tensors(pages * T.PAGE * 2, 1, device, layers, 2, 0, False)
        cls = build_elastic_block_pool(T.FakeBlockPool, T.FakeKVCacheBlock)
        pool = cls(num_gpu_blocks=n, block_size=block_tokens, cell_size=cell, num_layers=layers,
                   enable_caching=cfg["enable_caching"], max_cached_blocks=cfg["max_cached_blocks"])
        assert pool.kv_cache_manager._post_init_done.wait(20)
        assert pool.null_block.block_id == case["null_block"]
        peak = [0]

        def on_hit(hash_ids, blocks):   # the point of the GPU run: a hit hands back a block whose CONTENT is what was written for that hash
            for h_id, b in zip(hash_ids, blocks):
                for v in _blk_view(raw, b.block_id, block_bytes):
                    assert int(v[0]) == h_id and int(v[-1]) == ~h_id, (h_id, b.block_id)

        def on_new(hash_ids, blocks):   # "compute" the new blocks: sign them
            for h_id, b in zip(hash_ids, blocks):
                for v in _blk_view(raw, b.block_id, block_bytes):
                    v[0] = h_id
                    v[-1] = ~h_id

        def after_req():
            st = capi.get_stats()
            peak[0] = max(peak[0], st["pages_mapped"] - st["pages_unmapped"])

        out = T.replay_prefix_over_real_manager(pool, case["ops"], on_hit if on_gpu else None, on_new if on_gpu else None,
                                                after_req if on_gpu else None)
        pa = pool.kv_cache_manager.page_allocator
        ends = [pa.get_num_inuse_pages(), pa.get_num_reserved_pages(), pa.get_num_free_pages()]
        if on_gpu:
            torch.cuda.synchronize()
            assert peak[0] > 0
        del pool
    finally:
        vmm_ops.shutdown_kvcached()
        capi.set_mem_info_override(0, 0)
    return out, ends


@pytest.mark.parametrize("name", ["default_cap_1000", "cap_5", "tiny_pool_pressure"])
def test_prefix_cache_trace_over_a_gpu_backed_manager(monkeypatch, name):
    """f1 on the GPU against the reference itself: tests/golden/prefix_cache_real_manager.json is what the
    reference's ElasticBlockPool (kvcached/integration/vllm/patches.py:308-614) did on these request traces over its OWN
    KVCacheManager (oracle/gen_golden.py: gen_prefix_cache_real_manager). The product's pool over the product's manager on
    cuda:0, every map executed, must give the same record for EVERY op - hits, block ids, free blocks, cached keys, the order
    of the evictable set, the errors of an exhausted pool - and the same page counts at the end; and what only a GPU can
    show: every hit finds its block's contents intact in device memory."""
    case = next(c for c in PREFIX_REAL["cases"] if c["config"]["name"] == name)
    got, ends = _run_prefix_trace(monkeypatch, case, DEV)
    assert len(got) == len(case["records"]) == len(case["ops"])
    for i, (g, want) in enumerate(zip(got, case["records"])):
        assert g == want, f"op {i} {case['ops'][i]}: cuda:0 {g} != reference {want}"
    assert ends == case["pages_at_end"]
    assert sum(r[0]["hit"] for r in got if isinstance(r[0], dict)) > 0
