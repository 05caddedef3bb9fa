"""ElasticBlockPool (kvcached_amd/integration/vllm/block_pool.py) against traces recorded from the reference's
own class (kvcached/integration/vllm/patches.py:308-614), built the way its tests/test_prefix_cache.py builds it.
No vLLM, no GPU: fake BlockPool / KVCacheBlock / manager (tests/kvc_testlib.py)."""
import json
import os
from unittest import mock

import pytest

import kvc_testlib as T
import kvc_traces

GOLD = json.load(open(os.path.join(T.GOLDEN_DIR, "prefix_cache.json")))


def _pool(cfg):
    from kvcached_amd.integration.vllm.block_pool import build_elastic_block_pool
    manager = T.FifoBlockManager(cfg["num_blocks"])
    cls = build_elastic_block_pool(T.FakeBlockPool, T.FakeKVCacheBlock)
    with mock.patch("kvcached_amd.integration.vllm.interfaces.get_kv_cache_manager", return_value=manager):
        pool = cls(num_gpu_blocks=cfg["num_blocks"], block_size=16, cell_size=1024, num_layers=1,
                   enable_caching=cfg["enable_caching"], max_cached_blocks=cfg["max_cached_blocks"])
    return pool, manager


def test_cache_key_bytes():
    from kvcached_amd.integration.vllm.block_pool import make_cache_key
    for h, g, want in GOLD["cache_keys"]:
        for cand in (h, bytes.fromhex(h) if all(c in "0123456789abcdef" for c in h) and len(h) % 2 == 0 and h else None):
            if cand is not None and make_cache_key(cand, g).hex() == want:
                break
        else:
            pytest.fail(f"cache key mismatch for {h!r}, group {g}")


@pytest.mark.parametrize("idx", range(len(GOLD["cases"])))
def test_trace_matches_reference(idx):
    case = GOLD["cases"][idx]
    cfg = case["config"]
    assert kvc_traces.prefix_cache_ops(cfg["n_ops"], cfg["seed"], cfg["num_blocks"]) == case["ops"]
    pool, manager = _pool(cfg)
    assert pool.null_block.block_id == case["null_block"] and pool.null_block.is_null
    got = T.replay_prefix_cache(pool, manager, case["ops"])
    for i, (g, want) in enumerate(zip(got, case["records"])):
        assert g == want, f"{cfg['name']} op {i} {case['ops'][i]}: {g} != {want}"


def test_goldens_cover_hits_evictions_and_failures():
    recs = [(r, op) for c in GOLD["cases"] for r, op in zip(c["records"], c["ops"])]
    assert any(isinstance(r["r"], dict) and r["r"]["hit"] > 0 for r, _ in recs)              # prefix hits
    assert any(isinstance(r["r"], str) and r["r"].startswith("ValueError") for r, _ in recs)  # pool exhausted
    assert any(len(r["s"][1]) > 5 for r, _ in recs)                                            # evictable set grows
    cap5 = next(c for c in GOLD["cases"] if c["config"]["name"] == "cap_5")
    assert max(len(r["s"][1]) for r in cap5["records"]) == 5                                   # the cap binds


def test_call_forms_of_cache_full_blocks():
    """The positional/keyword forms vLLM has used over time register the same blocks."""
    results = []
    for form in range(4):
        pool, manager = _pool(dict(num_blocks=32, enable_caching=True, max_cached_blocks=1000))
        blocks = pool.get_new_blocks(3)
        hs = [b"a", b"b", b"c"]
        req = T.FakeRequest(hs)
        if form == 0:
            pool.cache_full_blocks(req, blocks, 0, 3, 16, 0)
        elif form == 1:
            pool.cache_full_blocks(req, blocks, hs, 0, 3, 16, 0, None)
        elif form == 2:
            pool.cache_full_blocks(req, blocks, num_cached_blocks=0, num_full_blocks=3, block_size=16, kv_cache_group_id=0)
        else:
            pool.cache_full_blocks(req, blocks, block_hashes=hs, num_cached_blocks=0, num_full_blocks=3, block_size=16,
                                   hash_fn=hash)
        results.append([pool.get_cached_block(h).block_id for h in hs])
        assert pool.get_cached_block(b"a", [0, 1]) is None and pool.get_cached_block(b"a", 0)[0].block_id == results[-1][0]
    assert results[0] == results[1] == results[2] == results[3] == [1, 2, 3]
    with pytest.raises(TypeError):
        pool.cache_full_blocks(req, blocks)


def test_pool_over_the_real_manager(monkeypatch):
    """End to end without fakes below the pool: ElasticBlockPool -> get_kv_cache_manager -> KVCacheManager ->
    vmm_ops -> C ABI, on the "cpu" device. Block 0 is the null block; evictable blocks are reused before new pages."""
    import kvcached_amd.integration.vllm.interfaces as vi
    from kvcached_amd import capi, vmm_ops
    from kvcached_amd.integration.vllm.block_pool import build_elastic_block_pool
    vmm_ops.init_kvcached("cpu", T.PAGE, False)
    T.set_product_phys_pages(1 << 30, T.PAGE, 2, 2)
    monkeypatch.setattr(vi, "_kvcached_initialized", True)
    monkeypatch.setattr(vi, "_is_worker", True)
    try:
        vmm_ops.create_kv_tensors(64 * T.PAGE * 2, 1, "cpu", 2, 2, 0, False)
        cls = build_elastic_block_pool(T.FakeBlockPool, T.FakeKVCacheBlock)
        pool = cls(num_gpu_blocks=64 * 64, block_size=16, cell_size=2048, num_layers=2, enable_caching=True)
        assert pool.kv_cache_manager._post_init_done.wait(10)
        assert pool.null_block.block_id == 0
        free0 = pool.get_num_free_blocks()
        a = pool.get_new_blocks(70)
        assert [b.block_id for b in a] == list(range(1, 71)) and pool.get_num_free_blocks() == free0 - 70
        hs = [b"p%d" % i for i in range(70)]
        pool.cache_full_blocks(T.FakeRequest(hs), a, 0, 70, 16, 0)
        pool.free_blocks(reversed(a))
        assert pool.get_num_free_blocks() == free0 and len(pool._evictable_blocks) == 70     # kept, but count as free
        assert pool.kv_cache_manager.page_allocator.get_num_inuse_pages() == 2                # their pages stay backed
        hit = pool.get_cached_block(b"p3", [0])
        assert hit and hit[0].block_id == 4
        pool.touch(hit)
        assert pool.reset_prefix_cache() and len(pool._cached_blocks) == 0
        pool.free_blocks(hit)
        assert pool.kv_cache_manager.page_allocator.get_num_inuse_pages() == 1                # only the null block's page
        del pool
    finally:
        vmm_ops.shutdown_kvcached()
        capi.set_mem_info_override(0, 0)


REAL = json.load(open(os.path.join(T.GOLDEN_DIR, "prefix_cache_real_manager.json")))


@pytest.mark.parametrize("idx", range(len(REAL["cases"])))
def test_trace_over_the_real_manager_matches_the_reference(monkeypatch, idx):
    """The reference's ElasticBlockPool over the reference's OWN KVCacheManager (oracle/gen_golden.py:
    gen_prefix_cache_real_manager) against the product's pool over the product's manager, on the cpu device of each: every op
    gives the same record - block ids now come from a page-granular manager, not from the stand-in of the traces above.
    (tests/test_gpu_integration_golden.py runs the same comparison on cuda:0.)"""
    import kvcached_amd.integration.vllm.interfaces as vi
    from kvcached_amd import capi, vmm_ops
    from kvcached_amd.integration.vllm.block_pool import build_elastic_block_pool
    case = REAL["cases"][idx]
    cfg, g = case["config"], case["geometry"]
    assert g == T.PREFIX_REAL_GEOMETRY and kvc_traces.prefix_cache_ops(cfg["n_ops"], cfg["seed"], cfg["num_blocks"]) == case["ops"]
    block_bytes = g["block_tokens"] * g["cell"]
    pages = -(-cfg["num_blocks"] * block_bytes // T.PAGE)
    vmm_ops.init_kvcached("cpu", T.PAGE, False)
    T.set_product_phys_pages(1 << 30, T.PAGE, g["layers"], 2)
    monkeypatch.setattr(vi, "_kvcached_initialized", True)
    monkeypatch.setattr(vi, "_is_worker", True)
    try:
        vmm_ops.create_kv_tensors(pages * T.PAGE * 2, 1, "cpu", g["layers"], 2, 0, False)
        cls = build_elastic_block_pool(T.FakeBlockPool, T.FakeKVCacheBlock)
        pool = cls(num_gpu_blocks=cfg["num_blocks"], block_size=g["block_tokens"], cell_size=g["cell"], num_layers=g["layers"],
                   enable_caching=cfg["enable_caching"], max_cached_blocks=cfg["max_cached_blocks"])
        assert pool.kv_cache_manager._post_init_done.wait(20)
        assert pool.null_block.block_id == case["null_block"]
        got = T.replay_prefix_over_real_manager(pool, case["ops"])
        for i, (a, want) in enumerate(zip(got, case["records"])):
            assert a == want, f"{cfg['name']} op {i} {case['ops'][i]}: {a} != {want}"
        pa = pool.kv_cache_manager.page_allocator
        assert [pa.get_num_inuse_pages(), pa.get_num_reserved_pages(), pa.get_num_free_pages()] == case["pages_at_end"]
        del pool
    finally:
        vmm_ops.shutdown_kvcached()
        capi.set_mem_info_override(0, 0)


# ---- make_cache_key: the reference's tests/test_make_cache_key.py restated against our function
def _gid(group_id: int) -> bytes:
    return group_id.to_bytes(4, "big", signed=False)


@pytest.mark.parametrize("group_id", [0, 1, 255, 256, 65535, 2 ** 31 - 1])
def test_make_cache_key_encoding(group_id):
    from kvcached_amd.integration.vllm.block_pool import make_cache_key
    assert make_cache_key(b"h", group_id) == b"h" + _gid(group_id)
    assert make_cache_key("deadbeef", group_id) == make_cache_key(b"deadbeef", group_id) == b"deadbeef" + _gid(group_id)
    make_cache_key("0a1b2c3d" * 8, 0)                      # a 64-char hex digest (str) must not raise
    assert make_cache_key(b"samehash", 0) != make_cache_key(b"samehash", 1)
