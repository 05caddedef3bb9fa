"""Token-index glue on a real MI355X: the gfx950 kernels (csrc/index_kernels.hip, through the C ABI) against
the numpy oracle (oracle/sglang_glue.py) on the same seeded inputs — bit-exact, int64 — and the two SGLang
allocator classes end to end on a live KVCacheManager."""
import os
import sys

import numpy as np
import pytest
import torch

import kvc_glue_cases as G
import kvc_testlib as T

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))
import sglang_glue as O  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture()
def gpu_lib():
    from kvcached_amd import capi, vmm_ops
    vmm_ops.init_kvcached(DEV, T.PAGE, False)
    yield capi
    vmm_ops.shutdown_kvcached()


def dev(a):
    return torch.tensor(np.asarray(a), dtype=torch.int64, device=DEV)


@pytest.mark.parametrize("n,tpb", [(0, 16), (1, 1), (1, 16), (300, 16), (1024, 64), (1025, 16), (5000, 3), (70000, 1)])
def test_expand_block_ids(gpu_lib, n, tpb):
    from kvcached_amd.integration.sglang.allocators import expand_block_ids
    ids = np.random.default_rng(n * 31 + tpb).permutation(1 << 22)[:n].astype(np.int64)
    got = expand_block_ids([int(x) for x in ids], tpb, DEV)
    torch.cuda.synchronize()
    assert got.dtype == torch.int64 and np.array_equal(got.cpu().numpy(), O.expand_block_ids(ids, tpb))


EXTEND_GPU_CASES = G.EXTEND_CASES + [
    dict(seed=11, bs=1, tpb=16, max_prefix=5, max_extend=20000),           # one long prompt: several token groups
    dict(seed=12, bs=700, tpb=16, max_prefix=64, max_extend=64),           # > 256 requests: strided prefix sums
    dict(seed=13, bs=40, tpb=16, max_prefix=200, max_extend=2000),         # > 1024 new blocks: staged id table
    dict(seed=14, bs=2000, tpb=1, max_prefix=3, max_extend=3),             # page_size 1
]


@pytest.mark.parametrize("cfg", EXTEND_GPU_CASES, ids=lambda c: f"s{c['seed']}_bs{c['bs']}_t{c['tpb']}")
def test_alloc_extend_indices(gpu_lib, cfg):
    from kvcached_amd.integration.sglang.allocators import alloc_extend_indices
    case = G.extend_case(**cfg)
    want = O.alloc_extend(case["prefix_lens"], case["seq_lens"], case["last_loc"], case["free_pages"], case["tpb"])
    if cfg["seed"] == 13:
        assert len(case["free_pages"]) > 1024
    got = alloc_extend_indices(dev(case["prefix_lens"]), dev(case["seq_lens"]), dev(case["last_loc"]),
                               [int(x) for x in case["free_pages"]], case["tpb"], case["extend_num_tokens"])
    torch.cuda.synchronize()
    got = got.cpu().numpy()
    assert np.array_equal(got, want)
    G.check_extend_invariants(case, got)


def test_alloc_extend_accepts_int32_lengths(gpu_lib):
    """SGLang hands int32 lengths on some paths: converted on the caller's stream, same results. Many small seeded
    batches back to back (nothing synchronises in between: the temporaries of one call are recycled by the next)."""
    from kvcached_amd.integration.sglang.allocators import alloc_extend_indices
    pending = []
    for seed in range(21, 61):
        case = G.extend_case(seed=seed, bs=1 + seed % 13, tpb=16, max_prefix=50, max_extend=50 + 20 * (seed % 5))
        want = O.alloc_extend(case["prefix_lens"], case["seq_lens"], case["last_loc"], case["free_pages"], 16)
        got = alloc_extend_indices(dev(case["prefix_lens"]).to(torch.int32), dev(case["seq_lens"]).to(torch.int32),
                                   dev(case["last_loc"]), [int(x) for x in case["free_pages"]], 16, case["extend_num_tokens"])
        pending.append((seed, got, want))
    torch.cuda.synchronize()
    for seed, got, want in pending:
        got = got.cpu().numpy()
        bad = np.flatnonzero(got != want)
        assert bad.size == 0, (seed, bad[:8].tolist(), got[bad[:8]].tolist(), want[bad[:8]].tolist(), len(want))


DECODE_GPU_CASES = G.DECODE_CASES + [
    dict(seed=15, bs=3000, tpb=1, max_len=20),                             # every request opens a block: staged ids
    dict(seed=16, bs=5000, tpb=16, max_len=4000, force_boundary_every=3),
]


@pytest.mark.parametrize("cfg", DECODE_GPU_CASES, ids=lambda c: f"s{c['seed']}_bs{c['bs']}_t{c['tpb']}")
def test_alloc_decode_indices(gpu_lib, cfg):
    from kvcached_amd.integration.sglang.allocators import alloc_decode_indices
    case = G.decode_case(**cfg)
    want = O.alloc_decode(case["seq_lens"], case["last_loc"], case["free_pages"], case["tpb"])
    got = alloc_decode_indices(dev(case["seq_lens"]), dev(case["last_loc"]), [int(x) for x in case["free_pages"]], case["tpb"])
    torch.cuda.synchronize()
    assert np.array_equal(got.cpu().numpy(), want)


@pytest.mark.parametrize("n,tpb,n_blocks", [(1, 16, 10), (3000, 16, 5000), (100000, 16, 200000), (5000, 1, 1 << 20),
                                             (4096, 48, 3_000_000)])
def test_unique_block_ids(gpu_lib, n, tpb, n_blocks):
    from kvcached_amd.integration.sglang.allocators import unique_block_ids
    rng = np.random.default_rng(n + tpb)
    idx = rng.integers(0, n_blocks * tpb, size=n)
    idx[0], idx[-1] = 0, n_blocks * tpb - 1                     # first and last block
    got = unique_block_ids(dev(idx), tpb, n_blocks)
    assert got == O.unique_block_ids(idx, tpb).tolist()
    # the scratch bitmap is clean again: a second, different query is not polluted by the first
    idx2 = rng.integers(0, min(n_blocks, 50) * tpb, size=200)
    assert unique_block_ids(dev(idx2), tpb, n_blocks) == O.unique_block_ids(idx2, tpb).tolist()


def test_unique_block_ids_rejects_out_of_range_and_recovers(gpu_lib):
    from kvcached_amd import capi
    from kvcached_amd.integration.sglang.allocators import unique_block_ids
    with pytest.raises(RuntimeError, match="outside"):
        unique_block_ids(dev([5, 16 * 100, 7]), 16, 100)
    with pytest.raises(RuntimeError, match="outside"):
        unique_block_ids(dev([-1]), 16, 100)
    assert unique_block_ids(dev([33, 1, 17, 34]), 16, 100) == [0, 1, 2]


# ------------------------------------------------------------------ the allocator classes on a live manager
class FakeBase:
    """The attributes SGLang's BaseTokenToKVPoolAllocator gives its subclasses."""

    def __init__(self, size, page_size, dtype, device, kvcache, *a, **k):
        self.size, self.page_size, self.dtype, self.device, self._kvcache = size, page_size, dtype, device, kvcache
        self.is_not_in_free_group, self.free_group = True, []


class FakePool:
    def __init__(self, manager):
        self.kvcached_allocator = manager


def _manager(num_tokens, tpb, monkeypatch):
    import kvcached_amd.kv_cache_manager as kcm
    from kvcached_amd import vmm_ops
    monkeypatch.setattr(kcm, "CONTIGUOUS_LAYOUT", False)
    monkeypatch.setattr(kcm, "PAGE_PREALLOC_ENABLED", False)
    cell = 8 * 64 * 2
    nblocks = num_tokens // tpb + 1
    size = 2 * (-(-nblocks * tpb * cell // T.PAGE)) * T.PAGE        # K and V halves, page aligned
    vmm_ops.create_kv_tensors(size, 1, DEV, 2, 2, 0, False)
    m = kcm.KVCacheManager(num_blocks=nblocks, block_size=tpb, cell_size=cell, num_layers=2, reserve_null_block=True)
    assert m._post_init_done.wait(10)
    return m


def test_paged_allocator_end_to_end(gpu_lib, monkeypatch):
    """A small serving loop: prefill (alloc_extend), decode steps (alloc_decode), completion (free). Every token
    position of every live request must sit in a block that request owns, at offset position % page_size; no slot
    is shared; the block ids are exactly what KVCacheManager handed out; after freeing everything is back."""
    from kvcached_amd.integration.sglang.allocators import build_elastic_allocators
    tpb, num_tokens = 16, 65536
    m = _manager(num_tokens, tpb, monkeypatch)
    _, Paged = build_elastic_allocators(FakeBase)
    a = Paged(num_tokens, tpb, torch.bfloat16, DEV, FakePool(m))
    avail0 = a.available_size()
    assert avail0 == m.available_size() * tpb
    seen = []
    orig_alloc = m.alloc
    monkeypatch.setattr(m, "alloc", lambda n: (seen.append(orig_alloc(n)) or seen[-1]))
    rng = np.random.default_rng(0)
    live = {}                                                   # rid -> np.array of token slots (position order)
    next_rid = 0
    for step in range(30):
        # ---- prefill a few new requests
        bs = int(rng.integers(1, 6))
        lens = rng.integers(1, 400, size=bs)
        pre = torch.zeros(bs, dtype=torch.int64)
        seq = torch.tensor(lens, dtype=torch.int64)
        out = a.alloc_extend(pre.to(DEV), pre, seq.to(DEV), seq, torch.full((bs,), -1, dtype=torch.int64, device=DEV),
                             int(lens.sum()))
        assert out is not None
        out = out.cpu().numpy()
        k = 0
        for n in lens:
            live[next_rid] = out[k:k + n]
            k += n
            next_rid += 1
        # ---- three decode steps for everybody
        for _ in range(3):
            rids = list(live)
            seq = torch.tensor([len(live[r]) + 1 for r in rids], dtype=torch.int64)
            last = torch.tensor([int(live[r][-1]) for r in rids], dtype=torch.int64, device=DEV)
            out = a.alloc_decode(seq.to(DEV), seq, last)
            assert out is not None
            out = out.cpu().numpy()
            for r, slot in zip(rids, out):
                live[r] = np.append(live[r], slot)
        # ---- invariants
        all_slots = np.concatenate(list(live.values()))
        assert len(np.unique(all_slots)) == len(all_slots)
        for r, slots in live.items():
            pos = np.arange(len(slots))
            assert np.array_equal(slots % tpb, pos % tpb), r
            blocks = slots // tpb
            assert all(len(set(blocks[i:i + tpb])) == 1 for i in range(0, len(slots), tpb))
            assert 0 not in blocks                                # the null block is never handed out
        handed = {b for lst in seen for b in lst}
        assert set(np.unique(all_slots // tpb)) <= handed
        # ---- finish a random half
        for r in list(live):
            if rng.random() < 0.5:
                a.free(torch.tensor(live.pop(r), dtype=torch.int64, device=DEV))
    for r in list(live):
        a.free(torch.tensor(live.pop(r), dtype=torch.int64, device=DEV))
    assert a.available_size() == avail0
    # free-group mode defers, like the reference
    a.is_not_in_free_group = False
    a.free(torch.tensor([16, 17], device=DEV))
    assert len(a.free_group) == 1
    a.clear()
    assert a.free_group == [] and a.is_not_in_free_group
    del a, m


def test_token_allocator_end_to_end(gpu_lib, monkeypatch):
    from kvcached_amd.integration.sglang.allocators import build_elastic_allocators
    m = _manager(32768, 1, monkeypatch)
    Token, _ = build_elastic_allocators(FakeBase)
    a = Token(32768, torch.bfloat16, DEV, FakePool(m))
    assert a.available_size() <= 32768
    x = a.alloc(1000)
    y = a.alloc(10)
    assert x.device.type == "cuda" and x.dtype == torch.int64
    got = x.cpu().tolist() + y.cpu().tolist()
    assert got == list(range(1, 1011))                           # block 0 is the reserved null block
    a.free(x[::2])
    a.free(x[1::2])
    a.free(y)
    assert m.page_allocator.get_num_inuse_pages() == 1           # only the null block's page
    with pytest.raises(ValueError, match="GPU devices"):
        Token(10, torch.bfloat16, "cpu", FakePool(m))
    with pytest.raises(ValueError, match="elastic MHA pool"):
        Token(10, torch.bfloat16, DEV, object())
    del a, m
