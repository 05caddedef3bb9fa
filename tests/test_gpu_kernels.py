"""Parity of the two gfx950 kernels against the CPU oracle, through the C ABI (ctypes).

zero_fill_pages  vs  okvc_zero_fill_pages (memset)      — bit-exact
compact_blocks   vs  okvc_compact_blocks  (memcpy loop)  — bit-exact
plus size-independent properties at BASELINE.json's full batch size (1024 x 2 MiB).
"""
import ctypes

import numpy as np
import pytest
import torch

import kvc_testlib as T

pytestmark = pytest.mark.gpu

KiB, MiB = 1 << 10, 1 << 20


@pytest.fixture(scope="module")
def capi():
    from kvcached_amd import capi as c
    c.init("cuda:0", 2 * MiB, False)
    yield c
    c.shutdown()


def _oracle_fill(lib, host: np.ndarray, offsets, page_bytes):
    base = host.ctypes.data
    ptrs = (ctypes.c_void_p * len(offsets))(*[base + o for o in offsets])
    lib.okvc_zero_fill_pages(ptrs, len(offsets), page_bytes)


@pytest.mark.parametrize("page_bytes,n_pages,n_sel,seed", [
    (64 * KiB, 8, 1, 0),            # one slab
    (64 * KiB, 64, 17, 1),          # ragged selection of minimal pages
    (2 * MiB, 16, 5, 2),            # real page size, scattered
    (2 * MiB, 300, 257, 3),         # a ragged count that is not a multiple of the 8-page XCD group
    (64 * KiB, 2100, 1500, 6),      # more pages than one launch carries (1024)
    (6 * MiB, 6, 3, 4),             # page size that is not a power of two (3 x 2 MiB)
    (2 * MiB, 4, 0, 5),             # empty batch
])
@pytest.mark.parametrize("variant", [0, 1, 2, 3, 4, 5])   # 0: XCD x owns pages x, x+8, ... of the table in its fixed pseudo-random order; 5: of the caller's order; 4: a contiguous eighth; 1-3: A/B forms
def test_zero_fill_matches_oracle(capi, oracle_lib, page_bytes, n_pages, n_sel, seed, variant):
    capi.set_option(capi.OPT_FILL_VARIANT, variant)
    try:
        _zero_fill_case(capi, oracle_lib, page_bytes, n_pages, n_sel, seed)
    finally:
        capi.set_option(capi.OPT_FILL_VARIANT, 0)


def _zero_fill_case(capi, oracle_lib, page_bytes, n_pages, n_sel, seed):
    rng = np.random.default_rng(seed)
    total = page_bytes * n_pages
    dev = torch.empty(total, dtype=torch.uint8, device="cuda:0")
    # random poison, so that a kernel writing the wrong place is visible
    host = rng.integers(1, 256, size=total, dtype=np.uint8)
    dev.copy_(torch.from_numpy(host))
    sel = sorted(int(x) for x in rng.choice(n_pages, size=n_sel, replace=False)) if n_sel else []
    rng.shuffle(sel)
    offsets = [s * page_bytes for s in sel]
    base = dev.data_ptr()
    if base % 16:
        pytest.skip("allocator returned an unaligned buffer")
    torch.cuda.synchronize()  # our kernels run on the library's own stream
    capi.zero_fill_pages([base + o for o in offsets], page_bytes)
    _oracle_fill(oracle_lib, host, offsets, page_bytes)
    got = dev.cpu().numpy()
    assert np.array_equal(got, host)


def test_zero_fill_full_batch_properties(capi):
    """1024 x 2 MiB (the bench batch): everything selected is zero, guards on both sides intact,
    idempotent."""
    page, n = 2 * MiB, 1024
    guard = 64 * KiB
    dev = torch.full((guard + page * n + guard,), 0xA5, dtype=torch.uint8, device="cuda:0")
    base = dev.data_ptr() + guard
    assert base % 16 == 0
    perm = np.random.default_rng(0).permutation(n)
    ptrs = [base + int(i) * page for i in perm]
    capi.reset_stats()
    torch.cuda.synchronize()
    capi.zero_fill_pages(ptrs, page)
    body = dev[guard:guard + page * n]
    assert int(torch.count_nonzero(body)) == 0
    assert bool((dev[:guard] == 0xA5).all()) and bool((dev[-guard:] == 0xA5).all())
    capi.zero_fill_pages(ptrs, page)
    assert int(torch.count_nonzero(body)) == 0
    st = capi.get_stats()
    assert st["fill_bytes"] == 2 * page * n and st["fill_launches"] == 2      # 1024 pages = one launch each time


def test_zero_fill_rejects_bad_arguments(capi):
    dev = torch.zeros(4 * MiB, dtype=torch.uint8, device="cuda:0")
    with pytest.raises(capi.KvcError):
        capi.zero_fill_pages([dev.data_ptr()], 100 * KiB)      # not a multiple of 64 KiB
    with pytest.raises(capi.KvcError):
        capi.zero_fill_pages([dev.data_ptr() + 8], 64 * KiB)   # misaligned pointer


def _oracle_compact(lib, hosts, src, dst, block_bytes):
    bases = (ctypes.c_void_p * len(hosts))(*[h.ctypes.data for h in hosts])
    lib.okvc_compact_blocks(bases, len(hosts), T._arr(src), T._arr(dst), len(src), block_bytes)


@pytest.mark.parametrize("variant", [0, 1, 2, 3, 4, 5, 7, 8, 9, 10, 11])   # 9: an eighth of the pairs per XCD; 0 = 6: the 32 KiB-tile default; 8: the 16 KiB-tile default of rounds 1-2
@pytest.mark.parametrize("block_bytes,n_blocks,n_regions,n_moves,seed", [
    (32 * KiB, 256, 4, 40, 0),       # Llama-3-8B block (16 tok x 2048 B): two full 16 KiB tiles
    (16 * KiB, 128, 3, 7, 1),        # cfg-1 block: exactly one tile
    (48 * KiB, 64, 2, 9, 2),         # three tiles
    (18432, 113, 2, 30, 3),          # MLA block 16 x 1152 B: one full tile + ragged 2 KiB tail
    (16, 1000, 1, 100, 4),           # minimum block: one lane
    (1040, 1000, 5, 449, 5),         # ragged inside the first piece; more moves than one launch (448)
    (32 * KiB, 64, 130, 3, 6),       # more regions than one launch (128)
    (32 * KiB, 64, 2, 0, 7),         # empty move list
])
def test_compact_blocks_matches_oracle(capi, oracle_lib, variant, block_bytes, n_blocks, n_regions, n_moves, seed):
    rng = np.random.default_rng(seed)
    capi.set_option(capi.OPT_COMPACT_VARIANT, variant)
    try:
        hosts = [rng.integers(0, 256, size=n_blocks * block_bytes, dtype=np.uint8) for _ in range(n_regions)]
        devs = [torch.from_numpy(h).to("cuda:0") for h in hosts]
        ids = rng.choice(n_blocks, size=2 * n_moves, replace=False)  # disjoint sources and destinations
        src, dst = [int(x) for x in ids[:n_moves]], [int(x) for x in ids[n_moves:]]
        torch.cuda.synchronize()
        capi.compact_blocks([d.data_ptr() for d in devs], src, dst, block_bytes)
        _oracle_compact(oracle_lib, hosts, src, dst, block_bytes)
        for d, h in zip(devs, hosts):
            assert np.array_equal(d.cpu().numpy(), h)
    finally:
        capi.set_option(capi.OPT_COMPACT_VARIANT, 0)


def test_compact_blocks_llama_geometry_properties(capi):
    """Full Llama-3-8B geometry: 64 regions (32 layers x K/V), 32 KiB blocks, 512 moves. Moved blocks
    equal their sources, nothing else changes (checksum of the untouched blocks)."""
    block, n_blocks, regions, moves = 32 * KiB, 2048, 64, 512
    g = torch.Generator(device="cuda:0").manual_seed(2)
    devs = [torch.randint(0, 2**31 - 1, (n_blocks * block // 4,), dtype=torch.int32, device="cuda:0", generator=g)
            for _ in range(regions)]
    before = [d.clone() for d in devs]
    ids = np.random.default_rng(2).permutation(n_blocks)[:2 * moves]
    src, dst = [int(x) for x in ids[:moves]], [int(x) for x in ids[moves:]]
    capi.reset_stats()
    torch.cuda.synchronize()
    capi.compact_blocks([d.data_ptr() for d in devs], src, dst, block)
    w = block // 4
    untouched = torch.ones(n_blocks, dtype=torch.bool, device="cuda:0")
    untouched[torch.tensor(dst, device="cuda:0")] = False
    for d, b in zip(devs, before):
        d2, b2 = d.view(n_blocks, w), b.view(n_blocks, w)
        assert torch.equal(d2[dst], b2[src])
        assert torch.equal(d2[untouched], b2[untouched])
    st = capi.get_stats()
    assert st["compact_bytes"] == 2 * block * regions * moves


def test_compact_blocks_rejects_bad_arguments(capi):
    dev = torch.zeros(1 * MiB, dtype=torch.uint8, device="cuda:0")
    with pytest.raises(capi.KvcError):
        capi.compact_blocks([dev.data_ptr()], [0], [1], 24)      # not a multiple of 16
    with pytest.raises(capi.KvcError):
        capi.compact_blocks([dev.data_ptr()], [-1], [1], 32)     # negative id
    with pytest.raises(ValueError):
        capi.compact_blocks([dev.data_ptr()], [0, 1], [2], 32)
