"""Seeded inputs for the SGLang token-index glue (shared by the CPU oracle tests and the GPU parity tests)."""
from __future__ import annotations

import numpy as np


def extend_case(seed: int, bs: int, tpb: int, max_prefix: int, max_extend: int, n_blocks: int = 1 << 20,
                aligned_prefix: bool = False, zero_extend_every: int = 0):
    """A consistent alloc_extend input: each request owns distinct blocks for its prefix; last_loc is the slot of
    its last prefix token (-1 when the prefix is empty, as SGLang passes it)."""
    rng = np.random.default_rng(seed)
    pre = rng.integers(0, max_prefix + 1, size=bs)
    if aligned_prefix:
        pre = pre // tpb * tpb
    ext = rng.integers(1, max_extend + 1, size=bs)
    if zero_extend_every:
        ext[::zero_extend_every] = 0
    seq = pre + ext
    pre_blocks = (pre + tpb - 1) // tpb
    new_blocks = (seq + tpb - 1) // tpb - pre_blocks
    ids = rng.permutation(n_blocks)[: int(pre_blocks.sum() + new_blocks.sum())].astype(np.int64)
    owned, k = [], 0
    for n in pre_blocks:
        owned.append(ids[k:k + n])
        k += n
    free_pages = ids[k:]
    last_loc = np.array([(owned[i][-1] * tpb + (pre[i] - 1) % tpb) if pre[i] else -1 for i in range(bs)], dtype=np.int64)
    return dict(prefix_lens=pre.astype(np.int64), seq_lens=seq.astype(np.int64), last_loc=last_loc,
                free_pages=free_pages, tpb=tpb, owned=owned, extend_num_tokens=int(ext.sum()))


def decode_case(seed: int, bs: int, tpb: int, max_len: int, n_blocks: int = 1 << 20, force_boundary_every: int = 0):
    rng = np.random.default_rng(seed)
    seq = rng.integers(1, max_len + 1, size=bs)
    if force_boundary_every:
        seq[::force_boundary_every] = (seq[::force_boundary_every] // tpb) * tpb + 1
    pre = seq - 1
    pre_blocks = (pre + tpb - 1) // tpb
    need = (seq + tpb - 1) // tpb - pre_blocks
    ids = rng.permutation(n_blocks)[: int(pre_blocks.sum() + need.sum())].astype(np.int64)
    owned, k = [], 0
    for n in pre_blocks:
        owned.append(ids[k:k + n])
        k += n
    free_pages = ids[k:]
    last_loc = np.array([(owned[i][-1] * tpb + (pre[i] - 1) % tpb) if pre[i] else -1 for i in range(bs)], dtype=np.int64)
    return dict(seq_lens=seq.astype(np.int64), last_loc=last_loc, free_pages=free_pages, tpb=tpb, owned=owned)


def check_extend_invariants(case, out):
    """Layout-independent properties of a correct alloc_extend result."""
    tpb = case["tpb"]
    pos0 = 0
    used_new = []
    k = 0
    for i, (pre, seq) in enumerate(zip(case["prefix_lens"], case["seq_lens"])):
        pre, seq = int(pre), int(seq)
        n_new = (seq + tpb - 1) // tpb - (pre + tpb - 1) // tpb
        blocks = list(case["owned"][i]) + list(case["free_pages"][k:k + n_new])
        k += n_new
        used_new += blocks[len(case["owned"][i]):]
        for t in range(seq - pre):
            p = pre + t
            assert out[pos0 + t] == blocks[p // tpb] * tpb + p % tpb, (i, t)
        pos0 += seq - pre
    assert pos0 == len(out) and k == len(case["free_pages"])
    assert len(set(used_new)) == len(used_new)


EXTEND_CASES = [
    dict(seed=1, bs=1, tpb=16, max_prefix=0, max_extend=40),
    dict(seed=2, bs=7, tpb=16, max_prefix=100, max_extend=100),
    dict(seed=3, bs=64, tpb=16, max_prefix=300, max_extend=50, aligned_prefix=True),
    dict(seed=4, bs=33, tpb=1, max_prefix=20, max_extend=20),
    dict(seed=5, bs=50, tpb=3, max_prefix=40, max_extend=40, zero_extend_every=7),
    dict(seed=6, bs=300, tpb=64, max_prefix=500, max_extend=300),
]
DECODE_CASES = [
    dict(seed=1, bs=1, tpb=16, max_len=100),
    dict(seed=2, bs=257, tpb=16, max_len=1000, force_boundary_every=5),
    dict(seed=3, bs=64, tpb=1, max_len=50),
    dict(seed=4, bs=1000, tpb=3, max_len=200),
]
