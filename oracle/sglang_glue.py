"""oracle/sglang_glue.py — CPU restatement (numpy + plain loops) of the token-index glue that sits between
KVCacheManager and SGLang's token pools. TEST INFRASTRUCTURE ONLY: imported by tests/ and __graft_entry__.smoke(),
never by the product (kvcached_amd/ computes these on the GPU, csrc/index_kernels.hip).

What is restated, and how it is pinned:
  expand_block_ids   the reference's own expression, kvcached/integration/sglang/patches.py:192-196
                     (`page_ids[:, None] * page_size + arange(page_size)`), and :100-102 for page_size 1.
                     Pinned: tests/test_sglang_glue.py evaluates that expression with torch on the same inputs.
  unique_block_ids   patches.py:283 `torch.unique(free_index // page_size)` -> sorted distinct block ids.
                     Pinned the same way (torch.unique on CPU).
  alloc_extend / alloc_decode
                     the reference calls SGLang's Triton kernels `alloc_extend_kernel` / `alloc_decode_kernel`
                     (patches.py:229-242, 264-275; sglang >= 0.4.9 per patches.py:20, module
                     sglang/srt/mem_cache/allocator.py). SGLang is a third-party dependency that is NOT in
                     /root/reference and NOT installed in this image, and Triton kernels cannot run on CPU:
                     PARITY UNPINNED. What follows restates the published three-part algorithm of those kernels
                     (part 1: finish the last partially filled page after last_loc; part 2: whole new pages;
                     part 3: the final partial new page) request by request; tests additionally check the
                     layout-independent invariants (every token position lands in the page its sequence owns at
                     the right in-page offset; new pages are consumed in request order, each exactly once).
  get_num_new_pages  sglang.srt.utils.get_num_new_pages as used at patches.py:215-219,254-258.
"""
from __future__ import annotations

import numpy as np


def expand_block_ids(block_ids, tokens_per_block: int) -> np.ndarray:
    ids = np.asarray(block_ids, dtype=np.int64)
    return (ids[:, None] * tokens_per_block + np.arange(tokens_per_block, dtype=np.int64)).reshape(-1)


def unique_block_ids(token_indices, tokens_per_block: int) -> np.ndarray:
    return np.unique(np.asarray(token_indices, dtype=np.int64) // tokens_per_block)


def get_num_new_pages(seq_lens, page_size: int, prefix_lens=None, decode: bool = False) -> int:
    seq = np.asarray(seq_lens, dtype=np.int64)
    if decode:
        pre = seq - 1
    else:
        pre = np.asarray(prefix_lens, dtype=np.int64)
    after = (seq + page_size - 1) // page_size
    before = (pre + page_size - 1) // page_size
    return int((after - before).sum())


def alloc_extend(prefix_lens, seq_lens, last_loc, free_pages, page_size: int) -> np.ndarray:
    pre_all = [int(x) for x in prefix_lens]
    seq_all = [int(x) for x in seq_lens]
    out = np.full(sum(s - p for s, p in zip(seq_all, pre_all)), -1, dtype=np.int64)
    ps = page_size
    out_start = 0
    new_page_start = 0
    for pre, seq, loc in zip(pre_all, seq_all, [int(x) for x in last_loc]):
        extend = seq - pre
        n_new = (seq + ps - 1) // ps - (pre + ps - 1) // ps
        # part 1: the rest of the partially filled last page
        n1 = min(seq, (pre + ps - 1) // ps * ps) - pre
        for o in range(n1):
            out[out_start + o] = loc + 1 + o
        if pre + n1 != seq:
            # part 2: whole new pages
            n2 = seq // ps * ps - (pre + ps - 1) // ps * ps
            for o in range(n2):
                out[out_start + n1 + o] = int(free_pages[new_page_start + o // ps]) * ps + o % ps
            if pre + n1 + n2 != seq:
                # part 3: the final, partially filled new page
                n3 = seq - seq // ps * ps
                start = int(free_pages[new_page_start + n_new - 1])
                for o in range(n3):
                    out[out_start + n1 + n2 + o] = start * ps + o
        out_start += extend
        new_page_start += n_new
    return out


def alloc_decode(seq_lens, last_loc, free_pages, page_size: int) -> np.ndarray:
    ps = page_size
    out = np.empty(len(seq_lens), dtype=np.int64)
    k = 0
    for i, (seq, loc) in enumerate(zip([int(x) for x in seq_lens], [int(x) for x in last_loc])):
        n_new = (seq + ps - 1) // ps - (seq - 1 + ps - 1) // ps
        if n_new == 0:
            out[i] = loc + 1
        else:
            out[i] = int(free_pages[k]) * ps
            k += 1
    return out
