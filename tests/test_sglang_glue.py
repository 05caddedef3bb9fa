"""CPU tests of the SGLang token-index glue: the numpy oracle against the reference's own torch expressions
(kvcached/integration/sglang/patches.py:192-196, :283) and against the layout-independent invariants of
alloc_extend / alloc_decode (SGLang's kernels themselves are not in this image: parity unpinned for those two,
see oracle/sglang_glue.py), plus the host helper num_new_pages."""
import os
import sys

import numpy as np
import pytest
import torch

import kvc_glue_cases as G

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))
import sglang_glue as O  # noqa: E402


@pytest.mark.parametrize("n,tpb", [(0, 16), (1, 1), (5, 16), (1025, 64), (17, 3)])
def test_expand_matches_the_reference_expression(n, tpb):
    ids = np.random.default_rng(n + tpb).permutation(1 << 20)[:n].astype(np.int64)
    page_ids = torch.tensor(ids, dtype=torch.int64)
    want = (page_ids[:, None] * tpb + torch.arange(tpb)).reshape(-1)       # the reference's expression
    assert np.array_equal(O.expand_block_ids(ids, tpb), want.numpy())


@pytest.mark.parametrize("tpb", [1, 16, 48])
def test_unique_matches_the_reference_expression(tpb):
    idx = np.random.default_rng(tpb).integers(0, 5000 * tpb, size=3000)
    want = torch.unique(torch.tensor(idx) // tpb)
    assert np.array_equal(O.unique_block_ids(idx, tpb), want.numpy())


@pytest.mark.parametrize("cfg", G.EXTEND_CASES, ids=lambda c: f"s{c['seed']}_bs{c['bs']}_t{c['tpb']}")
def test_oracle_alloc_extend_invariants(cfg):
    case = G.extend_case(**cfg)
    out = O.alloc_extend(case["prefix_lens"], case["seq_lens"], case["last_loc"], case["free_pages"], case["tpb"])
    assert (out >= 0).all()
    G.check_extend_invariants(case, out)
    assert O.get_num_new_pages(case["seq_lens"], case["tpb"], case["prefix_lens"]) == len(case["free_pages"])


@pytest.mark.parametrize("cfg", G.DECODE_CASES, ids=lambda c: f"s{c['seed']}_bs{c['bs']}_t{c['tpb']}")
def test_oracle_alloc_decode_is_a_one_token_extend(cfg):
    case = G.decode_case(**cfg)
    out = O.alloc_decode(case["seq_lens"], case["last_loc"], case["free_pages"], case["tpb"])
    ext = O.alloc_extend(case["seq_lens"] - 1, case["seq_lens"], case["last_loc"], case["free_pages"], case["tpb"])
    assert np.array_equal(out, ext)
    assert O.get_num_new_pages(case["seq_lens"], case["tpb"], decode=True) == len(case["free_pages"])


def test_num_new_pages_host_helper():
    from kvcached_amd.integration.sglang.allocators import num_new_pages
    for cfg in G.EXTEND_CASES:
        c = G.extend_case(**cfg)
        assert num_new_pages(torch.tensor(c["seq_lens"]), c["tpb"], torch.tensor(c["prefix_lens"])) == len(c["free_pages"])
    for cfg in G.DECODE_CASES:
        c = G.decode_case(**cfg)
        assert num_new_pages(torch.tensor(c["seq_lens"]), c["tpb"], decode=True) == len(c["free_pages"])


def test_index_ops_need_a_gpu_device():
    """No CPU fallback: on the cpu device the C ABI refuses instead of computing something."""
    from kvcached_amd import capi, vmm_ops
    vmm_ops.init_kvcached("cpu", 2 << 20, False)
    try:
        buf = (capi._i64 * 16)()
        with pytest.raises(capi.KvcError) as e:
            capi.expand_block_ids([1, 2], 4, capi.ctypes.addressof(buf))
        assert e.value.code == capi.E_NO_GPU if hasattr(capi, "E_NO_GPU") else e.value.code < 0
        with pytest.raises(capi.KvcError):
            capi.unique_block_ids(capi.ctypes.addressof(buf), 4, 4, 16)
    finally:
        vmm_ops.shutdown_kvcached()
