"""alloc_kv_cache of the vLLM and SGLang interfaces: the size handed to create_kv_tensors and the
shape / strides / byte offsets of every returned view, against golden vectors produced by the
reference's own integration code (tests/golden/alloc_kv_cache_layouts.json). Pure integer math:
runs on the "cpu" device with torch.cuda.get_device_properties stubbed, like the reference's
tests/test_alloc_kv_cache_alignment.py does."""
import json
import os

import pytest
import torch

import kvc_testlib as T

CASES = json.load(open(os.path.join(T.GOLDEN_DIR, "alloc_kv_cache_layouts.json")))["cases"]


class _Props:
    def __init__(self, total):
        self.total_memory = total


def _desc(t, base):
    return {"shape": list(t.shape), "stride": list(t.stride()), "dtype": str(t.dtype),
            "offset_bytes": int(t.data_ptr() - base), "storage_offset": int(t.storage_offset())}


@pytest.mark.parametrize("idx", range(len(CASES)))
def test_layout_matches_reference(monkeypatch, idx):
    rec = CASES[idx]
    c = rec["case"]
    import kvcached_amd.integration.sglang.interfaces as sg
    import kvcached_amd.integration.vllm.interfaces as vl
    from kvcached_amd import vmm_ops
    mod = vl if c["engine"] == "vllm" else sg
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
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
        return ts

    monkeypatch.setattr(mod, "create_kv_tensors", spy)
    vmm_ops.init_kvcached("cpu", 2 << 20, c["contiguous"])
    try:
        dtype = getattr(torch, c["dtype"])
        if c["engine"] == "vllm":
            out = mod.alloc_kv_cache(tuple(c["shape"]), c["block_size"], dtype, "cpu", c["num_layers"],
                                     attention_type=c["attention_type"], kernel_block_size=c.get("kernel_block_size"))
        else:
            out = mod.alloc_kv_cache(tuple(c["shape"]), dtype, "cpu", c["num_layers"], page_size=c["block_size"],
                                     attention_type=c["attention_type"])
        bases = captured.pop("bases")
        assert captured == rec["create_kv_tensors"]
        extra = None
        if c["attention_type"] == "HYBRID_LINEAR":
            out, extra = out

        def descs(tensors):
            return [_desc(t, bases[0] if c["contiguous"] else bases[i]) for i, t in enumerate(tensors)][:3]

        if isinstance(out, tuple):
            assert descs(out[0]) == rec["k"] and descs(out[1]) == rec["v"]
            assert len(out[0]) == len(out[1]) == c["num_layers"]
        else:
            assert descs(out) == rec["kv"] and len(out) == c["num_layers"]
        if extra is not None:
            got = {k: (v if not isinstance(v, list) else [list(b.shape) for b in v][:2]) for k, v in extra.items()}
            assert got == rec["raw_info"]
    finally:
        vmm_ops.shutdown_kvcached()


@pytest.mark.parametrize("integration", ["vllm", "sglang"])
@pytest.mark.parametrize("attention_type", ["MLA", "MHA"])
@pytest.mark.parametrize("gpu_gb,num_layers", [(8, 24), (8, 48), (20, 48), (24, 32), (40, 61), (80, 81), (16, 17),
                                               (288, 32), (288, 61)])
def test_ftensor_bytes_aligned_to_2x_page_size(monkeypatch, integration, attention_type, gpu_gb, num_layers):
    """The invariant of the reference's tests/test_alloc_kv_cache_alignment.py, plus MI355X's 288 GB."""
    import importlib
    from kvcached_amd.utils import PAGE_SIZE
    mod = importlib.import_module(f"kvcached_amd.integration.{integration}.interfaces")

    class Captured(Exception):
        pass

    def fake(size, *a, **k):
        raise Captured(size)

    monkeypatch.setattr(mod, "_kvcached_initialized", True, raising=False)
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "get_device_properties", lambda dev=None: _Props(gpu_gb * (1 << 30)))
    monkeypatch.setattr(mod, "create_kv_tensors", fake)
    shape = (8, 16, 576) if attention_type == "MLA" else ((2, 8, 16, 8, 128) if integration == "vllm" else (1024, 8, 128))
    with pytest.raises(Captured) as e:
        if integration == "vllm":
            mod.alloc_kv_cache(shape, 16, torch.float16, "cuda:0", num_layers, attention_type=attention_type)
        else:
            mod.alloc_kv_cache(shape, torch.float16, "cuda:0", num_layers, page_size=16, attention_type=attention_type)
    assert e.value.args[0] % (2 * PAGE_SIZE) == 0


def test_mamba_states_layout(monkeypatch):
    """alloc_mamba_states: packed super-cells, cell size is a divisor of the page, views do not overlap."""
    import types
    import kvcached_amd.integration.sglang.interfaces as sg
    from kvcached_amd import vmm_ops
    from kvcached_amd.utils import PAGE_SIZE
    params = types.SimpleNamespace(shape=types.SimpleNamespace(conv=[(4, 96), (3, 8)], temporal=(2, 16, 8)),
                                   dtype=types.SimpleNamespace(conv=torch.bfloat16, temporal=torch.float32))
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(sg, "_kvcached_initialized", True)
    for contiguous in (False, True):
        monkeypatch.setattr(sg, "_contiguous_layout", contiguous)
        vmm_ops.init_kvcached("cpu", 2 << 20, contiguous)
        try:
            conv, temporal, info = sg.alloc_mamba_states(num_slots=100, num_mamba_layers=3, cache_params=params,
                                                         device="cpu")
            raw = 4 * 96 * 2 + 3 * 8 * 2 + 2 * 16 * 8 * 4
            assert info["conv_offsets"] == [0, 768] and info["temporal_offset"] == 816
            assert info["cell_size"] >= raw and PAGE_SIZE % info["cell_size"] == 0 and info["is_contiguous"] == contiguous
            cs = info["cell_size"]
            if contiguous:
                assert conv[0].shape == (3, 100, 4, 96) and temporal.shape == (3, 100, 2, 16, 8)
                assert conv[0].stride()[:2] == (cs // 2, 3 * cs // 2) and temporal.stride()[:2] == (cs // 4, 3 * cs // 4)
                assert temporal.data_ptr() - conv[0].data_ptr() == 816
            else:
                assert len(conv) == 2 and len(conv[0]) == 3 and len(temporal) == 3
                assert conv[1][2].shape == (100, 3, 8) and conv[1][2].stride()[0] == cs // 2
                assert temporal[0].data_ptr() - conv[0][0].data_ptr() == 816
                conv[0][1][5].fill_(1.0)      # writes stay inside their own (slot, kind) span
                assert float(temporal[1][5].abs().sum()) == 0.0 and float(conv[1][1][5].abs().sum()) == 0.0
        finally:
            vmm_ops.shutdown_kvcached()
