"""The C-ABI library loads without a GPU and exports exactly what include/kvcached_amd.h declares
(no compute calls here). The header is parsed with a small regex, the library with ctypes/nm."""
import ctypes
import os
import re
import subprocess

import pytest

import kvc_testlib as T

HEADER = os.path.join(T.REPO, "include", "kvcached_amd.h")


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = set(re.findall(r"\b(kvc_[a-z0-9_]+)\s*\(", src))
    names -= {"kvc_broadcast_cb", "kvc_bool_cb"}  # function-pointer typedefs
    return names


def test_library_exports_every_declared_symbol():
    from kvcached_amd import capi
    names = declared_symbols()
    assert len(names) >= 55
    for n in sorted(names):
        assert hasattr(capi.lib, n), f"{n} is declared in include/kvcached_amd.h but not exported"
    assert names == set(capi.SIGNATURES), sorted(names ^ set(capi.SIGNATURES))
    out = subprocess.check_output(["nm", "-D", "--defined-only", capi.LIB_PATH], text=True)
    exported = {l.split()[-1] for l in out.splitlines() if " T " in l and l.split()[-1].startswith("kvc_")}
    assert exported == names, f"exported but undeclared: {sorted(exported - names)}"


def test_abi_version_and_error_channel():
    from kvcached_amd import capi
    want = int(re.search(r"#define\s+KVC_ABI_VERSION\s+(\d+)", open(HEADER).read()).group(1))
    assert capi.lib.kvc_abi_version() == want == 5          # (5: kvc_quiesce_*, KVC_OPT_UNMAP_INVALIDATION_US)
    assert capi.lib.kvc_set_option(999, 1) == capi.KVC_E_INVALID
    assert capi.last_error() == "unknown option"
    assert capi.lib.kvc_get_device(None, None) == capi.KVC_E_INVALID  # not initialised


def test_gpu_only_entry_points_refuse_the_cpu_device():
    """No CPU fallback: on the explicit "cpu" device the kernels and the memory query fail loudly."""
    from kvcached_amd import capi
    capi.init("cpu", 2 << 20, False)
    try:
        buf = (ctypes.c_char * (1 << 16))()
        with pytest.raises(capi.KvcError) as e:
            capi.zero_fill_pages([ctypes.addressof(buf)], 1 << 16)
        assert e.value.code == capi.KVC_E_NO_GPU
        with pytest.raises(capi.KvcError) as e:
            capi.compact_blocks([ctypes.addressof(buf)], [0], [1], 16)
        assert e.value.code == capi.KVC_E_NO_GPU
        with pytest.raises(capi.KvcError) as e:
            capi.mem_get_info()
        assert e.value.code == capi.KVC_E_NO_GPU
        with pytest.raises(capi.KvcError) as e:
            capi.export_mapped_slots([0])
        assert e.value.code in (capi.KVC_E_NO_GPU, capi.KVC_E_INVALID)
    finally:
        capi.shutdown()


def test_argument_validation_matches_reference_messages():
    from kvcached_amd import capi, vmm_ops
    with pytest.raises(RuntimeError, match="must be a multiple of 2MB"):
        vmm_ops.init_kvcached("cpu", 3 << 20, False)           # reference aborts here; we raise
    with pytest.raises(RuntimeError, match="Unsupported device string"):
        vmm_ops.init_kvcached("tpu:0", 2 << 20, False)
    vmm_ops.init_kvcached("cpu", 2 << 20, False)
    try:
        with pytest.raises(RuntimeError, match="Unsupported dtype size: 3"):
            vmm_ops.create_kv_tensors(4 << 20, 3, "cpu", 1)
        assert vmm_ops.map_to_kv_tensors([0]) is False          # KV tensors not created: False, like the reference
        assert vmm_ops.unmap_from_kv_tensors([0]) is False
        assert vmm_ops.kv_tensors_created() is False
        ts = vmm_ops.create_kv_tensors(3 << 20, 1, "cpu", 2)     # not page aligned: rounded up (warning)
        assert [t.numel() for t in ts] == [4 << 20, 4 << 20] and vmm_ops.kv_tensors_created()
        with pytest.raises(RuntimeError, match="not a valid page offset"):
            vmm_ops.map_to_kv_tensors([12345])
        assert vmm_ops.map_to_kv_tensors([0]) and vmm_ops.map_to_kv_tensors([0])   # double map tolerated
        st = capi.get_stats()
        assert st["pages_mapped"] >= 4
    finally:
        vmm_ops.shutdown_kvcached()
    with pytest.raises(RuntimeError, match="init"):
        vmm_ops.kv_tensors_created()


def test_backend_names_are_validated(monkeypatch):
    """KVCACHED_VMM_BACKEND is read at init on every device; on "cpu" every valid name is accepted and has no effect
    (no driver is touched), an unknown one is refused with the list."""
    from kvcached_amd import capi, vmm_ops
    monkeypatch.setenv("KVCACHED_VMM_BACKEND", "cuda")
    with pytest.raises(RuntimeError, match="'drm', 'hybrid' or 'hip'"):
        vmm_ops.init_kvcached("cpu", 2 << 20, False)
    for name in ("drm", "hybrid", "hip"):
        monkeypatch.setenv("KVCACHED_VMM_BACKEND", name)
        vmm_ops.init_kvcached("cpu", 2 << 20, False)
        try:
            ts = vmm_ops.create_kv_tensors(4 << 20, 1, "cpu", 1)
            assert vmm_ops.map_to_kv_tensors([0]) and vmm_ops.unmap_from_kv_tensors([0])
            assert capi.get_option(110) == 0                     # pages straight from KFD: only ever on a GPU
        finally:
            vmm_ops.shutdown_kvcached()


def test_page_allocator_surface_and_errors():
    from kvcached_amd import vmm_ops
    vmm_ops.init_kvcached("cpu", 2 << 20, False)
    try:
        pa = vmm_ops.PageAllocator(num_layers=1, mem_size_per_layer=4 << 20, page_size=2 << 20, enable_page_prealloc=False,
                                   contiguous_layout=False, ipc_name=os.environ["KVCACHED_IPC_NAME"] + "_abi")
        vmm_ops.create_kv_tensors(8 << 20, 1, "cpu", 1, 2)
        a, b = pa.alloc_page(), pa.alloc_page()
        assert (a.page_id, b.page_id, a.page_size) == (0, 1, 2 << 20)
        with pytest.raises(RuntimeError, match="No free pages left"):
            pa.alloc_page()
        assert pa.get_page_id(130, 32 << 10) == 2 and pa.get_resize_target() == -1
        # failing broadcast callback: the page id is rolled back and the message names the page
        pa.free_pages([0, 1])
        pa.trim()
        pa.set_should_use_worker_ipc_callback(lambda: True)

        def boom(ws, offs):
            raise ValueError("worker down")
        pa.set_broadcast_map_callback(boom)
        with pytest.raises(RuntimeError, match="Failed to map page 0"):
            pa.alloc_page()
        assert pa.get_num_free_pages() == 2 and pa._page_list(0)[0] == 0
        pa.set_broadcast_map_callback(None)
        assert pa.alloc_page().page_id == 0
        with pytest.raises(TypeError):
            a.alloc()                        # num_blocks has no default in the binding, like the reference
        del pa
    finally:
        vmm_ops.shutdown_kvcached()


def test_the_shipped_library_has_no_test_hooks():
    """The switches that remove a safety step for the benefit of a test (KVCACHED_TEST_*: skip the TLB invalidation, the
    rewrite of split mappings, fail a self test) are compiled into a second build that only the tests' own child processes load
    (kvcached_amd/_testhooks/, build.py); the library that ships has neither the code nor the names, so no environment can
    switch page privacy off in a deployment."""
    from kvcached_amd import capi
    here = os.path.join(T.REPO, "kvcached_amd")
    with open(os.path.join(here, "libkvcached_amd.so"), "rb") as f:
        shipped = f.read()
    assert re.search(rb"KVCACHED_TEST_[A-Z_]+", shipped) is None
    with open(os.path.join(here, "_testhooks", "libkvcached_amd.so"), "rb") as f:
        hooked = f.read()
    names = {m.decode() for m in re.findall(rb"KVCACHED_TEST_[A-Z_]+", hooked)}
    assert {"KVCACHED_TEST_BREAK_TLB_FLUSH", "KVCACHED_TEST_SKIP_PRT_REMAINDER_REFRESH", "KVCACHED_TEST_FAIL_DRM_SELFTEST"} <= names, names
    # and the hooks build is the same library otherwise: same ABI
    lib = ctypes.CDLL(os.path.join(here, "_testhooks", "libkvcached_amd.so"))
    lib.kvc_abi_version.restype = ctypes.c_int
    assert lib.kvc_abi_version() == capi.lib.kvc_abi_version()
