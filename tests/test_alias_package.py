"""`import kvcached.…` (what the reference's engine patches do) resolves to kvcached_amd."""
import os
import subprocess
import sys

import kvc_testlib as T


def test_reference_import_paths_resolve_to_the_native_implementation():
    code = r"""
import os, sys
sys.path.insert(0, %r)
os.environ["KVCACHED_IPC_NAME"] = "kvc_test_alias_%%d" %% os.getpid()
from kvcached.vmm_ops import create_kv_tensors, init_kvcached, kv_tensors_created, map_to_kv_tensors, shutdown_kvcached, unmap_from_kv_tensors
import kvcached.vmm_ops as ops
from kvcached.kv_cache_manager import KVCacheManager
from kvcached.tp_ipc_util import broadcast_map_to_kv_tensors, start_worker_listener_thread
from kvcached.utils import PAGE_SIZE, CONTIGUOUS_LAYOUT, DEFAULT_IPC_NAME, KVCachedConfigError, get_kvcached_logger
from kvcached.locks import NoOpLock
from kvcached.integration.vllm.interfaces import alloc_kv_cache, get_kv_cache_manager, init_kvcached as vinit, should_use_worker_ipc
from kvcached.integration.sglang.interfaces import alloc_mamba_states
import kvcached_amd.vmm_ops, kvcached_amd.kv_cache_manager
assert ops is kvcached_amd.vmm_ops and KVCacheManager is kvcached_amd.kv_cache_manager.KVCacheManager
assert "kvcached_amd" in ops.__file__ and ops.__file__.endswith(".so")
assert {"PageAllocator", "InternalPage", "init_kvcached", "create_kv_tensors"} <= set(dir(ops))
print("alias ok", PAGE_SIZE)
""" % T.REPO
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120,
                         env={**os.environ, "PYTHONPATH": ""})
    assert out.returncode == 0, out.stderr[-800:]
    assert "alias ok 2097152" in out.stdout
