"""kvcached_amd — MI355X-native elastic KV-cache virtual-memory manager.

Same public Python API as the reference `kvcached` package for the allocator hot path
(`vmm_ops`, `kv_cache_manager`, `tp_ipc_util`, `utils`, `integration.{vllm,sglang}.interfaces`),
implemented over hand-written HIP for gfx950. The sibling top-level package
`kvcached/` only aliases these modules so that `import kvcached.…` in existing engine patches
resolves here unchanged.

The native extension is mandatory: there is no pure-Python or CPU fallback for it.
"""
import torch  # noqa: F401  (must be loaded before the extension: it links libtorch)

__version__ = "0.1.0"


def _require_native():
    try:
        from . import vmm_ops  # noqa: F401
    except ImportError as e:  # fail loudly, never degrade
        raise ImportError(
            "kvcached_amd: the native extension (libkvcached_amd.so + vmm_ops) is missing or does not load: "
            f"{e}. Build it in-tree with `python kvcached_amd/build.py` (needs hipcc, --offload-arch=gfx950).") from e


_require_native()
