"""`kvcached` — import-compatibility alias of kvcached_amd.

The reference's engine patches, tools and tests import `kvcached.vmm_ops`,
`kvcached.kv_cache_manager`, `kvcached.tp_ipc_util`, `kvcached.utils`, `kvcached.locks`, `kvcached.mem_info_tracker`, `kvcached.cli.utils` and
`kvcached.integration.{vllm,sglang}.interfaces`. This package contains no logic: it registers the
kvcached_amd modules under those names so that the existing imports resolve to the MI355X-native
implementation unchanged. Out-of-scope reference modules (engine patches, CLI, controller) are not
provided here; installed side by side from the reference they keep working on top of these names.
"""
import importlib
import sys

import kvcached_amd as _impl

__version__ = _impl.__version__

_ALIASES = {
    "kvcached.vmm_ops": "kvcached_amd.vmm_ops",
    "kvcached.utils": "kvcached_amd.utils",
    "kvcached.locks": "kvcached_amd.locks",
    "kvcached.tp_ipc_util": "kvcached_amd.tp_ipc_util",
    "kvcached.kv_cache_manager": "kvcached_amd.kv_cache_manager",
    "kvcached.cli": "kvcached_amd.cli",
    "kvcached.cli.utils": "kvcached_amd.cli.utils",
    "kvcached.mem_info_tracker": "kvcached_amd.mem_info_tracker",
    "kvcached.integration": "kvcached_amd.integration",
    "kvcached.integration.vllm": "kvcached_amd.integration.vllm",
    "kvcached.integration.vllm.interfaces": "kvcached_amd.integration.vllm.interfaces",
    "kvcached.integration.vllm.block_pool": "kvcached_amd.integration.vllm.block_pool",
    "kvcached.integration.sglang": "kvcached_amd.integration.sglang",
    "kvcached.integration.sglang.interfaces": "kvcached_amd.integration.sglang.interfaces",
    "kvcached.integration.sglang.allocators": "kvcached_amd.integration.sglang.allocators",
}
for _alias, _target in _ALIASES.items():
    _mod = importlib.import_module(_target)
    sys.modules[_alias] = _mod
    _parent, _, _leaf = _alias.rpartition(".")
    if _parent in sys.modules:
        setattr(sys.modules[_parent], _leaf, _mod)
