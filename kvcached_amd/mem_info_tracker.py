"""Python view of one engine's memory record (kvcached/mem_info_tracker.py:50-102).

In this implementation the native PageAllocator owns its record (csrc/page_allocator.cpp creates the segment,
accounts into it and polls it for resizes); this class exists for the Python callers the reference has — tools and
tests that create, charge and poll a record without an allocator — with the reference's names and behaviour:
group 0 uses DEFAULT_IPC_NAME, group g>0 appends `_g<g>`; the segment is unlinked at exit and on
SIGINT/SIGTERM/SIGHUP/SIGQUIT through ONE process-wide handler (signal.signal replaces, it does not chain)."""
from __future__ import annotations

import atexit
import os
import signal
from typing import List, Optional

from kvcached_amd.cli.utils import MemInfoStruct, RwLockedShm, get_ipc_name, get_ipc_path, init_kv_cache_limit
from kvcached_amd.utils import DEFAULT_IPC_NAME

_active_trackers: List["MemInfoTracker"] = []
_cleanup_installed = False


def _cleanup_all(*args) -> None:
    while _active_trackers:
        _active_trackers.pop()._unlink_segment()
    if args and isinstance(args[0], int):                    # called as a signal handler: die of the same signal
        signal.signal(args[0], signal.SIG_DFL)
        os.kill(os.getpid(), args[0])


def _install_cleanup_handlers() -> None:
    global _cleanup_installed
    if _cleanup_installed:
        return
    _cleanup_installed = True
    atexit.register(_cleanup_all)
    for sig in (signal.SIGINT, signal.SIGTERM, signal.SIGHUP, signal.SIGQUIT):
        try:
            signal.signal(sig, _cleanup_all)
        except Exception:                                    # not the main thread, or the signal is not settable
            pass


class MemInfoTracker:
    def __init__(self, total_mem_size: int, group_id: int = 0):
        base = DEFAULT_IPC_NAME if group_id == 0 else f"{DEFAULT_IPC_NAME}_g{group_id}"
        self.ipc_name = get_ipc_name(base)
        init_kv_cache_limit(self.ipc_name, total_mem_size)
        _active_trackers.append(self)
        _install_cleanup_handlers()

    def check_and_get_resize_target(self, current_mem_size: int, num_layers: int,
                                    num_kv_buffers: int = 2) -> Optional[int]:
        """Per-layer, per-buffer share of the record's total if it differs from `current_mem_size`, else None."""
        with RwLockedShm(self.ipc_name, MemInfoStruct.SHM_SIZE, RwLockedShm.RLOCK) as mm:
            target = MemInfoStruct.from_buffer(mm).total_size // num_layers // num_kv_buffers
        return target if target != current_mem_size else None

    def update_memory_usage(self, used_size: int, prealloc_size: int) -> None:
        with RwLockedShm(self.ipc_name, MemInfoStruct.SHM_SIZE, RwLockedShm.WLOCK) as mm:
            info = MemInfoStruct.from_buffer(mm)
            info.used_size, info.prealloc_size = used_size, prealloc_size
            info.write_to_buffer(mm)

    def _unlink_segment(self) -> None:
        try:
            os.unlink(get_ipc_path(self.ipc_name))
        except FileNotFoundError:
            pass
