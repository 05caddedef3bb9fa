"""The `/dev/shm/<ipc_name>` memory record, seen from Python — the format `kvctl limit` writes and the native
PageAllocator (csrc/page_allocator.cpp: update_memory_usage, resize watcher) reads and writes.

Same names, arguments and error behaviour as the reference's kvcached/cli/utils.py:16-210, so its tools and its
tests/test_shm_info_tracker.py run on this module; implemented on plain files under SHM_DIR (which is all a POSIX
shared-memory object is on Linux), so `posix_ipc` — absent from this image — is not needed.

Record (cli/utils.py:28-50; csrc/inc/mem_info_tracker.hpp:152-244): three little-endian int64
`total_size, used_size, prealloc_size`, 24 bytes, every access under `flock` (shared to read, exclusive to write).
"""
from __future__ import annotations

import fcntl
import mmap
import os
import struct
from dataclasses import dataclass
from typing import ClassVar, Optional

from kvcached_amd.utils import SHM_DIR

_RECORD = struct.Struct("<3q")


def get_ipc_path(ipc_name: str) -> str:
    """Segment name -> file path; an absolute path is taken as it is (cli/utils.py:16-20)."""
    return ipc_name if ipc_name.startswith("/") else os.path.join(SHM_DIR, ipc_name)


def get_ipc_name(ipc_path: str) -> str:
    """File path -> segment name (cli/utils.py:23-25)."""
    return os.path.basename(ipc_path)


@dataclass
class MemInfoStruct:
    """One snapshot of the record. `from_buffer` copies out of a mapping, `write_to_buffer` stores all three
    fields back (cli/utils.py:28-50)."""
    total_size: int
    used_size: int
    prealloc_size: int

    N_FIELDS: ClassVar[int] = 3
    SHM_SIZE: ClassVar[int] = _RECORD.size

    @classmethod
    def from_buffer(cls, buf) -> "MemInfoStruct":
        return cls(*_RECORD.unpack_from(buf, 0))

    def write_to_buffer(self, buf) -> None:
        _RECORD.pack_into(buf, 0, self.total_size, self.used_size, self.prealloc_size)


class RwLockedShm:
    """`with RwLockedShm(name, size, RLOCK|WLOCK) as mm:` — the mapped record with the file lock held
    (cli/utils.py:53-98). A missing segment is created (and sized) for a writer and is FileNotFoundError for a
    reader, so that "no limit set yet" stays distinguishable."""
    RLOCK = fcntl.LOCK_SH
    WLOCK = fcntl.LOCK_EX

    def __init__(self, file_path: str, size: int, lock_type: int):
        self.file_path = get_ipc_path(file_path)
        self.size = size
        self.lock_type = lock_type
        self._fd = -1
        self.mm: Optional[mmap.mmap] = None

    def __enter__(self) -> mmap.mmap:
        writer = self.lock_type == RwLockedShm.WLOCK
        try:
            self._fd = os.open(self.file_path, os.O_RDWR | os.O_CLOEXEC)
        except FileNotFoundError:
            if not writer:
                raise
            self._fd = os.open(self.file_path, os.O_RDWR | os.O_CREAT | os.O_CLOEXEC, 0o666)
        try:
            fcntl.flock(self._fd, self.lock_type)
            if writer and os.fstat(self._fd).st_size < self.size:
                os.ftruncate(self._fd, self.size)            # grown under the exclusive lock: no reader maps a short file
            self.mm = mmap.mmap(self._fd, self.size, access=mmap.ACCESS_WRITE if writer else mmap.ACCESS_READ)
        except BaseException:
            os.close(self._fd)                               # closing drops the lock
            self._fd = -1
            raise
        return self.mm

    def __exit__(self, exc_type, exc_value, traceback) -> None:
        try:
            self.mm.close()
        finally:
            fcntl.flock(self._fd, fcntl.LOCK_UN)
            os.close(self._fd)
            self._fd, self.mm = -1, None


def init_kv_cache_limit(ipc_name: str, kv_cache_limit: int) -> MemInfoStruct:
    """Create (or reset) the segment as `{limit, 0, 0}`; it outlives the process (cli/utils.py:101-118).
    Mode 0666 like the reference's segment, whatever the umask: other users' engines share one GPU."""
    path = get_ipc_path(get_ipc_name(ipc_name))
    fd = os.open(path, os.O_RDWR | os.O_CREAT | os.O_CLOEXEC, 0o666)
    try:
        try:
            os.fchmod(fd, 0o666)
        except PermissionError:
            pass                                             # somebody else's segment: use it as it is
    finally:
        os.close(fd)
    with RwLockedShm(get_ipc_name(ipc_name), MemInfoStruct.SHM_SIZE, RwLockedShm.WLOCK) as mm:
        info = MemInfoStruct(kv_cache_limit, 0, 0)
        info.write_to_buffer(mm)
        return info


def get_kv_cache_limit(ipc_name: str) -> Optional[MemInfoStruct]:
    """The current record, or None when the segment does not exist (cli/utils.py:121-130)."""
    try:
        with RwLockedShm(get_ipc_name(ipc_name), MemInfoStruct.SHM_SIZE, RwLockedShm.RLOCK) as mm:
            return MemInfoStruct.from_buffer(mm)
    except FileNotFoundError:
        return None


def update_kv_cache_limit(ipc_name: str, kv_cache_limit: int) -> Optional[MemInfoStruct]:
    """Set `total_size` (what `kvctl limit` does); the engine's resize watcher picks it up within 100 ms
    (page_allocator.cpp:764-778). Shrinking below what is in use is announced but still written — the allocator
    then refuses the resize and finishes it as blocks are freed (cli/utils.py:133-157)."""
    try:
        with RwLockedShm(get_ipc_name(ipc_name), MemInfoStruct.SHM_SIZE, RwLockedShm.WLOCK) as mm:
            info = MemInfoStruct.from_buffer(mm)
            if kv_cache_limit < info.total_size and info.used_size > kv_cache_limit:
                print(f"No enough free space to decrease for the new kv_cache_limit for {ipc_name}")
            info.total_size = kv_cache_limit
            info.write_to_buffer(mm)
            print(f"Updated kv cache limit for {ipc_name} to {_format_size(kv_cache_limit)} ({kv_cache_limit} bytes)")
            return info
    except FileNotFoundError:
        return None


def delete_kv_cache_segment(ipc_name: str) -> bool:
    """Remove the segment; False when there was none (cli/utils.py:165-188)."""
    try:
        os.unlink(get_ipc_path(get_ipc_name(ipc_name)))
        return True
    except FileNotFoundError:
        return False


def get_total_gpu_memory() -> int:
    """Total memory of device 0, or 0 without a GPU (cli/utils.py:191-200)."""
    try:
        import torch
        if torch.cuda.is_available():
            return torch.cuda.get_device_properties(0).total_memory
    except Exception:
        pass
    return 0


def _format_size(num_bytes: int) -> str:
    """1024-based, two decimals, B..TB (cli/utils.py:203-210)."""
    size = float(num_bytes)
    for unit in ("B", "KB", "MB", "GB"):
        if size < 1024:
            return f"{size:.2f} {unit}"
        size /= 1024
    return f"{size:.2f} TB"
