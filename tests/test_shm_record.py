"""The memory record `/dev/shm/<name>` through its Python helpers (kvcached_amd/cli/utils.py, mem_info_tracker.py):
the behaviours the reference pins in tests/test_shm_info_tracker.py (charge/uncharge/reserve arithmetic on the
record, exclusive-lock read-modify-write across processes, tracker update and resize-target maths), the error
behaviour of its helpers (cli/utils.py:53-188), and — what the reference cannot test without a GPU — that these
helpers and the native PageAllocator talk to each other through the same 24 bytes."""
import multiprocessing as mp
import os
import struct
import time

import pytest

from kvcached_amd.cli.utils import (MemInfoStruct, RwLockedShm, _format_size, delete_kv_cache_segment, get_ipc_name,
                                    get_ipc_path, get_kv_cache_limit, init_kv_cache_limit, update_kv_cache_limit)

MiB = 1 << 20
PAGE = 2 * MiB
TOTAL = 10_000_000


@pytest.fixture()
def seg():
    name = f"kvc_rec_{os.getpid()}_{time.monotonic_ns()}"
    init_kv_cache_limit(name, TOTAL)
    yield name
    delete_kv_cache_segment(name)


def _rmw(name, field, delta, hold=0.0):
    with RwLockedShm(name, MemInfoStruct.SHM_SIZE, RwLockedShm.WLOCK) as mm:
        info = MemInfoStruct.from_buffer(mm)
        if hold:
            time.sleep(hold)
        setattr(info, field, getattr(info, field) + delta)
        info.write_to_buffer(mm)


def test_record_layout_and_names(seg):
    assert MemInfoStruct.SHM_SIZE == 24 and MemInfoStruct.N_FIELDS == 3
    assert get_ipc_path(seg) == "/dev/shm/" + seg and get_ipc_path("/tmp/x") == "/tmp/x"
    assert get_ipc_name("/dev/shm/" + seg) == seg
    assert struct.unpack("<3q", open(get_ipc_path(seg), "rb").read()) == (TOTAL, 0, 0)
    assert os.stat(get_ipc_path(seg)).st_mode & 0o777 == 0o666
    assert get_kv_cache_limit(seg) == MemInfoStruct(TOTAL, 0, 0)
    # negative and 63-bit values survive (the native side stores int64)
    _rmw(seg, "used_size", -5)
    _rmw(seg, "prealloc_size", (1 << 62) + 3)
    assert get_kv_cache_limit(seg) == MemInfoStruct(TOTAL, -5, (1 << 62) + 3)
    # init on an existing segment resets it
    init_kv_cache_limit(seg, 77)
    assert get_kv_cache_limit(seg) == MemInfoStruct(77, 0, 0)


@pytest.mark.parametrize("field,other", [("used_size", "prealloc_size"), ("prealloc_size", "used_size")])
def test_charge_and_uncharge_touch_one_field(seg, field, other):
    _rmw(seg, field, 300)
    got = get_kv_cache_limit(seg)
    assert getattr(got, field) == 300 and getattr(got, other) == 0 and got.total_size == TOTAL
    _rmw(seg, field, -300)
    assert get_kv_cache_limit(seg) == MemInfoStruct(TOTAL, 0, 0)


def _charger(name, amount, barrier):
    barrier.wait()
    _rmw(name, "used_size", amount, hold=0.05)


def test_exclusive_lock_serialises_processes(seg):
    """Five processes enter together, each holds the write lock across a sleep inside its read-modify-write:
    without mutual exclusion updates would be lost (tests/test_shm_info_tracker.py:150-173)."""
    ctx = mp.get_context("fork")
    n, amount = 5, 500
    barrier = ctx.Barrier(n)
    procs = [ctx.Process(target=_charger, args=(seg, amount, barrier)) for _ in range(n)]
    t0 = time.perf_counter()
    for p in procs:
        p.start()
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    assert get_kv_cache_limit(seg).used_size == n * amount
    assert time.perf_counter() - t0 >= n * 0.05                  # they really queued up


def test_readers_share_writers_exclude(seg):
    import fcntl
    with RwLockedShm(seg, 24, RwLockedShm.RLOCK) as a, RwLockedShm(seg, 24, RwLockedShm.RLOCK) as b:
        assert MemInfoStruct.from_buffer(a) == MemInfoStruct.from_buffer(b)
        with pytest.raises(TypeError):
            MemInfoStruct(1, 2, 3).write_to_buffer(a)             # a reader's mapping is read-only
        fd = os.open(get_ipc_path(seg), os.O_RDWR)
        try:
            with pytest.raises(BlockingIOError):
                fcntl.flock(fd, fcntl.LOCK_EX | fcntl.LOCK_NB)
        finally:
            os.close(fd)
    fd = os.open(get_ipc_path(seg), os.O_RDWR)
    try:
        fcntl.flock(fd, fcntl.LOCK_EX | fcntl.LOCK_NB)            # released on exit
    finally:
        os.close(fd)


def test_missing_segment(seg):
    name = seg + "_absent"
    assert get_kv_cache_limit(name) is None
    assert delete_kv_cache_segment(name) is False
    with pytest.raises(FileNotFoundError):
        with RwLockedShm(name, 24, RwLockedShm.RLOCK):
            pass
    assert not os.path.exists(get_ipc_path(name))                 # a failed read creates nothing
    with RwLockedShm(name, 24, RwLockedShm.WLOCK) as mm:          # a writer creates and sizes it
        assert MemInfoStruct.from_buffer(mm) == MemInfoStruct(0, 0, 0)
    assert os.path.getsize(get_ipc_path(name)) == 24
    assert delete_kv_cache_segment(name) is True
    # the reference's update takes the WRITE lock, which creates a missing segment: `kvctl limit` before the engine
    # is up leaves {limit, 0, 0} behind (its `except FileNotFoundError` cannot trigger, cli/utils.py:137-157)
    assert update_kv_cache_limit(name, 5) == MemInfoStruct(5, 0, 0) and get_kv_cache_limit(name).total_size == 5
    assert delete_kv_cache_segment(name) is True


def test_update_limit_messages(seg, capsys):
    _rmw(seg, "used_size", 4 * MiB)
    got = update_kv_cache_limit(seg, 8 * MiB)
    assert got == MemInfoStruct(8 * MiB, 4 * MiB, 0)
    out = capsys.readouterr().out
    assert "No enough free space" not in out and f"to 8.00 MB ({8 * MiB} bytes)" in out
    got = update_kv_cache_limit(seg, 2 * MiB)                     # below what is in use: announced, still written
    assert got.total_size == 2 * MiB and get_kv_cache_limit(seg).total_size == 2 * MiB
    assert "No enough free space to decrease" in capsys.readouterr().out


def test_format_size():
    assert [_format_size(v) for v in (0, 1023, 1024, 1536 * 1024, 3 << 30, 5 << 40, 1 << 60)] == \
        ["0.00 B", "1023.00 B", "1.00 KB", "1.50 MB", "3.00 GB", "5.00 TB", "1048576.00 TB"]


def test_tracker(monkeypatch):
    import kvcached_amd.mem_info_tracker as mit
    base = f"kvc_trk_{os.getpid()}"
    monkeypatch.setattr(mit, "DEFAULT_IPC_NAME", base)
    t0 = mit.MemInfoTracker(TOTAL)
    t3 = mit.MemInfoTracker(2 * TOTAL, group_id=3)
    assert (t0.ipc_name, t3.ipc_name) == (base, base + "_g3")
    t0.update_memory_usage(used_size=600, prealloc_size=700)
    assert get_kv_cache_limit(base) == MemInfoStruct(TOTAL, 600, 700)
    assert get_kv_cache_limit(base + "_g3") == MemInfoStruct(2 * TOTAL, 0, 0)
    layers = 10
    share = TOTAL // layers // 2
    assert t0.check_and_get_resize_target(share - 1, layers) == share
    assert t0.check_and_get_resize_target(share, layers) is None
    assert t0.check_and_get_resize_target(TOTAL // layers, layers, num_kv_buffers=1) is None
    update_kv_cache_limit(base, TOTAL // 2)
    assert t0.check_and_get_resize_target(share, layers) == share // 2
    # one process-wide cleanup for every tracker
    assert mit._active_trackers[-2:] == [t0, t3]
    mit._cleanup_all()
    assert not os.path.exists(get_ipc_path(base)) and not os.path.exists(get_ipc_path(base + "_g3"))
    assert mit._active_trackers == []
    t0._unlink_segment()                                           # idempotent


def _sigterm_child(base, conn):
    import kvcached_amd.mem_info_tracker as mit
    mit.DEFAULT_IPC_NAME = base
    mit.MemInfoTracker(123)
    mit.MemInfoTracker(456, group_id=1)
    conn.send("up")
    time.sleep(30)


def test_tracker_unlinks_on_sigterm():
    import signal
    ctx = mp.get_context("spawn")
    base = f"kvc_sig_{os.getpid()}"
    parent, child = ctx.Pipe()
    p = ctx.Process(target=_sigterm_child, args=(base, child))
    p.start()
    try:
        assert parent.poll(60) and parent.recv() == "up"
        assert get_kv_cache_limit(base).total_size == 123 and get_kv_cache_limit(base + "_g1").total_size == 456
        os.kill(p.pid, signal.SIGTERM)
        p.join(20)
        assert p.exitcode == -signal.SIGTERM                       # re-raised with the default action
        assert get_kv_cache_limit(base) is None and get_kv_cache_limit(base + "_g1") is None
    finally:
        if p.is_alive():
            p.kill()
        for n in (base, base + "_g1"):
            delete_kv_cache_segment(n)


# ------------------------------------------------------------------ both sides of the same 24 bytes
@pytest.fixture()
def cpu_ops():
    import kvc_testlib as T
    from kvcached_amd import capi, vmm_ops
    vmm_ops.init_kvcached("cpu", PAGE, False)
    T.set_product_phys_pages(1 << 30, PAGE, 2, 2)
    yield vmm_ops
    vmm_ops.shutdown_kvcached()
    capi.set_mem_info_override(0, 0)


def test_native_allocator_and_python_helpers_share_the_record(cpu_ops):
    ops = cpu_ops
    layers, kv, pages = 2, 2, 64
    ops.create_kv_tensors(pages * PAGE * kv, 1, "cpu", layers, kv, 0, False)
    pa = ops.PageAllocator(layers, pages * PAGE, PAGE, 1, 0, False, False, False, kv, 0,
                           os.environ["KVCACHED_IPC_NAME"] + "_rec")
    name = pa._ipc_name()
    unit = PAGE * layers * kv
    assert get_kv_cache_limit(name) == MemInfoStruct(pages * unit, 0, 0)          # written by the C++ constructor
    held = [pa.alloc_page().page_id for _ in range(3)]
    assert get_kv_cache_limit(name) == MemInfoStruct(pages * unit, 3 * unit, 0)
    pa.free_page(held.pop())
    assert get_kv_cache_limit(name) == MemInfoStruct(pages * unit, 2 * unit, unit)
    # `kvctl limit`: Python writes field 0 under the exclusive lock, the allocator reads it under the shared one
    update_kv_cache_limit(name, 16 * unit)
    assert pa.check_and_get_resize_target(pages * PAGE) == 16 * PAGE
    assert pa.resize(16 * PAGE) and pa.get_num_total_pages() == 16
    # the allocator publishes used/prealloc WITHOUT the file lock (one aligned 8-byte store each):
    # an external lock holder never stalls the serving path ...
    with RwLockedShm(name, 24, RwLockedShm.WLOCK) as mm:
        stale = MemInfoStruct.from_buffer(mm)
        t0 = time.perf_counter()
        pa.free_pages(held)
        assert time.perf_counter() - t0 < 0.5
        assert MemInfoStruct.from_buffer(mm).used_size == 0                        # visible through the same pages
        # ... and a controller's read-modify-write that loses this race writes the old values back
        stale.total_size = 16 * unit
        stale.write_to_buffer(mm)
    assert get_kv_cache_limit(name).used_size == 2 * unit
    pa.trim()                                                                     # the next page event republishes
    assert get_kv_cache_limit(name) == MemInfoStruct(16 * unit, 0, 0)
    del pa
    assert get_kv_cache_limit(name) is None                                       # the owner unlinks its segment


def test_stale_controller_write_is_repaired_by_the_watcher(cpu_ops, monkeypatch):
    """With no page event to republish, the 10 Hz watcher re-asserts the engine's own fields."""
    ops = cpu_ops
    import kvcached_amd.kv_cache_manager as kcm
    monkeypatch.setattr(kcm, "PAGE_PREALLOC_ENABLED", True)                        # the watcher thread runs with it
    ops.create_kv_tensors(64 * PAGE * 2, 1, "cpu", 2, 2, 0, False)
    m = kcm.KVCacheManager(num_blocks=64 * 64, block_size=16, cell_size=2048, num_layers=2)
    assert m._post_init_done.wait(10)
    pa = m.page_allocator
    name, unit = pa._ipc_name(), PAGE * 2 * 2
    ids = m.alloc(64 * 3)
    time.sleep(0.3)                                                               # the prealloc refill has settled

    def truth():
        return MemInfoStruct(64 * unit, pa.get_num_inuse_pages() * unit, pa.get_num_reserved_pages() * unit)
    assert get_kv_cache_limit(name) == truth()
    with RwLockedShm(name, 24, RwLockedShm.WLOCK) as mm:
        MemInfoStruct(64 * unit, 12345, 678).write_to_buffer(mm)
    t0 = time.time()
    while get_kv_cache_limit(name) != truth() and time.time() - t0 < 3:
        time.sleep(0.02)
    assert get_kv_cache_limit(name) == truth()
    m.free(ids)
    del m, pa
