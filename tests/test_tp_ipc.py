"""The multi-GPU leg of the path on CPU: scheduler -> worker fan-out over Unix sockets (same wire
format as the reference) and the SPMD CollectiveFanout over torch.distributed (gloo, world_size 2;
on the GPU box the same code runs over RCCL)."""
import os
import socket
import sys
import threading
import time

import pytest
import torch
import torch.multiprocessing as mp

import kvc_testlib as T

MiB = 1 << 20
PAGE = 2 * MiB


@pytest.fixture()
def cpu_pool():
    from kvcached_amd import capi, vmm_ops
    vmm_ops.init_kvcached("cpu", PAGE, False)
    vmm_ops.create_kv_tensors(32 * MiB, 1, "cpu", 2, 2, 0, False)
    capi.reset_stats()
    yield vmm_ops, capi
    vmm_ops.shutdown_kvcached()


def test_socket_paths_and_framing(tmp_path):
    from kvcached_amd import tp_ipc_util as tp
    assert tp.get_worker_socket_path(3).endswith("/w3.sock") and "/pp2/" in tp.get_worker_socket_path(1, 2)
    assert tp.SOCKET_DIR.startswith("/tmp/kvcached-tp-")
    a, b = socket.socketpair()
    tp.send_msg(a, {"cmd": "map_to_kv_tensors", "offsets": [0, PAGE], "group_id": 0})
    assert tp.recv_msg(b) == {"cmd": "map_to_kv_tensors", "offsets": [0, PAGE], "group_id": 0}
    # wire format: 4-byte big-endian length + pickle (what the reference's clients/workers speak)
    import pickle
    payload = pickle.dumps({"status": "success"})
    a.sendall(len(payload).to_bytes(4, "big") + payload)
    assert tp.recv_msg(b) == {"status": "success"}
    a.close()
    with pytest.raises(ConnectionError):
        tp.recv_msg(b)


def test_fd_passing_over_unix_socket(tmp_path):
    from kvcached_amd import tp_ipc_util as tp
    a, b = socket.socketpair(socket.AF_UNIX, socket.SOCK_STREAM)
    files = []
    for i in range(5):
        f = open(tmp_path / f"f{i}", "w+")
        f.write(f"payload {i}")
        f.flush()
        files.append(f)
    tp.send_fds(a, {"n": 5}, [f.fileno() for f in files])
    msg, fds = tp.recv_fds(b, 5)
    assert msg == {"n": 5} and len(fds) == 5
    for i, fd in enumerate(fds):
        assert os.pread(fd, 100, 0) == f"payload {i}".encode()
        os.close(fd)


def test_unix_socket_broadcast_to_two_workers(cpu_pool):
    """Two listener threads stand in for two TP workers; the scheduler-side API is the reference's."""
    ops, capi = cpu_pool
    from kvcached_amd import tp_ipc_util as tp
    servers = [tp.start_worker_listener_thread(r, pp_rank=0) for r in range(2)]
    try:
        assert tp.broadcast_kv_tensors_created(2) is True
        tp.broadcast_map_to_kv_tensors(2, [0, 2 * PAGE], pp_rank=0, group_id=0)
        # every "worker" executed the command against this process's allocator: the second one finds the
        # slots already mapped (logged, tolerated) — 2 offsets x 2 layers x K/V backed once
        assert capi.get_stats()["pages_mapped"] == 8
        tp.broadcast_unmap_from_kv_tensors(2, [0, 2 * PAGE])
        assert capi.get_stats()["pages_unmapped"] == 8
        t0 = time.perf_counter()
        for _ in range(200):                      # persistent connections: no connect/asyncio per call
            tp.broadcast_kv_tensors_created(2)
        per_call_us = (time.perf_counter() - t0) / 200 * 1e6
        assert per_call_us < 2000, per_call_us
        with pytest.raises(RuntimeError, match="failed to map"):
            tp.broadcast_map_to_kv_tensors(2, [12345])      # invalid offset -> worker error -> RuntimeError
        # a reference-style one-shot client (connect, one message, close) is still served
        s = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
        s.connect(tp.get_worker_socket_path(1))
        tp.send_msg(s, {"cmd": "kv_tensors_created", "group_id": 0})
        assert tp.recv_msg(s) == {"status": "success", "created": True}
        tp.send_msg(s, {"cmd": "nonsense"})
        assert tp.recv_msg(s)["status"] == "error"
        s.close()
    finally:
        tp._channels.close()
        for srv in servers:
            srv.close()


def test_manager_drives_workers_through_broadcast_callbacks(cpu_pool, monkeypatch):
    """world_size 2: KVCacheManager installs the broadcast callbacks itself; page offsets reach the workers."""
    ops, capi = cpu_pool
    import kvcached_amd.kv_cache_manager as kcm
    from kvcached_amd import tp_ipc_util as tp
    servers = [tp.start_worker_listener_thread(r) for r in range(2)]
    T.set_product_phys_pages(1 << 30, PAGE, 2, 2)
    try:
        m = kcm.KVCacheManager(num_blocks=512, block_size=16, cell_size=2048, num_layers=2, world_size=2)
        assert m._post_init_done.wait(10)
        got = m.alloc(100)
        assert got == list(range(100))
        assert capi.get_stats()["pages_mapped"] == 2 * 2 * 2     # 2 page ids x 2 layers x K/V
        m.free(got)
        m.trim()
        assert capi.get_stats()["pages_unmapped"] == 8
        del m
    finally:
        capi.set_mem_info_override(0, 0)
        tp._channels.close()
        for srv in servers:
            srv.close()


# ------------------------------------------------------------------ collective fan-out, 2 ranks, gloo
def _collective_worker(rank, world, port, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), KVCACHED_LOG_LEVEL="ERROR",
                          KVCACHED_IPC_NAME=f"kvc_test_coll_{port}_{rank}")
        sys.path.insert(0, T.REPO)
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from kvcached_amd import capi, vmm_ops
        from kvcached_amd.tp_ipc_util import CollectiveFanout
        vmm_ops.init_kvcached("cpu", PAGE, False)
        vmm_ops.create_kv_tensors(32 * MiB, 1, "cpu", 2, 2, 0, False)
        capi.reset_stats()
        fan = CollectiveFanout()
        # only rank 0 knows the offsets; every rank ends up mapping exactly those
        offs = fan.map_to_kv_tensors([PAGE, 3 * PAGE, 5 * PAGE] if rank == 0 else [])
        mapped = capi.get_stats()["pages_mapped"]
        fan.unmap_from_kv_tensors([3 * PAGE] if rank == 0 else [7 * PAGE])
        unmapped = capi.get_stats()["pages_unmapped"]
        # a failure on ANY rank is seen by every rank (status all-reduce)
        failed = False
        try:
            if rank == 1:
                vmm_ops.shutdown_kvcached()   # rank 1 loses its allocator -> its local map raises
            fan.map_to_kv_tensors([7 * PAGE] if rank == 0 else [])
        except RuntimeError:
            failed = True
        q.put((rank, offs, mapped, unmapped, failed))
        if rank == 0:
            vmm_ops.shutdown_kvcached()
        dist.destroy_process_group()
    except Exception as e:  # surface child errors in the parent
        q.put((rank, "ERR", repr(e), 0, False))


def _deferred_worker(rank, world, port, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), KVCACHED_LOG_LEVEL="ERROR",
                          KVCACHED_IPC_NAME=f"kvc_test_defer_{port}_{rank}")
        sys.path.insert(0, T.REPO)
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from kvcached_amd import capi, vmm_ops
        from kvcached_amd.tp_ipc_util import CollectiveFanout
        vmm_ops.init_kvcached("cpu", PAGE, False)
        vmm_ops.create_kv_tensors(32 * MiB, 1, "cpu", 2, 2, 0, False)
        capi.reset_stats()
        fan = CollectiveFanout(deferred_status=True, status_every=4)
        seen = []
        for i in range(5):                                      # 10 calls back to back: two windows close on the way
            seen.append([int(o) for o in fan.map_to_kv_tensors([i * PAGE, 7 * PAGE] if rank == 0 else [])])
            fan.unmap_from_kv_tensors([7 * PAGE, i * PAGE] if rank == 0 else [])
        fan.finish()
        st = capi.get_stats()
        events = []
        if rank == 1:
            vmm_ops.shutdown_kvcached()                         # rank 1 loses its allocator: its next local map fails
        try:
            fan.map_to_kv_tensors([6 * PAGE] if rank == 0 else [])        # raises on rank 1 at once; returns on rank 0
            events.append("call returned")
        except RuntimeError:
            events.append("call raised")
        try:
            fan.finish()                                                   # ... and here every rank learns of it
            events.append("finish returned")
        except RuntimeError:
            events.append("finish raised")
        q.put((rank, seen, st["pages_mapped"], st["pages_unmapped"], events))
        if rank == 0:
            vmm_ops.shutdown_kvcached()
        dist.destroy_process_group()
    except Exception as e:
        q.put((rank, "ERR", repr(e), 0, []))


def test_collective_fanout_with_the_agreement_pipelined():
    """deferred_status (bench.py at N > 1): the outcome of the local (un)maps goes into one all-reduce per window of calls, read
    when the next window closes or in finish(). Every rank still maps exactly rank 0's offsets; a rank's failure raises on that
    rank at once and on EVERY rank in finish() at the latest."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = [ctx.Process(target=_deferred_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    for rank, seen, mapped, unmapped, events in results:
        assert seen == [[i * PAGE, 7 * PAGE] for i in range(5)], results
        assert mapped == 10 * 2 * 2 and unmapped == 10 * 2 * 2
        assert events == (["call returned", "finish raised"] if rank == 0 else ["call raised", "finish raised"]), (rank, events)


def test_collective_fanout_two_ranks_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    import socket
    with socket.socket() as sk:          # a port nobody listens on right now
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = [ctx.Process(target=_collective_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    for rank, offs, mapped, unmapped, failed in results:
        assert offs == [PAGE, 3 * PAGE, 5 * PAGE], results
        assert mapped == 3 * 2 * 2 and unmapped == 1 * 2 * 2
        assert failed is True


# ------------------------------------------------------------------ persistent channels must not lose request/reply sync
def test_a_failing_rank_does_not_leave_stale_replies_behind(cpu_pool):
    """The connections to the workers are persistent (the reference opens one per message). If rank 0 answers "error"
    and the caller raised before reading rank 1's reply, that reply would be taken for the answer to the NEXT request -
    the scheduler would believe pages mapped that are not. Every rank that was written to is read before anything is
    raised (tp_ipc_util.py request_all)."""
    ops, capi = cpu_pool
    from kvcached_amd import tp_ipc_util as tp
    # rank 0: a worker that refuses every map and serves everything else; rank 1: the real listener
    path0 = tp.get_worker_socket_path(0)
    os.makedirs(os.path.dirname(path0), exist_ok=True)
    if os.path.exists(path0):
        os.remove(path0)
    srv0 = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
    srv0.bind(path0)
    srv0.listen()
    seen0 = []

    def fake_rank0():
        while True:
            try:
                conn, _ = srv0.accept()
            except OSError:
                return
            with conn:
                while True:
                    try:
                        msg = tp.recv_msg(conn)
                    except (ConnectionError, OSError):
                        break
                    seen0.append(msg["cmd"])
                    if msg["cmd"] == "map_to_kv_tensors":
                        tp.send_msg(conn, {"status": "error", "message": "out of memory on this rank"})
                    else:
                        tp.send_msg(conn, {"status": "success", "created": True})

    threading.Thread(target=fake_rank0, daemon=True).start()
    srv1 = tp.start_worker_listener_thread(1)
    try:
        with pytest.raises(RuntimeError, match="Worker 0 failed to map"):
            tp.broadcast_map_to_kv_tensors(2, [0])
        assert capi.get_stats()["pages_mapped"] == 4            # rank 1 did map (2 layers x K/V): its reply was waiting
        # the next exchange is answered by its own replies: rank 1's "created" flag, not its stale map reply
        assert tp.broadcast_kv_tensors_created(2) is True
        tp.broadcast_unmap_from_kv_tensors(2, [0])
        assert capi.get_stats()["pages_unmapped"] == 4
        assert seen0 == ["map_to_kv_tensors", "kv_tensors_created", "unmap_from_kv_tensors"]
        # a rank whose connection breaks mid-exchange is dropped and reconnected, the others are still heard out
        srv0.close()
        with pytest.raises(RuntimeError, match="Worker 0 failed"):
            for _ in range(3):                                   # the kernel may accept one more write on the dead socket
                tp.broadcast_map_to_kv_tensors(2, [2 * PAGE])
        assert tp._channels.request_all(2, 0, {"cmd": "kv_tensors_created", "group_id": 0}, "check", ranks=(1,))[0]["created"] is True
    finally:
        tp._channels.close()
        srv0.close()
        srv1.close()
