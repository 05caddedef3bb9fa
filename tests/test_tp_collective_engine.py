"""The engine path over the collective transport (KVCACHED_TP_TRANSPORT=collective): a scheduler OUTSIDE the
workers' process group (vLLM V1's EngineCore) drives a KVCacheManager with world_size 2; its broadcast callbacks - the
unchanged `broadcast_*` names - make one Unix hop to rank 0's listener, and rank 0 relays every command to its TP group
with CollectiveFanout (gloo here, RCCL over xGMI on GPUs). Both ranks must execute exactly the page offsets the
reference's own recording holds for the same alloc/free trace, in the same order.
Reference dispatch: csrc/page_allocator.cpp:633-635 -> kvcached/tp_ipc_util.py:173-192 (scheduler), :96-145 (worker)."""
import json
import os
import socket
import sys
import time

import pytest
import torch.multiprocessing as mp

import kvc_testlib as T

PAGE = 2 << 20


def _worker(rank, world, port, ipc_name, cfg, conn):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), KVCACHED_LOG_LEVEL="ERROR", KVCACHED_IPC_NAME=ipc_name,
                          KVCACHED_TP_TRANSPORT="collective")
        sys.path.insert(0, T.REPO)
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from kvcached_amd import tp_ipc_util as tp
        from kvcached_amd import vmm_ops
        vmm_ops.init_kvcached("cpu", PAGE, False)
        mem = cfg["num_blocks"] * cfg["block_size"] * cfg["cell_size"]
        mem = (mem + 2 * PAGE - 1) // (2 * PAGE) * (2 * PAGE)
        log = []
        real_map, real_unmap = tp.map_to_kv_tensors, tp.unmap_from_kv_tensors

        def logged_map(offs, group_id=0):
            log.append([0, [int(o) for o in offs]])
            return real_map(offs, group_id=group_id)

        def logged_unmap(offs, group_id=0):
            log.append([1, [int(o) for o in offs]])
            return real_unmap(offs, group_id=group_id)

        tp.map_to_kv_tensors, tp.unmap_from_kv_tensors = logged_map, logged_unmap
        group = dist.new_group(list(range(world)), backend="gloo")    # a group of its own for the helper threads
        srv = tp.start_worker_listener_thread(rank)
        tp.start_collective_worker(group=group)
        conn.send("listening")
        assert conn.recv() == "create"                                 # the scheduler first sees "not created yet"
        vmm_ops.create_kv_tensors(mem * cfg["num_kv_buffers"], 1, "cpu", cfg["num_layers"], cfg["num_kv_buffers"], 0, False)
        conn.send("created")
        assert conn.recv() == "dump"
        conn.send(log)
        assert conn.recv() == "stop"
        tp.stop_collective_worker()
        srv.close()
        vmm_ops.shutdown_kvcached()
        dist.destroy_process_group()
        conn.send("bye")
    except Exception as e:   # surface child errors in the parent
        import traceback
        conn.send(("ERR", repr(e), traceback.format_exc()))


class _SchedulerSide(T.Adapter):
    """Just enough of kvc_testlib's adapter protocol for T.replay: the manager's own callbacks stay in place."""

    def __init__(self, m, geom):
        self.m, self.geom = m, geom

    def alloc(self, n): return self.m.alloc(n)
    def free(self, ids): self.m.free(ids)
    def try_to_reserve(self, n): return self.m.try_to_reserve(n)
    def free_reserved(self): self.m.free_reserved()
    def resize(self, mem): return self.m.resize(mem)
    def trim(self): self.m.trim()
    def set_phys(self, pages): T.set_product_phys_pages(pages, *self.geom)
    def available_size(self): return self.m.available_size()
    def snapshot(self): return []
    def drain_events(self): return []


@pytest.mark.parametrize("case_name", ["partial_page_lifo", "resize_deferred_in_shrink"])
def test_manager_over_collective_transport_two_ranks_gloo(monkeypatch, case_name):
    case = next(c for c in json.load(open(os.path.join(T.REPO, "tests", "golden", "manager_small.json")))["cases"]
                if c["name"] == case_name)
    cfg = case["config"]
    assert case["full"] and not cfg["contiguous"]
    ipc = os.environ["KVCACHED_IPC_NAME"]
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ctx = mp.get_context("spawn")
    pipes, procs = [], []
    for r in range(2):
        a, b = ctx.Pipe()
        p = ctx.Process(target=_worker, args=(r, 2, port, ipc, cfg, b))
        p.start()
        pipes.append(a)
        procs.append(p)
    import kvcached_amd.kv_cache_manager as kcm
    from kvcached_amd import capi, vmm_ops
    from kvcached_amd import tp_ipc_util as tp
    monkeypatch.setenv("KVCACHED_TP_TRANSPORT", "collective")
    monkeypatch.setattr(kcm, "CONTIGUOUS_LAYOUT", False)
    monkeypatch.setattr(kcm, "BATCH_PAGE_ALLOC", False)    # call for call like the reference's recording
    try:
        for c in pipes:
            assert c.poll(120), "worker did not come up"
            got = c.recv()
            assert got == "listening", got
        vmm_ops.init_kvcached("cpu", PAGE, False)          # the scheduler's own process: bookkeeping only, no KV tensors
        geom = (PAGE, cfg["num_layers"], cfg["num_kv_buffers"])
        T.set_product_phys_pages(cfg["phys_pages"], *geom)
        assert tp.broadcast_kv_tensors_created(2) is False  # one collective: min over the ranks
        for c in pipes:
            c.send("create")
        for c in pipes:
            assert c.poll(60) and c.recv() == "created"
        m = kcm.KVCacheManager(cfg["num_blocks"], cfg["block_size"], cfg["cell_size"], cfg["num_layers"], world_size=2,
                               reserve_null_block=cfg["reserve_null_block"], num_kv_buffers=cfg["num_kv_buffers"])
        assert m._post_init_done.wait(30)                   # waits on broadcast_kv_tensors_created through the relay
        got = T.replay(_SchedulerSide(m, geom), case["ops"], full=True)
        want_results = [rec["r"] for rec in case["records"]]
        assert [rec["r"] for rec in got] == want_results    # block ids as recorded from the reference
        want_events = [e for rec in [case["init"]] + case["records"] for e in rec["e"]]
        assert want_events, "the trace maps nothing: it would prove nothing"
        # a failure on a rank reaches the scheduler as an exception (status all-reduce -> rank 0's reply)
        with pytest.raises(RuntimeError, match="failed to map"):
            tp.broadcast_map_to_kv_tensors(2, [12345])
        del m
        logs = []
        for c in pipes:
            c.send("dump")
            assert c.poll(60)
            logs.append(c.recv())
        for rank, log in enumerate(logs):
            assert isinstance(log, list), log
            assert log[-1] == [0, [12345]]                  # the bad request was broadcast too (and refused by every rank)
            assert log[:-1] == want_events, f"rank {rank} executed other offsets than the reference recorded"
        for c in pipes:
            c.send("stop")
        for c in pipes:
            assert c.poll(60) and c.recv() == "bye"
    finally:
        tp._channels.close()
        vmm_ops.shutdown_kvcached()
        capi.set_mem_info_override(0, 0)
        for p in procs:
            p.join(20)
            if p.is_alive():
                p.kill()
