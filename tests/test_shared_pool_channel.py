"""SharedPoolChannel (kvcached_amd/tp_ipc_util.py): the shared pool between the ranks of one torch.distributed group - offsets over
the group's collective, one file descriptor per slot over SCM_RIGHTS, a status all-reduce at the end. 3 ranks over gloo with
stand-ins for the library's export / import (the real ones need a GPU: tests/test_gpu_vmm.py, bench.py --gpus N): every peer
gets exactly the descriptors rank 0 exported, in order, for exactly the offsets rank 0 named; a failure on ANY rank - the
exporter's or one importer's - is an exception on EVERY rank, and the channel is usable afterwards (nothing is left unread)."""
import os
import socket
import sys

import torch.multiprocessing as mp

import kvc_testlib as T


def _rank(rank, world, port, ipc, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), KVCACHED_LOG_LEVEL="ERROR", KVCACHED_IPC_NAME=ipc)
        sys.path.insert(0, T.REPO)
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from kvcached_amd.tp_ipc_util import CollectiveFanout, SharedPoolChannel
        made, got = [], []

        def exporter(offs, gid):
            if gid == 13:
                raise RuntimeError("nothing to export [injected]")
            fds = []
            for o in offs:                                    # two "slots" per offset, each a file that names its origin
                for half in (0, 1):
                    fd = os.memfd_create("kvc_share_test")
                    os.write(fd, f"{o}:{half}:{gid}".encode())
                    fds.append(fd)
            made.append(len(fds))
            return fds

        def importer(offs, fds, gid, meta):
            if gid == 7 and rank == 2:
                raise RuntimeError("import refused [injected]")
            got.append((list(offs), [os.pread(fd, 64, 0).decode() for fd in fds], gid))

        chan = SharedPoolChannel(CollectiveFanout(device="cpu"), exporter=exporter, importer=importer, timeout=30)
        res = []
        for offs, gid in (([4096, 0, 8192], 0), ([2 << 20], 7), ([1, 2], 13), (list(range(0, 400 * 4096, 4096)), 1)):
            try:
                t = chan.share(offs if rank == 0 else (), gid)
                res.append(("ok", sorted(t)))
            except RuntimeError as e:
                res.append(("raised", str(e)[:60]))
        chan.close()
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, res, made, got))
    except Exception as e:
        import traceback
        q.put((rank, "ERR", repr(e), traceback.format_exc()))


def test_offsets_travel_through_the_collective_and_descriptors_through_scm_rights():
    ctx = mp.get_context("spawn")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    q = ctx.Queue()
    world = 3
    procs = [ctx.Process(target=_rank, args=(r, world, port, f"kvc_share_{os.getpid()}", q)) for r in range(world)]
    for p in procs:
        p.start()
    out = {}
    for _ in procs:
        item = q.get(timeout=120)
        assert item[1] != "ERR", item
        out[item[0]] = item[1:]
    for p in procs:
        p.join(timeout=30)
    for r in range(world):
        res = out[r][0]
        assert [x[0] for x in res] == ["ok", "raised", "raised", "ok"], (r, res)   # rank 2's refusal and rank 0's failed export reach everybody
    assert out[0][1] == [6, 2, 800]                                              # what rank 0 exported (the failed export made nothing)
    for r in (1, 2):
        got = out[r][2]
        first = got[0]
        assert first[0] == [4096, 0, 8192] and first[2] == 0
        assert first[1] == [f"{o}:{h}:0" for o in (4096, 0, 8192) for h in (0, 1)]  # the very descriptors, in order
        big = got[-1]
        assert len(big[0]) == 400 and len(big[1]) == 800 and big[1][-1] == f"{399 * 4096}:1:1"   # beyond one SCM_RIGHTS packet (250 fds)
    assert len(out[1][2]) == 3 and len(out[2][2]) == 2                           # rank 2 imported nothing in the round it refused
