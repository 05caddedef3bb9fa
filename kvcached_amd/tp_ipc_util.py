"""Fan-out of map/unmap commands from the scheduler to the tensor-parallel workers.

Same public API as the reference's kvcached/tp_ipc_util.py (socket naming :16-53, framing :61-93,
worker listener :96-145, broadcast_* wrappers :250-267), two transports:

1. Unix-domain sockets (scheduler outside the TP process group — vLLM V1's EngineCore). Wire
   format unchanged: 4-byte big-endian length + pickle(dict), one socket per rank at
   /tmp/kvcached-tp-<ipc>-<hash>/[pp<k>/]w<rank>.sock. What changed is the cost per call: the
   reference opens a fresh connection to every rank and spins up an asyncio event loop for every
   message (2.1 ms per page id at TP=4, benchmarks/bench_tp_ipc/README.md:162); here connections
   are kept open, the request is written to all ranks before any reply is awaited (workers map
   in parallel), and no event loop is involved. The listener still accepts one-shot clients.

2. CollectiveFanout (all ranks call together — SGLang-style SPMD schedulers, bench.py): rank 0's
   offsets are broadcast as one int64 vector over the TP group's torch.distributed backend
   (RCCL over xGMI on GPUs, gloo in the CPU tests), every rank maps locally, one all-reduce(min)
   collects the status. Payloads are < 10 KB: latency-bound, not link-bound.

Shared physical pool (north-star addition, `send_fds`/`recv_fds`): rank 0 exports one POSIX fd per
backed slot (hipMemExportToShareableHandle) and ships them with SCM_RIGHTS over the same Unix
sockets — fds cannot travel through RCCL; peers import and map them.
"""
from __future__ import annotations

import array
import os
import pickle
import socket
import threading
import uuid
from typing import Any, Dict, List, Optional, Sequence, Tuple, cast

from kvcached_amd.utils import DEFAULT_IPC_NAME
from kvcached_amd.vmm_ops import kv_tensors_created, map_to_kv_tensors, unmap_from_kv_tensors

Message = Dict[str, Any]


def _get_socket_dir_name() -> str:
    """Readable IPC name + a short deterministic hash, so all workers of one engine agree."""
    suffix = uuid.uuid5(uuid.NAMESPACE_DNS, DEFAULT_IPC_NAME).hex[:8]
    return f"kvcached-tp-{DEFAULT_IPC_NAME}-{suffix}"


SOCKET_DIR = os.path.join("/tmp", _get_socket_dir_name())


def get_worker_socket_path(rank: int, pp_rank: int = 0) -> str:
    """w<rank>.sock, under pp<k>/ for pipeline stage k > 0; must fit sun_path (108 chars)."""
    parts = [SOCKET_DIR] + ([f"pp{pp_rank}"] if pp_rank > 0 else []) + [f"w{rank}.sock"]
    path = os.path.join(*parts)
    if len(path) > 108:
        raise RuntimeError(f"Socket path too long ({len(path)} chars, max 108): {path}")
    return path


# ------------------------------------------------------------------ framing
def send_msg(sock: socket.socket, msg: Message) -> None:
    data = pickle.dumps(msg)
    sock.sendall(len(data).to_bytes(4, 'big') + data)


def _recv_exact(sock: socket.socket, n: int) -> bytes:
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(n - len(buf))
        if not chunk:
            raise ConnectionError("Socket connection closed" if not buf else
                                  "Socket connection closed while receiving data")
        buf += chunk
    return bytes(buf)


def recv_msg(sock: socket.socket) -> Message:
    length = int.from_bytes(_recv_exact(sock, 4), 'big')
    if length <= 0:
        raise ValueError("Received invalid length for message")
    return cast(Message, pickle.loads(_recv_exact(sock, length)))


def send_fds(sock: socket.socket, msg: Message, fds: Sequence[int]) -> None:
    """A framed message whose first byte carries `fds` as SCM_RIGHTS ancillary data."""
    data = pickle.dumps(msg)
    payload = len(data).to_bytes(4, 'big') + data
    MAX = 250  # SCM_MAX_FD is 253
    first = list(fds[:MAX])
    sock.sendmsg([payload], [(socket.SOL_SOCKET, socket.SCM_RIGHTS, array.array("i", first))] if first else [])
    for i in range(MAX, len(fds), MAX):  # further fds ride on 1-byte continuation packets
        sock.sendmsg([b"\x00"], [(socket.SOL_SOCKET, socket.SCM_RIGHTS, array.array("i", list(fds[i:i + MAX])))])


def recv_fds(sock: socket.socket, n_fds: int) -> Tuple[Message, List[int]]:
    fds: List[int] = []

    def take(bufsize: int) -> bytes:
        data, anc, _, _ = sock.recvmsg(bufsize, socket.CMSG_LEN(256 * 4))
        for level, typ, cdata in anc:
            if level == socket.SOL_SOCKET and typ == socket.SCM_RIGHTS:
                a = array.array("i")
                a.frombytes(cdata[:len(cdata) - (len(cdata) % a.itemsize)])
                fds.extend(a)
        if not data:
            raise ConnectionError("Socket connection closed")
        return data

    head = take(4)
    while len(head) < 4:
        head += take(4 - len(head))
    length = int.from_bytes(head, 'big')
    body = b""
    while len(body) < length:
        body += take(length - len(body))
    while len(fds) < n_fds:
        take(1)
    return cast(Message, pickle.loads(body)), fds


# ------------------------------------------------------------------ worker side
def _execute(msg: Message) -> Message:
    group_id: int = msg.get("group_id", 0)
    cmd = msg["cmd"]
    if cmd == "map_to_kv_tensors":
        map_to_kv_tensors(msg["offsets"], group_id=group_id)
        return {"status": "success"}
    if cmd == "unmap_from_kv_tensors":
        unmap_from_kv_tensors(msg["offsets"], group_id=group_id)
        return {"status": "success"}
    if cmd == "kv_tensors_created":
        return {"status": "success", "created": bool(kv_tensors_created(group_id=group_id))}
    return {"status": "error", "message": "Unknown command"}


def _serve_connection(rank: int, conn: socket.socket) -> None:
    """Messages until the peer hangs up (persistent clients) — or just one (reference clients)."""
    with conn:
        while True:
            try:
                msg = recv_msg(conn)
            except (ConnectionError, OSError):
                return
            try:
                if msg.get("cmd") == "map_imported_slots":  # shared-pool: fds follow as SCM_RIGHTS
                    from kvcached_amd import capi
                    _, fds = recv_fds(conn, msg["n_fds"])
                    try:
                        capi.map_imported_slots(msg["offsets"], fds, msg.get("group_id", 0))
                    finally:
                        for fd in fds:
                            os.close(fd)
                    reply: Message = {"status": "success"}
                else:
                    reply = _execute(msg)
            except Exception as e:
                print(f"Worker {rank} error processing message: {e}")
                reply = {"status": "error", "message": str(e)}
            try:
                send_msg(conn, reply)
            except OSError:
                return


def start_worker_listener_thread(rank: int, pp_rank: int = 0):
    """Bind w<rank>.sock and serve map/unmap/created requests on daemon threads."""
    socket_path = get_worker_socket_path(rank, pp_rank)
    os.makedirs(os.path.dirname(socket_path), exist_ok=True)
    if os.path.exists(socket_path):
        try:
            os.remove(socket_path)
        except OSError as e:
            print(f"Error removing existing socket file {socket_path}: {e}")
    server = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
    server.bind(socket_path)
    server.listen()

    def accept_loop():
        print(f"Worker {rank} IPC listener started at {socket_path}")
        while True:
            try:
                conn, _ = server.accept()
            except OSError:
                return
            threading.Thread(target=_serve_connection, args=(rank, conn), daemon=True).start()

    t = threading.Thread(target=accept_loop, daemon=True)
    t.start()
    return server


# ------------------------------------------------------------------ scheduler side
class _Channels:
    """Persistent client sockets, keyed by (rank, pp_rank); thread-safe."""

    def __init__(self):
        self._socks: Dict[Tuple[int, int], socket.socket] = {}
        self._lock = threading.Lock()

    def _connect(self, key) -> socket.socket:
        s = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
        s.connect(get_worker_socket_path(*key))
        self._socks[key] = s
        return s

    def drop(self, key) -> None:
        s = self._socks.pop(key, None)
        if s is not None:
            try:
                s.close()
            except OSError:
                pass

    def request_all(self, tp_size: int, pp_rank: int, msg: Message, what: str) -> List[Message]:
        """Write `msg` to every rank, then collect every reply; one reconnect attempt per rank."""
        data = pickle.dumps(msg)
        frame = len(data).to_bytes(4, 'big') + data
        with self._lock:
            keys = [(r, pp_rank) for r in range(tp_size)]
            for key in keys:
                for attempt in (0, 1):
                    try:
                        (self._socks.get(key) or self._connect(key)).sendall(frame)
                        break
                    except OSError as e:
                        self.drop(key)
                        if attempt:
                            raise RuntimeError(f"Worker {key[0]} failed to {what}: {e}")
            replies: List[Message] = []
            for key in keys:
                try:
                    reply = recv_msg(self._socks[key])
                except Exception as e:
                    self.drop(key)
                    raise RuntimeError(f"Worker {key[0]} failed to {what}: {e}")
                if not isinstance(reply, dict) or reply.get("status") != "success":
                    raise RuntimeError(f"Worker {key[0]} failed to {what}: {reply}")
                replies.append(reply)
            return replies

    def close(self) -> None:
        with self._lock:
            for key in list(self._socks):
                self.drop(key)


_channels = _Channels()


def broadcast_map_to_kv_tensors(tp_size: int, offsets: List[int], pp_rank: int = 0, group_id: int = 0) -> None:
    _channels.request_all(tp_size, pp_rank, {"cmd": "map_to_kv_tensors", "offsets": list(offsets),
                                             "group_id": group_id}, "map")


def broadcast_unmap_from_kv_tensors(tp_size: int, offsets: List[int], pp_rank: int = 0, group_id: int = 0) -> None:
    _channels.request_all(tp_size, pp_rank, {"cmd": "unmap_from_kv_tensors", "offsets": list(offsets),
                                             "group_id": group_id}, "unmap")


def broadcast_kv_tensors_created(tp_size: int, pp_rank: int = 0, group_id: int = 0) -> bool:
    replies = _channels.request_all(tp_size, pp_rank, {"cmd": "kv_tensors_created", "group_id": group_id},
                                    "check KV tensors created")
    return all(r.get("created", False) for r in replies)


def share_mapped_slots(tp_size: int, offsets: List[int], pp_rank: int = 0, group_id: int = 0,
                       src_rank: int = 0) -> None:
    """Shared-pool mode: the calling process (rank `src_rank`, which has just backed `offsets`
    with exportable handles) exports them and every other rank maps the same physical pages."""
    from kvcached_amd import capi
    fds = capi.export_mapped_slots(offsets, group_id)
    try:
        msg = {"cmd": "map_imported_slots", "offsets": list(offsets), "group_id": group_id, "n_fds": len(fds)}
        with _channels._lock:
            peers = [(r, pp_rank) for r in range(tp_size) if r != src_rank]
            for key in peers:
                s = _channels._socks.get(key) or _channels._connect(key)
                send_msg(s, msg)
                send_fds(s, {"fds": len(fds)}, fds)
            for key in peers:
                reply = recv_msg(_channels._socks[key])
                if reply.get("status") != "success":
                    raise RuntimeError(f"Worker {key[0]} failed to map shared slots: {reply}")
    finally:
        for fd in fds:
            os.close(fd)


# ------------------------------------------------------------------ collective transport
CMD_MAP, CMD_UNMAP = 1, 2


class CollectiveFanout:
    """SPMD fan-out over a torch.distributed group: every rank calls the same method; rank
    `src`'s offsets are authoritative. Backend nccl (= RCCL over xGMI) moves the vector through
    device memory, gloo through host memory."""

    MAX_OFFSETS = 4093  # header + payload = 4096 int64 = 32 KiB, one fixed-size broadcast

    def __init__(self, group=None, src: int = 0, device: Optional[str] = None):
        import torch
        import torch.distributed as dist
        self._torch, self._dist = torch, dist
        self.group, self.src = group, src
        self.rank = dist.get_rank(group)
        self.world_size = dist.get_world_size(group)
        backend = dist.get_backend(group)
        if device is None:
            device = f"cuda:{torch.cuda.current_device()}" if backend == "nccl" else "cpu"
        self.device = device
        self._buf = torch.zeros(3 + self.MAX_OFFSETS, dtype=torch.int64, device=device)
        self._status = torch.zeros(1, dtype=torch.int64, device=device)
        # host staging (pinned when the collective runs through device memory): the message is packed and unpacked with
        # numpy, one copy each way - the Python-list <-> tensor conversions were a third of the call at 1024 offsets
        on_gpu = str(device).startswith("cuda")
        self._stage = torch.zeros(3 + self.MAX_OFFSETS, dtype=torch.int64, pin_memory=on_gpu) if on_gpu else self._buf
        self._stage_np = self._stage.numpy()
        self._status_host = torch.zeros(1, dtype=torch.int64, pin_memory=on_gpu) if on_gpu else self._status
        self._on_gpu = on_gpu

    def _exchange(self, cmd: int, offsets: Sequence[int], group_id: int) -> Tuple[int, int, List[int]]:
        torch, dist = self._torch, self._dist
        if self.rank == self.src:
            n = len(offsets)
            if n > self.MAX_OFFSETS:
                raise ValueError(f"at most {self.MAX_OFFSETS} offsets per collective call")
            st = self._stage_np
            st[0], st[1], st[2] = cmd, group_id, n
            st[3:3 + n] = offsets
            if self._on_gpu:
                self._buf[:3 + n].copy_(self._stage[:3 + n], non_blocking=True)
        dist.broadcast(self._buf, src=dist.get_global_rank(self.group, self.src) if self.group else self.src,
                       group=self.group)
        if self._on_gpu and self.rank != self.src:
            # the payload length is not known before the copy: bring the whole (32 KiB) message over, once
            self._stage.copy_(self._buf, non_blocking=True)
            torch.cuda.current_stream().synchronize()
        st = self._stage_np
        n = int(st[2])
        return int(st[0]), int(st[1]), st[3:3 + n].tolist()

    def _finish(self, ok: bool) -> None:
        if self._on_gpu:
            self._status_host[0] = 1 if ok else 0
            self._status.copy_(self._status_host, non_blocking=True)
        else:
            self._status[0] = 1 if ok else 0
        self._dist.all_reduce(self._status, op=self._dist.ReduceOp.MIN, group=self.group)
        if int(self._status.item()) != 1:
            raise RuntimeError("a tensor-parallel rank failed to (un)map KV pages")

    def run(self, cmd: int, offsets: Sequence[int] = (), group_id: int = 0) -> List[int]:
        """Broadcast (cmd, group_id, offsets) from `src`, apply locally, agree on success."""
        cmd, group_id, offs = self._exchange(cmd, offsets, group_id)
        ok = True
        try:
            if cmd == CMD_MAP:
                ok = bool(map_to_kv_tensors(offs, group_id=group_id))
            elif cmd == CMD_UNMAP:
                ok = bool(unmap_from_kv_tensors(offs, group_id=group_id))
            else:
                ok = False
        except Exception as e:
            print(f"rank {self.rank}: collective (un)map failed: {e}")
            ok = False
        self._finish(ok)
        return offs

    def map_to_kv_tensors(self, offsets: Sequence[int] = (), group_id: int = 0) -> List[int]:
        return self.run(CMD_MAP, offsets, group_id)

    def unmap_from_kv_tensors(self, offsets: Sequence[int] = (), group_id: int = 0) -> List[int]:
        return self.run(CMD_UNMAP, offsets, group_id)
