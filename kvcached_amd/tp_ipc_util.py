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

Engine wiring of (2) - KVCACHED_TP_TRANSPORT=collective: the scheduler of vLLM V1 lives outside the workers' process
group, so `broadcast_*` (unchanged names and signatures) make ONE Unix hop to rank 0's listener, and rank 0 relays the
command to its TP group with CollectiveFanout (`start_collective_worker()` on every worker, next to
`start_worker_listener_thread()`): one RCCL broadcast + one status all-reduce instead of tp_size socket round trips
(reference dispatch: csrc/page_allocator.cpp:633-635 -> kvcached/tp_ipc_util.py:173-192 -> worker loop :96-145).
KVCACHED_SHARED_POOL=1 on top: rank 0 backs the slots and ships its pages to the peers (`share_mapped_slots`).

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
def tp_transport() -> str:
    """"unix" (default; the reference's shape: the scheduler talks to every rank's socket) or "collective" (scheduler ->
    rank 0 over its socket, rank 0 -> TP group over torch.distributed). Read at call time: tests flip it."""
    t = os.environ.get("KVCACHED_TP_TRANSPORT", "unix").lower()
    if t not in ("unix", "collective"):
        raise ValueError("KVCACHED_TP_TRANSPORT must be 'unix' or 'collective'")
    return t


def shared_pool_enabled() -> bool:
    return os.environ.get("KVCACHED_SHARED_POOL", "0").lower() in ("1", "true", "yes", "on")


_collective: Optional["CollectiveFanout"] = None   # this worker's relay (the src rank of its TP group holds it)
_collective_lock = threading.Lock()
_follower: Optional[threading.Thread] = None


def _execute(msg: Message) -> Message:
    group_id: int = msg.get("group_id", 0)
    cmd = msg["cmd"]
    if msg.get("relay"):   # KVCACHED_TP_TRANSPORT=collective: this rank passes the command on to its TP group
        fan = _collective
        if fan is None:
            return {"status": "error", "message": "relay requested but start_collective_worker() has not run on this rank"}
        code = {"map_to_kv_tensors": CMD_MAP, "unmap_from_kv_tensors": CMD_UNMAP, "kv_tensors_created": CMD_CREATED}.get(cmd)
        if code is None:
            return {"status": "error", "message": "Unknown command"}
        with _collective_lock:
            if code == CMD_MAP and msg.get("shared_pool"):
                # rank 0 backs the slots with its own (exportable) pages, every peer maps those very pages
                map_to_kv_tensors(msg["offsets"], group_id=group_id)
                share_mapped_slots(fan.world_size, msg["offsets"], msg.get("pp_rank", 0), group_id, src_rank=fan.rank)
                return {"status": "success"}
            ok = fan.run(code, msg.get("offsets", ()), group_id, raise_on_failure=False)
        if code == CMD_CREATED:
            return {"status": "success", "created": ok}
        return {"status": "success"} if ok else {"status": "error", "message": "a tensor-parallel rank failed to " + cmd}
    if cmd == "map_to_kv_tensors":
        map_to_kv_tensors(msg["offsets"], group_id=group_id)
        return {"status": "success"}
    if cmd == "unmap_from_kv_tensors":
        unmap_from_kv_tensors(msg["offsets"], group_id=group_id)
        return {"status": "success"}
    if cmd == "kv_tensors_created":
        return {"status": "success", "created": bool(kv_tensors_created(group_id=group_id))}
    return {"status": "error", "message": "Unknown command"}


def start_collective_worker(group=None, src: int = 0, device: Optional[str] = None) -> "CollectiveFanout":
    """Every TP worker calls this once torch.distributed is up (KVCACHED_TP_TRANSPORT=collective). `group` should be a
    group of its own (dist.new_group over the TP ranks): its collectives are issued from helper threads and must not
    interleave with the engine's. The src rank keeps the fan-out for its socket listener to relay with; every other
    rank parks a daemon thread in the collective, applying what src broadcasts, until stop_collective_worker()."""
    global _collective, _follower
    fan = CollectiveFanout(group=group, src=src, device=device, park_on_host=True)
    if fan.rank == src:
        _collective = fan
    else:
        def follow():
            # a new thread's current device is 0: bind it to the GPU the fan-out's buffers live on, or the stream this
            # thread synchronises (and the RCCL work it enqueues) would be another GPU's
            if fan._on_gpu:
                fan._torch.cuda.set_device(fan._torch.device(fan.device))
            while fan.serve_one():
                pass
        _follower = threading.Thread(target=follow, name="kvcached-tp-follower", daemon=True)
        _follower.start()
    return fan


def stop_collective_worker(timeout: float = 10.0) -> None:
    """src rank: releases the followers (one last broadcast). Other ranks: wait for the follower thread to leave."""
    global _collective
    fan, _collective = _collective, None
    if fan is not None:
        with _collective_lock:
            fan.run(CMD_STOP, (), 0, raise_on_failure=False)
    if _follower is not None:
        _follower.join(timeout)


def _serve_connection(rank: int, conn: socket.socket) -> None:
    """Messages until the peer hangs up (persistent clients) — or just one (reference clients)."""
    with conn:
        while True:
            try:
                msg = recv_msg(conn)
            except (ConnectionError, OSError):
                return
            try:
                if msg.get("cmd") in ("map_imported_slots", "map_imported_page_ids"):  # shared-pool: fds follow as SCM_RIGHTS
                    from kvcached_amd import capi
                    _, fds = recv_fds(conn, msg["n_fds"])
                    try:
                        if msg["cmd"] == "map_imported_page_ids":   # one dmabuf per buffer of lanes + (fd index, lanes, lane) per page id
                            capi.map_imported_page_ids(msg["offsets"], fds, msg["meta"], msg.get("group_id", 0))
                        else:
                            capi.map_imported_slots(msg["offsets"], fds, msg.get("group_id", 0))
                    finally:
                        for fd in fds:
                            os.close(fd)
                    reply: Message = {"status": "success"}
                elif msg.get("cmd") == "share_mapped_slots":  # shared pool, unix transport: this rank ships its pages
                    share_mapped_slots(msg["tp_size"], msg["offsets"], msg.get("pp_rank", 0), msg.get("group_id", 0), src_rank=rank)
                    reply = {"status": "success"}
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

    def request_all(self, tp_size: int, pp_rank: int, msg: Message, what: str, ranks: Optional[Sequence[int]] = None) -> List[Message]:
        """Write `msg` to every rank (or to `ranks`), then collect every reply; one reconnect attempt per rank.
        The connections are persistent, so a reply left unread would be taken for the answer to the NEXT request: every
        rank that was written to is read, whatever the others answered, and a rank whose exchange broke is dropped
        (reconnected next time) - then the first failure is raised."""
        data = pickle.dumps(msg)
        frame = len(data).to_bytes(4, 'big') + data
        with self._lock:
            keys = [(r, pp_rank) for r in (range(tp_size) if ranks is None else ranks)]
            sent: List[Tuple[int, int]] = []
            failure: Optional[str] = None
            for key in keys:
                for attempt in (0, 1):
                    try:
                        (self._socks.get(key) or self._connect(key)).sendall(frame)
                        sent.append(key)
                        break
                    except OSError as e:
                        self.drop(key)
                        if attempt:
                            failure = failure or f"Worker {key[0]} failed to {what}: {e}"
                if failure:
                    break   # the ranks already written to still answer below
            replies: List[Message] = []
            for key in sent:
                try:
                    reply = recv_msg(self._socks[key])
                except Exception as e:
                    self.drop(key)
                    failure = failure or f"Worker {key[0]} failed to {what}: {e}"
                    continue
                if not isinstance(reply, dict) or reply.get("status") != "success":
                    failure = failure or f"Worker {key[0]} failed to {what}: {reply}"
                    continue
                replies.append(reply)
            if failure:
                raise RuntimeError(failure)
            return replies

    def close(self) -> None:
        with self._lock:
            for key in list(self._socks):
                self.drop(key)


_channels = _Channels()


def _fan_out(tp_size: int, pp_rank: int, msg: Message, what: str) -> List[Message]:
    if tp_transport() == "collective":   # one hop to rank 0, which relays to its TP group (RCCL broadcast + status all-reduce)
        return _channels.request_all(tp_size, pp_rank, dict(msg, relay=True, pp_rank=pp_rank), what, ranks=(0,))
    return _channels.request_all(tp_size, pp_rank, msg, what)


def broadcast_map_to_kv_tensors(tp_size: int, offsets: List[int], pp_rank: int = 0, group_id: int = 0) -> None:
    msg: Message = {"cmd": "map_to_kv_tensors", "offsets": list(offsets), "group_id": group_id}
    if shared_pool_enabled():
        if tp_transport() == "collective":   # rank 0 backs and ships its pages to the peers itself
            _fan_out(tp_size, pp_rank, dict(msg, shared_pool=True), "map")
        else:                                  # rank 0 backs, then this process asks it to share
            _channels.request_all(tp_size, pp_rank, msg, "map", ranks=(0,))
            _channels.request_all(tp_size, pp_rank, dict(msg, cmd="share_mapped_slots", tp_size=tp_size, pp_rank=pp_rank),
                                  "share mapped slots", ranks=(0,))
        return
    _fan_out(tp_size, pp_rank, msg, "map")


def broadcast_unmap_from_kv_tensors(tp_size: int, offsets: List[int], pp_rank: int = 0, group_id: int = 0) -> None:
    _fan_out(tp_size, pp_rank, {"cmd": "unmap_from_kv_tensors", "offsets": list(offsets), "group_id": group_id}, "unmap")


def broadcast_kv_tensors_created(tp_size: int, pp_rank: int = 0, group_id: int = 0) -> bool:
    replies = _fan_out(tp_size, pp_rank, {"cmd": "kv_tensors_created", "group_id": group_id}, "check KV tensors created")
    return all(r.get("created", False) for r in replies)


def share_mapped_slots(tp_size: int, offsets: List[int], pp_rank: int = 0, group_id: int = 0,
                       src_rank: int = 0) -> None:
    """Shared-pool mode: the calling process (rank `src_rank`, which has just backed `offsets`
    with exportable handles) exports them and every other rank maps the same physical pages."""
    from kvcached_amd import capi
    # page ids as units where the library backs them as lanes (one dmabuf per buffer: DESIGN.md §4.11), slot by slot otherwise
    # (single-row geometries, other backends, KVCACHED_EXPORTABLE_HANDLES=1)
    meta = None
    if capi.get_option(129) > 0:
        try:
            fds, meta = capi.export_page_ids(offsets, group_id)
        except capi.KvcError:
            meta = None
    if meta is None:
        fds = capi.export_mapped_slots(offsets, group_id)
    try:
        msg = {"cmd": "map_imported_slots", "offsets": list(offsets), "group_id": group_id, "n_fds": len(fds)}
        if meta is not None:
            msg.update(cmd="map_imported_page_ids", meta=meta)
        with _channels._lock:
            peers = [(r, pp_rank) for r in range(tp_size) if r != src_rank]
            sent, failure = [], None
            for key in peers:
                try:
                    s = _channels._socks.get(key) or _channels._connect(key)
                    send_msg(s, msg)
                    send_fds(s, {"fds": len(fds)}, fds)
                    sent.append(key)
                except OSError as e:
                    _channels.drop(key)   # half a request may be on the wire: this connection is of no use any more
                    failure = failure or f"Worker {key[0]} failed to map shared slots: {e}"
                    break
            for key in sent:   # every peer that got the request is heard out (see request_all)
                try:
                    reply = recv_msg(_channels._socks[key])
                except Exception as e:
                    _channels.drop(key)
                    failure = failure or f"Worker {key[0]} failed to map shared slots: {e}"
                    continue
                if reply.get("status") != "success":
                    failure = failure or f"Worker {key[0]} failed to map shared slots: {reply}"
            if failure:
                raise RuntimeError(failure)
    finally:
        for fd in fds:
            os.close(fd)


# ------------------------------------------------------------------ collective transport
CMD_MAP, CMD_UNMAP, CMD_CREATED, CMD_STOP, CMD_SHARE = 1, 2, 3, 4, 5


class CollectiveFanout:
    """SPMD fan-out over a torch.distributed group: every rank calls the same method; rank
    `src`'s offsets are authoritative. Backend nccl (= RCCL over xGMI) moves the vector through
    device memory, gloo through host memory."""

    MAX_OFFSETS = 4093  # header + payload = 4096 int64 = 32 KiB, one fixed-size broadcast

    def __init__(self, group=None, src: int = 0, device: Optional[str] = None, park_on_host: bool = False,
                 deferred_status: bool = False, status_every: int = 8):
        """deferred_status (SPMD callers that issue commands back to back: bench.py): the ranks' agreement is pipelined instead of
        standing between two calls. Every rank folds the outcome of its local (un)maps into one flag; every `status_every` calls
        that flag goes into an asynchronous all-reduce(min), whose verdict is read when the NEXT window closes (or in finish()).
        A rank's failure raises on that rank at once (its own exception is not swallowed in this mode) and on every other rank
        within two windows, at the latest in finish(); nothing is left unchecked once finish() has returned. Measured at world
        size 1 over RCCL (benchmarks/probe_fanout_cost.py): a synchronous status costs 57 us per call, a per-call deferred one 70
        (its read-back waits for a kernel that has to get onto a GPU busy zeroing pages), the window 3. The engine relay
        (start_collective_worker) keeps the synchronous form: a scheduler must know before it hands the pages out.
        park_on_host: the non-src ranks wait for the NEXT command in a helper thread (start_collective_worker). A rank
        parked inside an RCCL broadcast has that collective sitting enqueued on its GPU, where any device-wide
        synchronisation (torch.cuda.synchronize, the library's own hipDeviceSynchronize) would wait for the scheduler's next
        KV command; so a one-word gloo broadcast wakes the followers first and the RCCL broadcast is only entered when src
        has a message in hand."""
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
        self._deferred = bool(deferred_status)
        self._pending = None          # (work, status tensor) of the previous call's all-reduce, deferred mode
        # deferred mode: two status words taken in turn (the previous one may still be travelling), refreshed from constants
        self._status2 = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(2)]
        self._const = {True: torch.ones(1, dtype=torch.int64, device=device), False: torch.zeros(1, dtype=torch.int64, device=device)}
        self._turn = 0
        self._every = max(1, int(status_every))
        self._calls_in_window, self._ok_in_window = 0, True
        self._stage_free = None       # event behind the last H2D copy out of the pinned staging buffer (src rank, GPU)
        self._wake_group = None
        if park_on_host and on_gpu and self.world_size > 1:
            ranks = dist.get_process_group_ranks(group) if group is not None else list(range(dist.get_world_size()))
            # (every member of the TP group calls this constructor; no other process has to)
            self._wake_group = dist.new_group(ranks=ranks, backend="gloo", use_local_synchronization=True)
            self._wake = torch.zeros(1, dtype=torch.int64)

    def _exchange(self, cmd: int, offsets: Sequence[int], group_id: int) -> Tuple[int, int, List[int]]:
        torch, dist = self._torch, self._dist
        if self.rank == self.src:
            n = len(offsets)
            if n > self.MAX_OFFSETS:
                raise ValueError(f"at most {self.MAX_OFFSETS} offsets per collective call")
            if self._stage_free is not None:   # the previous message has left the staging buffer (it has, as a rule, long ago)
                self._stage_free.synchronize()
            st = self._stage_np
            st[0], st[1], st[2] = cmd, group_id, n
            st[3:3 + n] = offsets
            if self._on_gpu:
                self._buf[:3 + n].copy_(self._stage[:3 + n], non_blocking=True)
                if self._deferred:             # (synchronous mode: the status read-back of every call is that synchronisation)
                    if self._stage_free is None:
                        self._stage_free = torch.cuda.Event()
                    self._stage_free.record(torch.cuda.current_stream(self._buf.device))
        if self._wake_group is not None:   # host-side wake-up: nobody waits inside a GPU collective for a command that is not there yet
            dist.broadcast(self._wake, src=dist.get_global_rank(self.group, self.src) if self.group else self.src, group=self._wake_group)
        dist.broadcast(self._buf, src=dist.get_global_rank(self.group, self.src) if self.group else self.src,
                       group=self.group)
        if self._on_gpu and self.rank != self.src:
            # the payload length is not known before the copy: bring the whole (32 KiB) message over, once
            self._stage.copy_(self._buf, non_blocking=True)
            torch.cuda.current_stream(self._buf.device).synchronize()   # the stream of the buffer's GPU, whatever this thread's current device is
        st = self._stage_np
        n = int(st[2])
        if self._deferred and int(st[0]) in (CMD_MAP, CMD_UNMAP):
            return int(st[0]), int(st[1]), st[3:3 + n]       # (a view of the staging buffer: valid until the next call)
        return int(st[0]), int(st[1]), st[3:3 + n].tolist()

    def _check_pending(self) -> None:
        """Deferred mode: the verdict of the previous call (raises on every rank if any rank had failed)."""
        pending, self._pending = self._pending, None
        if pending is None:
            return
        work, status = pending
        work.wait()
        if int(status.item()) != 1:
            raise RuntimeError("a tensor-parallel rank failed to (un)map KV pages (reported by the call that followed)")

    def _close_window(self) -> None:
        self._check_pending()          # the verdict of the window before this one (its all-reduce has had a whole window to finish)
        self._turn ^= 1
        status = self._status2[self._turn]
        status.copy_(self._const[self._ok_in_window], non_blocking=True)
        work = self._dist.all_reduce(status, op=self._dist.ReduceOp.MIN, group=self.group, async_op=True)
        self._pending = (work, status)
        self._calls_in_window, self._ok_in_window = 0, True

    def finish(self) -> None:
        """Deferred mode: agree on everything up to now (every rank calls it). A no-op otherwise."""
        if not self._deferred:
            return
        self._close_window()
        self._check_pending()

    def _finish(self, ok: bool) -> None:
        if self._deferred:
            self._ok_in_window = self._ok_in_window and bool(ok)
            self._calls_in_window += 1
            if self._calls_in_window >= self._every:
                self._close_window()
            return
        if self._on_gpu:
            self._status_host[0] = 1 if ok else 0
            self._status.copy_(self._status_host, non_blocking=True)
        else:
            self._status[0] = 1 if ok else 0
        self._dist.all_reduce(self._status, op=self._dist.ReduceOp.MIN, group=self.group)
        if int(self._status.item()) != 1:
            raise RuntimeError("a tensor-parallel rank failed to (un)map KV pages")

    def _apply_staged(self, cmd: int, n: int, group_id: int) -> bool:
        """deferred mode: the offsets go to the C ABI as they lie in the staging buffer (no list, no pybind conversion: 40 us of a
        1024-offset call)."""
        try:
            from kvcached_amd import capi
            import ctypes
            ptr = ctypes.cast(self._stage_np[3:].ctypes.data, ctypes.POINTER(ctypes.c_int64))
            if cmd == CMD_MAP:
                capi.check(capi.lib.kvc_map_to_kv_tensors(ptr, n, group_id))
                return True
            if cmd == CMD_UNMAP:
                capi.check(capi.lib.kvc_unmap_from_kv_tensors(ptr, n, group_id))
                return True
            return self._apply(cmd, [], group_id)
        except Exception as e:
            print(f"rank {self.rank}: collective (un)map failed: {e}")
            self._ok_in_window = False   # the others learn of it when the window closes ...
            self._local_failure = e      # ... this rank, from run(), now
            return False

    def _apply(self, cmd: int, offs: List[int], group_id: int) -> bool:
        try:
            if cmd == CMD_MAP:
                return bool(map_to_kv_tensors(offs, group_id=group_id))
            if cmd == CMD_UNMAP:
                return bool(unmap_from_kv_tensors(offs, group_id=group_id))
            if cmd == CMD_CREATED:
                return bool(kv_tensors_created(group_id=group_id))
            return cmd == CMD_STOP
        except Exception as e:
            print(f"rank {self.rank}: collective (un)map failed: {e}")
            return False

    def run(self, cmd: int, offsets: Sequence[int] = (), group_id: int = 0, raise_on_failure: bool = True):
        """Broadcast (cmd, group_id, offsets) from `src`, apply locally, agree on success. Returns the offsets - a list, or in
        deferred mode a numpy view of the message that is valid until the next call - (or, with raise_on_failure=False, whether
        every rank succeeded)."""
        cmd, group_id, offs = self._exchange(cmd, offsets, group_id)
        if self._deferred and cmd in (CMD_MAP, CMD_UNMAP):
            self._local_failure = None
            ok = self._apply_staged(cmd, len(offs), group_id)
            self._finish(ok)
            if self._local_failure is not None:
                raise RuntimeError(f"rank {self.rank} failed to (un)map KV pages: {self._local_failure}")
            return offs
        ok = self._apply(cmd, offs, group_id)
        if raise_on_failure:
            self._finish(ok)
            return offs
        try:
            self._finish(ok)
            return True
        except RuntimeError:
            return False

    def serve_one(self) -> bool:
        """A non-src rank's share of one command, whatever src sends; False once src says stop."""
        cmd, group_id, offs = self._exchange(0, (), 0)
        ok = self._apply(cmd, offs, group_id)
        try:
            self._finish(ok)
        except RuntimeError:
            pass   # src reports the failure to the scheduler
        return cmd != CMD_STOP

    def map_to_kv_tensors(self, offsets: Sequence[int] = (), group_id: int = 0) -> List[int]:
        return self.run(CMD_MAP, offsets, group_id)

    def unmap_from_kv_tensors(self, offsets: Sequence[int] = (), group_id: int = 0) -> List[int]:
        return self.run(CMD_UNMAP, offsets, group_id)


class SharedPoolChannel:
    """The shared pool between the ranks of ONE torch.distributed group (north star: rank 0 creates pages, peers map them over
    xGMI; no counterpart in the reference, whose ranks each back their own pages - kvcached/tp_ipc_util.py:173-192 only
    fans offsets out). SPMD: every rank calls share() together.
      * the METADATA - (group id, offsets) - travels through the group's collective: one RCCL broadcast from `src`
        (CollectiveFanout's message), and one all-reduce(min) of a status word at the end;
      * the HANDLES are POSIX file descriptors (hipMemExportToShareableHandle / AMDKFD_IOC_EXPORT_DMABUF) and cannot travel
        through RCCL: src ships them with SCM_RIGHTS over one persistent Unix socket per peer
        (<socket dir>/[pp<k>/]fds<rank>.sock, next to the workers' command sockets).
    units="page_ids" (default where the library backs page ids as lanes): ONE descriptor per page id - the buffer
    its lane lives in - and two numbers per page id (lanes in the buffer, lane index) that ride in the header of the descriptor
    message: 64x fewer descriptors, exports and imports for Llama-3-8B than units="slots" (one per 2 MiB slot;
    KVCACHED_EXPORTABLE_HANDLES=1 on the source), which stays for single-row geometries and the other backends.
    `exporter(offsets, group_id) -> fds | (fds, meta)` and `importer(offsets, fds, group_id, meta)` default to the library's
    kvc_export_page_ids / kvc_map_imported_page_ids or kvc_export_mapped_slots / kvc_map_imported_slots; tests and the CPU
    rehearsal of bench.py pass stand-ins."""

    def __init__(self, fan: "CollectiveFanout", pp_rank: int = 0, exporter=None, importer=None, timeout: float = 60.0,
                 units: str = "page_ids"):
        self.fan, self.rank, self.src, self.world = fan, fan.rank, fan.src, fan.world_size
        if units not in ("page_ids", "slots"):
            raise ValueError("units must be 'page_ids' or 'slots'")
        self.units = units
        if exporter is None or importer is None:
            from kvcached_amd import capi
            if units == "page_ids":
                exporter = exporter or (lambda offs, gid: capi.export_page_ids(offs, gid))
                importer = importer or (lambda offs, fds, gid, meta: capi.map_imported_page_ids(offs, fds, meta, gid))
            else:
                exporter = exporter or (lambda offs, gid: capi.export_mapped_slots(offs, gid))
                importer = importer or (lambda offs, fds, gid, meta: capi.map_imported_slots(offs, fds, gid))
        self._export, self._import = exporter, importer
        self._peers: Dict[int, socket.socket] = {}
        self._up: Optional[socket.socket] = None
        base = os.path.dirname(get_worker_socket_path(0, pp_rank))
        os.makedirs(base, exist_ok=True)
        path = lambda r: os.path.join(base, f"fds{r}.sock")   # noqa: E731
        server = None
        if self.rank != self.src:
            if os.path.exists(path(self.rank)):
                os.remove(path(self.rank))
            server = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
            server.bind(path(self.rank))
            server.listen(1)
            server.settimeout(timeout)
        fan.run(CMD_CREATED, (), 0, raise_on_failure=False)   # (a barrier: every peer listens before src connects)
        if self.rank == self.src:
            for r in range(self.world):
                if r != self.src:
                    s = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
                    s.settimeout(timeout)
                    s.connect(path(r))
                    self._peers[r] = s
        else:
            self._up, _ = server.accept()
            self._up.settimeout(timeout)
            server.close()
            os.remove(path(self.rank))

    def share(self, offsets: Sequence[int] = (), group_id: int = 0) -> Dict[str, float]:
        """src has just backed `offsets` with exportable pages; afterwards every other rank shows the SAME physical pages at the
        same offsets. Returns this rank's timings (seconds): export, ship (src) / import+map (peers), total. Raises on every
        rank if any rank failed."""
        import time
        t0 = time.perf_counter()
        fds: List[int] = []
        meta: List[int] = []
        ok, err = True, None
        t_export = t_ship = t_import = 0.0
        try:
            if self.rank == self.src:
                got_ = self._export(list(offsets), group_id)
                if isinstance(got_, tuple):
                    fds, meta = list(got_[0]), list(got_[1])
                else:
                    fds = list(got_)
                t_export = time.perf_counter() - t0
        except Exception as e:   # the others are waiting in the broadcast: tell them there is nothing to come
            ok, err = False, e
        _, gid, offs = self.fan._exchange(CMD_SHARE, offsets if ok else (), group_id)
        try:
            t1 = time.perf_counter()
            if self.rank == self.src:
                for s in self._peers.values():   # (also when the export failed: the peers are waiting for this header)
                    send_msg(s, {"n_fds": len(fds) if ok else 0, "failed": not ok, "meta": meta})
                    if ok and fds:
                        send_fds(s, {}, fds)
                t_ship = time.perf_counter() - t1
            else:
                head = recv_msg(self._up)
                got: List[int] = []
                try:
                    if head.get("failed"):
                        raise RuntimeError("the source rank could not export its pages")
                    if head["n_fds"]:
                        _, got = recv_fds(self._up, head["n_fds"])
                        self._import(offs, got, gid, head.get("meta") or [])
                    elif offs:
                        raise RuntimeError("no handles came with the offsets")
                finally:
                    for fd in got:
                        os.close(fd)
                t_import = time.perf_counter() - t1
        except Exception as e:
            ok, err = False, err or e
        finally:
            for fd in fds:
                os.close(fd)
        try:
            self.fan._finish(ok)
        except RuntimeError:
            raise RuntimeError(f"sharing {len(offs)} offsets failed on some rank" + (f" (this one: {err})" if err else "")) from err
        return {"export_s": t_export, "ship_s": t_ship, "import_map_s": t_import, "total_s": time.perf_counter() - t0}

    def close(self) -> None:
        for s in list(self._peers.values()) + ([self._up] if self._up else []):
            try:
                s.close()
            except OSError:
                pass
        self._peers, self._up = {}, None
