"""Shared test infrastructure: a tiny trace language for KVCacheManager / PageAllocator
behaviour, a replay engine, and adapters for (a) the CPU oracle (oracle/libkvc_oracle.so)
and (b) the product (kvcached_amd). oracle/gen_golden.py reuses the replay engine with a
third adapter that drives the REAL reference, so that goldens, oracle and product are all
compared through exactly the same code path.

Trace ops (JSON arrays, first element is the opcode):
  ["a", req, n]            blocks[req] = alloc(n)                (None allowed)
  ["f", req]               free(blocks[req]); forget req
  ["fp", req, lo, hi]      free(blocks[req][lo:hi]); keep the rest under req
  ["rs", n]                try_to_reserve(n)
  ["fr"]                   free_reserved()
  ["rz", mem_bytes]        resize(mem_bytes)
  ["tr"]                   trim()
  ["ph", pages]            set what get_avail_physical_pages() reports
  ["pre"]                  run one synchronous prealloc pass (oracle/product-cpu only)
Each op yields a record {"r": result, "s": snapshot, "e": events}.
"""
from __future__ import annotations

import ctypes
import hashlib
import os
import struct
from typing import Any, Dict, List, Optional, Sequence

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN_DIR = os.path.join(REPO, "tests", "golden")
ORACLE_SO = os.path.join(REPO, "oracle", "libkvc_oracle.so")

MiB = 1 << 20
PAGE = 2 * MiB


def h64(ints: Sequence[int]) -> str:
    """Short stable hash of an int64 list (little-endian bytes, sha256, 16 hex chars)."""
    b = struct.pack("<%dq" % len(ints), *ints) if len(ints) else b""
    return hashlib.sha256(b).hexdigest()[:16]


def summarise(ints: Optional[Sequence[int]], full: bool) -> Any:
    if ints is None:
        return None
    ints = list(ints)
    if full:
        return ints
    return {"n": len(ints), "h": h64(ints), "head": ints[:4], "tail": ints[-2:]}


class Adapter:
    """Interface every implementation under comparison provides."""

    def alloc(self, n: int) -> Optional[List[int]]: ...
    def free(self, ids: List[int]) -> None: ...
    def try_to_reserve(self, n: int) -> bool: ...
    def free_reserved(self) -> None: ...
    def resize(self, mem: int) -> bool: ...
    def trim(self) -> None: ...
    def set_phys(self, pages: int) -> None: ...
    def prealloc_step(self) -> int: raise NotImplementedError
    def available_size(self) -> int: ...
    def snapshot(self) -> List[int]: ...
    def drain_events(self) -> List[List[Any]]: ...
    def close(self) -> None: ...


def replay(ad: Adapter, ops: List[List[Any]], full: bool = False) -> List[Dict[str, Any]]:
    blocks: Dict[int, List[int]] = {}
    out = []
    for op in ops:
        code = op[0]
        r: Any = None
        if code == "a":
            got = ad.alloc(op[2])
            if got is not None:
                blocks[op[1]] = list(got)
            r = summarise(got, full)
        elif code == "f":
            ad.free(blocks.pop(op[1], []))
        elif code == "fp":
            cur = blocks.get(op[1], [])
            part = cur[op[2]:op[3]]
            blocks[op[1]] = cur[:op[2]] + cur[op[3]:]
            ad.free(part)
        elif code == "rs":
            r = bool(ad.try_to_reserve(op[1]))
        elif code == "fr":
            ad.free_reserved()
        elif code == "rz":
            r = bool(ad.resize(op[1]))
        elif code == "tr":
            ad.trim()
        elif code == "ph":
            ad.set_phys(op[1])
        elif code == "pre":
            r = ad.prealloc_step()
        else:
            raise ValueError(f"bad op {op}")
        ev = ad.drain_events()
        if not full:
            ev = [[k, summarise(o, len(o) <= 8)] for k, o in ev]
        out.append({"r": r, "s": ad.snapshot(), "e": ev})
    return out


def merge_events(ev: List[List[Any]]) -> List[List[Any]]:
    """Consecutive map (or unmap) calls folded into one with the offsets concatenated: what is left when only the
    call boundaries differ (KVCACHED_BATCH_PAGE_ALLOC)."""
    out: List[List[Any]] = []
    for kind, offs in ev:
        if out and out[-1][0] == kind:
            out[-1][1] = out[-1][1] + list(offs)
        else:
            out.append([kind, list(offs)])
    return out


def chain_hash(records: List[Dict[str, Any]], every: int = 100) -> Dict[str, Any]:
    """Running sha256 over the JSON of each record; checkpoints localise a mismatch."""
    import json
    h = hashlib.sha256()
    cps = []
    for i, rec in enumerate(records):
        h.update(json.dumps(rec, sort_keys=True, separators=(",", ":")).encode())
        if (i + 1) % every == 0:
            cps.append(h.hexdigest()[:16])
    return {"final": h.hexdigest(), "checkpoints": cps, "every": every, "n": len(records)}


# --------------------------------------------------------------------------- oracle adapter
_I64P = ctypes.POINTER(ctypes.c_int64)


def load_oracle() -> ctypes.CDLL:
    if not os.path.exists(ORACLE_SO):
        raise RuntimeError(f"{ORACLE_SO} missing: run `make -C oracle oracle` (or __graft_entry__.build())")
    lib = ctypes.CDLL(ORACLE_SO)
    vp, i64, ci = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
    sig = {
        "okvc_last_error": (ctypes.c_char_p, []),
        "okvc_get_block_range": (None, [i64, i64, i64, _I64P, _I64P]),
        "okvc_get_num_blocks": (i64, [i64, i64]),
        "okvc_ref_avail_physical_pages": (i64, [ctypes.c_uint64, ctypes.c_uint64, ctypes.c_double, i64, i64, i64]),
        "okvc_page_new": (vp, [i64, i64]), "okvc_page_delete": (None, [vp]),
        "okvc_page_init": (None, [vp, i64]), "okvc_page_alloc": (i64, [vp, i64, _I64P]),
        "okvc_page_free": (None, [vp, i64]), "okvc_page_free_batch": (None, [vp, _I64P, i64]),
        "okvc_page_empty": (ci, [vp]), "okvc_page_full": (ci, [vp]), "okvc_page_num_free": (i64, [vp]),
        "okvc_page_free_blocks": (i64, [vp, _I64P, i64]),
        "okvc_pa_new": (vp, [i64, i64, i64, i64, ci, ci, i64, i64, i64]), "okvc_pa_delete": (None, [vp]),
        "okvc_pa_alloc_page": (i64, [vp]), "okvc_pa_free_page": (None, [vp, i64]),
        "okvc_pa_free_pages": (None, [vp, _I64P, i64]), "okvc_pa_resize": (ci, [vp, i64]),
        "okvc_pa_trim": (None, [vp]), "okvc_pa_reset_free_page_order": (None, [vp]),
        "okvc_pa_prealloc_step": (i64, [vp]), "okvc_pa_set_prealloc_needed": (None, [vp, ci]),
        "okvc_pa_set_avail_phys_pages": (None, [vp, i64]), "okvc_pa_set_shm_total": (None, [vp, i64]),
        "okvc_pa_watcher_tick": (None, [vp]), "okvc_pa_get_resize_target": (i64, [vp]),
        "okvc_pa_get_page_id": (i64, [vp, i64, i64]), "okvc_pa_stats": (None, [vp, _I64P]),
        "okvc_pa_list": (i64, [vp, ci, _I64P, i64]),
        "okvc_pa_group_indices": (i64, [vp, _I64P, i64, i64, _I64P, _I64P, _I64P]),
        "okvc_pa_drain_log": (i64, [vp, _I64P, i64]),
        "okvc_mgr_new": (vp, [i64, i64, i64, i64, i64, ci, i64, i64, ci, ci, i64, i64]),
        "okvc_mgr_delete": (None, [vp]), "okvc_mgr_pa": (vp, [vp]), "okvc_mgr_post_init": (ci, [vp]),
        "okvc_mgr_alloc": (i64, [vp, i64, _I64P]), "okvc_mgr_free": (ci, [vp, _I64P, i64]),
        "okvc_mgr_try_to_reserve": (ci, [vp, i64]), "okvc_mgr_free_reserved": (ci, [vp]),
        "okvc_mgr_resize": (ci, [vp, i64]), "okvc_mgr_trim": (None, [vp]), "okvc_mgr_clear": (ci, [vp]),
        "okvc_mgr_available_size": (i64, [vp]), "okvc_mgr_reserved_blocks": (i64, [vp, _I64P, i64]),
        "okvc_mgr_stats": (None, [vp, _I64P]),
        "okvc_zero_fill_pages": (None, [ctypes.POINTER(vp), i64, i64]),
        "okvc_compact_blocks": (None, [ctypes.POINTER(vp), i64, _I64P, _I64P, i64, i64]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    return lib


def _arr(vals: Sequence[int]):
    return (ctypes.c_int64 * max(1, len(vals)))(*vals)


def _drain_flat(buf, n) -> List[List[Any]]:
    ev, i = [], 0
    while i < n:
        kind, cnt = buf[i], buf[i + 1]
        ev.append([int(kind), [int(x) for x in buf[i + 2:i + 2 + cnt]]])
        i += 2 + cnt
    return ev


class OraclePA:
    """ctypes view of the oracle's PageAllocator (used directly and through OracleAdapter)."""

    def __init__(self, lib, handle, owned=True):
        self.lib, self.h, self.owned = lib, handle, owned

    @classmethod
    def create(cls, lib, num_layers, mem, page_size=PAGE, world_size=1, contiguous=False, prealloc=False,
               num_kv_buffers=2, min_res=5, max_res=10):
        return cls(lib, lib.okvc_pa_new(num_layers, mem, page_size, world_size, int(contiguous), int(prealloc),
                                        num_kv_buffers, min_res, max_res))

    def alloc_page(self) -> int:
        pid = self.lib.okvc_pa_alloc_page(self.h)
        if pid < 0:
            raise RuntimeError(self.lib.okvc_last_error().decode())
        return pid

    def free_page(self, pid): self.lib.okvc_pa_free_page(self.h, pid)
    def free_pages(self, pids): self.lib.okvc_pa_free_pages(self.h, _arr(pids), len(pids))
    def resize(self, mem) -> bool: return bool(self.lib.okvc_pa_resize(self.h, mem))
    def trim(self): self.lib.okvc_pa_trim(self.h)
    def reset_free_page_order(self): self.lib.okvc_pa_reset_free_page_order(self.h)
    def prealloc_step(self) -> int: return self.lib.okvc_pa_prealloc_step(self.h)
    def set_prealloc_needed(self, v=True): self.lib.okvc_pa_set_prealloc_needed(self.h, int(v))
    def set_phys(self, pages): self.lib.okvc_pa_set_avail_phys_pages(self.h, pages)
    def get_page_id(self, b, B): return self.lib.okvc_pa_get_page_id(self.h, b, B)

    def stats(self) -> List[int]:
        o = (ctypes.c_int64 * 7)()
        self.lib.okvc_pa_stats(self.h, o)
        return list(o)

    def lst(self, which: int) -> List[int]:
        n = self.lib.okvc_pa_list(self.h, which, None, 0)
        buf = (ctypes.c_int64 * max(1, n))()
        self.lib.okvc_pa_list(self.h, which, buf, n)
        return list(buf[:n])

    def group_indices_by_page(self, idx: Sequence[int], B: int) -> Dict[int, List[int]]:
        n = len(idx)
        keys, counts, vals = _arr([0] * n), _arr([0] * n), _arr([0] * n)
        k = self.lib.okvc_pa_group_indices(self.h, _arr(idx), n, B, keys, counts, vals)
        out, w = {}, 0
        for i in range(k):
            out[int(keys[i])] = [int(v) for v in vals[w:w + counts[i]]]
            w += counts[i]
        return out

    def drain_events(self):
        n = self.lib.okvc_pa_drain_log(self.h, None, 0)
        buf = (ctypes.c_int64 * max(1, n))()
        self.lib.okvc_pa_drain_log(self.h, buf, n)
        return _drain_flat(buf, n)

    def close(self):
        if self.owned and self.h:
            self.lib.okvc_pa_delete(self.h)
            self.h = None


class OracleAdapter(Adapter):
    def __init__(self, lib, num_blocks, block_size, cell_size, num_layers, world_size=1, reserve_null_block=False,
                 num_kv_buffers=2, page_size=PAGE, contiguous=False, prealloc=False, min_res=5, max_res=10,
                 phys_pages=1 << 40):
        self.lib = lib
        self.h = lib.okvc_mgr_new(num_blocks, block_size, cell_size, num_layers, world_size, int(reserve_null_block),
                                  num_kv_buffers, page_size, int(contiguous), int(prealloc), min_res, max_res)
        self.pa = OraclePA(lib, lib.okvc_mgr_pa(self.h), owned=False)
        self.pa.set_phys(phys_pages)
        rc = lib.okvc_mgr_post_init(self.h)
        if rc != 0:
            raise RuntimeError("Failed to reserve null block at index 0")

    def alloc(self, n):
        buf = (ctypes.c_int64 * max(1, n))()
        k = self.lib.okvc_mgr_alloc(self.h, n, buf)
        if k == -1:
            return None
        if k < 0:
            raise RuntimeError(self.lib.okvc_last_error().decode())
        return list(buf[:k])

    def free(self, ids):
        if self.lib.okvc_mgr_free(self.h, _arr(ids), len(ids)) != 0:
            raise RuntimeError(self.lib.okvc_last_error().decode())

    def try_to_reserve(self, n): return bool(self.lib.okvc_mgr_try_to_reserve(self.h, n))
    def free_reserved(self): self.lib.okvc_mgr_free_reserved(self.h)

    def resize(self, mem):
        rc = self.lib.okvc_mgr_resize(self.h, mem)
        if rc < 0:
            raise AssertionError("Reserved blocks must be freed before resizing.")
        return bool(rc)

    def trim(self): self.lib.okvc_mgr_trim(self.h)
    def clear(self): self.lib.okvc_mgr_clear(self.h)
    def set_phys(self, pages): self.pa.set_phys(pages)
    def prealloc_step(self): return self.pa.prealloc_step()
    def available_size(self): return self.lib.okvc_mgr_available_size(self.h)

    def reserved_blocks(self):
        n = self.lib.okvc_mgr_reserved_blocks(self.h, None, 0)
        buf = (ctypes.c_int64 * max(1, n))()
        self.lib.okvc_mgr_reserved_blocks(self.h, buf, n)
        return list(buf[:n])

    def mgr_stats(self):
        o = (ctypes.c_int64 * 7)()
        self.lib.okvc_mgr_stats(self.h, o)
        return list(o)

    def snapshot(self):
        free, inuse, total, reserved, st, su, sp = self.pa.stats()
        ms = self.mgr_stats()
        return [self.available_size(), free, inuse, total, reserved, ms[3], ms[4], st, su, sp]

    def drain_events(self): return self.pa.drain_events()

    def close(self):
        if self.h:
            self.lib.okvc_mgr_delete(self.h)
            self.h = None


# --------------------------------------------------------------------------- product adapter
PHYS_CAP = 1 << 30  # "unlimited" physical pages, kept small enough that bytes fit in size_t


def set_product_phys_pages(pages: int, page_size: int, num_layers: int, num_kv_buffers: int) -> None:
    """Make the product's get_avail_physical_pages() report exactly `pages` through the C ABI's
    mem-info override: total=20 gives a headroom of size_t(20*0.05)=1 byte."""
    from kvcached_amd import capi
    pages = min(pages, PHYS_CAP)
    capi.set_mem_info_override(pages * page_size * num_layers * num_kv_buffers + 1, 20)


class ProductAdapter(Adapter):
    """kvcached_amd.KVCacheManager (Python) over vmm_ops (pybind11) over the C ABI.

    device="cpu": the reference's own host device path — bookkeeping only, map/unmap requests are
    captured through the broadcast-callback hook exactly like oracle/gen_golden.py does with the
    reference. device="cuda:0": requests are captured AND executed on the GPU (real hipMemMap +
    zero fill), so the same golden traces also drive the HIP path."""

    def __init__(self, num_blocks, block_size, cell_size, num_layers, world_size=1, reserve_null_block=False,
                 num_kv_buffers=2, contiguous=False, phys_pages=1 << 40, group_id=0, device="cpu", execute=False,
                 page_size=PAGE, batch_page_alloc=False):
        import kvcached_amd.kv_cache_manager as kcm
        from kvcached_amd import vmm_ops
        self.kcm, self.ops = kcm, vmm_ops
        self.geom = (page_size, num_layers, num_kv_buffers)
        self._saved_globals = (kcm.CONTIGUOUS_LAYOUT, kcm.BATCH_PAGE_ALLOC)   # restored by close()
        kcm.CONTIGUOUS_LAYOUT = contiguous
        # False = the reference's page-by-page map calls (what the goldens record, call by call); True = the
        # product default: one map call per alloc(), same offsets in the same order (compare with merge_events)
        kcm.BATCH_PAGE_ALLOC = batch_page_alloc
        vmm_ops.init_kvcached(device, page_size, contiguous)
        self.real_phys = device != "cpu" and phys_pages >= PHYS_CAP
        if not self.real_phys:
            set_product_phys_pages(phys_pages, *self.geom)
        self.events: List[List[Any]] = []
        self.execute = execute
        self.group_id = group_id
        self.m = kcm.KVCacheManager(num_blocks, block_size, cell_size, num_layers, world_size=world_size,
                                    reserve_null_block=reserve_null_block, num_kv_buffers=num_kv_buffers,
                                    group_id=group_id)
        pa = self.m.page_allocator
        pa.set_should_use_worker_ipc_callback(lambda: True)
        pa.set_broadcast_map_callback(self._on_map)
        pa.set_broadcast_unmap_callback(self._on_unmap)
        mem = num_blocks * block_size * cell_size
        mem = (mem + 2 * page_size - 1) // (2 * page_size) * (2 * page_size)
        self.tensors = vmm_ops.create_kv_tensors(mem * num_kv_buffers, 1, device, num_layers, num_kv_buffers,
                                                 group_id, False)
        assert self.m._post_init_done.wait(20), "_post_init did not finish"

    def _on_map(self, ws, offs):
        self.events.append([0, [int(o) for o in offs]])
        if self.execute:
            assert self.ops.map_to_kv_tensors(list(offs), self.group_id)

    def _on_unmap(self, ws, offs):
        self.events.append([1, [int(o) for o in offs]])
        if self.execute:
            assert self.ops.unmap_from_kv_tensors(list(offs), self.group_id)

    def alloc(self, n): return self.m.alloc(n)
    def free(self, ids): self.m.free(ids)
    def try_to_reserve(self, n): return self.m.try_to_reserve(n)
    def free_reserved(self): self.m.free_reserved()
    def resize(self, mem): return self.m.resize(mem)
    def trim(self): self.m.trim()

    def set_phys(self, pages):
        if not self.real_phys or pages < PHYS_CAP:
            self.real_phys = False
            set_product_phys_pages(pages, *self.geom)

    def available_size(self): return self.m.available_size()

    def snapshot(self):
        import numpy as np
        pa = self.m.page_allocator
        t, u, p = [int(x) for x in np.fromfile("/dev/shm/" + pa._ipc_name(), dtype=np.int64)[:3]]
        return [self.m.available_size(), pa.get_num_free_pages(), pa.get_num_inuse_pages(), pa.get_num_total_pages(),
                pa.get_num_reserved_pages(), len(self.m.reserved_blocks), int(self.m.in_shrink), t, u, p]

    def drain_events(self):
        ev, self.events = self.events, []
        return ev

    def close(self):
        from kvcached_amd import capi
        self.m = None
        self.tensors = None
        self.kcm.CONTIGUOUS_LAYOUT, self.kcm.BATCH_PAGE_ALLOC = self._saved_globals
        self.ops.shutdown_kvcached()
        capi.set_mem_info_override(0, 0)


# --------------------------------------------------------------------------- prefix-cache harness (no vLLM)
class FakeBlockPool:
    """Stand-in for vllm.v1.core.block_pool.BlockPool (base class only)."""


class FakeKVCacheBlock:
    def __init__(self, block_id: int, ref_cnt: int = 0):
        self.block_id, self.ref_cnt, self.is_null = block_id, ref_cnt, False


class FakeRequest:
    def __init__(self, block_hashes):
        self.block_hashes = block_hashes


class FifoBlockManager:
    """Deterministic stand-in for KVCacheManager: hands out the lowest free ids, frees to the back."""

    def __init__(self, num_blocks: int):
        self.free_ids = list(range(num_blocks))

    def alloc(self, n: int):
        if len(self.free_ids) < n:
            return None
        out, self.free_ids = self.free_ids[:n], self.free_ids[n:]
        return out

    def free(self, ids):
        self.free_ids.extend(ids)

    def available_size(self) -> int:
        return len(self.free_ids)


def replay_prefix_cache(pool, manager, ops):
    """Drives an ElasticBlockPool like vLLM's KVCacheManager does. Ops:
      ["req", rid, [hash ids], group]   look the prefix up (all-or-nothing per block), touch the hits, allocate the
                                        rest, register the full blocks                          -> block ids | error
      ["fin", rid]                      free the request's blocks in reverse order
      ["evict", [ids]] / ["reset"] / ["stat"]
    Returns one record per op: result + (num_free_blocks, evictable ids in LRU order, #cached keys, manager free list)."""
    live, out = {}, []
    for op in ops:
        r = None
        try:
            if op[0] == "req":
                _, rid, hashes, group = op
                hs = [b"h%06d" % h for h in hashes]
                hits = []
                for h in hs:
                    got = pool.get_cached_block(h, [group])
                    if not got:
                        break
                    hits.append(got[0])
                if hits:
                    pool.touch(hits)
                need = len(hs) - len(hits)
                new = pool.get_new_blocks(need) if need else []
                blocks = hits + new
                pool.cache_full_blocks(FakeRequest(hs), blocks, len(hits), len(blocks), 16, group)
                live[rid] = blocks
                r = {"hit": len(hits), "ids": [b.block_id for b in blocks]}
            elif op[0] == "fin":
                blocks = live.pop(op[1], [])
                pool.free_blocks(reversed(blocks))
            elif op[0] == "evict":
                pool.evict_blocks(set(op[1]))
            elif op[0] == "reset":
                r = pool.reset_prefix_cache()
            elif op[0] == "stat":
                r = [pool.get_num_free_blocks(), round(pool.get_usage(), 9), len(pool.take_events())]
        except ValueError as e:
            r = "ValueError: " + str(e)
        out.append({"r": r, "s": [pool.get_num_free_blocks(), list(getattr(pool, "_evictable_blocks", {}).keys()),
                                  len(getattr(pool, "_cached_blocks", {})), h64(manager.free_ids),
                                  [b.ref_cnt for bl in live.values() for b in bl][:16]]})
    return out


# geometry of the prefix-cache traces over a REAL manager (tests/golden/prefix_cache_real_manager.json): 256 KiB blocks, 8 per page
PREFIX_REAL_GEOMETRY = dict(layers=2, block_tokens=16, cell=16384)


def replay_prefix_over_real_manager(pool, ops, on_hit=None, on_new=None, after_req=None):
    """The request trace of replay_prefix_cache through an ElasticBlockPool that sits on a REAL KVCacheManager (the reference's in
    oracle/gen_golden.py, the product's in the tests - on its cpu device and on cuda:0). One record per op:
    [hits + block ids | error | reset result | stat, free blocks, cached keys, evictable ids in LRU order].
    on_hit(hash ids, blocks) / on_new(hash ids, blocks): where a GPU run checks / writes the blocks' contents."""
    live, out = {}, []
    for op in ops:
        r = None
        if op[0] == "req":
            _, rid, hashes, group = op
            hs = [b"h%06d" % h for h in hashes]
            hit_blocks = []
            for h in hs:
                got = pool.get_cached_block(h, [group])
                if not got:
                    break
                hit_blocks.append(got[0])
            if hit_blocks:
                pool.touch(hit_blocks)
                if on_hit:
                    on_hit(hashes[:len(hit_blocks)], hit_blocks)
            need = len(hs) - len(hit_blocks)
            try:
                new = pool.get_new_blocks(need) if need else []
            except ValueError as e:
                r = "ValueError: " + str(e)
            else:
                if on_new:
                    on_new(hashes[len(hit_blocks):], new)
                blocks = hit_blocks + new
                pool.cache_full_blocks(FakeRequest(hs), blocks, len(hit_blocks), len(blocks), 16, group)
                live[rid] = blocks
                r = {"hit": len(hit_blocks), "ids": [b.block_id for b in blocks]}
                if after_req:
                    after_req()
        elif op[0] == "fin":
            pool.free_blocks(reversed(live.pop(op[1], [])))
        elif op[0] == "evict":
            pool.evict_blocks(set(op[1]))
        elif op[0] == "reset":
            r = pool.reset_prefix_cache()
        elif op[0] == "stat":
            r = [pool.get_num_free_blocks(), len(pool.take_events())]
        out.append([r, pool.get_num_free_blocks(), len(pool._cached_blocks), list(pool._evictable_blocks.keys())])
    for blocks in live.values():
        pool.free_blocks(reversed(blocks))
    return out
