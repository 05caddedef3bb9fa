"""ctypes binding of the C ABI (include/kvcached_amd.h) — the same symbols the pybind11 module
`vmm_ops` sits on. Used for the entry points that have no counterpart in the reference's
`vmm_ops` surface (HIP kernels, statistics, options, shared-pool export/import) and by the
tests/bench, which drive the hot path through the C ABI directly.

There is no fallback: if libkvcached_amd.so is missing, importing this module raises.
"""
from __future__ import annotations

import ctypes
import os
from typing import List, Sequence, Tuple

_HERE = os.path.dirname(os.path.abspath(__file__))
# KVCACHED_AMD_LIBRARY: the tests' build with hooks compiled in (kvcached_amd/_testhooks/, see build.py) - an explicit path,
# never a fallback: the library named must exist.
LIB_PATH = os.environ.get("KVCACHED_AMD_LIBRARY") or os.path.join(_HERE, "libkvcached_amd.so")

if not os.path.exists(LIB_PATH):
    raise ImportError(f"{LIB_PATH} is missing: build it with `python -m kvcached_amd.build` "
                      "(hipcc --offload-arch=gfx950). kvcached_amd has no CPU fallback.")

lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)

KVC_OK, KVC_E_INVALID, KVC_E_GPU, KVC_E_NO_PAGES, KVC_E_RUNTIME, KVC_E_NO_GPU, KVC_E_CALLBACK = 0, -1, -2, -3, -4, -5, -6
KVC_E_NOT_CREATED = -7
OPT_ZERO_BACKFILL, OPT_ZERO_FILL, OPT_POOL_BYTES, OPT_PROFILE, OPT_TLB_SHOOTDOWN, OPT_DEFER_UNMAP_SHOOTDOWN = 1, 2, 3, 4, 5, 6
OPT_ASYNC_UNMAP = 7
OPT_UNMAP_INVALIDATION_US = 8   # compat: the invalidation an unmap owes may trail the call by this many microseconds (0: inside the call)
OPT_FILL_VARIANT, OPT_COMPACT_VARIANT = 100, 101  # tuning only
# read-only: the VMM backend in effect after init's self tests (0 hip, 2 hybrid, 3 drm), and whether physical
# pages come straight from KFD (drm backend only)
OPT_EFFECTIVE_BACKEND, OPT_KFD_CREATE_ACTIVE = 108, 110
OPT_BACKGROUND_SHOOTDOWNS = 111  # read-only: TLB invalidations performed by the library's own threads so far
# read-only: 112-117 KFD allocation / dmabuf export / DRM import ns, creations, free ns, frees; 118 the direct KFD TLB flush
# is in use; 119 pages per physical extent at most; 120-123 pool footprint: pages held from the driver, pages handed out,
# free pieces inside partly used extents (the waste), the extent size new runs get right now
OPT_KFD_TLB_FLUSH_ACTIVE, OPT_MAX_EXTENT_PAGES = 118, 119
OPT_POOL_HELD_PAGES, OPT_POOL_OUT_PAGES, OPT_POOL_FREE_PIECES, OPT_POOL_EXTENT_PAGES_NOW = 120, 121, 122, 123
# read-only: pages zeroed on their way back to the pool / handed out by map calls that had nothing left to zero (§4.9)
OPT_PAGES_SCRUBBED, OPT_PAGES_PRESCRUBBED = 125, 126
OPT_ZERO_EXTENT_PAGES = 127  # read-only: compat mode aliases a zero extent of this many pages (0: PRT, or sharded zero pages through ROCr)
OPT_PRT = 128                # read-only: unbacked slots are PRT mappings (reads 0, writes dropped, no fault): the compat default on drm/gfx950
OPT_HOST_SEGMENT_0 = 130     # read-only, 130..149: host ns of the map/unmap calls by segment since the last reset (bench.py HOST_SEGMENTS)

_vp, _i64, _int, _sz = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_size_t
_I64P = ctypes.POINTER(ctypes.c_int64)
_VPP = ctypes.POINTER(ctypes.c_void_p)
_INTP = ctypes.POINTER(ctypes.c_int)
_SZP = ctypes.POINTER(ctypes.c_size_t)

BROADCAST_CB = ctypes.CFUNCTYPE(_int, _vp, _i64, _I64P, _sz)
BOOL_CB = ctypes.CFUNCTYPE(_int, _vp)


class Stats(ctypes.Structure):
    _fields_ = [("pages_mapped", _i64), ("pages_unmapped", _i64),
                ("handles_created", _i64), ("handles_released", _i64), ("handles_reused", _i64),
                ("map_calls", _i64), ("unmap_calls", _i64), ("map_ns", _i64), ("unmap_ns", _i64),
                ("fill_launches", _i64), ("fill_bytes", _i64), ("fill_ms", ctypes.c_double),
                ("compact_launches", _i64), ("compact_bytes", _i64), ("compact_ms", ctypes.c_double),
                ("tlb_shootdowns", _i64), ("shootdown_ns", _i64), ("index_launches", _i64),
                ("unmaps_queued", _i64), ("unmaps_cancelled", _i64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


# symbol -> (restype, argtypes); this table is also what tests/test_c_abi.py checks against the header
SIGNATURES = {
    "kvc_last_error": (ctypes.c_char_p, []),
    "kvc_abi_version": (_int, []),
    "kvc_init": (_int, [ctypes.c_char_p, _sz, _int]),
    "kvc_shutdown": (_int, []),
    "kvc_create_kv_tensors": (_int, [_sz, _sz, ctypes.c_char_p, _i64, _i64, _i64, _int, _VPP, _SZP, _I64P]),
    "kvc_kv_tensors_created": (_int, [_i64]),
    "kvc_get_device": (_int, [_INTP, _INTP]),
    "kvc_map_to_kv_tensors": (_int, [_I64P, _sz, _i64]),
    "kvc_unmap_from_kv_tensors": (_int, [_I64P, _sz, _i64]),
    "kvc_set_option": (_int, [_int, _i64]),
    "kvc_get_option": (_i64, [_int]),
    "kvc_get_stats": (_int, [ctypes.POINTER(Stats)]),
    "kvc_reset_stats": (_int, []),
    "kvc_flush_unmaps": (_int, []),
    "kvc_quiesce_begin": (_int, []),
    "kvc_quiesce_end": (_int, []),
    "kvc_get_driver_breakdown": (_int, [_I64P]),
    "kvc_mem_get_info": (_int, [_SZP, _SZP]),
    "kvc_set_mem_info_override": (_int, [_sz, _sz]),
    "kvc_page_new": (_vp, [_i64, _i64]),
    "kvc_page_delete": (None, [_vp]),
    "kvc_page_id": (_i64, [_vp]),
    "kvc_page_size": (_i64, [_vp]),
    "kvc_page_init": (None, [_vp, _i64]),
    "kvc_page_alloc": (_i64, [_vp, _i64, _I64P]),
    "kvc_page_free": (None, [_vp, _i64]),
    "kvc_page_free_batch": (None, [_vp, _I64P, _sz]),
    "kvc_page_empty": (_int, [_vp]),
    "kvc_page_full": (_int, [_vp]),
    "kvc_page_num_free_blocks": (_i64, [_vp]),
    "kvc_page_get_free_blocks": (_i64, [_vp, _I64P, _i64]),
    "kvc_page_get_block_range": (None, [_i64, _i64, _i64, _I64P, _I64P]),
    "kvc_page_get_num_blocks": (_i64, [_i64, _i64]),
    "kvc_pa_new": (_vp, [_i64, _i64, _i64, _i64, _i64, _int, _int, _int, _i64, _i64, ctypes.c_char_p]),
    "kvc_pa_delete": (None, [_vp]),
    "kvc_pa_start_prealloc_thread": (_int, [_vp]),
    "kvc_pa_stop_prealloc_thread": (_int, [_vp]),
    "kvc_pa_alloc_page": (_i64, [_vp]),
    "kvc_pa_alloc_pages": (_i64, [_vp, _i64, _I64P]),
    "kvc_pa_free_page": (_int, [_vp, _i64]),
    "kvc_pa_free_pages": (_int, [_vp, _I64P, _sz]),
    "kvc_pa_resize": (_int, [_vp, _i64]),
    "kvc_pa_trim": (_int, [_vp]),
    "kvc_pa_reset_free_page_order": (_int, [_vp]),
    "kvc_pa_get_num_free_pages": (_i64, [_vp]),
    "kvc_pa_get_num_inuse_pages": (_i64, [_vp]),
    "kvc_pa_get_num_total_pages": (_i64, [_vp]),
    "kvc_pa_get_num_reserved_pages": (_i64, [_vp]),
    "kvc_pa_get_avail_physical_pages": (_i64, [_vp]),
    "kvc_pa_check_and_get_resize_target": (_i64, [_vp, _i64]),
    "kvc_pa_get_resize_target": (_i64, [_vp]),
    "kvc_pa_get_page_id": (_i64, [_vp, _i64, _i64]),
    "kvc_pa_group_indices_by_page": (_i64, [_vp, _I64P, _sz, _i64, _I64P, _I64P, _I64P]),
    "kvc_pa_set_broadcast_map_callback": (_int, [_vp, BROADCAST_CB, _vp]),
    "kvc_pa_set_broadcast_unmap_callback": (_int, [_vp, BROADCAST_CB, _vp]),
    "kvc_pa_set_should_use_worker_ipc_callback": (_int, [_vp, BOOL_CB, _vp]),
    "kvc_pa_get_page_list": (_i64, [_vp, _int, _I64P, _i64]),
    "kvc_pa_ipc_name": (ctypes.c_char_p, [_vp]),
    "kvc_zero_fill_pages": (_int, [_VPP, _sz, _sz, _vp, _int]),
    "kvc_compact_blocks": (_int, [_VPP, _sz, _I64P, _I64P, _sz, _sz, _vp, _int]),
    "kvc_get_region_bases": (_i64, [_i64, _VPP, _i64]),
    "kvc_expand_block_ids": (_int, [_I64P, _sz, _i64, _vp, _vp]),
    "kvc_alloc_extend_indices": (_int, [_vp, _vp, _vp, _sz, _I64P, _sz, _i64, _vp, _sz, _vp]),
    "kvc_alloc_decode_indices": (_int, [_vp, _vp, _sz, _I64P, _sz, _i64, _vp, _vp]),
    "kvc_unique_block_ids": (_i64, [_vp, _sz, _i64, _i64, _I64P, _sz, _vp]),
    "kvc_export_mapped_slots": (_int, [_I64P, _sz, _i64, _INTP, _i64]),
    "kvc_map_imported_slots": (_int, [_I64P, _sz, _i64, _INTP, _sz]),
    "kvc_export_page_ids": (_int, [_I64P, _sz, _i64, _INTP, _I64P, _i64]),
    "kvc_map_imported_page_ids": (_int, [_I64P, _sz, _i64, _INTP, _sz, _I64P]),
}
for _name, (_res, _args) in SIGNATURES.items():
    _fn = getattr(lib, _name)  # AttributeError here = the library does not match the header
    _fn.restype, _fn.argtypes = _res, _args


class KvcError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(msg or f"kvcached_amd error {code}")
        self.code = code


def last_error() -> str:
    return (lib.kvc_last_error() or b"").decode()


def check(rc: int) -> int:
    if rc < 0:
        raise KvcError(rc, last_error())
    return rc


def i64_array(vals: Sequence[int]):
    return (ctypes.c_int64 * max(1, len(vals)))(*vals)


def ptr_array(vals: Sequence[int]):
    return (ctypes.c_void_p * max(1, len(vals)))(*vals)


# ------------------------------------------------------------------ convenience wrappers
def init(dev: str, page_size: int = 0, contiguous_layout: bool = False) -> None:
    check(lib.kvc_init(dev.encode(), page_size, int(contiguous_layout)))


def shutdown() -> None:
    check(lib.kvc_shutdown())


def create_kv_tensors(size: int, dtype_size: int, dev: str, num_layers: int, num_kv_buffers: int = 2,
                      group_id: int = 0, unified_pool: bool = False) -> List[Tuple[int, int]]:
    """Returns [(device_ptr, nbytes), ...]."""
    cap = max(1, num_layers)
    ptrs, nbytes, cnt = (ctypes.c_void_p * cap)(), (ctypes.c_size_t * cap)(), ctypes.c_int64(cap)
    check(lib.kvc_create_kv_tensors(size, dtype_size, dev.encode(), num_layers, num_kv_buffers, group_id,
                                    int(unified_pool), ptrs, nbytes, ctypes.byref(cnt)))
    return [(int(ptrs[i] or 0), int(nbytes[i])) for i in range(cnt.value)]


def map_to_kv_tensors(offsets: Sequence[int], group_id: int = 0) -> None:
    """`offsets`: a sequence of ints, or an int64 ctypes array made ahead of time (i64_array) - handed over as it is."""
    arr = offsets if isinstance(offsets, ctypes.Array) else i64_array(offsets)
    check(lib.kvc_map_to_kv_tensors(arr, len(offsets), group_id))


def unmap_from_kv_tensors(offsets: Sequence[int], group_id: int = 0) -> None:
    arr = offsets if isinstance(offsets, ctypes.Array) else i64_array(offsets)
    check(lib.kvc_unmap_from_kv_tensors(arr, len(offsets), group_id))


def set_option(opt: int, value: int) -> None:
    check(lib.kvc_set_option(opt, value))


def get_option(opt: int) -> int:
    return lib.kvc_get_option(opt)


def get_stats() -> dict:
    s = Stats()
    check(lib.kvc_get_stats(ctypes.byref(s)))
    return s.as_dict()


DRIVER_CALLS = ("unmap_alias", "acquire", "map", "set_access", "unmap", "release", "realias", "fill_sync")


def get_driver_breakdown() -> dict:
    """Host ns inside each driver-call class since the last reset_stats()."""
    out = (ctypes.c_int64 * 8)()
    check(lib.kvc_get_driver_breakdown(out))
    return dict(zip(DRIVER_CALLS, (int(x) for x in out)))


def flush_unmaps() -> None:
    check(lib.kvc_flush_unmaps())


class quiesced:
    """`with capi.quiesced():` - no page-table update of this process's KV regions happens inside the block (map / unmap calls of
    other threads, the prealloc thread and the reclaimer wait): the way to read or copy WHOLE KV tensors, unbacked parts
    included, next to background mapping (a slot in transition is not at rest: include/kvcached_amd.h, kvc_quiesce_begin)."""

    def __enter__(self):
        check(lib.kvc_quiesce_begin())
        return self

    def __exit__(self, *exc):
        check(lib.kvc_quiesce_end())
        return False


def reset_stats() -> None:
    check(lib.kvc_reset_stats())


def mem_get_info() -> Tuple[int, int]:
    f, t = ctypes.c_size_t(), ctypes.c_size_t()
    check(lib.kvc_mem_get_info(ctypes.byref(f), ctypes.byref(t)))
    return f.value, t.value


def set_mem_info_override(free_bytes: int, total_bytes: int) -> None:
    check(lib.kvc_set_mem_info_override(free_bytes, total_bytes))


def zero_fill_pages(page_ptrs: Sequence[int], page_bytes: int, stream: int = 0, sync: bool = True) -> None:
    """Zero `page_bytes` at each device pointer with the gfx950 fill kernel."""
    check(lib.kvc_zero_fill_pages(ptr_array(page_ptrs), len(page_ptrs), page_bytes, stream or None, int(sync)))


def compact_blocks(region_bases: Sequence[int], src_blocks: Sequence[int], dst_blocks: Sequence[int],
                   block_bytes: int, stream: int = 0, sync: bool = True) -> None:
    """In every region copy block src[m] onto block dst[m] (HIP kernel, LDS-staged)."""
    if len(src_blocks) != len(dst_blocks):
        raise ValueError("src_blocks and dst_blocks differ in length")
    check(lib.kvc_compact_blocks(ptr_array(region_bases), len(region_bases), i64_array(src_blocks),
                                 i64_array(dst_blocks), len(src_blocks), block_bytes, stream or None, int(sync)))


def get_region_bases(group_id: int = 0) -> List[int]:
    n = check(lib.kvc_get_region_bases(group_id, None, 0))
    buf = (ctypes.c_void_p * max(1, n))()
    check(lib.kvc_get_region_bases(group_id, buf, n))
    return [int(buf[i] or 0) for i in range(n)]


# ---- block ids <-> token slot indices. Device arrays are passed as raw addresses (tensor.data_ptr()),
# `stream` as the raw hipStream_t (torch.cuda.current_stream().cuda_stream); 0 = the device's default (null) stream, which
# is what torch's current stream is unless the caller switched.
def expand_block_ids(block_ids: Sequence[int], tokens_per_block: int, out_ptr: int, stream: int = 0) -> None:
    check(lib.kvc_expand_block_ids(i64_array(block_ids), len(block_ids), tokens_per_block, out_ptr, stream))


def alloc_extend_indices(prefix_lens_ptr: int, seq_lens_ptr: int, last_loc_ptr: int, bs: int, new_block_ids: Sequence[int],
                         tokens_per_block: int, out_ptr: int, extend_num_tokens: int, stream: int = 0) -> None:
    check(lib.kvc_alloc_extend_indices(prefix_lens_ptr, seq_lens_ptr, last_loc_ptr, bs, i64_array(new_block_ids),
                                       len(new_block_ids), tokens_per_block, out_ptr, extend_num_tokens, stream))


def alloc_decode_indices(seq_lens_ptr: int, last_loc_ptr: int, bs: int, new_block_ids: Sequence[int], tokens_per_block: int,
                         out_ptr: int, stream: int = 0) -> None:
    check(lib.kvc_alloc_decode_indices(seq_lens_ptr, last_loc_ptr, bs, i64_array(new_block_ids), len(new_block_ids),
                                       tokens_per_block, out_ptr, stream))


def unique_block_ids(token_indices_ptr: int, n: int, tokens_per_block: int, num_blocks: int, stream: int = 0) -> List[int]:
    """Blocking. Sorted distinct block ids of n device token indices."""
    cap = min(n, num_blocks)
    out = (_i64 * max(1, cap))()
    cnt = check(lib.kvc_unique_block_ids(token_indices_ptr, n, tokens_per_block, num_blocks, out, cap, stream))
    return list(out[:cnt])


def export_mapped_slots(offsets: Sequence[int], group_id: int = 0) -> List[int]:
    n = check(lib.kvc_export_mapped_slots(i64_array(offsets), len(offsets), group_id, None, 0))
    fds = (ctypes.c_int * max(1, n))()
    k = check(lib.kvc_export_mapped_slots(i64_array(offsets), len(offsets), group_id, fds, n))
    return [int(fds[i]) for i in range(k)]


def map_imported_slots(offsets: Sequence[int], fds: Sequence[int], group_id: int = 0) -> None:
    arr = (ctypes.c_int * max(1, len(fds)))(*fds)
    check(lib.kvc_map_imported_slots(i64_array(offsets), len(offsets), group_id, arr, len(fds)))


def export_page_ids(offsets: Sequence[int], group_id: int = 0) -> Tuple[List[int], List[int]]:
    """Shared pool with page ids as units: one dmabuf fd per BUFFER (up to 8 page ids live in one) and, per page id, (which fd, lanes
    in that buffer, lane index). Raises KvcError(KVC_E_INVALID) if the page ids are not backed by lanes of this process."""
    n = len(offsets)
    fds, meta = (ctypes.c_int * max(1, n))(), (ctypes.c_int64 * max(1, 3 * n))()
    k = check(lib.kvc_export_page_ids(i64_array(offsets), n, group_id, fds, meta, n))
    return [int(fds[i]) for i in range(k)], [int(meta[i]) for i in range(3 * n)]


def map_imported_page_ids(offsets: Sequence[int], fds: Sequence[int], meta: Sequence[int], group_id: int = 0) -> None:
    if len(meta) != 3 * len(offsets):
        raise ValueError("one (fd index, lanes, lane) triple per page id")
    arr = (ctypes.c_int * max(1, len(fds)))(*fds)
    check(lib.kvc_map_imported_page_ids(i64_array(offsets), len(offsets), group_id, arr, len(fds), i64_array(meta)))
