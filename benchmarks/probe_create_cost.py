"""Where does the cost of CREATING physical pages go, and what does it depend on?

Round 1's growth burst ran at 300 GB/s on the builder's boxes (3.8 us per page allocated) and at 23.6 GB/s on the
driver's (86 us per page). This probe drives the library through its C ABI alone (ctypes, no torch, nothing else in
the process) and times page creation, split into KFD allocation / dmabuf export / DRM import, in the situations that
could differ between boxes:

  A  the very first creations of a fresh process               (VRAM that this boot may never have handed out)
  B  release of everything (pool off)                          (the kernel wipes on release)
  C  creations right after those releases                      (pending wipes?)
  D  the same after a pause                                    (wipes done)
  E  more creations while C/D's pages are still held           (new VRAM again)
  F  one 32 MiB buffer per 16 pages instead of 16 x 2 MiB      (per-call or per-byte?)

    python benchmarks/probe_create_cost.py [--batches 8] > gpurun_out/create_cost.jsonl
One JSON line per phase."""
import argparse
import ctypes
import json
import os
import sys
import time

REPO = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
PAGE, N = 2 << 20, 1024


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", type=int, default=8)
    ap.add_argument("--chunk", type=int, default=1)
    ap.add_argument("--pause", type=float, default=3.0)
    args = ap.parse_args()
    os.environ["KVCACHED_VMM_BACKEND"] = "drm"
    os.environ["KVCACHED_PHYS_POOL_MB"] = "0"            # every unmap gives the page back to the driver
    os.environ["KVCACHED_PHYS_CHUNK_PAGES"] = str(args.chunk)
    os.environ.setdefault("KVCACHED_LOG_LEVEL", "ERROR")
    lib = ctypes.CDLL(os.path.join(REPO, "kvcached_amd", "libkvcached_amd.so"), mode=ctypes.RTLD_GLOBAL)
    lib.kvc_get_option.restype = ctypes.c_int64
    lib.kvc_last_error.restype = ctypes.c_char_p

    def ck(rc):
        if rc < 0:
            raise RuntimeError((lib.kvc_last_error() or b"").decode())

    ck(lib.kvc_init(b"cuda:0", ctypes.c_size_t(PAGE), 0))
    window = args.batches * 3
    ptrs, nb, cnt = (ctypes.c_void_p * 1)(), (ctypes.c_size_t * 1)(), ctypes.c_int64(1)
    ck(lib.kvc_create_kv_tensors(ctypes.c_size_t(window * N * PAGE), ctypes.c_size_t(1), b"cuda:0", ctypes.c_int64(1),
                                 ctypes.c_int64(1), ctypes.c_int64(0), 1, ptrs, nb, ctypes.byref(cnt)))
    f, t = ctypes.c_size_t(), ctypes.c_size_t()
    lib.kvc_mem_get_info(ctypes.byref(f), ctypes.byref(t))
    print(json.dumps({"phase": "start", "free_GiB": round(f.value / 2**30, 1), "total_GiB": round(t.value / 2**30, 1),
                      "kfd_create": int(lib.kvc_get_option(110)), "backend": int(lib.kvc_get_option(108)),
                      "chunk_pages": args.chunk}), flush=True)

    def offs(b):
        return (ctypes.c_int64 * N)(*[(b * N + i) * PAGE for i in range(N)])

    def opts():
        return [int(lib.kvc_get_option(k)) for k in (112, 113, 114, 115, 116, 117)]

    def run(phase, first, n, unmap=False):
        o0 = opts()
        per = []
        t0 = time.perf_counter()
        for b in range(first, first + n):
            ta = time.perf_counter()
            ck((lib.kvc_unmap_from_kv_tensors if unmap else lib.kvc_map_to_kv_tensors)(offs(b), ctypes.c_size_t(N), ctypes.c_int64(0)))
            per.append(time.perf_counter() - ta)
        ck(lib.kvc_flush_unmaps())
        dt = time.perf_counter() - t0
        o1 = opts()
        d = [b - a for a, b in zip(o0, o1)]
        pages = n * N
        rec = {"phase": phase, "pages": pages, "us_per_page": round(dt / pages * 1e6, 2),
               "per_batch_ms": [round(x * 1e3, 1) for x in per]}
        if d[3]:
            rec.update(creates=d[3], kfd_alloc_us=round(d[0] / d[3] / 1e3, 2), kfd_export_us=round(d[1] / d[3] / 1e3, 2),
                       drm_import_us=round(d[2] / d[3] / 1e3, 2))
        if d[5]:
            rec.update(frees=d[5], free_us=round(d[4] / d[5] / 1e3, 2))
        print(json.dumps(rec), flush=True)

    nb_ = args.batches
    run("A first creations of the process", 0, nb_)
    run("B release everything (pool off)", 0, nb_, unmap=True)
    run("C creations right after the releases", 0, nb_)
    run("B2 release again", 0, nb_, unmap=True)
    time.sleep(args.pause)
    run(f"D creations {args.pause:.0f} s after the releases", 0, nb_)
    run("E more creations while D's pages are held", nb_, nb_)
    run("E2 and more", 2 * nb_, nb_)
    run("G release all three", 0, 3 * nb_, unmap=True)
    time.sleep(args.pause)
    run("H creations after releasing 3x and a pause", 0, 3 * nb_)
    run("I release", 0, 3 * nb_, unmap=True)
    ck(lib.kvc_shutdown())


if __name__ == "__main__":
    sys.exit(main())
