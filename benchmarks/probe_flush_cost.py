"""What does one TLB invalidation cost, and what does the cost depend on?

A bare KFD map+unmap pair of a 4 KiB buffer measured 170-200 us in round 1's probe process (`tools/drm_vmm_probe.cpp`,
profiles/r01_drm_flush_modes.log); inside the round-2 bench the same pair costs 390-410 us, and it is two thirds of the
default mode's unmap call. This probe drives the library through its C ABI (ctypes, no torch) and times the invalidation
an unmap of ONE page performs in the calling thread (default mode), in states that differ in what the process's GPU VM
holds:

  A  a fresh window, a handful of pages ever created
  B  after the whole window was backed once and released into the pool (every extent + its alias mapping live)
  C  with the whole window backed (every slot a real mapping instead of a PRT one)
  D  as B, while a fill kernel keeps the memory system busy

and, in separate child processes, with the alias mappings (KVCACHED_SCRUB_ON_RELEASE=false), the PRT rest state
(KVCACHED_PRT=false) or the multi-page extents (KVCACHED_PHYS_CHUNK_PAGES=1) switched off, and a smaller window.

    python benchmarks/probe_flush_cost.py > gpurun_out/flush_cost.jsonl
One JSON line per phase."""
import argparse
import ctypes
import json
import os
import subprocess
import sys
import time

REPO = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, REPO)
PAGE, N = 2 << 20, 1024


def child(args):
    from kvcached_amd import capi
    capi.init("cuda:0", PAGE, False)
    window = args.window_gib * 512 // N                 # batches of 1024 pages
    capi.create_kv_tensors(window * N * PAGE, 1, "cuda:0", 1, 1)
    tag = {"config": args.tag, "window_GiB": args.window_gib, "prt": capi.get_option(capi.OPT_PRT),
           "max_extent_pages": capi.get_option(119), "kfd_flush": capi.get_option(118)}

    def flushes(phase, off, reps=30, busy=False):
        capi.flush_unmaps()
        per = []
        if busy:                                        # 2 GiB of stores in flight (0.3 ms) while the invalidation runs
            base = capi.get_region_bases(0)[0]
            hot = [(N + i) * PAGE for i in range(N)]
            capi.map_to_kv_tensors(hot)
            ptrs = [base + o for o in hot]
        for _ in range(reps):
            capi.map_to_kv_tensors([off])
            if busy:
                capi.zero_fill_pages(ptrs, PAGE, 0, False)
            s0 = capi.get_stats()
            capi.unmap_from_kv_tensors([off])
            capi.flush_unmaps()                         # lazy mode: the invalidation runs behind the call
            s1 = capi.get_stats()
            n = s1["tlb_shootdowns"] - s0["tlb_shootdowns"]
            if n:
                per.append((s1["shootdown_ns"] - s0["shootdown_ns"]) / n / 1e3)
        if busy:
            capi.unmap_from_kv_tensors(hot)
        capi.flush_unmaps()
        per.sort()
        rec = dict(tag, phase=phase, flushes=len(per))
        if per:
            rec.update(min_us=round(per[0], 1), p50_us=round(per[len(per) // 2], 1), max_us=round(per[-1], 1))
        print(json.dumps(rec), flush=True)

    flushes("A fresh window", 0)
    t0 = time.perf_counter()
    for b in range(window):
        capi.map_to_kv_tensors([(b * N + i) * PAGE for i in range(N)])
    flushes(f"C whole window backed ({window * N} pages, {time.perf_counter() - t0:.1f} s)", 0)
    for b in range(window):
        capi.unmap_from_kv_tensors([(b * N + i) * PAGE for i in range(N)])
    capi.flush_unmaps()
    flushes("B window released into the pool", 0)
    flushes("B' same, another slot", (window * N // 2) * PAGE)
    if args.busy:
        flushes("D as B with a fill kernel in flight", 0, busy=True)
    capi.shutdown()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--child", action="store_true")
    ap.add_argument("--tag", default="default")
    ap.add_argument("--window-gib", type=int, default=64)
    ap.add_argument("--busy", action="store_true")
    args = ap.parse_args()
    if args.child:
        return child(args)
    base = dict(os.environ, KVCACHED_VMM_BACKEND="drm", KVCACHED_LOG_LEVEL="ERROR", KVCACHED_PHYS_POOL_MB=str(80 << 10))
    runs = [("default", 64, {}, True),
            ("window 8 GiB", 8, {}, False),
            ("no alias mappings (scrub off)", 64, {"KVCACHED_SCRUB_ON_RELEASE": "false"}, False),
            ("one buffer per page", 64, {"KVCACHED_PHYS_CHUNK_PAGES": "1"}, False),
            ("unmapped VA instead of PRT (lazy)", 64, {"KVCACHED_PRT": "false", "KVCACHED_ZERO_EXTENT": "false",
                                                       "KVCACHED_ZERO_BACKFILL": "false",
                                                       "KVCACHED_ASYNC_SHOOTDOWN": "false"}, False)]
    for tag, gib, env, busy in runs:
        cmd = [sys.executable, os.path.abspath(__file__), "--child", "--tag", tag, "--window-gib", str(gib)]
        if busy:
            cmd.append("--busy")
        r = subprocess.run(cmd, env=dict(base, **env), timeout=300)
        if r.returncode != 0:
            print(json.dumps({"config": tag, "failed": r.returncode}), flush=True)
            return r.returncode        # after a failed GPU step, start no further one
    return 0


if __name__ == "__main__":
    sys.exit(main())
