import os, sys, time, json, statistics
sys.path.insert(0, os.getcwd())
os.environ["KVCACHED_LOG_LEVEL"] = "ERROR"; os.environ.setdefault("KVCACHED_IPC_NAME", "kvc_probe")
import torch
from kvcached_amd import capi
PAGE = 2 << 20
for async_on in (0, 1):
    capi.init("cuda:0", PAGE, False)
    capi.create_kv_tensors(256 * PAGE * 2, 1, "cuda:0", 32, 2, 0, False)
    capi.set_option(capi.OPT_ASYNC_UNMAP, async_on)
    for n in (1, 8, 64):
        tm, tu = [], []
        capi.reset_stats()
        for it in range(30):
            offs = [((it * n + i) % 192) * PAGE for i in range(n)]
            t0 = time.perf_counter(); capi.map_to_kv_tensors(offs); tm.append(time.perf_counter() - t0)
            t0 = time.perf_counter(); capi.unmap_from_kv_tensors(offs); tu.append(time.perf_counter() - t0)
        capi.flush_unmaps()
        st = capi.get_stats(); drv = capi.get_driver_breakdown()
        slots = 30 * n * 64
        print(json.dumps(dict(async_unmap=async_on, page_ids=n, map_ms_p50=round(statistics.median(tm) * 1e3, 3), map_ms_max=round(max(tm) * 1e3, 3),
              unmap_ms_p50=round(statistics.median(tu) * 1e3, 3), created=st["handles_created"], reused=st["handles_reused"],
              cancelled=st["unmaps_cancelled"], shootdowns=st["tlb_shootdowns"], shoot_us=round(st["shootdown_ns"] / 1e3 / max(1, st["tlb_shootdowns"])),
              drv_us_per_slot={k: round(v / 1e3 / slots, 2) for k, v in drv.items() if v})), flush=True)
    capi.set_option(capi.OPT_ASYNC_UNMAP, 0)
    capi.shutdown()
