#!/usr/bin/env python3
"""Copy the newest rocprofv3 outputs of gpurun_out/ into profiles/ (tracked) and derive
profiles/zero_fill_traffic.json (read by bench.py for roofline.traffic)."""
import csv, glob, json, os, shutil, statistics, sys
R = sys.argv[1] if len(sys.argv) > 1 else "r01"
def newest(pattern):
    f = sorted(glob.glob(pattern), key=os.path.getmtime)
    return f[-1] if f else None
os.makedirs("profiles", exist_ok=True)
ks, kt = newest("gpurun_out/prof_kt/runc/*_kernel_stats.csv"), newest("gpurun_out/prof_kt/runc/*_kernel_trace.csv")
shutil.copy(ks, f"profiles/{R}_rocprofv3_kernel_stats.csv")
shutil.copy(newest("gpurun_out/prof_kt/runc/*_domain_stats.csv"), f"profiles/{R}_rocprofv3_domain_stats.csv")
rows = list(csv.DictReader(open(kt)))
# the bench's fill launches (grid = pages x 32 workgroups x 512 threads): round 1 filled a 1024-page batch as 768 + 256 pages inside
# the map call; since round 2 it is ONE 1024-page launch, queued behind the unmap (pages are zeroed on their way back)
SHAPES = {"12582912": 768, "4194304": 256, "16777216": 1024}
big = [r for r in rows if "zero_fill" in r["Kernel_Name"] and r["Grid_Size_X"] in SHAPES]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in big]
algo = [SHAPES[r["Grid_Size_X"]] * 2097152 for r in big]
w = list(csv.DictReader(open(newest("gpurun_out/prof_pmc_w/runc/*_counter_collection.csv"))))
rd = list(csv.DictReader(open(newest("gpurun_out/prof_pmc_r/runc/*_counter_collection.csv"))))
ws = [float(x["Counter_Value"]) for x in w if x["Counter_Name"] == "WRITE_SIZE" and x["Grid_Size"] in SHAPES and "zero_fill" in x["Kernel_Name"]]
fs = [float(x["Counter_Value"]) for x in rd if x["Counter_Name"] == "FETCH_SIZE" and x["Grid_Size"] in SHAPES and "zero_fill" in x["Kernel_Name"]]
wa = [SHAPES[x["Grid_Size"]] * 2097152 for x in w if x["Counter_Name"] == "WRITE_SIZE" and x["Grid_Size"] in SHAPES and "zero_fill" in x["Kernel_Name"]]
bench = json.loads(open("gpurun_out/bench.log").read().strip().splitlines()[-1])
json.dump(bench, open(f"profiles/{R}_bench_n1.json", "w"))
per_shape = {}
for r, us in zip(big, d):
    per_shape.setdefault(SHAPES[r["Grid_Size_X"]], []).append(us)
out = {"kernel": "kvc::zero_fill_pages_kernel<512,false,true>",
       "launch_shape": "a 1024-page batch = ONE launch (2 GiB), queued on the library's scrub stream when the batch is unmapped; 32 workgroups x 512 threads per 2 MiB page",
       "algorithmic_bytes_per_launch": int(sum(algo) / len(algo)),
       "rocprofv3_kernel_trace": {"launches": len(d), "avg_us": round(sum(d) / len(d), 2), "median_us": round(statistics.median(d), 2),
                                  "min_us": round(min(d), 2), "max_us": round(max(d), 2), "GBps_at_avg": round(sum(algo) / sum(d) / 1e3, 1),
                                  "per_shape": {f"{k}_pages": {"launches": len(v), "avg_us": round(sum(v) / len(v), 2),
                                                               "GBps": round(k * 2097152 / (sum(v) / len(v)) / 1e3, 1)} for k, v in sorted(per_shape.items())}},
       "bench_hip_events": {"avg_launch_us": bench["roofline"]["avg_launch_us"], "achieved_GBps": bench["roofline"]["achieved"]},
       "pmc": {"WRITE_SIZE_KiB_per_launch": statistics.mean(ws), "FETCH_SIZE_KiB_per_launch_raw": round(statistics.mean(fs), 2),
               "WRITE_SIZE_over_algorithmic": round(sum(ws) * 1024 / sum(wa), 6),
               "note": "separate --pmc passes; WRITE_SIZE exact for 16 B/lane streaming stores; FETCH_SIZE doubled (gfx950 tallies 128 B requests as 64 B), MI355X_MICROARCH.md"},
       "write_bytes_per_launch": int(statistics.mean(ws) * 1024), "read_bytes_per_launch": int(2 * statistics.mean(fs) * 1024)}
out["hbm_bytes_per_launch"] = out["write_bytes_per_launch"] + out["read_bytes_per_launch"]
json.dump(out, open("profiles/zero_fill_traffic.json", "w"), indent=1)
for src, dst in (("bench_elastic.log", f"{R}_bench_elastic.jsonl"), ("bench_tp_ipc.log", f"{R}_bench_tp_ipc.jsonl"),
                 ("create_diag.log", f"{R}_create_release_order.log"), ("compact_bench3.log", f"{R}_compact_bench.jsonl"),
                 ("bench_n2_rehearsal.log", f"{R}_bench_n2_gloo_rehearsal.log"), ("pytest_gpu.log", f"{R}_pytest_gpu.log"),
                 ("bench_vmm.log", f"{R}_bench_vmm.jsonl"), ("bench_sglang_glue.log", f"{R}_bench_sglang_glue.jsonl")):
    if os.path.exists("gpurun_out/" + src):
        shutil.copy("gpurun_out/" + src, "profiles/" + dst)
# ---- compact_blocks (benchmarks/bench_compact.py --profile-shape under rocprofv3, build/prof_compact.sh)
ckt = newest("gpurun_out/prof_compact_kt/runc/*_kernel_trace.csv")
if ckt:
    from collections import defaultdict
    dur, wr, rd_ = defaultdict(list), defaultdict(list), defaultdict(list)
    for r in csv.DictReader(open(ckt)):
        if "compact_blocks" in r["Kernel_Name"]:
            dur[r["Grid_Size_X"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for path, name, dst in (("gpurun_out/prof_compact_w/runc/*_counter_collection.csv", "WRITE_SIZE", wr),
                            ("gpurun_out/prof_compact_r/runc/*_counter_collection.csv", "FETCH_SIZE", rd_)):
        for r in csv.DictReader(open(newest(path))):
            if r["Counter_Name"] == name and "compact_blocks" in r["Kernel_Name"]:
                dst[r["Grid_Size"]].append(float(r["Counter_Value"]))
    shapes = []
    for grid, d in dur.items():
        moves = int(grid) // 256 // 64               # grid = moves x 64 regions x 1 tile (32 KiB block = one 32 KiB tile since round 3) x 256 threads
        algo = 32768 * 64 * moves
        w, f = int(statistics.mean(wr[grid]) * 1024), int(2 * statistics.mean(rd_[grid]) * 1024)
        shapes.append({"moves": moves, "launches": len(d), "avg_us": round(sum(d) / len(d), 1),
                       "algorithmic_read_bytes": algo, "algorithmic_write_bytes": algo, "pmc_write_bytes": w, "pmc_read_bytes": f,
                       "traffic_over_algorithmic": round((w + f) / (2 * algo), 4),
                       "GBps_read_plus_write": round(2 * algo / (sum(d) / len(d)) / 1e3, 1)})
    json.dump({"kernel": "kvc::compact_blocks_big_kernel<true, true> (LDS-staged, a contiguous eighth of the pairs per XCD, non-temporal, 32 KiB tiles)", "geometry": "Llama-3-8B: 64 regions x 32 KiB blocks",
               "note": "rocprofv3 --kernel-trace for durations; WRITE_SIZE and FETCH_SIZE in separate --pmc passes; FETCH_SIZE doubled (gfx950 tallies 128 B requests as 64 B)",
               "per_launch": shapes}, open(f"profiles/{R}_compact_traffic.json", "w"), indent=1)
    shutil.copy(newest("gpurun_out/prof_compact_kt/runc/*_kernel_stats.csv"), f"profiles/{R}_rocprofv3_compact_kernel_stats.csv")
print(open(f"profiles/{R}_rocprofv3_kernel_stats.csv").read())
print(json.dumps(out["rocprofv3_kernel_trace"]), json.dumps(out["bench_hip_events"]))
