#!/bin/bash
# One GPU session (run through gpurun): stage 1 = parity tests, smoke, the driver's bench command, kernel-trace profile, PMC
# passes; stage 2 = soaks with the footprint sampled, configs 2-4 in full.
# usage: ./tools_gpu_round.sh <stage: 1|2|tests> [round tag, default r02]
set -o pipefail
export TMPDIR=/tmp
STAGE=${1:-1}; R=${2:-r03}
mkdir -p gpurun_out
if [ "$STAGE" = "1" ] || [ "$STAGE" = "tests" ]; then
  echo "== pytest -m gpu"
  timeout -k 10 1000 python -u -m pytest tests -m gpu -q --timeout=300 > gpurun_out/pytest_gpu.log 2>&1; rc=$?
  grep -E "^(FAILED|ERROR)|passed|failed|^E  " gpurun_out/pytest_gpu.log | tail -12 | cut -c1-300; echo "pytest rc=$rc"
  [ $rc -ne 0 ] && exit $rc
  [ "$STAGE" = "tests" ] && exit 0
  echo "== smoke"
  timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; rc=$?
  tail -1 gpurun_out/smoke.log | cut -c1-300; echo "smoke rc=$rc"
  [ $rc -ne 0 ] && exit $rc
  echo "== bench (the driver's command)"
  timeout -k 10 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench.log 2> gpurun_out/bench.err; rc=$?
  tail -1 gpurun_out/bench.log | cut -c1-1200; grep -v amdgpu.ids gpurun_out/bench.err | tail -5; echo "bench rc=$rc"
  [ $rc -ne 0 ] && exit $rc
  echo "== rocprofv3 kernel trace (same command, context legs off)"
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_kt -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-variants --no-cpu-baseline > gpurun_out/prof_kt.log 2>&1; rc=$?
  grep '"metric"' gpurun_out/prof_kt.log | cut -c1-300; echo "rocprof kt rc=$rc"
  echo "== rocprofv3 PMC WRITE_SIZE (own pass)"
  timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/prof_pmc_w -- python3 bench.py --no-variants --no-cpu-baseline --steps 4 --warmup 2 > gpurun_out/prof_pmc_w.log 2>&1; echo "pmc write rc=$?"
  echo "== rocprofv3 PMC FETCH_SIZE (own pass)"
  timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/prof_pmc_r -- python3 bench.py --no-variants --no-cpu-baseline --steps 4 --warmup 2 > gpurun_out/prof_pmc_r.log 2>&1; echo "pmc fetch rc=$?"
  find gpurun_out/prof_kt gpurun_out/prof_pmc_w gpurun_out/prof_pmc_r -name "*.csv" | head -20
fi
if [ "$STAGE" = "2" ]; then
  echo "== soak: default pool, then pool off (fragmentation alone), then prealloc + watcher + resizes, then one buffer per page"
  : > gpurun_out/${R}_soak.jsonl
  for args in "--seconds 60 --compat --touch-unbacked" "--seconds 60 --touch-unbacked" "--seconds 45 --compat --prealloc --touch-unbacked" "--seconds 45 --compat --touch-unbacked --unmap-invalidation-us 300" "--seconds 60" "--seconds 60 --pool-mb 0" "--seconds 60 --prealloc" "--seconds 45 --pool-mb 0 --extent-pages 32" "--seconds 45 --pool-mb 0 --extent-pages 1" "--seconds 30 --extent-pages 1" "--seconds 45 --compat" "--seconds 45 --async-unmap --prealloc" "--seconds 30 --compat --touch-unbacked --no-prt" "--seconds 30 --compat --touch-unbacked --backend hybrid"; do
    timeout -k 10 200 python benchmarks/soak_manager.py $args 2> gpurun_out/soak.err | tail -1 >> gpurun_out/${R}_soak.jsonl; rc=$?
    tail -1 gpurun_out/${R}_soak.jsonl | cut -c1-700; echo "soak ($args) rc=$rc"
    [ $rc -ne 0 ] && { grep -v amdgpu.ids gpurun_out/soak.err | tail -20; exit $rc; }
  done
  echo "== elastic (config 3)"
  timeout -k 10 600 python benchmarks/bench_elastic.py 2>&1 | grep -v "amdgpu.ids\|IPC listener" > gpurun_out/bench_elastic.log; cut -c1-700 gpurun_out/bench_elastic.log
  echo "== tp ipc (config 4)"
  timeout -k 10 300 python benchmarks/bench_tp_ipc.py 2>&1 | grep -v "amdgpu.ids\|IPC listener" > gpurun_out/bench_tp_ipc.log; cut -c1-400 gpurun_out/bench_tp_ipc.log
  echo "== bench_vmm (config 2 in full)"
  timeout -k 10 600 python benchmarks/bench_vmm.py 2>&1 | grep -v "amdgpu.ids" > gpurun_out/bench_vmm.log; cut -c1-900 gpurun_out/bench_vmm.log
  echo "== sglang glue"
  timeout -k 10 300 python benchmarks/bench_sglang_glue.py 2>&1 | grep -v "amdgpu.ids" > gpurun_out/bench_sglang_glue.log; tail -3 gpurun_out/bench_sglang_glue.log | cut -c1-300
fi
