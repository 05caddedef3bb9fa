#!/bin/bash
# one GPU session: parity tests, smoke, bench, kernel-trace profile, PMC passes
# usage: ./tools_gpu_round1.sh [nopytest]
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out
if [ "$1" != "nopytest" ]; then
  echo "== pytest -m gpu"
  timeout -k 10 900 python -u -m pytest tests -m gpu -x -v --timeout=240 > gpurun_out/pytest_gpu.log 2>&1; rc=$?
  tail -4 gpurun_out/pytest_gpu.log; echo "pytest rc=$rc"
  [ $rc -ne 0 ] && exit $rc
fi
echo "== smoke"
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; rc=$?
tail -1 gpurun_out/smoke.log; echo "smoke rc=$rc"
[ $rc -ne 0 ] && exit $rc
echo "== bench"
timeout -k 10 600 python bench.py > gpurun_out/bench.log 2> gpurun_out/bench.err; rc=$?
tail -1 gpurun_out/bench.log; grep -v amdgpu.ids gpurun_out/bench.err | tail -5; echo "bench rc=$rc"
[ $rc -ne 0 ] && exit $rc
echo "== rocprofv3 kernel trace (same command as the bench, context legs off)"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_kt -- python3 bench.py --no-variants --no-cpu-baseline > gpurun_out/prof_kt.log 2>&1; rc=$?
grep '"metric"' gpurun_out/prof_kt.log | cut -c1-300; echo "rocprof kt rc=$rc"
echo "== rocprofv3 PMC WRITE_SIZE (own pass)"
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/prof_pmc_w -- python3 bench.py --no-variants --no-cpu-baseline --steps 4 --warmup 2 > gpurun_out/prof_pmc_w.log 2>&1; echo "pmc write rc=$?"
echo "== rocprofv3 PMC FETCH_SIZE (own pass)"
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/prof_pmc_r -- python3 bench.py --no-variants --no-cpu-baseline --steps 4 --warmup 2 > gpurun_out/prof_pmc_r.log 2>&1; echo "pmc fetch rc=$?"
find gpurun_out/prof_kt gpurun_out/prof_pmc_w gpurun_out/prof_pmc_r -name "*.csv" | head -20
echo "== elastic (config 3)"
timeout -k 10 600 python benchmarks/bench_elastic.py 2>&1 | grep -v "amdgpu.ids\|IPC listener" > gpurun_out/bench_elastic.log; cut -c1-700 gpurun_out/bench_elastic.log
echo "== tp ipc (config 4)"
timeout -k 10 300 python benchmarks/bench_tp_ipc.py 2>&1 | grep -v "amdgpu.ids\|IPC listener" > gpurun_out/bench_tp_ipc.log; cut -c1-400 gpurun_out/bench_tp_ipc.log
echo "== bench_vmm (config 2 in full)"
timeout -k 10 600 python benchmarks/bench_vmm.py 2>&1 | grep -v "amdgpu.ids" > gpurun_out/bench_vmm.log; cut -c1-900 gpurun_out/bench_vmm.log
echo "== sglang glue"
timeout -k 10 300 python benchmarks/bench_sglang_glue.py 2>&1 | grep -v "amdgpu.ids" > gpurun_out/bench_sglang_glue.log; tail -3 gpurun_out/bench_sglang_glue.log | cut -c1-300
