#!/bin/bash
# one GPU session: parity tests, smoke, bench, kernel-trace profile
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "== pytest -m gpu"
[ "$1" = "nopytest" ] || timeout -k 10 900 python -u -m pytest tests -m gpu -x -v --timeout=240 > gpurun_out/pytest_gpu.log 2>&1; rc=$?
[ "$1" = "nopytest" ] && rc=0
tail -5 gpurun_out/pytest_gpu.log; echo "pytest rc=$rc"
[ $rc -ne 0 ] && exit $rc
echo "== smoke"
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; rc=$?
tail -3 gpurun_out/smoke.log; echo "smoke rc=$rc"
[ $rc -ne 0 ] && exit $rc
echo "== bench"
timeout -k 10 600 python bench.py > gpurun_out/bench.log 2> gpurun_out/bench.err; rc=$?
tail -2 gpurun_out/bench.log; tail -5 gpurun_out/bench.err; echo "bench rc=$rc"
[ $rc -ne 0 ] && exit $rc
echo "== rocprofv3 kernel trace"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_kt -- python3 bench.py --no-variants --no-cpu-baseline > gpurun_out/prof_kt.log 2>&1; rc=$?
tail -3 gpurun_out/prof_kt.log; echo "rocprof rc=$rc"
find gpurun_out/prof_kt -name "*stats*" | head
