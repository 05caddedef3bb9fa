#!/bin/bash
# Host-side sanitizer passes over the library (GPU ASAN/XNACK are not available on the pool: CPU build only).
# Builds libkvcached_amd.so with -Xarch_host -fsanitize=<address|thread>, swaps it in for the in-tree library,
# runs the CPU test suite under the matching runtime, and restores the normal build.
# (The tests' hooks build, kvcached_amd/_testhooks/, is loaded only by children of GPU tests and is left alone; the time stamp
# of the swapped library is kept older than vmm_ops' so that tests/conftest.py does not rebuild it in the middle of the run.)
# usage: ./tools_sanitize_cpu.sh address|thread
set -e
SAN=${1:-address}
RT=/opt/rocm/lib/llvm/lib/clang/22/lib/linux/libclang_rt.$([ "$SAN" = address ] && echo asan || echo tsan)-x86_64.so
cd "$(dirname "$0")"
mkdir -p build/$SAN
(cd kvcached_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -shared -pthread \
    -Xarch_host -fsanitize=$SAN -Xarch_host -fno-omit-frame-pointer -x hip \
    c_api.cpp kv_allocator.cpp gpu_context.cpp page_allocator.cpp kernels.hip index_kernels.hip -o ../../build/$SAN/libkvcached_amd.so -L/opt/rocm/lib -lhsa-runtime64 -ldl)
cp kvcached_amd/libkvcached_amd.so build/libkvcached_amd.so.bak
restore() { cp build/libkvcached_amd.so.bak kvcached_amd/libkvcached_amd.so; sleep 1; touch kvcached_amd/vmm_ops.*.so; }
trap restore EXIT
cp build/$SAN/libkvcached_amd.so kvcached_amd/libkvcached_amd.so; sleep 1; touch kvcached_amd/vmm_ops.*.so
# (exitcode=0: a report does not fail the process it was made in - torch's own gloo backend has races of its own that would fail the
# 2-rank tests; every process writes its reports to build/$SAN/reports.<pid>, and the ones that name this library are counted)
rm -f build/$SAN/reports.*
LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:halt_on_error=0:log_path=$PWD/build/$SAN/reports \
    TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0 exitcode=0 log_path=$PWD/build/$SAN/reports" \
    python -m pytest tests -q -m "not gpu" -p no:cacheprovider 2>&1 | tee build/$SAN/run.log | tail -3
all=$(cat build/$SAN/reports.* 2>/dev/null | grep -c 'ERROR: AddressSanitizer\|WARNING: ThreadSanitizer')
ours=$(cat build/$SAN/reports.* 2>/dev/null | awk '/ERROR: AddressSanitizer|WARNING: ThreadSanitizer/{if(hit)n++; hit=0} /libkvcached_amd|kvcached_amd\/csrc|vmm_ops/{hit=1} END{if(hit)n++; print n+0}')
echo "sanitizer reports: $all in all processes, $ours naming this library"
