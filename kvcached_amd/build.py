"""In-tree build of the native code (no setup.py, no JIT cache: the .so files must sit next to the
sources so that they travel to the GPU box with the repo snapshot).

  libkvcached_amd.so   HIP VMM backend + gfx950 kernels + page allocator, C ABI (include/kvcached_amd.h)
                       built with hipcc --offload-arch=gfx950
  vmm_ops.<abi>.so     pybind11/torch module with the reference's `kvcached.vmm_ops` surface,
                       a thin layer over the C ABI; built with g++ against the installed torch

Usage: python kvcached_amd/build.py [--force]
"""
from __future__ import annotations

import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
REPO = os.path.dirname(HERE)
LIB = os.path.join(HERE, "libkvcached_amd.so")
# The same library with the test hooks compiled in (-DKVC_TEST_HOOKS: switches that remove a safety step so that the tests can
# show they would notice). Never loaded by the product: only child processes of the tests that ask for it by path
# (KVCACHED_AMD_LIBRARY + LD_LIBRARY_PATH, tests/kvc_testlib.py: hooks_env()).
HOOKS_DIR = os.path.join(HERE, "_testhooks")
HOOKS_LIB = os.path.join(HOOKS_DIR, "libkvcached_amd.so")
EXT = os.path.join(HERE, "vmm_ops" + sysconfig.get_config_var("EXT_SUFFIX"))
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")
HIPCC = os.path.join(ROCM, "bin", "hipcc")
ARCH = "gfx950"

LIB_SRCS = ["c_api.cpp", "kv_allocator.cpp", "gpu_context.cpp", "page_allocator.cpp", "kernels.hip", "index_kernels.hip"]
LIB_DEPS = LIB_SRCS + ["common.hpp", "hip_vmm.hpp", "drm_vm.hpp", "extent_pool.hpp", "kernels.hpp", "kv_allocator.hpp", "mem_info.hpp",
                       "page_allocator.hpp", "run_scan.hpp", "../../include/kvcached_amd.h"]
EXT_SRCS = ["vmm_ops.cpp"]


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in deps)


def _run(cmd):
    print("[kvcached_amd.build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def build_lib(force: bool = False) -> str:
    for target, extra in ((LIB, []), (HOOKS_LIB, ["-DKVC_TEST_HOOKS=1"])):
        if force or _stale(target, LIB_DEPS):
            os.makedirs(os.path.dirname(target), exist_ok=True)
            cmd = [HIPCC, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-shared", "-pthread",
                   "-Wall", "-Wno-unused-result"] + extra + ["-x", "hip"]
            cmd += [os.path.join(CSRC, s) for s in LIB_SRCS]
            cmd += ["-o", target, f"-L{os.path.join(ROCM, 'lib')}", "-lhsa-runtime64", "-ldl"]  # ROCr directly (hybrid/drm VMM backends); libdrm_amdgpu is dlopen()ed
            _run(cmd)
    return LIB


def build_ext(force: bool = False) -> str:
    if force or _stale(EXT, EXT_SRCS + ["../../include/kvcached_amd.h"]) or os.path.getmtime(LIB) > os.path.getmtime(EXT):
        import pybind11
        import torch
        tdir = os.path.dirname(torch.__file__)
        cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-w",
               "-DTORCH_EXTENSION_NAME=vmm_ops", "-DTORCH_API_INCLUDE_EXTENSION_H",
               f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}",
               "-DUSE_ROCM=1", "-D__HIP_PLATFORM_AMD__=1",
               f"-I{os.path.join(REPO, 'include')}", f"-I{tdir}/include", f"-I{tdir}/include/torch/csrc/api/include",
               f"-I{ROCM}/include", f"-I{pybind11.get_include()}", f"-I{sysconfig.get_paths()['include']}"]
        cmd += [os.path.join(CSRC, s) for s in EXT_SRCS]
        cmd += ["-o", EXT, f"-L{HERE}", "-lkvcached_amd", f"-L{tdir}/lib", "-lc10", "-ltorch", "-ltorch_cpu",
                "-ltorch_python", "-Wl,-rpath,$ORIGIN", f"-Wl,-rpath,{tdir}/lib", f"-Wl,-rpath,{ROCM}/lib"]
        _run(cmd)
    return EXT


def build_all(force: bool = False):
    return build_lib(force), build_ext(force)


if __name__ == "__main__":
    build_all("--force" in sys.argv)
