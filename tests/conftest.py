"""pytest configuration: markers, process-wide environment, and one-time native builds.

`-m "not gpu"`: oracle vs golden fixtures, host logic of the product on its explicit "cpu" device,
C-ABI symbol checks, 2-rank gloo tests. `-m gpu`: parity of the HIP path on a real MI355X.
"""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))

# must be set before kvcached_amd.utils is imported (the config is read once at import)
os.environ.setdefault("KVCACHED_PAGE_PREALLOC_ENABLED", "false")
os.environ.setdefault("KVCACHED_IPC_NAME", f"kvc_test_{os.getpid()}")
os.environ.setdefault("KVCACHED_LOG_LEVEL", "ERROR")
os.environ.setdefault("KVCACHED_CONTIGUOUS_LAYOUT", "false")

import pytest  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")
    # checkers and product are built once per session; both are no-ops when up to date
    if not os.path.exists(os.path.join(REPO, "oracle", "libkvc_oracle.so")):
        subprocess.check_call(["make", "-C", os.path.join(REPO, "oracle"), "oracle"])
    import importlib.util
    spec = importlib.util.spec_from_file_location("kvcached_amd_build", os.path.join(REPO, "kvcached_amd", "build.py"))
    kb = importlib.util.module_from_spec(spec)   # by path: the package itself refuses to import without the extension
    spec.loader.exec_module(kb)
    try:
        stale = kb._stale(kb.LIB, kb.LIB_DEPS) or kb._stale(kb.EXT, kb.EXT_SRCS)
    except OSError:
        stale = True
    if stale:
        kb.build_all()


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container (runs on the MI355X box with -m gpu)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def oracle_lib():
    import kvc_testlib
    return kvc_testlib.load_oracle()


def pytest_sessionfinish(session, exitstatus):
    # leave /dev/shm clean (segments are unlinked by PageAllocator destructors; this catches aborts)
    prefix = os.environ.get("KVCACHED_IPC_NAME", "")
    if prefix:
        for f in os.listdir("/dev/shm"):
            if f.startswith(prefix):
                try:
                    os.unlink(os.path.join("/dev/shm", f))
                except OSError:
                    pass
