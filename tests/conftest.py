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


HOOKS_DIR = os.path.join(REPO, "kvcached_amd", "_testhooks")


def hooks_env(base=None):
    """Environment of a child process that loads the library build with the test hooks compiled in
    (kvcached_amd/_testhooks/libkvcached_amd.so, -DKVC_TEST_HOOKS: the KVCACHED_TEST_* switches remove a safety step so that a
    test can show it would notice). The shipped library has neither the code nor the names; nothing but such children ever
    loads the hooks build."""
    env = dict(os.environ if base is None else base)
    env["KVCACHED_AMD_LIBRARY"] = os.path.join(HOOKS_DIR, "libkvcached_amd.so")
    env["LD_LIBRARY_PATH"] = HOOKS_DIR + os.pathsep + env.get("LD_LIBRARY_PATH", "")   # vmm_ops' DT_NEEDED resolves to the same file
    env["KVC_HOOKS_CHILD"] = "1"
    return env


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")
    config.addinivalue_line("markers", "hooks_build: the test body needs KVCACHED_TEST_* hooks: it runs in a child pytest that loads "
                                       "the library build with the hooks compiled in (the parent session keeps the shipped library)")
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


@pytest.hookimpl(tryfirst=True)
def pytest_pyfunc_call(pyfuncitem):
    """A test marked `hooks_build` is executed by a child pytest session that loads the hooks build; this session only checks
    the child's verdict (and never sees a hook itself)."""
    if pyfuncitem.get_closest_marker("hooks_build") is None or os.environ.get("KVC_HOOKS_CHILD") == "1":
        return None
    env = hooks_env()
    env["KVCACHED_IPC_NAME"] = f"kvc_test_hooks_{os.getpid()}"
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", "-m", "gpu or not gpu", pyfuncitem.nodeid],
                       cwd=REPO, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, "in the hooks build:\n" + r.stdout[-3000:] + r.stderr[-1500:]
    assert "1 passed" in r.stdout, r.stdout[-800:]
    return True


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
