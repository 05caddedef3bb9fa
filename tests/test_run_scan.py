"""Batches without sorting (kvcached_amd/csrc/run_scan.hpp, KeyGroups in extent_pool.hpp): a map/unmap call turns its
slots into runs of neighbours - one page-table ioctl per run - through a bitmap per region, and a release groups its pages
by extent through a small hash table; both replaced std::sort on the hot path. Checked here against the
sort-based statement of the same thing on 6000 random batches, under ASan + UBSan, with g++ and no GPU. The reference has
no counterpart: it issues its driver calls slot by slot in the caller's order (csrc/allocator.cpp:168-207)."""
import json
import os
import subprocess

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_runs_and_groups_match_what_sorting_gives(tmp_path):
    exe = str(tmp_path / "run_scan_check")
    env = {k: v for k, v in os.environ.items() if k != "LD_PRELOAD"}   # (tools_sanitize_cpu.sh preloads clang's runtime: this binary has gcc's)
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-Wall", "-Werror", "-fsanitize=address,undefined", "-o", exe,
                           os.path.join(REPO, "tests", "native", "run_scan_check.cpp")], env=env)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    rep = json.loads(out.stdout)
    assert rep["scan_cases"] == 4000 and rep["group_cases"] == 2000 and rep["runs"] > 0
