"""The physical-extent pool (kvcached_amd/csrc/extent_pool.hpp) on the CPU: the header has no HIP in it, so its placement
policy - run-sized extents, waste used up before anything is created, the governor, cap / decay / pressure / reserve -
is compiled with g++ against a fake driver (tests/native/extent_pool_check.cpp) and run here. The same class serves
the GPU path (tests/test_gpu_vmm.py, the soak). What it replaces in the reference: one cuMemCreate per mapped page and one
cuMemRelease per unmapped page (csrc/page.cpp:10-17, csrc/ftensor.cpp:100-140)."""
import json
import os
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def report(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("native") / "extent_pool_check")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-Wall", "-Werror", "-fsanitize=address,undefined", "-o", exe,
                           os.path.join(REPO, "tests", "native", "extent_pool_check.cpp")],
                          env={k: v for k, v in os.environ.items() if k != "LD_PRELOAD"})
    env = {k: v for k, v in os.environ.items() if k != "LD_PRELOAD"}   # (tools_sanitize_cpu.sh preloads clang's runtime: this binary has gcc's)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-3000:]   # every REQUIRE in the program is an invariant
    return json.loads(out.stdout)


def test_invariants_hold_under_asan_and_ubsan(report):
    assert set(report) >= {"bench_shape", "requests_with_5pct_stragglers", "tides_governed", "single_pages"}


def test_the_bench_shape_needs_one_ioctl_per_64_pages_and_no_creation_after_the_first_batch(report):
    assert report["bench_shape"]["creates"] == 16
    assert report["bench_shape"]["map_ioctls_per_page"] == pytest.approx(1 / 64)


def test_request_shaped_churn_stays_within_ten_percent_of_what_is_mapped(report):
    for name in ("requests_with_5pct_stragglers", "requests_with_5pct_stragglers_sorted_free_list"):
        assert report[name]["p90"] <= 1.10, report[name]
        assert report[name]["held_over_mapped_p50"] <= 1.05


def test_single_page_extents_cannot_fragment(report):
    assert report["single_pages"]["max"] == 1.0


def test_the_governor_halves_the_adversarial_case(report):
    """Memory that grows in runs and ebbs page by page at random pins extents; extents that exist cannot be undone, but
    none are made once the waste shows."""
    assert report["tides_governed"]["p90"] < 0.75 * report["tides_ungoverned"]["p90"]
