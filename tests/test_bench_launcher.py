"""`python bench.py --gpus N` must start N ranks by itself: the driver's command form has no launcher
in front of it. The rehearsal mode drives the whole N>1 path - launcher, rendezvous, CollectiveFanout over gloo, barrier,
max-over-ranks timing, the JSON contract - on the library's `cpu` device (bookkeeping only: its line is labelled as no
measurement). The reference's harness spawns its ranks itself as well
(benchmarks/bench_tp_ipc/kvcached_tp_ipc_benchmark.py:131,212)."""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env, *argv, timeout=300):
    env = dict(os.environ, KVC_BENCH_REHEARSAL="cpu", **extra_env)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    return subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), *argv], env=env, capture_output=True, text=True,
                          timeout=timeout)


def test_gpus_2_spawns_two_ranks_and_prints_one_line():
    out = _run({}, "--gpus", "2", "--steps", "3", "--warmup", "1")
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    assert out.stdout.strip().splitlines() == lines, out.stdout     # nothing else on stdout (gloo's connection notice goes to stderr)
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["ranks"] == [0, 1]
    assert line["steps"] == 3 and line["warmup"] == 1 and line["scaling"] == "weak"
    assert "gloo" in line["config"]["fanout"]
    assert "REHEARSAL" in line["data"]          # nobody can take this line for a measurement
    # whole-job value: both ranks' bytes over the slowest rank's time
    per_gpu = line["config"]["per_gpu_bytes_per_step"]
    assert abs(line["value"] - 2 * per_gpu / (line["ms_per_step"] * 1e-3) / 1e9) / line["value"] < 0.02
    # the shared-pool leg (BASELINE config 4) rides in the same line: rank 0's offsets reached rank 1 through the collective, one
    # file descriptor per 2 MiB slot through SCM_RIGHTS, both ranks agreed on the result - on the cpu device nothing is mapped,
    # and the line says so
    sp = line["shared_pool"]
    assert "error" not in sp, sp
    assert sp["ranks"] == 2 and sp["rccl_world_size"] == 2 and "REHEARSAL" in sp["what"]
    for mode, per in (("page_ids_as_units", 1), ("slot_by_slot", 64)):   # one descriptor per page id, or one per 2 MiB slot
        assert "error" not in sp[mode], sp[mode]
        for k in (1, 8):
            leg = sp[mode][f"{k}_page_ids"]
            assert leg["slots_2MiB"] == k * 64 and str(leg["descriptors"]) == str(k * per) and leg["signature_seen_by_every_peer"] is True
            assert leg["export_ship_import_map_ms_p50_slowest_rank"] > 0
    assert line["config"]["shared_pool_8_page_ids_ms"] == sp["page_ids_as_units"]["8_page_ids"]["export_ship_import_map_ms_p50_slowest_rank"]
    assert line["rccl_world_size"] == 2


def test_under_a_launcher_it_does_not_spawn_again():
    """torch.distributed.run sets WORLD_SIZE: the script must then be a rank, not a launcher."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, KVC_BENCH_REHEARSAL="cpu", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(port), KVC_BENCH_FORCE_DIST="1")
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1"],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["ranks"] == [0]


def test_a_dying_rank_fails_the_whole_run():
    out = _run({"KVC_BENCH_TEST_FAIL_RANK": "1"}, "--gpus", "2", "--steps", "2", "--warmup", "1", timeout=120)
    assert out.returncode == 7
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert "rank 1 exited with status 7" in out.stderr


def test_without_a_gpu_and_without_the_rehearsal_flag_it_refuses():
    env = {k: v for k, v in os.environ.items() if k not in ("KVC_BENCH_REHEARSAL", "WORLD_SIZE", "RANK")}
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("a GPU is present")
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "1", "--warmup", "0"], env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "needs a GPU" in (out.stderr + out.stdout)
