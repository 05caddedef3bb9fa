"""CollectiveFanout over RCCL (backend "nccl") on the GPU: a 1-rank group is all a 1-GPU box allows (RCCL refuses
two ranks on one device), but it exercises what the multi-GPU bench relies on with this torch build — process-group
creation with device_id, a device-memory int64 broadcast, the status all-reduce — plus bench.py's own N>1 code path
(`KVC_BENCH_FORCE_DIST=1`). World size 2 is covered on CPU with gloo (tests/test_tp_ipc.py)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ["KVC_REPO"])
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda:0"))
from kvcached_amd import capi
from kvcached_amd.tp_ipc_util import CollectiveFanout
PAGE = 2 << 20
capi.init("cuda:0", PAGE, False)
ts = capi.create_kv_tensors(64 * PAGE, 1, "cuda:0", 1, 1, 0, True)
f = CollectiveFanout(device="cuda:0")
assert f.world_size == 1 and str(f._buf.device) == "cuda:0"
offs = [i * PAGE for i in (5, 1, 9, 2)]
capi.reset_stats()
assert f.map_to_kv_tensors(offs) == offs
assert capi.get_stats()["pages_mapped"] == 4
assert f.unmap_from_kv_tensors(offs) == offs
assert capi.get_stats()["pages_unmapped"] == 4
big = [i * PAGE for i in range(64)]
assert f.map_to_kv_tensors(big) == big and f.unmap_from_kv_tensors(big) == big
try:
    f.unmap_from_kv_tensors([PAGE * 1000])          # outside the region: the rank reports failure, everybody raises
    raise SystemExit("expected a failure")
except RuntimeError as e:
    assert "failed to (un)map" in str(e)
capi.shutdown()
dist.barrier()
dist.destroy_process_group()
print("FANOUT_OK")
"""


def _free_port() -> str:
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return str(sk.getsockname()[1])


def _env():
    env = dict(os.environ, KVC_REPO=REPO, MASTER_ADDR="127.0.0.1", MASTER_PORT=_free_port(), RANK="0", WORLD_SIZE="1",
               LOCAL_RANK="0", KVCACHED_LOG_LEVEL="ERROR", HSA_ENABLE_IPC_MODE_LEGACY="0")
    return env


def test_collective_fanout_over_rccl_single_rank():
    out = subprocess.run([sys.executable, "-c", SCRIPT], env=_env(), capture_output=True, text=True, timeout=240)
    assert out.returncode == 0 and "FANOUT_OK" in out.stdout, (out.stdout[-500:], out.stderr[-1500:])


def test_bench_distributed_path_single_rank():
    """bench.py with the process group forced on: the line must come out with the N>1 fan-out in `config`."""
    env = _env()
    env.update(KVC_BENCH_FORCE_DIST="1")
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
                          "--no-variants", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=300)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and lines, (out.stdout[-500:], out.stderr[-1500:])
    d = json.loads(lines[-1])
    assert d["n_gpus"] == 1 and d["value"] > 0 and "nccl broadcast" in d["config"]["fanout"]
    assert d["roofline"]["frac"] > 0.3
