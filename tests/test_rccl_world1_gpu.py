"""GPU: the RCCL code paths on the one GPU of the box (round-2 verdict item 4) — a ONE-rank `backend='nccl'` process
group runs (a) bench.py's timing barrier, per-rank all_gather and MAX / SUM reductions on device tensors and (b) the
update-time exchange of ic3net_amd.sharding (gradient / stat all-reduce, seed and parameter broadcast:
/root/reference/multi_processing.py:74-98, main.py:157-159,177-178).  The N > 1 control flow (global env ids, one JSON line)
is covered on two ranks over gloo by tests/test_multirank_gpu.py; an 8-GPU RCCL run belongs to the driver."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")


def test_sharding_collectives_over_a_one_rank_rccl_group():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_world1_worker.py")], cwd=ROOT, env=_env(),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RCCL_WORLD1_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])


def test_bench_timing_path_over_rccl():
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "1", "--rccl", "1", "--steps", "12", "--warmup", "4",
                        "--nenvs", "512", "--no-cpu-baseline"], cwd=ROOT, env=_env(), capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["collectives"] == "nccl", d["collectives"]          # RCCL came up (no silent gloo fallback)
    assert d["n_gpus"] == 1 and len(d["ms_per_step_ranks"]) == 1
    assert d["ms_per_step_ranks"][0] == pytest.approx(d["ms_per_step"], rel=1e-6)
    assert d["timing"]["launch_ms_min"] <= d["timing"]["launch_ms_median"] <= d["timing"]["launch_ms_max"]
