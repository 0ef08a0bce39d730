"""GPU: the N > 1 control flow on ONE device (the 8-GPU box belongs to the driver): two ranks launched with
torch.distributed.run share GPU 0.  RCCL refuses two ranks on one device, so the collectives of these tests go over
gloo (bench.py falls back to it by itself for its timing barrier; the trainer takes --dist_backend gloo); what is
exercised is everything else: rank-dependent global env ids, one seed for all replicas, rank-0 parameters everywhere,
gradient / stat all-reduce (multi_processing.py:74-98), one JSON line from rank 0."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _torchrun(nproc, args, env=None, timeout=600):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port())] + args
    e = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    e.update(env or {})
    return subprocess.run(cmd, cwd=ROOT, env=e, capture_output=True, text=True, timeout=timeout)


def test_bench_two_ranks_share_one_device():
    r = _torchrun(2, ["bench.py", "--gpus", "2", "--steps", "12", "--warmup", "4", "--nenvs", "512", "--no-cpu-baseline"],
                  env={"IC3_BENCH_DEVICE": "0"})
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 12 and d["scaling"] == "weak"
    assert d["config"]["parallelism"] == "env-shard x2" and d["config"]["envs_per_gpu"] == 512
    assert d["live_frac"] == pytest.approx(1.0)                 # PP-hard with a random-init policy never ends early
    # whole-job aggregate: agents x (envs of BOTH ranks) x steps / max-over-ranks time
    assert d["value"] == pytest.approx(10 * 1024 * 12 / (d["ms_per_step"] * 12 * 1e-3), rel=1e-3)
    assert d["cpu_baseline"] is None and d["roofline"]["bound"] == "hbm"


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` WITHOUT torch.distributed.run (WORLD_SIZE unset): bench.py starts the two ranks itself
    (round 3: it died on an assert) and rank 0 prints the one line with per-rank times."""
    e = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", IC3_BENCH_DEVICE="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "12", "--warmup", "4", "--nenvs", "512",
                        "--no-cpu-baseline"], cwd=ROOT, env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and len(d["ms_per_step_ranks"]) == 2 and d["collectives"] == "gloo"   # (nccl on 2 real GPUs)
    assert d["config"]["parallelism"] == "env-shard x2"
    assert d["value"] == pytest.approx(10 * 1024 * 12 / (d["ms_per_step"] * 12 * 1e-3), rel=1e-3)


def test_bench_gpus_8_self_launch_on_one_device():
    """Round-4 verdict item 8: the driver's 8-GPU command, `python bench.py --gpus 8 --steps K --warmup W`, kept runnable on a
    one-GPU box: bench.py starts the EIGHT ranks itself (WORLD_SIZE unset), the IC3_BENCH_DEVICE hook puts them on one GPU
    (gloo stands in for RCCL, which refuses duplicate devices), 256 envs per rank.  One JSON line from rank 0: eight per-rank
    times, value = agents x (sum of all ranks' envs) x steps / the MAX-over-ranks time, weak scaling."""
    e = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", IC3_BENCH_DEVICE="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "8", "--steps", "10", "--warmup", "3", "--nenvs", "256"],
                       cwd=ROOT, env=e, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["steps"] == 10 and d["warmup"] == 3 and d["scaling"] == "weak"
    assert len(d["ms_per_step_ranks"]) == 8 and d["collectives"] == "gloo"
    assert d["ms_per_step"] == pytest.approx(max(d["ms_per_step_ranks"]), rel=1e-3)          # MAX over ranks
    assert d["config"]["parallelism"] == "env-shard x8" and d["config"]["envs_per_gpu"] == 256
    assert d["live_frac"] == pytest.approx(1.0)
    assert d["value"] == pytest.approx(10 * 8 * 256 * 10 / (d["ms_per_step"] * 10 * 1e-3), rel=1e-3)
    assert d["cpu_baseline"] is None                              # rank 0 at N = 1 only


PP = ['--env_name', 'predator_prey', '--nagents', '3', '--nprocesses', '1', '--num_epochs', '2', '--epoch_size', '1',
      '--hid_size', '64', '--detach_gap', '10', '--lrate', '0.001', '--dim', '5', '--max_steps', '20', '--ic3net',
      '--vision', '0', '--recurrent', '--dist_backend', 'gloo', '--device', '0',
      '--batch_size', '100']      # one get_episode() per update on every rank (batch_size counts per process, like the
                                  # reference's per-worker batch_size): 2 x 8 envs and 1 x 16 envs then play the same episodes


def _load(prefix, rank):
    return torch.load('%s.rank%d.pt' % (prefix, rank), weights_only=False)


def test_replicas_stay_identical_with_the_default_seed(tmp_path):
    """--seed -1 (the reference's default, main.py:157-159): rank 0's draw is adopted by every rank and rank 0's
    parameters are broadcast, so that identical all-reduced gradients keep the replicas bit-identical."""
    prefix = str(tmp_path / "r")
    r = _torchrun(2, ["-m", "ic3net_amd.main"] + PP + ['--nenvs', '8', '--seed', '-1'], env={"IC3_DUMP_PARAMS": prefix})
    assert r.returncode == 0, r.stderr[-3000:]
    a, b = _load(prefix, 0), _load(prefix, 1)
    for k in a['params']:
        assert torch.equal(a['params'][k], b['params'][k]), k
    assert sum(l.startswith('Epoch') for l in r.stdout.splitlines()) == 2      # rank 0 alone prints


def test_two_ranks_equal_one_process_with_twice_the_envs(tmp_path):
    """Global env ids make the sharded run the same job: 2 ranks x 8 envs == 1 process x 16 envs (env and sampling
    streams are keyed by (seed, global env id); gradients and stats are summed over ranks and divided by the global
    num_steps) — up to fp32 summation order."""
    p2, p1 = str(tmp_path / "two"), str(tmp_path / "one")
    r2 = _torchrun(2, ["-m", "ic3net_amd.main"] + PP + ['--nenvs', '8', '--seed', '5'], env={"IC3_DUMP_PARAMS": p2})
    assert r2.returncode == 0, r2.stderr[-3000:]
    r1 = _torchrun(1, ["-m", "ic3net_amd.main"] + PP + ['--nenvs', '16', '--seed', '5'], env={"IC3_DUMP_PARAMS": p1})
    assert r1.returncode == 0, r1.stderr[-3000:]
    two, one = _load(p2, 0), _load(p1, 0)
    for k in one['params']:
        np.testing.assert_allclose(two['params'][k].numpy(), one['params'][k].numpy(), rtol=0, atol=2e-5, err_msg=k)
    for k in ('reward', 'success', 'steps_taken', 'comm_action'):
        np.testing.assert_allclose(np.asarray(two['log'][k], np.float64), np.asarray(one['log'][k], np.float64),
                                   rtol=1e-5, atol=1e-6, err_msg=k)
    for k in ('value_loss', 'action_loss', 'entropy'):
        np.testing.assert_allclose(np.asarray(two['log'][k], np.float64), np.asarray(one['log'][k], np.float64),
                                   rtol=2e-4, atol=1e-5, err_msg=k)
