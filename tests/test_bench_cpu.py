"""bench.py pieces that run without a GPU: the CPU-baseline leg (oracle env + fp64 numpy policy in forked workers) for
every workload, and the refusal to run the timed path without an MI355X (no CPU fallback)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


@pytest.mark.parametrize("workload", ["pp_easy", "pp_hard", "tj_medium", "tj_hard"])
def test_cpu_baseline_leg(workload):
    import bench
    out = bench.cpu_baseline(workload, envs_per_proc=2, episodes=1, budget_s=0.3)
    pp = bench.WORKLOADS[workload][0] == 'predator_prey'
    assert set(out) == {"value", "unit", "cores", "kind", "sample", "reference_probe"} | ({"reference_shaped"} if pp else set())
    if pp:      # leg (ii): the reference-shaped numpy env (dense one-hot grid copy per step)
        assert out["reference_shaped"]["value"] > 0 and out["reference_shaped"]["unit"] == "agent-steps/s"
    assert out["reference_probe"]["rollout_1proc"] > 0 and "build container" in out["reference_probe"]["host"]
    assert out["unit"] == "agent-steps/s" and out["kind"] == "port" and out["cores"] >= 1
    assert out["value"] > 0
    N, T = bench.WORKLOADS[workload][1]['nagents'], bench.WORKLOADS[workload][1]['max_steps']
    steps = int(out["sample"].split("policy + C oracle env: ")[1].split(" env-steps")[0])
    assert steps >= out["cores"]                       # every worker played at least one step of one episode
    assert steps <= out["cores"] * 2 * T * 50          # and stopped near its budget
    assert N > 0


def test_bench_refuses_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a machine without a GPU")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "no CPU fallback" in (p.stderr + p.stdout)
