"""The two tile plans of ic3_policy_step (full tiles only / full + half tiles) are chosen by a cost model
per shape; the library reads IC3_PS_HALF once per process, so each forced plan runs the launch-chain equivalence tests
in a process of its own."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("plan", ["0", "1"])
def test_launch_chain_equivalence_under_a_forced_tile_plan(plan):
    env = dict(os.environ, IC3_PS_HALF=plan)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_policy_step_gpu.py"), "-q", "-x",
                        "-p", "no:cacheprovider", "-k", "equals_the_launch_chain or masks_and_dead"],
                       cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    tail = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else ""
    assert r.returncode == 0 and " passed" in tail, r.stdout[-2000:]


def test_wave_specialised_schedule_gives_the_same_results():
    """IC3_PS_WS=1 (csrc/policy_step_ws.hpp: one persistent workgroup per CU, matrix waves + helper waves, LDS counters instead
    of s_barrier) is the same arithmetic on another schedule: the one-hop tests against the fp64 policy + oracle env at the
    BASELINE shapes — full episodes, the E = 8192 launch geometries, auto-reset — pass unchanged under it, and the CRC of
    everything a few steps produce equals the default kernel's."""
    env = dict(os.environ, IC3_PS_WS="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_policy_step_onehop_gpu.py"),
                        os.path.join(ROOT, "tests", "test_auto_reset_gpu.py"), "-q", "-x", "-p", "no:cacheprovider", "-m", "gpu",
                        "-k", "(bf16x9 and not pp_scaled) or auto_reset_stream"],
                       cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1500)
    tail = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else ""
    assert r.returncode == 0 and " passed" in tail, r.stdout[-2000:]
    crc = {}
    for ws in ("0", "1"):
        w = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "ps_checksum_worker.py")], cwd=ROOT,
                           env=dict(os.environ, IC3_PS_WS=ws), capture_output=True, text=True, timeout=600)
        assert w.returncode == 0, w.stderr[-2000:]
        crc[ws] = [l for l in w.stdout.splitlines() if l.startswith("PS_CRC")][0]
    assert crc["0"] == crc["1"]
