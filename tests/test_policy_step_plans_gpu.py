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
