"""Worst error of ic3_policy_step against the fp64 policy (oracle/policy_ref.py) + oracle env over full free-running
episodes, default fp32 gate product vs the gate_split experiment:  python tools/exp/gate_split_accuracy.py
(the machinery of tests/test_policy_step_onehop_gpu.py; this is a measurement script, it imports the test helper and,
through it, the oracle — it is not part of the product)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import test_policy_step_onehop_gpu as t  # noqa: E402

t.TOL = 1.0                                        # measure, do not assert
for workload, E, T in (("pp_hard", 13, 80), ("tj_hard", 7, 80), ("tj_medium", 13, 40), ("pp_scaled", 3, 20)):
    e32 = t._free_run(workload, E, T, seed=5, offset=300, check_envs=list(range(E)), gate_split=False)
    esp = t._free_run(workload, E, T, seed=5, offset=300, check_envs=list(range(E)), gate_split=True)
    print("%-10s %2d envs x %2d steps: worst |error| vs fp64 (log-probs, value, h, c)   fp32 MFMA %.3e   bf16 split x9 %.3e"
          % (workload, E, T, e32, esp), flush=True)
