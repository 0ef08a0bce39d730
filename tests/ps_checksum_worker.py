"""Worker of test_policy_step_gpu.py::test_result_changing_environment_knobs_are_gone: CRC of everything a few
ic3_policy_step iterations produce (PP-hard shape, E = 40)."""
import os
import sys
import zlib

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

tr, a = bench.build_trainer('pp_hard', 40, 7, 0, 0)
a.max_steps = 6
ep, _ = tr.get_episode(0)
assert getattr(tr.policy_net, 'mega_steps', 0) == 6
crc = 0
for t in ep:
    for x in [t.action, t.reward, t.value] + list(t.action_out):
        crc = zlib.crc32(x.detach().cpu().numpy().tobytes(), crc)
crc = zlib.crc32(tr.env.env._obs.cpu().numpy().tobytes(), crc)
print("PS_CRC %08x" % crc)
