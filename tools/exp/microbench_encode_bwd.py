"""Time of ic3_env_encode_backward at PP-hard (E = 8192 by default): python tools/exp/microbench_encode_bwd.py [E]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import bench  # noqa: E402

E = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
wl = sys.argv[2] if len(sys.argv) > 2 else 'pp_hard'
tr, a = bench.build_trainer(wl, E, 0, 0, 0)
raw = tr.env.env
R, H = raw.nenvs * raw.nagents_env, a.hid_size
tr.env.reset(0)                                 # a played state: random placement, then a few random steps
nact = 5 if wl.startswith('pp') else 2
for _ in range(6):
    raw.step(torch.randint(0, nact, (raw.nenvs, raw.nagents_env), device='cuda', dtype=torch.int32))
wide = torch.randn(R, 2 * H, device='cuda')
g = wide[:, :H]
for _ in range(5):
    raw.encode_backward(g)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
torch.cuda.synchronize()
ev[0].record()
n = 50
for _ in range(n):
    raw.encode_backward(g)
ev[1].record()
torch.cuda.synchronize()
print("%s E=%d encode_backward: %.1f us per call (%s)" % (wl, E, ev[0].elapsed_time(ev[1]) * 1000 / n,
                                                          os.path.basename(os.environ.get('IC3_ROLLOUT_LIB', 'default'))))
