"""Wall time of the phases of Trainer.train_batch (synchronised at the phase borders, so the sum is a little above an
unsynchronised update): rollout (run_batch), losses (bptt.loss_gradients), backward through time, optimizer.
python tools/train_phases.py [nenvs] [workload] [updates]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from ic3net_amd import bptt, trainer as trainer_mod  # noqa: E402

E = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
WL = sys.argv[2] if len(sys.argv) > 2 else 'pp_hard'
U = int(sys.argv[3]) if len(sys.argv) > 3 else 4
tr, a = bench.build_trainer(WL, E, 0, 0, 0)
a.__dict__.update(gamma=1.0, normalize_rewards=False, entr=0, value_coeff=0.01, advantages_per_action=False,
                  batch_size=E * a.max_steps)
acc = {}


def timed(name, fn):
    def wrapper(*args, **kw):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn(*args, **kw)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        h, w = acc.get(name, (0.0, 0.0))
        acc[name] = (h + t1 - t0, w + t2 - t0)
        return r
    return wrapper


tr.run_batch = timed('rollout (run_batch)', tr.run_batch)
bptt.loss_gradients = timed('losses (loss_gradients)', bptt.loss_gradients)
bptt.backward_episode = timed('backward through time', bptt.backward_episode)
tr.optimizer.step = timed('optimizer.step', tr.optimizer.step)
tr.train_batch(0)
tr.train_batch(1)
acc.clear()
torch.cuda.synchronize()
t0 = time.perf_counter()
for u in range(U):
    tr.train_batch(2 + u)
torch.cuda.synchronize()
tot = (time.perf_counter() - t0) / U
print("%s E=%d: %.2f ms per update (synchronised at the phase borders)" % (WL, E, tot * 1e3))
for k, (h, w) in acc.items():
    print("  %-28s host %7.2f ms   until the GPU is done %7.2f ms" % (k, h / U * 1e3, w / U * 1e3))
print("  %-28s %7.2f ms" % ("everything else", (tot - sum(w for _, w in acc.values()) / U) * 1e3))
