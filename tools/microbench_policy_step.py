#!/usr/bin/env python
"""Stand-alone timing of the one-launch rollout iteration (ic3_policy_step) against the launch chain it replaces,
without the obs-assembly launch: python tools/microbench_policy_step.py [workload] [nenvs] [steps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

import bench  # noqa: E402


def time_steps(workload, nenvs, steps, mega):
    tr, a = bench.build_trainer(workload, nenvs, 0, 0, 0)
    a.mega_policy = bool(mega)
    a.dense_obs = os.environ.get('IC3_MB_OBS', '0') == '1'   # default obs=NULL: only the policy + sampling + step work
    T = a.max_steps
    tr.begin_episode(0)
    for t in range(8):                        # warm-up (also fills the packed-weight cache)
        tr.step_episode(t)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    t = 8
    for i in range(steps):
        if t == T:
            tr.end_episode()
            tr.begin_episode(0)
            t = 0
        ev[i][0].record()
        tr.step_episode(t)
        ev[i][1].record()
        t += 1
    torch.cuda.synchronize()
    ms = sorted(s.elapsed_time(e) for s, e in ev)
    R, H = nenvs * a.nagents, a.hid_size
    OT = sum(a.naction_heads) + 1
    flops = 2.0 * R * (2 * H * 4 * H + H * H + H * OT)
    med = ms[len(ms) // 2]
    return med, ms[0], flops / (med * 1e-3) / 1e12, getattr(tr.policy_net, 'mega_steps', 0)


if __name__ == '__main__':
    workload = sys.argv[1] if len(sys.argv) > 1 else 'pp_hard'
    nenvs = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 60
    modes = (1,) if os.environ.get('IC3_PS_DEBUG') or (len(sys.argv) > 4 and sys.argv[4] == 'mega') else (1, 0)
    for mega in modes:
        med, best, tf, n = time_steps(workload, nenvs, steps, mega)
        print("[IC3_PS_DEBUG=%s] %s E=%d %s: median %.1f us, best %.1f us per step (policy+draws+env.step, no obs) = %.1f TFLOP/s fp32 "
              "[one-launch calls: %d]" % (os.environ.get('IC3_PS_DEBUG', '0'), workload, nenvs, "ic3_policy_step" if mega else "launch chain", med * 1e3,
                                          best * 1e3, tf, n))
