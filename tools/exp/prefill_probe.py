#!/usr/bin/env python
"""Round-4 experiment: what the policy launch costs in its obs modes, with and without a fill launch beside it.
   python tools/exp/prefill_probe.py MODE [workload] [gate_split]
MODE: noobs | fused | patch_longfill (patch_nofill + ONE slow fill launch of a third buffer beside each episode) | patch_nofill (rows marked prefilled by ic3_obs_set_prefilled, NO fill launch at all) |
      patch_fill (ic3_obs_prefill of the other buffer on a second stream beside every launch; IC3_FILL_* pace it)
Prints the median / min launch time of ic3_policy_step (dispatch-stamped events) and the wall time per step."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import bench  # noqa: E402
from ic3net_amd.envs import DispatchEvent  # noqa: E402

mode = sys.argv[1]
workload = sys.argv[2] if len(sys.argv) > 2 else 'pp_hard'
gs = int(sys.argv[3]) if len(sys.argv) > 3 else 1
tr, a = bench.build_trainer(workload, 8192, 0, 0, 0)
a.gate_split = bool(gs)
a.prefill_obs = False
a.dense_obs = mode != 'noobs'
raw = tr.env.env
pair = raw.obs_pair()
side = torch.cuda.Stream()
main = torch.cuda.current_stream()
T = a.max_steps
tr.begin_episode(0)
for t in range(T):                       # warm-up episode on the ordinary path
    tr.step_episode(t)
tr.end_episode()
times, fills = [], []
wall0 = None
for ep in range(3):
    tr.begin_episode(0)
    if ep == 1:
        torch.cuda.synchronize()
        wall0 = time.perf_counter()
    if mode == 'patch_longfill':              # ONE slow fill launch (IC3_FILL_NAP large) running beside the whole episode
        side.wait_stream(main)
        third = torch.empty_like(pair[0]) if ep == 0 else third
        raw.prefill(third, side)
    for t in range(T):
        e0, e1 = DispatchEvent(), DispatchEvent()
        raw.set_step_events(e0, e1)
        if mode in ('patch_nofill', 'patch_fill', 'patch_longfill'):
            raw._obs = pair[t & 1]
            tr._state = raw._obs
            if mode in ('patch_nofill', 'patch_longfill'):
                raw.mark_prefilled(raw._obs)
            else:
                side.wait_stream(main)
                f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                f0.record(side)
                raw.prefill(pair[(t + 1) & 1], side)
                f1.record(side)
                fills.append((f0, f1))
        tr.step_episode(t)
        if mode == 'patch_fill':
            main.wait_stream(side)
        if ep >= 1:
            times.append((e0, e1))
    tr.end_episode()
torch.cuda.synchronize()
wall = (time.perf_counter() - wall0) / (2 * T) * 1e3
ms = sorted(s.elapsed_time(e) for s, e in times)
fm = sorted(s.elapsed_time(e) for s, e in fills[T:]) if fills else [0.0]
print("%-13s %-9s gs=%d env[%s]: policy launch median %.4f min %.4f ms | fill median %.4f | wall %.4f ms/step" % (
    mode, workload, gs, " ".join("%s=%s" % (k, v) for k, v in sorted(os.environ.items()) if k.startswith("IC3_FILL")),
    ms[len(ms) // 2], ms[0], fm[len(fm) // 2], wall))
