"""Soak of the obs rows the one-launch steps write (zero fill + patches / run-based stores) at the bench's launch geometry: before
every step the current state's rows are assembled by the stand-alone obs kernel (ic3_env_observe) and kept, the obs buffer is
overwritten with a marker, the one-launch step writes its rows, and the two must be the same bits.  A rare ordering defect
between a tile's zero stores and its patches (or a tile plan that skips rows) would show here.
    python tools/soak_obs_rows.py [episodes per workload, default 4] [envs, default 8192]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

episodes = int(sys.argv[1]) if len(sys.argv) > 1 else 4
E = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
bad = 0
for wl in ("tj_medium_commnet_mlp", "pp_hard_ic", "pp_hard_iric_tanh", "pp_hard", "tj_hard", "tj_medium"):
    tr, a = bench.build_trainer(wl, E, 1, 0, 0)
    raw = tr.env.env
    steps = 0
    for ep in range(episodes):
        tr.begin_episode(ep)
        for t in range(a.max_steps):
            ref = raw.observe().clone()
            raw._obs.fill_(-7.0)
            tr.step_episode(t)
            if not torch.equal(raw._obs, ref):
                bad += 1
                diff = (raw._obs != ref).nonzero()
                print("MISMATCH", wl, "episode", ep, "step", t, "entries", diff.shape[0], "first", diff[0].tolist(), flush=True)
            steps += 1
        tr.end_episode()
    one = getattr(tr.policy_net, 'commnet_steps', 0) + getattr(tr.policy_net, 'mega_steps', 0)
    print("%-24s E=%d: %d steps, one-launch steps %d, obs tensor %.2f GB: %s" %
          (wl, E, steps, one, raw._obs.numel() * 4 / 1e9, "rows equal bit for bit" if not bad else "MISMATCHES"), flush=True)
    assert one == steps, "the one-launch path did not run"
    del tr, raw
    torch.cuda.empty_cache()
sys.exit(1 if bad else 0)
