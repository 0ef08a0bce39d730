#!/usr/bin/env python
"""Throughput of the ACTUAL reference (/root/reference, fp64, CPU) on this container's cores, for the record in
DESIGN.md / BASELINE.md: Trainer.run_batch (rollout only) on one process.  Not used by tests or bench.py (the
reference does not exist on the GPU box).   PYTHONDONTWRITEBYTECODE=1 OMP_NUM_THREADS=1 python tests/golden/bench_reference.py
"""
import os
import sys
import time

sys.dont_write_bytecode = True
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch

import ref_harness as rh

CONFIGS = {
    'pp_hard': ('predator_prey', dict(nagents=10, dim=20, vision=1, max_steps=80, hid_size=128, ic3net=True,
                                      recurrent=True, detach_gap=10)),
    'tj_medium': ('traffic_junction', dict(nagents=10, dim=14, vision=1, max_steps=40, hid_size=128, commnet=True,
                                           recurrent=True, detach_gap=10, difficulty='medium', add_rate_min=0.05,
                                           add_rate_max=0.05)),
    'tj_hard': ('traffic_junction', dict(nagents=20, dim=18, vision=1, max_steps=80, hid_size=128, ic3net=True,
                                         recurrent=True, detach_gap=10, difficulty='hard', add_rate_min=0.05,
                                         add_rate_max=0.05)),
}


def main():
    ref = rh.load_reference()
    torch.set_default_dtype(torch.float64)
    torch.set_num_threads(1)
    import ic3net_envs.predator_prey_env as pp
    import ic3net_envs.traffic_junction_env as tj
    pp.np = np                      # plain numpy RNG: this is a speed probe, not a parity run
    tj.np = np
    for name in sys.argv[1:] or ['pp_hard', 'tj_medium', 'tj_hard']:
        env_name, flags = CONFIGS[name]
        a = rh.make_args(env_name, **flags)
        env = rh.make_env(env_name, a)
        rh.finish_args(a, env)
        torch.manual_seed(0)
        net = ref['comm'].CommNetMLP(a, a.num_inputs)
        tr = ref['trainer'].Trainer(a, net, env)
        tr.run_batch(0)
        t0, steps = time.perf_counter(), 0
        while time.perf_counter() - t0 < 8.0:
            _, st = tr.run_batch(0)
            steps += st['num_steps']
        dt = time.perf_counter() - t0
        print("%-10s reference rollout, 1 process: %7.1f env-steps/s = %8.1f agent-steps/s" %
              (name, steps / dt, a.nagents * steps / dt))


if __name__ == '__main__':
    main()
