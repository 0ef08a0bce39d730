#!/usr/bin/env python
"""Host-side cost of an episode boundary (end_episode + begin_episode + the first step's enqueue), with cProfile."""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

import torch  # noqa: E402

import bench  # noqa: E402

tr, a = bench.build_trainer('pp_hard', 8192, 0, 0, 0)
T = a.max_steps
for ep in range(3):
    tr.get_episode(ep)
torch.cuda.synchronize()
acc = [0.0, 0.0, 0.0, 0.0]
pr = cProfile.Profile()
n = 20
for ep in range(n):
    tr.begin_episode(0)
    for t in range(T):
        tr.step_episode(t)
    t0 = time.perf_counter()
    pr.enable()
    tr.end_episode()
    pr.disable()
    t1 = time.perf_counter()
    pr.enable()
    tr.begin_episode(0)
    pr.disable()
    t2 = time.perf_counter()
    tr.step_episode(0)
    t3 = time.perf_counter()
    for t in range(1, T):
        tr.step_episode(t)
    t4 = time.perf_counter()
    tr.end_episode()
    acc[0] += t1 - t0
    acc[1] += t2 - t1
    acc[2] += t3 - t2
    acc[3] += (t4 - t3) / (T - 1)
print("end_episode %.1f us (includes waiting for the GPU), begin_episode %.1f us, first step enqueue %.1f us, "
      "later steps enqueue %.1f us each" % tuple(1e6 * x / n for x in acc))
pstats.Stats(pr).sort_stats('tottime').print_stats(22)
