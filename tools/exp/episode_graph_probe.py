#!/usr/bin/env python
"""Round-4 experiment: one hipGraph for a WHOLE episode (T step launches as T kernel nodes) against eager launches and
against one graph per step:  python tools/exp/episode_graph_probe.py [workload] [nenvs]"""
import gc
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import bench  # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else 'pp_hard'
E = int(sys.argv[2]) if len(sys.argv) > 2 else 8192


def play(tr, T):
    tr.begin_episode(0)
    for t in range(T):
        tr.step_episode(t)


def timeit(fn, reps):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


# eager
tr, a = bench.build_trainer(workload, E, 0, 0, 0)
T = a.max_steps
gc.collect()
gc.disable()
eager = timeit(lambda: play(tr, T), 4) / T * 1e3
# one graph per step (the Trainer's hip_graph mode)
tr2, a2 = bench.build_trainer(workload, E, 0, 0, 0)
a2.hip_graph = True
play(tr2, T)
tr2._episodes_played = 1
play(tr2, T)
tr2._episodes_played = 2
per_step = timeit(lambda: play(tr2, T), 4) / T * 1e3
# ONE graph for the episode's T steps (reset outside: it reads the episode counter on the host)
tr3, a3 = bench.build_trainer(workload, E, 0, 0, 0)
play(tr3, T)                                   # eager warm-up: caches, static buffers
a3.hip_graph = True                            # (static buffers, no allocation per step)
tr3.begin_episode(0)
tr3._episodes_played = 0                       # _step_body eagerly ...
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, capture_error_mode="thread_local"):
    for t in range(T):
        tr3._step_body(t, observe=True)        # ... but inside ONE capture


def play_graph():
    tr3.begin_episode(0)
    g.replay()
whole = timeit(play_graph, 4) / T * 1e3
print("%s E=%d T=%d: eager %.4f ms/step | one graph per step %.4f | ONE graph per episode %.4f" % (workload, E, T, eager, per_step, whole))
