#!/usr/bin/env python
"""Experiment: the E envs as P independent groups, each stepped by its own launch chain on its own stream
(envs never interact, so group g's step t+1 only waits for group g's step t).  Measures whether the step launches of
different groups overlap usefully (tail of one launch filled by the next, phases of the workgroups spread out)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

import torch  # noqa: E402

import bench  # noqa: E402


def run(workload, nenvs, parts, steps):
    trs = []
    for p in range(parts):
        tr, a = bench.build_trainer(workload, nenvs // parts, 0, p * (nenvs // parts), 0)
        trs.append(tr)
    streams = [torch.cuda.Stream() for _ in range(parts)] if parts > 1 else [torch.cuda.current_stream()]
    T = a.max_steps

    def episode():
        for p, tr in enumerate(trs):
            with torch.cuda.stream(streams[p]):
                tr.begin_episode(0)
        for t in range(T):
            for p, tr in enumerate(trs):
                with torch.cuda.stream(streams[p]):
                    tr.step_episode(t)
        for p, tr in enumerate(trs):
            with torch.cuda.stream(streams[p]):
                tr.end_episode()
    episode()
    episode()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    neps = max(1, steps // T)
    for _ in range(neps):
        episode()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ms = dt / (neps * T) * 1e3
    print("%s E=%d parts=%d: %.4f ms per step of all envs, %.1f M agent-steps/s (upper bound: every slot live)"
          % (workload, nenvs, parts, ms, nenvs * a.nagents / ms / 1e3), flush=True)


if __name__ == '__main__':
    wl = sys.argv[1] if len(sys.argv) > 1 else 'pp_hard'
    E = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
    for parts in (1, 2, 4, 1, 2, 3):
        run(wl, E - E % parts, parts, 320)
