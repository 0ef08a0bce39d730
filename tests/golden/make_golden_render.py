#!/usr/bin/env python
"""Generates tests/golden/render_fixture.json: what the REFERENCE's `render()` draws (predator_prey_env.py:307-336,
traffic_junction_env.py:254-292) for a handful of states, captured by handing the env modules a recording stand-in
for `curses` (the calls `stdscr.addstr(row, x, text, color_pair)` in order).  Run in the build container:

    PYTHONDONTWRITEBYTECODE=1 OMP_NUM_THREADS=1 python tests/golden/make_golden_render.py

Data only (states + the recorded calls); /root/reference is not needed to run the tests.
"""
import json
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.environ.get('IC3_GOLDEN_OUT', HERE)   # where the fixtures are written (tests/test_golden_recipes_cpu.py: a tmp dir)
sys.path.insert(0, HERE)

import numpy as np  # noqa: E402

import ref_harness as rh  # noqa: E402
from oracle import philox  # noqa: E402


class Screen(object):
    def __init__(self):
        self.calls = []

    def clear(self):
        self.calls = []

    def addstr(self, row, x, text, attr=0):
        self.calls.append([int(row), int(x), str(text), int(attr)])

    def refresh(self):
        pass


class FakeCurses(object):
    COLOR_RED, COLOR_YELLOW, COLOR_CYAN, COLOR_GREEN, COLOR_BLUE = 1, 3, 6, 2, 4

    @staticmethod
    def color_pair(n):
        return n


def pp_cases(ref, out):
    ref['pp'].curses = FakeCurses
    for N, dim, layouts in [
        (3, 5, [[(0, 0), (2, 3), (4, 4), (1, 1)],               # all apart
                [(2, 2), (2, 2), (2, 2), (2, 2)],               # everyone on the prey's cell
                [(1, 4), (1, 4), (0, 0), (3, 3)],               # two predators share a cell
                [(0, 1), (4, 0), (3, 2), (3, 2)]]),             # one predator on the prey
        (5, 8, [[(7, 7), (7, 7), (7, 7), (0, 0), (0, 0), (0, 0)],
                [(3, 3), (3, 4), (4, 3), (4, 4), (5, 5), (6, 1)]]),
    ]:
        a = rh.make_args('predator_prey', nagents=N, dim=dim, vision=1, mode='mixed', max_steps=20)
        env = rh.make_env('predator_prey', a)
        raw = env.env
        st = philox.Stream(77, 0)
        ref['rnd'].begin(st, philox.DOMAIN_PP_RESET, 0, 0)
        env.reset(0)
        raw.stdscr = Screen()
        for lay in layouts:
            raw.predator_loc[:] = np.array(lay[:N])
            raw.prey_loc[:] = np.array(lay[N:])
            raw.render()
            out.append(dict(env='pp', N=N, dim=dim, loc_r=[p[0] for p in lay], loc_c=[p[1] for p in lay],
                            cells=[c for c in raw.stdscr.calls]))


def tj_cases(ref, out):
    ref['tj'].curses = FakeCurses
    for difficulty, dim, N, vision, T in [('easy', 6, 5, 1, 14), ('medium', 14, 10, 1, 25), ('hard', 18, 20, 1, 30)]:
        a = rh.make_args('traffic_junction', nagents=N, dim=dim, vision=vision, difficulty=difficulty, max_steps=T,
                         add_rate_min=0.3, add_rate_max=0.3)
        env = rh.make_env('traffic_junction', a)
        raw = env.env
        st = philox.Stream(99, 3)
        ref['rnd'].begin(st, philox.DOMAIN_TJ_ADD, 0, 0)
        env.reset(0)
        raw.stdscr = Screen()
        rs = np.random.RandomState(5)
        for t in range(T):
            ref['rnd'].begin(st, philox.DOMAIN_TJ_ADD, 0, t + 1)
            env.step([rs.randint(0, 2, size=N)])
            if t % 4 == 3 or t == T - 1:
                raw.render()
                out.append(dict(env='tj', difficulty=difficulty, dim=dim, vision=vision, N=N, t=t,
                                grid=np.asarray(raw.grid).astype(int).tolist(), outside=int(raw.OUTSIDE_CLASS),
                                alive=np.asarray(raw.alive_mask).astype(int).tolist(),
                                loc_r=[int(p[0]) for p in raw.car_loc], loc_c=[int(p[1]) for p in raw.car_loc],
                                last_act=np.asarray(raw.car_last_act).astype(int).tolist(),
                                cells=[c for c in raw.stdscr.calls]))


def main():
    ref = rh.load_reference()
    out = []
    pp_cases(ref, out)
    tj_cases(ref, out)
    path = os.path.join(OUT, 'render_fixture.json')
    with open(path, 'w') as f:
        json.dump(out, f, separators=(',', ':'))
    print("wrote %s: %d views, %d draw calls" % (path, len(out), sum(len(o['cells']) for o in out)))


if __name__ == '__main__':
    main()
