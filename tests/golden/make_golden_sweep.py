#!/usr/bin/env python
"""Checksum sweep: the REFERENCE on many random small configurations (injected Philox stream), stored as per-step
CRC32s of (positions / car state, reward as float32 bytes, flattened obs as float32 bytes).  A checksum of checksums
over the config space: both the oracle (CPU test) and the HIP path (GPU test) must reproduce every CRC.
    PYTHONDONTWRITEBYTECODE=1 OMP_NUM_THREADS=1 python tests/golden/make_golden_sweep.py
"""
import os
import sys
import zlib

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.environ.get('IC3_GOLDEN_OUT', HERE)   # where the fixtures are written (tests/test_golden_recipes_cpu.py: a tmp dir)
sys.path.insert(0, HERE)
import numpy as np

import ref_harness as rh
from oracle import philox

SEED = 4321


def crc(*arrays):
    c = 0
    for a in arrays:
        c = zlib.crc32(np.ascontiguousarray(a).tobytes(), c)
    return c


def main():
    ref = rh.load_reference()
    rs = np.random.RandomState(99)
    pp_cfg, pp_act, pp_crc = [], [], []
    T_PP, T_TJ = 6, 10
    for i in range(120):
        dim = int(rs.randint(2, 13))
        N = int(rs.randint(1, min(12, dim * dim - 1) + 1))
        v = int(rs.randint(0, 3))
        mode = int(rs.randint(3))
        ec, ns = int(rs.rand() < 0.3), int(rs.rand() < 0.2)
        gid = int(rs.randint(1 << 20))
        a = rh.make_args('predator_prey', nagents=N, dim=dim, vision=v, mode=['mixed', 'cooperative', 'competitive'][mode],
                         no_stay=bool(ns), enemy_comm=bool(ec))
        if ec:
            a.nagents += 1
        env = rh.make_env('predator_prey', a)
        raw = env.env
        st = philox.Stream(SEED, gid)
        ref['rnd'].begin(st, philox.DOMAIN_PP_RESET, 0, 0)
        o = env.reset(0)
        R = N + ec
        crcs = [crc(raw.predator_loc.astype(np.int32), raw.prey_loc.astype(np.int32), o[0].numpy().astype(np.float32))]
        acts = np.zeros((T_PP, 13), np.int32)
        for t in range(T_PP):
            act = rs.randint(0, 4 if ns else 5, size=R)
            acts[t, :R] = act
            if raw.episode_over:
                crcs.append(0)
                continue
            o, r, d, _ = env.step([act])
            crcs.append(crc(raw.predator_loc.astype(np.int32), raw.reached_prey.astype(np.int32),
                            np.asarray(r, np.float64).astype(np.float32), o[0].numpy().astype(np.float32),
                            np.int32(int(d))))
        pp_cfg.append([N, dim, v, mode, ec, ns, gid])
        pp_act.append(acts)
        pp_crc.append(crcs)
    tj_dims = {0: [6, 8, 10], 1: [6, 8, 10, 14], 2: [9, 12, 15, 18]}
    tj_cfg, tj_act, tj_crc = [], [], []
    for i in range(90):
        diff = int(rs.randint(3))
        dim = int(tj_dims[diff][rs.randint(len(tj_dims[diff]))])
        v = int(rs.randint(0, 3))
        if diff != 2 and dim < 4 + v:
            v = 0
        N = int(rs.randint(2, 25))          # the reference cannot step a 1-car env (len() of a squeezed scalar, TJ:226-230)
        rate_i = int(rs.randint(4))
        rate = [0.05, 0.3, 0.7, 1.0][rate_i]
        scalar = int(rs.rand() < 0.3)
        gid = int(rs.randint(1 << 20))
        a = rh.make_args('traffic_junction', nagents=N, dim=dim, vision=v, difficulty=['easy', 'medium', 'hard'][diff],
                         add_rate_min=rate, add_rate_max=rate, vocab_type='scalar' if scalar else 'bool')
        env = rh.make_env('traffic_junction', a)
        raw = env.env
        st = philox.Stream(SEED, gid)
        env.reset(0)
        acts = np.zeros((T_TJ, 24), np.int32)
        crcs = []
        for t in range(T_TJ):
            act = (rs.rand(N) < 0.4).astype(np.int64)
            acts[t, :N] = act
            ref['rnd'].begin(st, philox.DOMAIN_TJ_ADD, 0, t)
            o, r, d, info = env.step([act])
            crcs.append(crc(raw.alive_mask.astype(np.int32), raw.wait.astype(np.int32), raw.car_loc.astype(np.int32),
                            raw.car_last_act.astype(np.int32), raw.car_route_loc.astype(np.int32),
                            np.asarray(raw.route_id, np.int32), np.asarray(r, np.float64).astype(np.float32),
                            o[0].numpy().astype(np.float32)))
        tj_cfg.append([N, dim, v, diff, rate_i, scalar, gid])
        tj_act.append(acts)
        tj_crc.append(crcs)
    np.savez_compressed(os.path.join(OUT, 'sweep_checksums.npz'), seed=SEED,
                        pp_cfg=np.array(pp_cfg, np.int32), pp_act=np.array(pp_act, np.int32),
                        pp_crc=np.array(pp_crc, np.uint32), tj_cfg=np.array(tj_cfg, np.int32),
                        tj_act=np.array(tj_act, np.int32), tj_crc=np.array(tj_crc, np.uint32))
    print('sweep: %d PP configs, %d TJ configs' % (len(pp_cfg), len(tj_cfg)))


if __name__ == '__main__':
    main()
