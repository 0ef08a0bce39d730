"""TEST INFRASTRUCTURE ONLY.  ctypes view of oracle/libic3oracle.so (built from ic3_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get('IC3_ORACLE_SO') or os.path.join(_HERE, 'libic3oracle.so')   # (override: the ASan/UBSan build)
_lib = None


def build(force=False):
    src = os.path.join(_HERE, 'ic3_oracle.c')
    if os.environ.get('IC3_ORACLE_SO'):
        return _SO
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(['make', '-C', _HERE, '-s', '-B'])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.orc_x24.restype = C.c_uint32
        _lib.orc_x24.argtypes = [C.c_uint32] * 6
        _lib.orc_sample_one.restype = C.c_int32
    return _lib


class PPCfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ('N', 'nprey', 'dim', 'vision', 'mode', 'naction', 'enemy_comm')]


class TJCfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ('N', 'h', 'w', 'vision', 'vocab', 'outside', 'car_class', 'npath',
                                         'narrival', 'routes_per_arrival', 'scalar')]


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


MODES = {'mixed': 0, 'cooperative': 1, 'competitive': 2}


class PPOracle(object):
    """One Predator-Prey environment (predator_prey_env.py), state as int32 numpy arrays."""

    def __init__(self, N, dim, vision, mode='mixed', nprey=1, stay=True, seed=0, env_gid=0, enemy_comm=False):
        self.cfg = PPCfg(N, nprey, dim, vision, MODES[mode], 5 if stay else 4, int(enemy_comm))
        self.N, self.nprey, self.dim, self.vision = N, nprey, dim, vision
        self.rows = N + (nprey if enemy_comm else 0)       # agents the policy sees (main.py:125-130)
        self.vocab = dim * dim + 4
        self.obs_dim = (2 * vision + 1) ** 2 * self.vocab
        self.seed, self.env_gid = seed, env_gid
        self.episode = -1
        self.loc = np.zeros((N + nprey, 2), np.int32)
        self.reached = np.zeros(N, np.int32)
        self.over = C.c_int32(0)
        self.success = C.c_int32(0)

    def reset(self):
        self.episode += 1
        lib().orc_pp_reset(C.byref(self.cfg), C.c_uint32(self.seed), C.c_uint32(self.env_gid),
                           C.c_uint32(self.episode), _p(self.loc), _p(self.reached), C.byref(self.over))
        self.success = C.c_int32(0)
        return self.obs()

    def set_state(self, loc, reached=None, over=0):
        self.loc[:] = np.asarray(loc, np.int32).reshape(self.loc.shape)
        self.reached[:] = 0 if reached is None else np.asarray(reached, np.int32)
        self.over = C.c_int32(int(over))

    def obs(self):
        out = np.empty((self.rows, self.obs_dim), np.float32)
        lib().orc_pp_obs(C.byref(self.cfg), _p(self.loc), _p(out))
        return out

    def step(self, action):
        """-> obs (rows,obs_dim) f32, reward (rows,) f64, done bool; raises like the reference.
        The reference moves, takes obs, then computes the reward (which freezes) — PP:134-144; obs only reads the
        positions, which the reward pass does not change, so taking it afterwards is equivalent."""
        action = np.ascontiguousarray(np.asarray(action).reshape(-1), np.int32)
        if self.over.value:
            raise RuntimeError("Episode is done")
        if len(action) < self.N:
            raise AssertionError("Action for each agent should be provided.")
        reward = np.empty(self.rows, np.float64)
        rc = lib().orc_pp_step(C.byref(self.cfg), _p(action), _p(self.loc), _p(self.reached), _p(reward),
                               C.byref(self.over), C.byref(self.success))
        if rc == -2:
            raise AssertionError("Actions should be in the range [0,naction).")
        return self.obs(), reward, bool(self.over.value)


class TJOracle(object):
    """One Traffic-Junction environment (traffic_junction_env.py); tables from oracle.tj_tables."""

    def __init__(self, N, dim, vision, difficulty, add_rate_min=0.05, add_rate_max=0.2, curr_start=0,
                 curr_end=0, seed=0, env_gid=0, vocab_type='bool'):
        from . import tj_tables
        self.tab = tab = tj_tables.build(dim, vision, difficulty, vocab_type)
        self.N, self.vision = N, vision
        scalar = vocab_type == 'scalar'
        self.cfg = TJCfg(N, tab['h'], tab['w'], vision, tab['vocab'], tab['outside'], tab['car_class'],
                         tab['npath'], tab['narrival'], tab['routes_per_arrival'], int(scalar))
        self.grid = np.ascontiguousarray(tab['grid'], np.int32)
        self.route_off = np.ascontiguousarray(tab['route_off'], np.int32)
        self.route_rc = np.ascontiguousarray(tab['route_rc'], np.int32)
        self.obs_dim = (4 if scalar else 2) + (2 * vision + 1) ** 2 * tab['vocab']
        self.seed, self.env_gid = seed, env_gid
        self.episode = -1
        self.t = 0
        self.add_rate_min, self.add_rate_max = add_rate_min, add_rate_max
        self.curr_start, self.curr_end = curr_start, curr_end
        self.exact_rate = C.c_double(add_rate_min)      # TJ:103
        self.add_rate = C.c_double(add_rate_min)
        self.epoch_last_update = C.c_double(0)          # TJ:104
        z = lambda: np.zeros(N, np.int32)
        self.alive, self.wait, self.last_act, self.route_loc, self.route_id, self.is_completed = \
            z(), z(), z(), z(), z(), z()
        self.loc = np.zeros((N, 2), np.int32)
        self.cars_in_sys = C.c_int32(0)
        self.has_failed = C.c_int32(0)

    def reset(self, epoch=None):
        self.episode += 1
        self.t = 0
        lib().orc_tj_reset(C.byref(self.cfg), _p(self.alive), _p(self.wait), _p(self.loc), _p(self.last_act),
                           _p(self.route_loc), _p(self.route_id), C.byref(self.cars_in_sys),
                           C.byref(self.has_failed))
        self.is_completed[:] = 0
        lib().orc_tj_curriculum(C.c_double(self.add_rate_min), C.c_double(self.add_rate_max),
                                C.c_double(self.curr_start), C.c_double(self.curr_end),
                                C.c_int(epoch is not None), C.c_double(0 if epoch is None else epoch),
                                C.byref(self.exact_rate), C.byref(self.add_rate),
                                C.byref(self.epoch_last_update))
        return self.obs()

    def obs(self):
        out = np.empty((self.N, self.obs_dim), np.float32)
        lib().orc_tj_obs(C.byref(self.cfg), _p(self.grid), _p(self.alive), _p(self.loc), _p(self.last_act),
                         _p(self.route_id), _p(out))
        return out

    def step(self, action):
        from .philox import rate_threshold
        action = np.ascontiguousarray(np.asarray(action).reshape(-1), np.int32)
        assert len(action) == self.N, "Action for each agent should be provided."
        reward = np.empty(self.N, np.float64)
        rc = lib().orc_tj_step(C.byref(self.cfg), _p(self.grid), _p(self.route_off), _p(self.route_rc),
                               _p(action), C.c_int32(rate_threshold(self.add_rate.value)),
                               C.c_uint32(self.seed), C.c_uint32(self.env_gid), C.c_uint32(self.episode),
                               C.c_uint32(self.t), _p(self.alive), _p(self.wait), _p(self.loc),
                               _p(self.last_act), _p(self.route_loc), _p(self.route_id),
                               C.byref(self.cars_in_sys), C.byref(self.has_failed), _p(self.is_completed),
                               _p(reward))
        if rc == -2:
            raise AssertionError("Actions should be in the range [0,naction).")
        self.t += 1
        # obs is taken after _add_cars and before reward; reward does not mutate what obs reads
        return self.obs(), reward, False


def sample_one(logp, x24):
    logp = np.ascontiguousarray(logp, np.float32)
    return int(lib().orc_sample_one(_p(logp), C.c_int(len(logp)), C.c_uint32(int(x24))))
