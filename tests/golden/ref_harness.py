"""Harness that imports the *reference* (/root/reference, read-only, THIS container only) so that
golden vectors can be generated from it.  Never imported by tests / product: only by make_golden.py.

Accommodations (all applied from outside, reference files untouched):
  1. stub `gym` package (tests/golden/refshim)
  2. numpy>=2: np.ogrid returns a tuple -> patch `_all_idx`
  3. torch>=2: in-place mul on an expanded view in comm.py:175 -> `get_agent_mask` returns a clone
  4. RNG injection: the env modules' global `np` is rebound to a proxy whose `.random` draws from the
     build's counter-based Philox stream (oracle/philox.py)
  5. inspect.getargspec was removed in Python 3.11 (present in 3.10; aliased defensively)
"""
import os
import sys
import argparse
import inspect

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'

sys.dont_write_bytecode = True
for p in (os.path.join(HERE, 'refshim'), os.path.join(REF, 'ic3net-envs'), REF, REPO):
    if p not in sys.path:
        sys.path.insert(0, p)
if not hasattr(inspect, 'getargspec'):
    inspect.getargspec = inspect.getfullargspec

import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import philox  # noqa: E402


def _all_idx(self, idx, axis):
    grid = list(np.ogrid[tuple(map(slice, idx.shape))])
    grid.insert(axis, idx)
    return tuple(grid)


class RandomShim(object):
    """Replaces np.random inside the reference env modules; semantics = oracle/philox.py contract."""

    def __init__(self):
        self.stream = None
        self.r = -1          # current TJ arrival point
        self.d = 0           # sequential draw counter (PP reset)

    def begin(self, stream, domain, episode, t):
        self.stream = stream
        stream.at(domain, episode, t)
        self.r = -1
        self.d = 0

    # traffic_junction_env.py:375  `np.random.uniform() <= self.add_rate`
    def uniform(self):
        self.r += 1
        return self.stream.draw(3 * self.r) / 16777216.0

    def choice(self, a, size=None, replace=True):
        if np.ndim(a) == 0:
            n = int(a)
            if size is None:                       # traffic_junction_env.py:383 route pick
                return (self.stream.draw(3 * self.r + 2) * n) >> 24
            assert not replace                     # predator_prey_env.py:174
            out = []
            while len(out) < size:
                k = (self.stream.draw(self.d) * n) >> 24
                self.d += 1
                if k not in out:
                    out.append(k)
            return np.array(out)
        a = np.asarray(a)                          # traffic_junction_env.py:618 k-th dead slot
        return a[(self.stream.draw(3 * self.r + 1) * len(a)) >> 24]


class NpProxy(object):
    def __init__(self, rnd):
        self.random = rnd

    def __getattr__(self, k):
        return getattr(np, k)


_state = {}


def load_reference():
    """Import the reference modules with the compat patches; returns a dict of modules."""
    if _state:
        return _state
    import gym  # noqa: F401  (stub)
    import ic3net_envs  # noqa: F401  registers ids
    import ic3net_envs.predator_prey_env as pp
    import ic3net_envs.traffic_junction_env as tj
    import ic3net_envs.traffic_helper as th
    pp.PredatorPreyEnv._all_idx = _all_idx
    tj.TrafficJunctionEnv._all_idx = _all_idx
    rnd = RandomShim()
    pp.np = NpProxy(rnd)
    tj.np = NpProxy(rnd)
    import comm
    _orig = comm.CommNetMLP.get_agent_mask

    def get_agent_mask(self, batch_size, info):
        n, m = _orig(self, batch_size, info)
        return n, m.clone()
    comm.CommNetMLP.get_agent_mask = get_agent_mask
    import env_wrappers
    import data
    import action_utils
    import trainer
    import utils
    import models
    _state.update(models=models, pp=pp, tj=tj, th=th, comm=comm, env_wrappers=env_wrappers, data=data,
                  action_utils=action_utils, trainer=trainer, utils=utils, rnd=rnd)
    return _state


def make_args(env_name, **kw):
    """Replicates main.py:22-155 argument handling (main.py itself needs visdom and trains at import)."""
    ref = load_reference()
    a = argparse.Namespace(
        num_epochs=100, epoch_size=10, batch_size=500, nprocesses=1, hid_size=64, recurrent=False,
        gamma=1.0, tau=1.0, seed=0, normalize_rewards=False, lrate=0.001, entr=0, value_coeff=0.01,
        env_name=env_name, max_steps=20, nactions='1', action_scale=1.0, plot=False, plot_env='main',
        save='', save_every=0, load='', display=False, random=False, commnet=False, ic3net=False,
        nagents=1, comm_mode='avg', comm_passes=1, comm_mask_zero=False, mean_ratio=1.0,
        rnn_type='MLP', detach_gap=10000, comm_init='uniform', hard_attn=False, comm_action_one=False,
        advantages_per_action=False, share_weights=False)
    if env_name == 'predator_prey':       # predator_prey_env.py:55-70
        a.__dict__.update(nenemies=1, dim=5, vision=2, moving_prey=False, no_stay=False, mode='mixed',
                          enemy_comm=False)
    else:                                 # traffic_junction_env.py:60-77
        a.__dict__.update(dim=5, vision=1, add_rate_min=0.05, add_rate_max=0.2, curr_start=0, curr_end=0,
                          difficulty='easy', vocab_type='bool')
    a.__dict__.update(kw)
    if a.ic3net:                          # main.py:115-123
        a.commnet = 1
        a.hard_attn = 1
        a.mean_ratio = 0
        if a.env_name == 'traffic_junction':
            a.comm_action_one = True
    a.nfriendly = a.nagents               # main.py:125
    return a


def finish_args(a, env):
    """main.py:134-155 (derived flags) given the wrapped env."""
    ref = load_reference()
    a.num_actions = env.num_actions
    if not isinstance(a.num_actions, (list, tuple)):
        a.num_actions = [a.num_actions]
    a.dim_actions = env.dim_actions
    a.num_inputs = env.observation_dim
    if a.hard_attn and a.commnet:
        a.num_actions = [*a.num_actions, 2]
        a.dim_actions = env.dim_actions + 1
    if a.commnet and (a.recurrent or a.rnn_type == 'LSTM'):
        a.recurrent = True
        a.rnn_type = 'LSTM'
    ref['action_utils'].parse_action_args(a)
    return a


def make_env(env_name, args):
    ref = load_reference()
    return ref['data'].init(env_name, args, False)
