"""`GymWrapper` for the batched environments — same public surface as the reference adapter
(/root/reference/env_wrappers.py:7-107: observation_dim, num_actions, dim_actions, action_space, reset, step,
reward_terminal, get_stat, display, end_display), different job: the reference flattens per-agent tuples / arrays
into a (1, N, obs_dim) double tensor on every call (`_flatten_obs`, :88-100); here the kernels already write the
(E, N, obs_dim) float32 layout, so the wrapper only derives the three sizes once and forwards calls.
"""
from inspect import signature

import numpy as np
import torch


def _space_sizes(observation_space, action_space):
    """(obs_dim, num_actions, dim_actions) with the reference's rules:
      obs_dim   Box -> prod(shape); Tuple -> sum over members of prod(member.shape) if the ACTION space has a
                `shape` attribute, else 1 per member (sic, env_wrappers.py:21-29, quirk Q17 — harmless because
                every space class carries `shape`, and prod(()) == 1 for Discrete members);
      actions   MultiDiscrete -> (nvec[0], len(nvec)); Discrete -> (n, 1)   (env_wrappers.py:33-50)."""
    members = getattr(observation_space, 'spaces', None)
    if members is None:
        obs_dim = int(np.prod(observation_space.shape))
    else:
        counted = hasattr(action_space, 'shape')
        obs_dim = sum(int(np.prod(m.shape)) if counted else 1 for m in members)
    if hasattr(action_space, 'nvec'):
        return obs_dim, int(action_space.nvec[0]), action_space.shape[0]
    if hasattr(action_space, 'n'):
        return obs_dim, action_space.n, 1
    return obs_dim, None, None


class GymWrapper(object):
    def __init__(self, env):
        self.env = env
        self._takes_epoch = 'epoch' in signature(env.reset).parameters      # env_wrappers.py:57-61
        self._obs_dim, self._num_actions, self._dim_actions = _space_sizes(env.observation_space, env.action_space)

    observation_dim = property(lambda self: self._obs_dim)
    num_actions = property(lambda self: self._num_actions)
    dim_actions = property(lambda self: self._dim_actions)
    action_space = property(lambda self: self.env.action_space)
    nenvs = property(lambda self: self.env.nenvs)

    def reset(self, epoch):
        obs = self.env.reset(epoch) if self._takes_epoch else self.env.reset()
        return self._flatten_obs(obs)

    def step(self, action, observe=True):
        """`action` is the list of per-head (E, N) arrays the policy produced; an env with a single action
        dimension gets head 0 only (the IC3Net talk head is not an env action, env_wrappers.py:76-77)."""
        env_action = action[0] if self._dim_actions == 1 else action
        obs, reward, done, info = self.env.step(env_action) if observe else self.env.step(env_action, observe=False)
        return self._flatten_obs(obs), reward, done, info

    def reward_terminal(self):
        fn = getattr(self.env, 'reward_terminal', None)
        return fn() if fn is not None else torch.zeros(1)

    def _flatten_obs(self, obs):
        return obs.reshape(self.env.nenvs, -1, self._obs_dim)              # already flat: a view, no copy

    def get_stat(self):
        """env.stat of all E environments, summed (merge_stat sums the same keys over episodes in the reference):
        PP 'success' (predator_prey_env.py:284-288, absent in competitive mode); TJ 'success' = 1 - has_failed and
        'add_rate' (traffic_junction_env.py:249-250).  'steps_taken' is dropped like env_wrappers.py:104."""
        s = self.env.device_stats()
        stat = {k: v for k, v in self.env.stat.items() if k != 'steps_taken'}
        # auto-reset mode: the episodes that ended inside step launches + the ones cut at the end of the window
        success = s.success_sum + s.auto_success_sum
        episodes = self.env.nenvs + s.auto_episodes
        if self.env.dims.kind == 1:
            if self.env.mode != 'competitive':
                stat['success'] = success
        else:
            stat['success'] = success
            stat['add_rate'] = s.add_rate * episodes
        if getattr(self.env, 'auto_max_steps', 0):
            stat['_episodes'] = episodes                  # Trainer.run_batch counts these instead of nenvs
        return stat

    def display(self):
        """env_wrappers.py:66-68: render (here: env 0 of the batch as text, from a state readback)."""
        self.env.render()

    def end_display(self):
        pass
