"""GymWrapper mirror (/root/reference/env_wrappers.py:7-107) for the batched envs: the kernels already
write the flattened (E, N, obs_dim) float32 layout, so `_flatten_obs` is the identity."""
from inspect import signature

import numpy as np
import torch


class GymWrapper(object):
    def __init__(self, env):
        self.env = env

    @property
    def observation_dim(self):                 # env_wrappers.py:15-31 (incl. quirk Q17)
        if hasattr(self.env.observation_space, 'spaces'):
            total_obs_dim = 0
            for space in self.env.observation_space.spaces:
                if hasattr(self.env.action_space, 'shape'):
                    total_obs_dim += int(np.prod(space.shape))
                else:
                    total_obs_dim += 1
            return total_obs_dim
        return int(np.prod(self.env.observation_space.shape))

    @property
    def num_actions(self):                     # env_wrappers.py:33-40
        if hasattr(self.env.action_space, 'nvec'):
            return int(self.env.action_space.nvec[0])
        elif hasattr(self.env.action_space, 'n'):
            return self.env.action_space.n

    @property
    def dim_actions(self):                     # env_wrappers.py:42-50
        if hasattr(self.env.action_space, 'nvec'):
            return self.env.action_space.shape[0]
        elif hasattr(self.env.action_space, 'n'):
            return 1

    @property
    def action_space(self):
        return self.env.action_space

    @property
    def nenvs(self):
        return self.env.nenvs

    def reset(self, epoch):                    # env_wrappers.py:56-64
        if 'epoch' in signature(self.env.reset).parameters:
            obs = self.env.reset(epoch)
        else:
            obs = self.env.reset()
        return self._flatten_obs(obs)

    def display(self):
        raise NotImplementedError("rendering is outside the hot-path scope (SURVEY 8(f) f4)")

    def end_display(self):
        pass

    def step(self, action, observe=True):      # env_wrappers.py:73-80
        if self.dim_actions == 1:
            action = action[0]
        obs, r, done, info = self.env.step(action, observe) if not observe else self.env.step(action)
        return (self._flatten_obs(obs), r, done, info)

    def reward_terminal(self):                 # env_wrappers.py:82-86
        if hasattr(self.env, 'reward_terminal'):
            return self.env.reward_terminal()
        return torch.zeros(1)

    def _flatten_obs(self, obs):               # env_wrappers.py:88-100: already (E, N, obs_dim) float32
        return obs.reshape(self.env.nenvs, -1, self.observation_dim)

    def get_stat(self):                        # env_wrappers.py:102-107
        """env.stat summed over the E environments of the handle: PP 'success' (PP:284-288), TJ 'success',
        'add_rate' (TJ:249-250) — the reference sums these per episode in merge_stat."""
        s = self.env.device_stats()
        stat = dict(self.env.stat)
        kind = self.env.dims.kind
        if kind == 1:
            if self.env.mode != 'competitive':
                stat['success'] = s.success_sum
        else:
            stat['success'] = s.success_sum
            stat['add_rate'] = s.add_rate * self.env.nenvs
        stat.pop('steps_taken', None)
        return stat
