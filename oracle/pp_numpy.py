"""TEST / BENCH INFRASTRUCTURE ONLY.  A "reference-shaped" Predator-Prey environment in Python + numpy: ONE env per
object, every step re-creates the dense one-hot grid the way the reference does — copy the (dim+2v)^2 x vocab base
grid, add the predator / prey counts, slice one (2v+1)^2 x vocab window per agent, flatten — so that its cost
structure (interpreter overhead + a dense copy of the whole padded grid per step) is the reference's, unlike the C
oracle which assembles rows sparsely.  Used as leg (ii) of bench.py's cpu_baseline and pinned against
the reference's golden trajectories by tests/test_pp_numpy_cpu.py.

Semantics follow /root/reference/ic3net-envs/ic3net_envs/predator_prey_env.py (cited "PP:line") with the injected
Philox stream of oracle/philox.py in place of numpy's global RNG.
"""
import numpy as np

from . import philox


class PPNumpyEnv(object):
    def __init__(self, N, dim, vision, mode='mixed', stay=True, seed=0, env_gid=0):
        self.N, self.dim, self.v, self.mode = N, dim, vision, mode
        self.naction = 5 if stay else 4                                    # PP:90-93
        self.base = dim * dim                                              # PP:97
        self.OUTSIDE, self.PREY, self.PRED = self.base + 1, self.base + 2, self.base + 3   # PP:98-100
        self.vocab = self.base + 4                                         # PP:103
        self.obs_dim = (2 * vision + 1) ** 2 * self.vocab
        self.seed, self.env_gid = seed, env_gid
        self.episode = -1
        # PP:177-186: cell ids, padded with OUTSIDE, as a dense one-hot grid
        ids = np.arange(self.base).reshape(dim, dim)
        pad = np.pad(ids, vision, 'constant', constant_values=self.OUTSIDE)
        self.one_hot = np.zeros(pad.shape + (self.vocab,), dtype=np.int64)
        rr, cc = np.indices(pad.shape)
        self.one_hot[rr, cc, pad] = 1
        self.pad = pad
        self.loc = np.zeros((N + 1, 2), np.int64)                          # N predators, then the prey
        self.reached = np.zeros(N, np.int64)
        self.over = False
        self.success = 0

    # PP:146-175: distinct cells for predators and prey (np.random.choice(..., replace=False) on the injected stream)
    def reset(self):
        self.episode += 1
        cells, d = [], 0
        while len(cells) < self.N + 1:
            k = (philox.x24(self.seed, self.env_gid, philox.DOMAIN_PP_RESET, self.episode, 0, d) * self.base) >> 24
            d += 1
            if k not in cells:
                cells.append(k)
        self.loc[:, 0] = [k // self.dim for k in cells]
        self.loc[:, 1] = [k % self.dim for k in cells]
        self.reached[:] = 0
        self.over = False
        self.success = 0
        return self.obs()

    # PP:188-210 (+ env_wrappers.py:88-100): the dense grid is copied and updated on every call
    def obs(self):
        v = self.v
        grid = self.one_hot.copy()
        for r, c in self.loc[:self.N]:
            grid[r + v, c + v, self.PRED] += 1
        grid[self.loc[self.N, 0] + v, self.loc[self.N, 1] + v, self.PREY] += 1
        rows = []
        for r, c in self.loc[:self.N]:
            rows.append(grid[r:r + 2 * v + 1, c:c + 2 * v + 1].reshape(-1))
        return np.stack(rows).astype(np.float32)

    def _blocked(self, pr, pc):
        return self.pad[pr, pc] == self.OUTSIDE

    # PP:212-252 for one predator (quirks Q1, Q2: 'act == 5' guard, padded indices clamped in unpadded range)
    def _move(self, i, act):
        if self.reached[i] == 1 or act == 5:
            return
        r, c = self.loc[i]
        v, dim = self.v, self.dim
        if act == 0 and not self._blocked(max(0, r + v - 1), c + v):
            self.loc[i, 0] = max(0, r - 1)
        elif act == 1 and not self._blocked(r + v, min(dim - 1, c + v + 1)):
            self.loc[i, 1] = min(dim - 1, c + 1)
        elif act == 2 and not self._blocked(min(dim - 1, r + v + 1), c + v):
            self.loc[i, 0] = min(dim - 1, r + 1)
        elif act == 3 and not self._blocked(r + v, max(0, c + v - 1)):
            self.loc[i, 1] = max(0, c - 1)

    def step(self, action):
        if self.over:
            raise RuntimeError("Episode is done")                          # PP:129-130
        action = np.asarray(action).reshape(-1)
        assert np.all(action <= self.naction), "Actions should be in the range [0,naction)."   # PP:137 (<=, sic)
        for i in range(self.N):
            self._move(i, int(action[i]))
        obs = self.obs()
        # PP:254-290
        on = np.all(self.loc[:self.N] == self.loc[self.N], axis=1)
        n_on = int(on.sum())
        reward = np.full(self.N, -0.05)
        if self.mode == 'cooperative':
            reward[on] = 0.05 * n_on
        elif self.mode == 'competitive':
            reward[on] = 0.05 / n_on if n_on else 0.0
        else:
            reward[on] = 0.0
        self.reached[on] = 1
        if self.mode == 'mixed' and np.all(self.reached == 1):
            self.over = True
        if self.mode != 'competitive':
            self.success = int(n_on == self.N)
        return obs, reward, self.over
