"""Space descriptors (the reference uses gym.spaces only as descriptors: predator_prey_env.py:95,107,
traffic_junction_env.py:109,135-148, env_wrappers.py:15-50)."""
import numpy as np


class Box(object):
    def __init__(self, low, high, shape=None, dtype=float):
        self.low, self.high, self.shape, self.dtype = low, high, tuple(shape), dtype


class Discrete(object):
    def __init__(self, n):
        self.n = n
        self.shape = ()

    def sample(self):
        return np.random.randint(self.n)


class MultiDiscrete(object):
    def __init__(self, nvec):
        self.nvec = np.asarray(nvec)
        self.shape = self.nvec.shape


class MultiBinary(object):
    def __init__(self, n):
        self.n = n
        self.shape = tuple(n) if hasattr(n, '__len__') else (n,)


class Tuple(object):
    def __init__(self, spaces):
        self.spaces = tuple(spaces)
