"""Minimal stand-in for the `gym` package (not installed here, no network) so that the reference's
ic3net_envs / env_wrappers / data modules import in THIS container for golden-vector generation.
The reference uses gym only for registration and space descriptors. Test infrastructure only."""
import importlib
from . import spaces
from .envs import registration


class Env(object):
    def close(self):
        pass


def make(id):
    mod, cls = registration.registry[id].split(':')
    return getattr(importlib.import_module(mod), cls)()
