"""utils.py mirror: the helpers on the rollout path (/root/reference/utils.py:13-29)."""
import numbers
from collections import namedtuple

import numpy as np
import torch

LogField = namedtuple('LogField', ('data', 'plot', 'x_axis', 'divide_by'))


def merge_stat(src, dest):                     # utils.py:15-29
    for k, v in src.items():
        if k not in dest:
            dest[k] = v
        elif isinstance(v, numbers.Number):
            dest[k] = dest.get(k, 0) + v
        elif isinstance(v, np.ndarray) or torch.is_tensor(v):
            dest[k] = dest.get(k, 0) + v
        else:
            if isinstance(dest[k], list) and isinstance(v, list):
                dest[k].extend(v)
            elif isinstance(dest[k], list):
                dest[k].append(v)
            else:
                dest[k] = [dest[k], v]


def init_args_for_env(parser, argv=None):      # utils.py:107-132
    import sys
    from . import envs
    table = {'predator_prey': envs.PredatorPreyEnv, 'traffic_junction': envs.TrafficJunctionEnv}
    argv = sys.argv if argv is None else argv
    env_name = None
    for index, item in enumerate(argv):
        if item == '--env_name':
            env_name = argv[index + 1]
    if not env_name or env_name not in table:
        return
    table[env_name]().init_args(parser)
