"""Helpers of the rollout path with the reference's names and behaviour (/root/reference/utils.py:13-29,107-132):
`LogField`, `merge_stat`, `init_args_for_env`."""
import numbers
from collections import namedtuple

import numpy as np
import torch

LogField = namedtuple('LogField', ('data', 'plot', 'x_axis', 'divide_by'))

_ADDABLE = (numbers.Number, np.ndarray, torch.Tensor)


def merge_stat(src, dest):
    """Fold the statistics of one episode / worker into `dest` (same result as utils.py:15-29): values that can be
    added (python / numpy scalars, arrays, tensors) are summed per key, anything else is collected into a list."""
    for key, val in src.items():
        if key not in dest:
            dest[key] = val
            continue
        cur = dest[key]
        if isinstance(val, _ADDABLE):
            dest[key] = cur + val
        elif isinstance(cur, list):
            dest[key] = cur + (val if isinstance(val, list) else [val])
        else:
            dest[key] = [cur, val]


_ENV_TABLE = {'predator_prey': 'PredatorPreyEnv', 'traffic_junction': 'TrafficJunctionEnv'}


def init_args_for_env(parser, argv=None):
    """Let the env named by `--env_name` on the command line register its own flag group before parsing
    (what utils.py:107-132 does through gym's registry).  Unknown / absent env names add nothing."""
    import sys
    from . import envs
    argv = list(sys.argv if argv is None else argv)
    name = None
    for flag, value in zip(argv, argv[1:]):
        if flag == '--env_name':
            name = value                       # the last occurrence wins, as argparse would have it
    cls = getattr(envs, _ENV_TABLE.get(name, ''), None)
    if cls is not None:
        cls().init_args(parser)
