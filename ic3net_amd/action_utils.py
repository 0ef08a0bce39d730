"""action_utils.py mirror (/root/reference/action_utils.py:5-63), discrete branch only (PP and TJ are
discrete): `parse_action_args`, `select_action` (multinomial per head, as the `sample_actions` HIP op on
the Philox stream), `translate_action` (actions stay on the device as (E,N) int32 tensors)."""
import torch

from . import ops


def parse_action_args(args):                   # action_utils.py:5-24
    if args.num_actions[0] > 0:
        args.continuous = False
        args.naction_heads = [int(args.num_actions[i]) for i in range(args.dim_actions)]
    else:
        raise NotImplementedError("continuous / --nactions specs are outside the hot-path scope")


class SampleClock(object):
    """Where on the counter-based stream the next draw sits: (seed, env_id_offset, episode, t)."""

    def __init__(self, seed=0, env_id_offset=0, env=None):
        self.seed, self.env_id_offset, self.episode, self.t = seed, env_id_offset, 0, 0
        # when set (a batched env object), draws are positioned by the env handle's own device-side
        # (episode, t) counters — the same stream positions, but hipGraph-capturable
        self.env = env


def select_action(args, action_out, clock=None, out=None):
    """action_utils.py:32-36.  action_out: list of (E,N,A_k) log-probs -> (heads, E, N) int32.
    `out`: optional preallocated (heads, E, N) int32 tensor the sampling kernels write into."""
    if getattr(args, 'continuous', False):
        raise NotImplementedError
    clock = clock or getattr(args, 'sample_clock', None) or SampleClock(getattr(args, 'seed', 0))
    if out is None:
        E, N = action_out[0].shape[:2]
        out = torch.empty((len(action_out), E, N), dtype=torch.int32, device=action_out[0].device)
    for k, lp in enumerate(action_out):
        if clock.env is not None:
            ops.sample_actions_env(clock.env, lp, k, out=out[k])
        else:
            ops.sample_actions(lp, k, clock.seed, clock.env_id_offset, clock.episode, clock.t, out=out[k])
    return out


def translate_action(args, env, action):      # action_utils.py:39-43
    if args.num_actions[0] > 0:
        action = [x for x in action]           # per-head (E,N) int32 device tensors (the reference: numpy (N,))
        return action, action
    raise NotImplementedError
