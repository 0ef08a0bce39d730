"""Checkpoint / log compatibility: the reference saves
`{'policy_net': state_dict, 'log': {name: LogField(data, plot, x_axis, divide_by)}, 'trainer': optimizer.state_dict()}`
with torch.save (/root/reference/main.py:260-265) and restores it with `load` (:267-272).  Same layout here; the
loader also accepts files written by the reference itself: its `LogField` namedtuple pickles as `utils.LogField`
(a module that does not exist in this package) and its tensors are float64.
"""
import pickle

import torch

from .utils import LogField


def new_log():
    """The 11 fields of main.py:190-201."""
    log = dict()
    log['epoch'] = LogField(list(), False, None, None)
    for k in ('reward', 'enemy_reward', 'success', 'steps_taken', 'add_rate'):
        log[k] = LogField(list(), True, 'epoch', 'num_episodes')
    for k in ('comm_action', 'enemy_comm', 'value_loss', 'action_loss', 'entropy'):
        log[k] = LogField(list(), True, 'epoch', 'num_steps')
    return log


def save(path, policy_net, log, trainer):          # main.py:260-265
    d = dict()
    d['policy_net'] = policy_net.state_dict()
    d['log'] = {k: tuple(v) for k, v in log.items()}     # plain tuples: loadable without this package
    d['trainer'] = trainer.state_dict()
    torch.save(d, path)


class _RefUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if name == 'LogField' and module in ('utils', 'ic3net_amd.utils'):
            return LogField
        return super().find_class(module, name)


class _RefPickle(object):
    """pickle_module for torch.load that resolves the reference's `utils.LogField`."""
    Unpickler = _RefUnpickler
    load = staticmethod(pickle.load)
    loads = staticmethod(pickle.loads)
    __name__ = 'pickle'


def load(path, policy_net, log, trainer, map_location=None):   # main.py:267-272
    d = torch.load(path, map_location=map_location, weights_only=False, pickle_module=_RefPickle)
    own = policy_net.state_dict()
    sd = {k: v.to(dtype=own[k].dtype) if torch.is_tensor(v) and k in own else v for k, v in d['policy_net'].items()}
    policy_net.load_state_dict(sd)                               # strict: same keys as the reference
    log.update({k: LogField(*v) for k, v in d['log'].items()})
    if d.get('trainer'):
        trainer.load_state_dict(_cast_optimizer_state(d['trainer'], next(policy_net.parameters())))
    return d


def _cast_optimizer_state(state, like):
    """RMSprop state (square_avg, ...) saved in float64 by the reference -> the policy's dtype/device."""
    out = {'param_groups': state['param_groups'], 'state': {}}
    for pid, st in state.get('state', {}).items():
        out['state'][pid] = {k: (v.to(dtype=like.dtype, device=like.device) if torch.is_tensor(v) and v.is_floating_point()
                                 and v.dim() > 0 else v) for k, v in st.items()}
    return out
