"""Checkpoint / log compatibility, both ways: the reference saves
`{'policy_net': state_dict, 'log': {name: LogField(data, plot, x_axis, divide_by)}, 'trainer': optimizer.state_dict()}`
with torch.save (/root/reference/main.py:260-265) and restores it with `load` (:267-272).

  * load(): accepts files written by the reference itself — its `LogField` namedtuple pickles as `utils.LogField`
    (a module that does not exist in this package) and its tensors are float64.
  * save(): writes files the reference's `load` can consume: its `log.update(d['log'])` followed by
    `v.data.append(...)` / `v.divide_by` (main.py:219-225,267-272) needs real `utils.LogField` namedtuples, so the log
    entries are pickled under the global name `utils.LogField` (resolved by the reference's own utils.py at load time),
    and every tensor is moved to the CPU (the reference is a CPU program; `torch.load(path)` without map_location
    cannot open device tensors on a box without that GPU).  dtype stays float32: `load_state_dict` /
    `Optimizer.load_state_dict` copy into the reference's float64 parameters.
"""
import pickle
import sys
import types
from collections import namedtuple
from contextlib import contextmanager

import torch

from .utils import LogField

_FIELDS = ('data', 'plot', 'x_axis', 'divide_by')


@contextmanager
def _reference_logfield():
    """Yields a namedtuple class that pickles as the global `utils.LogField`.  pickle only writes a global it can
    re-import, so while saving a stand-in module named `utils` holds the class — unless a real `utils` module with
    a LogField (the reference's, when this code runs inside its tree) is already imported; then that one is used."""
    real = sys.modules.get('utils')
    if real is not None and hasattr(real, 'LogField') and tuple(getattr(real.LogField, '_fields', ())) == _FIELDS:
        yield real.LogField
        return
    cls = namedtuple('LogField', _FIELDS)
    cls.__module__ = 'utils'
    shim = types.ModuleType('utils')
    shim.LogField = cls
    sys.modules['utils'] = shim
    try:
        yield cls
    finally:
        if real is not None:
            sys.modules['utils'] = real
        else:
            del sys.modules['utils']


def _to_cpu(obj):
    if torch.is_tensor(obj):
        return obj.detach().cpu()
    if isinstance(obj, dict):
        return type(obj)((k, _to_cpu(v)) for k, v in obj.items())
    if isinstance(obj, (list, tuple)) and not hasattr(obj, '_fields'):
        return type(obj)(_to_cpu(v) for v in obj)
    return obj


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
    d['policy_net'] = _to_cpu(policy_net.state_dict())
    d['trainer'] = _to_cpu(trainer.state_dict())
    with _reference_logfield() as ref_field:
        d['log'] = {k: ref_field(*_to_cpu(list(v))) for k, v in log.items()}
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


def read(path, map_location=None):
    """The raw {'policy_net', 'log', 'trainer'} dict of a checkpoint written here or by the reference (its `utils.LogField`
    entries come back as this package's LogField; plain torch.load needs an importable `utils` module for them)."""
    return torch.load(path, map_location=map_location, weights_only=False, pickle_module=_RefPickle)


def load(path, policy_net, log, trainer, map_location=None):   # main.py:267-272
    d = read(path, map_location)
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
