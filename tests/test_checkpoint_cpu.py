"""CPU: checkpoint / log layout compatibility with /root/reference/main.py:260-272."""
import sys
import types
from collections import namedtuple

import numpy as np
import torch

from policy_util import PolicyCase


def _policy(pc):
    from ic3net_amd.comm import CommNetMLP
    return CommNetMLP(pc.args(), pc.obs_dim)


class _Opt(object):
    def __init__(self, net):
        self.optimizer = torch.optim.RMSprop(net.parameters(), lr=0.001, alpha=0.97, eps=1e-6)

    def state_dict(self):
        return self.optimizer.state_dict()

    def load_state_dict(self, s):
        self.optimizer.load_state_dict(s)


def test_roundtrip(tmp_path):
    from ic3net_amd import checkpoint
    pc = PolicyCase("policy_ic3net_small")
    net, tr, log = _policy(pc), None, checkpoint.new_log()
    tr = _Opt(net)
    assert list(log) == ['epoch', 'reward', 'enemy_reward', 'success', 'steps_taken', 'add_rate', 'comm_action',
                         'enemy_comm', 'value_loss', 'action_loss', 'entropy']
    log['epoch'].data.extend([1, 2])
    log['reward'].data.extend([np.array([-1.0, -2.0]), np.array([-0.5, -1.0])])
    checkpoint.save(str(tmp_path / "ck.pt"), net, log, tr)
    net2, log2 = _policy(pc), checkpoint.new_log()
    tr2 = _Opt(net2)
    d = checkpoint.load(str(tmp_path / "ck.pt"), net2, log2, tr2)
    assert set(d) == {'policy_net', 'log', 'trainer'}
    for (k, a), (_, b) in zip(net.state_dict().items(), net2.state_dict().items()):
        assert torch.equal(a, b), k
    assert log2['epoch'].data == [1, 2] and log2['reward'].divide_by == 'num_episodes'


def test_loads_a_reference_written_checkpoint(tmp_path):
    """A file exactly as the reference writes it: float64 state_dict with the reference's key set (from the
    policy fixtures), `log` holding `utils.LogField` namedtuples, RMSprop state in float64."""
    from ic3net_amd import checkpoint
    pc = PolicyCase("policy_ic3net_small")
    ref_utils = types.ModuleType('utils')                      # stands in for the reference's utils module
    ref_utils.LogField = namedtuple('LogField', ('data', 'plot', 'x_axis', 'divide_by'))
    ref_utils.LogField.__module__ = 'utils'
    sys.modules['utils'] = ref_utils
    try:
        sd = {k: torch.from_numpy(np.asarray(v)).double() for k, v in pc.params.items()}
        ref_net = _policy(pc).double()
        ref_net.load_state_dict(sd)
        opt = torch.optim.RMSprop(ref_net.parameters(), lr=0.001, alpha=0.97, eps=1e-6)
        (ref_net.encoder.weight.sum() + ref_net.value_head.weight.sum()).backward()
        opt.step()
        log = {'epoch': ref_utils.LogField([1], False, None, None),
               'reward': ref_utils.LogField([np.array([-1.5])], True, 'epoch', 'num_episodes')}
        torch.save({'policy_net': ref_net.state_dict(), 'log': log, 'trainer': opt.state_dict()},
                   str(tmp_path / "ref.pt"))
    finally:
        del sys.modules['utils']
    net, mylog = _policy(pc), checkpoint.new_log()
    tr = _Opt(net)
    checkpoint.load(str(tmp_path / "ref.pt"), net, mylog, tr)
    assert net.encoder.weight.dtype == torch.float32
    np.testing.assert_allclose(net.encoder.weight.detach().numpy(), ref_net.encoder.weight.detach().numpy(), atol=1e-6)
    assert mylog['epoch'].data == [1] and type(mylog['reward']).__module__ == 'ic3net_amd.utils'
    st = tr.optimizer.state_dict()['state']
    assert st and all(v['square_avg'].dtype == torch.float32 for v in st.values())


# ----------------------------------------------------------------------------------------------------------------
# Files written by the reference itself (tests/golden/make_golden_ckpt.py ran /root/reference/main.py with --save)
# ----------------------------------------------------------------------------------------------------------------
import argparse
import json
import os
import subprocess

import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
REF = '/root/reference'


def _pp_easy_policy():
    from ic3net_amd.comm import CommNetMLP
    a = argparse.Namespace(nagents=3, hid_size=16, comm_passes=1, recurrent=True, continuous=False,
                           naction_heads=[5, 2], comm_mask_zero=False, share_weights=False, comm_init='uniform',
                           hard_attn=True, comm_mode='avg', rnn_type='LSTM', init_std=0.2)
    return CommNetMLP(a, 29)                      # dim 5, vision 0: obs_dim = 5*5 + 4


def test_checkpoint_written_by_the_reference_loads_and_continues():
    from ic3net_amd import checkpoint
    from ic3net_amd.main import normalise_epoch
    net, log = _pp_easy_policy(), checkpoint.new_log()
    tr = _Opt(net)
    d = checkpoint.load(os.path.join(GOLDEN, "ref_ckpt_pp_easy.pt"), net, log, tr)
    for k, v in d['policy_net'].items():
        assert v.dtype == torch.float64                                    # the reference trains in fp64
        np.testing.assert_allclose(net.state_dict()[k].numpy(), v.numpy(), rtol=0, atol=1e-6)
    assert log['epoch'].data == [1, 2, 3] and len(log['reward'].data) == 3
    st = tr.optimizer.state_dict()['state']
    assert len(st) > 0 and all(v['square_avg'].dtype == torch.float32 for v in st.values())
    # continue: one more epoch is appended with the reference's normalisation (main.py:219-225)
    stat = {'num_episodes': 4, 'num_steps': 80, 'reward': np.array([-4.0, -2.0, -1.0]), 'success': 2,
            'steps_taken': 80, 'comm_action': np.array([40.0, 20.0, 10.0]), 'value_loss': 8.0, 'action_loss': -4.0,
            'entropy': 16.0}
    assert normalise_epoch(stat, log) == 4
    assert log['epoch'].data == [1, 2, 3, 4]
    np.testing.assert_allclose(log['reward'].data[-1], [-1.0, -0.5, -0.25])
    np.testing.assert_allclose(log['comm_action'].data[-1], [0.5, 0.25, 0.125])
    assert log['success'].data[-1] == 0.5 and log['enemy_reward'].data[-1] == 0
    # and one optimizer step on the restored RMSprop state works
    (net.encoder.weight.sum() + net.value_head.weight.sum()).backward()
    tr.optimizer.step()


def test_stdout_lines_equal_the_reference_print_out():
    """format_epoch() fed with the (normalised) stats the reference logged must reproduce, byte for byte, what the
    reference printed for those epochs (main.py:229-244) — except the wall-clock field."""
    from ic3net_amd import checkpoint
    from ic3net_amd.main import format_epoch
    import re
    net, log = _pp_easy_policy(), checkpoint.new_log()
    checkpoint.load(os.path.join(GOLDEN, "ref_ckpt_pp_easy.pt"), net, log, _Opt(net))
    want = open(os.path.join(GOLDEN, "ref_stdout_pp_easy.txt")).read().splitlines()
    got = []
    for i, epoch in enumerate(log['epoch'].data):
        stat = {k: log[k].data[i] for k in ('reward', 'success', 'steps_taken', 'comm_action')}
        got += format_epoch(epoch, stat, 0.0)
    strip = lambda ls: [re.sub(r'Time [0-9.]+s', 'Time Xs', l) for l in ls]
    assert strip(got) == strip(want)


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree only exists in the build container")
def test_plot_script_extracts_the_same_numbers_from_our_lines(tmp_path):
    """/root/reference/plot_script.py:15-57 `read_file`, run on lines printed by format_epoch, returns what it
    returned on the reference's own stdout (tests/golden/ref_plot_expect.json)."""
    import ast
    from ic3net_amd import checkpoint
    from ic3net_amd.main import format_epoch
    net, log = _pp_easy_policy(), checkpoint.new_log()
    checkpoint.load(os.path.join(GOLDEN, "ref_ckpt_pp_easy.pt"), net, log, _Opt(net))
    path = tmp_path / "ours.log"
    with open(path, 'w') as f:
        for i, epoch in enumerate(log['epoch'].data):
            stat = {k: log[k].data[i] for k in ('reward', 'success', 'steps_taken', 'comm_action')}
            f.write("\n".join(format_epoch(epoch, stat, 1.25)) + "\n")
    src = open(os.path.join(REF, 'plot_script.py')).read()
    fn = [n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == 'read_file'][0]
    ns = {'np': np, 'print': lambda *a, **k: None}
    exec(compile(ast.Module([fn], []), 'plot_script.read_file', 'exec'), ns)
    expect = json.load(open(os.path.join(GOLDEN, "ref_plot_expect.json")))
    for term, scalar in (('Epoch', False), ('Success', True), ('Steps-taken', True)):
        got = ns['read_file']([], str(path), scalar, term)
        np.testing.assert_allclose(np.array(got, float), np.array(expect[term], float), rtol=0, atol=1e-12)


_REF_LOADER = r'''
import sys, os
sys.path.insert(0, os.path.join(sys.argv[2], 'tests', 'golden'))
import ref_harness as rh
import torch
ref = rh.load_reference()
torch.set_default_dtype(torch.float64)
a = rh.make_args('predator_prey', nagents=3, hid_size=16, dim=5, vision=0, max_steps=20, ic3net=True, recurrent=True,
                 detach_gap=10)
ref['pp'].np = __import__('numpy'); ref['tj'].np = ref['pp'].np
env = rh.make_env('predator_prey', a)
rh.finish_args(a, env)
policy_net = ref['comm'].CommNetMLP(a, a.num_inputs)
trainer = ref['trainer'].Trainer(a, policy_net, env)
LogField = ref['utils'].LogField
log = {'epoch': LogField(list(), False, None, None), 'reward': LogField(list(), True, 'epoch', 'num_episodes')}
# main.py:267-272 `load`, verbatim semantics (torch>=2.6 needs weights_only=False for ANY pickled namedtuple)
d = torch.load(sys.argv[1], weights_only=False)
policy_net.load_state_dict(d['policy_net'])
log.update(d['log'])
trainer.load_state_dict(d['trainer'])
# main.py:219-225 on the restored log
epoch = len(log['epoch'].data) + 1
for k, v in log.items():
    if k == 'epoch':
        v.data.append(epoch)
    else:
        assert v.divide_by in (None, 'num_episodes', 'num_steps')
        v.data.append(0)
assert type(log['reward']).__module__ == 'utils' and policy_net.encoder.weight.dtype == torch.float64
s = trainer.train_batch(0)                      # and the reference trains on from the restored state
print('REF-LOAD-OK', epoch, float(policy_net.encoder.weight.sum()))
'''


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree only exists in the build container")
def test_the_reference_loads_a_checkpoint_written_here(tmp_path):
    """checkpoint.save -> the reference's own load path (main.py:267-272) + its log bookkeeping + a train_batch."""
    from ic3net_amd import checkpoint
    net, log = _pp_easy_policy(), checkpoint.new_log()
    tr = _Opt(net)
    (net.encoder.weight.sum() + net.value_head.weight.sum()).backward()
    tr.optimizer.step()                                    # non-empty RMSprop state
    log['epoch'].data.extend([1, 2])
    for k in log:
        if k != 'epoch':
            log[k].data.extend([0.5, np.array([1.0, 2.0, 3.0])] if k == 'reward' else [0, 0])
    path = str(tmp_path / "ours.pt")
    checkpoint.save(path, net, log, tr)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE='1', OMP_NUM_THREADS='1')
    r = subprocess.run([sys.executable, '-c', _REF_LOADER, path, root], capture_output=True, text=True, env=env,
                       timeout=300)
    assert r.returncode == 0 and 'REF-LOAD-OK 3' in r.stdout, r.stderr[-2000:]
    want = float(net.encoder.weight.double().sum())
    got = float(r.stdout.split('REF-LOAD-OK 3')[1].split()[0])
    assert abs(got - want) < 1e-2 * max(1.0, abs(want))     # the reference took one RMSprop step from our weights
