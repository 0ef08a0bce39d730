"""End-to-end runs of the command-line runner on the GPU: the reference's README commands (shortened) train, print the
epoch lines in the reference's format, save a checkpoint and resume from it (main.py:206-272)."""
import re

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

PP = ['--env_name', 'predator_prey', '--nagents', '3', '--nprocesses', '1', '--num_epochs', '2', '--epoch_size', '2',
      '--hid_size', '64', '--detach_gap', '10', '--lrate', '0.001', '--dim', '5', '--max_steps', '20', '--ic3net',
      '--vision', '0', '--recurrent', '--nenvs', '16', '--seed', '5']
TJ = ['--env_name', 'traffic_junction', '--nagents', '5', '--nprocesses', '1', '--num_epochs', '2', '--epoch_size',
      '2', '--hid_size', '64', '--detach_gap', '10', '--lrate', '0.001', '--dim', '6', '--max_steps', '20',
      '--ic3net', '--vision', '0', '--recurrent', '--add_rate_min', '0.1', '--add_rate_max', '0.3', '--curr_start',
      '0', '--curr_end', '2', '--difficulty', 'easy', '--nenvs', '16', '--seed', '5']


def run(argv):
    from ic3net_amd import main
    lines = []
    log = main.run(argv, out=lines.append)
    return lines, log


def test_pp_run_prints_reference_format(tmp_path):
    path = str(tmp_path / 'pp.pt')
    lines, log = run(PP + ['--save', path])
    epochs = [l for l in lines if l.startswith('Epoch')]
    assert len(epochs) == 2
    for i, l in enumerate(epochs):
        m = re.match(r'Epoch (\d+)\tReward \[(.*)\]\tTime ([\d.]+)s$', l)
        assert m and int(m.group(1)) == i + 1
        assert len(m.group(2).split()) == 3                      # one reward per predator
    assert sum(l.startswith('Success: ') for l in lines) == 2
    assert sum(l.startswith('Steps-taken: ') for l in lines) == 2
    assert sum(l.startswith('Comm-Action: ') for l in lines) == 2
    assert not any(l.startswith('Add-Rate') for l in lines)
    assert log['epoch'].data == [1, 2]
    for k in ('reward', 'value_loss', 'action_loss', 'entropy', 'comm_action', 'steps_taken', 'success'):
        assert len(log[k].data) == 2 and np.all(np.isfinite(np.asarray(log[k].data[-1], dtype=np.float64)))
    assert 0 < log['steps_taken'].data[-1] <= 20
    # resume: the log continues from epoch 3 and the weights are the saved ones
    from ic3net_amd import checkpoint
    saved = checkpoint.read(path)        # the log is pickled as the reference's utils.LogField (checkpoint.py)
    lines2, log2 = run(PP + ['--load', path, '--num_epochs', '1'])
    assert [l.split('\t')[0] for l in lines2 if l.startswith('Epoch')] == ['Epoch 3']
    assert log2['epoch'].data == [1, 2, 3]
    assert set(saved.keys()) == {'policy_net', 'log', 'trainer'}


def test_tj_run_prints_add_rate_and_success():
    lines, log = run(TJ)
    rates = [float(l.split(' ')[-1]) for l in lines if l.startswith('Add-Rate: ')]
    # curriculum (TJ:222-231, reset :178-181): epoch index 0 keeps add_rate_min (epoch_last_update starts at 0),
    # epoch index 1 adds one step of (max-min)/(end-start) = 0.1, floored to 0.01 units in float arithmetic
    assert rates[0] == 0.10 and rates[1] in (0.19, 0.20)
    succ = [float(l.split(' ')[-1]) for l in lines if l.startswith('Success: ')]
    assert len(succ) == 2 and all(0.0 <= s <= 1.0 for s in succ)
    assert any(l.startswith('Comm-Action: ') for l in lines)
    assert log['add_rate'].data[0] == pytest.approx(0.1) and 0.185 < log['add_rate'].data[1] < 0.205


def test_baselines_and_flags_run():
    base = ['--env_name', 'predator_prey', '--nagents', '3', '--num_epochs', '1', '--epoch_size', '1', '--nprocesses',
            '1', '--dim', '5', '--vision', '1', '--max_steps', '10', '--nenvs', '8', '--seed', '1']
    for extra in (['--commnet'], ['--commnet', '--recurrent', '--comm_passes', '2'], [], ['--recurrent'],
                  ['--random'], ['--commnet', '--rnn_type', 'LSTM', '--share_weights', '--comm_mode', 'sum'],
                  ['--ic3net', '--recurrent', '--enemy_comm', '--nenemies', '1', '--mode', 'competitive']):
        lines, log = run(base + extra)
        assert lines[0].startswith('Epoch 1\tReward '), (extra, lines)
        if '--enemy_comm' in extra:
            assert any(l.startswith('Enemy-Reward: ') for l in lines) and any(l.startswith('Enemy-Comm: ') for l in lines)
