"""Host logic of the command-line runner (flag set, derived flags, epoch normalisation, stdout format):
/root/reference/main.py:22-155, 219-244.  Expected strings are written out from the reference's format strings."""
import re
from types import SimpleNamespace

import numpy as np

from ic3net_amd import checkpoint, main


def parse(argv):
    argv = ['main.py'] + argv
    return main.build_parser(argv).parse_args(argv[1:])


def test_flag_defaults_match_reference():
    a = parse([])
    # main.py:25-109 defaults
    assert (a.num_epochs, a.epoch_size, a.batch_size, a.nprocesses, a.hid_size) == (100, 10, 500, 16, 64)
    assert (a.gamma, a.tau, a.seed, a.lrate, a.entr, a.value_coeff) == (1.0, 1.0, -1, 0.001, 0, 0.01)
    assert (a.env_name, a.max_steps, a.nactions, a.action_scale) == ('Cartpole', 20, '1', 1.0)
    assert (a.comm_mode, a.comm_passes, a.mean_ratio, a.rnn_type, a.detach_gap, a.comm_init) == \
        ('avg', 1, 1.0, 'MLP', 10000, 'uniform')
    for flag in ('recurrent', 'normalize_rewards', 'plot', 'display', 'random', 'commnet', 'ic3net', 'comm_mask_zero',
                 'hard_attn', 'comm_action_one', 'advantages_per_action', 'share_weights'):
        assert getattr(a, flag) is False


def test_env_flag_groups():
    a = parse(['--env_name', 'predator_prey', '--nenemies', '2', '--dim', '7', '--mode', 'cooperative'])
    assert (a.nenemies, a.dim, a.vision, a.mode, a.no_stay, a.enemy_comm) == (2, 7, 2, 'cooperative', False, False)
    a = parse(['--env_name', 'traffic_junction', '--difficulty', 'hard', '--add_rate_max', '0.05'])
    assert (a.dim, a.vision, a.difficulty, a.vocab_type, a.add_rate_min, a.add_rate_max) == \
        (5, 1, 'hard', 'bool', 0.05, 0.05)


def test_ic3net_derivations():
    a = main.derive_args(parse(['--env_name', 'traffic_junction', '--ic3net', '--nagents', '5']))
    assert (a.commnet, a.hard_attn, a.mean_ratio, a.comm_action_one, a.nfriendly) == (1, 1, 0, True, 5)
    a = main.derive_args(parse(['--env_name', 'predator_prey', '--ic3net', '--nagents', '3']))
    assert a.comm_action_one is False and a.nagents == 3
    a = main.derive_args(parse(['--env_name', 'predator_prey', '--commnet', '--nagents', '3', '--enemy_comm',
                                '--nenemies', '2']))
    assert (a.nfriendly, a.nagents, a.hard_attn) == (3, 5, False)


def test_finish_args():
    env = SimpleNamespace(num_actions=5, dim_actions=1, observation_dim=29)
    a = main.finish_args(main.derive_args(parse(['--env_name', 'predator_prey', '--ic3net', '--recurrent'])), env)
    assert a.num_actions == [5, 2] and a.dim_actions == 2 and a.num_inputs == 29
    assert a.recurrent is True and a.rnn_type == 'LSTM'
    assert a.naction_heads == [5, 2]                       # action_utils.py:5-24 (discrete)
    a = main.finish_args(main.derive_args(parse(['--env_name', 'predator_prey', '--commnet'])), env)
    assert a.num_actions == [5] and a.dim_actions == 1 and a.rnn_type == 'MLP'


def test_normalise_and_format():
    log = checkpoint.new_log()
    stat = {'num_episodes': 4, 'num_steps': 80, 'reward': np.array([2.0, -4.0, 1.0]), 'success': 3,
            'steps_taken': 60, 'add_rate': 0.8, 'comm_action': np.array([40.0, 20.0, 80.0]), 'value_loss': 8.0,
            'action_loss': -16.0, 'entropy': 160.0}
    epoch = main.normalise_epoch(stat, log)
    assert epoch == 1 and log['epoch'].data == [1]
    np.testing.assert_allclose(stat['reward'], [0.5, -1.0, 0.25])
    assert stat['success'] == 0.75 and stat['steps_taken'] == 15 and stat['add_rate'] == 0.2
    np.testing.assert_allclose(stat['comm_action'], [0.5, 0.25, 1.0])
    assert stat['value_loss'] == 0.1 and stat['entropy'] == 2.0
    assert log['enemy_reward'].data == [0] and log['success'].data == [0.75]      # missing stats log 0 (main.py:225)
    lines = main.format_epoch(epoch, stat, 1.2345)
    assert lines == ['Epoch 1\tReward [ 0.5  -1.    0.25]\tTime 1.23s', 'Add-Rate: 0.20', 'Success: 0.75',
                     'Steps-taken: 15.00', 'Comm-Action: [0.5  0.25 1.  ]']
    # the patterns plot_script.py scans for (plot_script.py:30-60): 'Epoch' lines carry Reward, others 'Name: value'
    assert re.match(r'Epoch \d+\tReward .*\tTime [\d.]+s$', lines[0])
    # second epoch appends
    assert main.normalise_epoch({'num_episodes': 2, 'num_steps': 10, 'reward': np.array([1.0])}, log) == 2
    assert log['epoch'].data == [1, 2] and len(log['reward'].data) == 2
