"""Command-line runner with the reference's flag set, derived flags, epoch loop, stat normalisation and stdout
format (/root/reference/main.py:22-155, 190-258), so the README commands of the reference run unchanged:

    python -m ic3net_amd.main --env_name predator_prey --nagents 3 --nprocesses 1 --num_epochs 5 --hid_size 128 \\
        --detach_gap 10 --lrate 0.001 --dim 5 --max_steps 20 --ic3net --vision 0 --recurrent --nenvs 400

Differences: `--nenvs` environments are simulated in lock-step on the GPU (one update uses nenvs x max_steps
env-steps, so `--batch_size` only matters when it exceeds that); `--nprocesses` must stay 1 — scale over GPUs with
`python -m torch.distributed.run --nproc-per-node G -m ic3net_amd.main ...` (gradients and stats are all-reduced,
multi_processing.py:74-98); `--plot` (visdom) and `--display` through curses are not available (`--display` prints a
text view).  The printed lines keep the format plot_script.py parses.
"""
import argparse
import gc
import os
import sys
import time

import numpy as np
import torch

from . import checkpoint, data, models, sharding
from .action_utils import parse_action_args
from .comm import CommNetMLP
from .trainer import Trainer
from .utils import init_args_for_env, merge_stat

# (flag, type or 'flag', default) — names, types and defaults of main.py:25-109
_FLAGS = [
    ('num_epochs', int, 100), ('epoch_size', int, 10), ('batch_size', int, 500), ('nprocesses', int, 16),
    ('hid_size', int, 64), ('recurrent', 'flag', False), ('gamma', float, 1.0), ('tau', float, 1.0), ('seed', int, -1),
    ('normalize_rewards', 'flag', False), ('lrate', float, 0.001), ('entr', float, 0), ('value_coeff', float, 0.01),
    ('env_name', str, 'Cartpole'), ('max_steps', int, 20), ('nactions', str, '1'), ('action_scale', float, 1.0),
    ('plot', 'flag', False), ('plot_env', str, 'main'), ('save', str, ''), ('save_every', int, 0), ('load', str, ''),
    ('display', 'flag', False), ('random', 'flag', False), ('commnet', 'flag', False), ('ic3net', 'flag', False),
    ('nagents', int, 1), ('comm_mode', str, 'avg'), ('comm_passes', int, 1), ('comm_mask_zero', 'flag', False),
    ('mean_ratio', float, 1.0), ('rnn_type', str, 'MLP'), ('detach_gap', int, 10000), ('comm_init', str, 'uniform'),
    ('hard_attn', 'flag', False), ('comm_action_one', 'flag', False), ('advantages_per_action', 'flag', False),
    ('share_weights', 'flag', False),
]
# engine flags (new)
def _hip_graph_mode(v):
    """--hip_graph [0 | 1 | episode | step]: 1 / episode (the default; also a bare `--hip_graph`): rollouts that feed no update
    (evaluation, --display off) replay ONE hipGraph per episode; step: one graph per step index; 0: eager launches."""
    v = str(v).lower()
    if v in ('0', 'false', 'off', 'eager'):
        return False
    if v in ('1', 'true', 'on', 'episode'):
        return True
    if v == 'step':
        return 'step'
    raise argparse.ArgumentTypeError("--hip_graph takes 0, 1, episode or step (got %r)" % v)


_ENGINE_FLAGS = [('nenvs', int, 32), ('device', int, -1), ('hip_graph', 'hip_graph', True), ('tune_gemm', 'flag', False),
                 ('dist_backend', str, 'nccl')]     # 'nccl' = RCCL over xGMI; 'gloo' for several ranks on one GPU (tests)


def build_parser(argv):
    parser = argparse.ArgumentParser(description='IC3Net trainer on the MI355X batched rollout engine')
    for name, typ, default in _FLAGS + _ENGINE_FLAGS:
        if typ == 'flag':
            parser.add_argument('--' + name, action='store_true', default=default)
        elif typ == 'hip_graph':
            parser.add_argument('--' + name, type=_hip_graph_mode, nargs='?', const=True, default=default)
        else:
            parser.add_argument('--' + name, type=typ, default=default)
    init_args_for_env(parser, argv)                 # main.py:112: the env adds its own flag group
    return parser


def derive_args(args):
    """main.py:115-130 (before the env exists)."""
    if args.ic3net:
        args.commnet = 1
        args.hard_attn = 1
        args.mean_ratio = 0
        if args.env_name == "traffic_junction":
            args.comm_action_one = True
    args.nfriendly = args.nagents
    if getattr(args, 'enemy_comm', False):
        if hasattr(args, 'nenemies'):
            args.nagents += args.nenemies
        else:
            raise RuntimeError("Env. needs to pass argument 'nenemy'.")
    return args


def finish_args(args, env):
    """main.py:134-155 (after the env exists)."""
    args.num_actions = env.num_actions
    if not isinstance(args.num_actions, (list, tuple)):
        args.num_actions = [args.num_actions]
    args.dim_actions = env.dim_actions
    args.num_inputs = env.observation_dim
    if args.hard_attn and args.commnet:
        args.num_actions = [*args.num_actions, 2]
        args.dim_actions = env.dim_actions + 1
    if args.commnet and (args.recurrent or args.rnn_type == 'LSTM'):
        args.recurrent = True
        args.rnn_type = 'LSTM'
    parse_action_args(args)
    return args


def make_policy(args, num_inputs):                   # main.py:164-171
    if args.commnet:
        return CommNetMLP(args, num_inputs)
    if args.random:
        return models.Random(args, num_inputs)
    if args.recurrent:
        return models.RNN(args, num_inputs)
    return models.MLP(args, num_inputs)


def format_epoch(epoch, stat, epoch_time):
    """The lines main.py:229-244 prints (np precision 2) — plot_script.py parses these."""
    np.set_printoptions(precision=2)
    lines = ['Epoch {}\tReward {}\tTime {:.2f}s'.format(epoch, stat['reward'], epoch_time)]
    if 'enemy_reward' in stat:
        lines.append('Enemy-Reward: {}'.format(stat['enemy_reward']))
    if 'add_rate' in stat:
        lines.append('Add-Rate: {:.2f}'.format(stat['add_rate']))
    if 'success' in stat:
        lines.append('Success: {:.2f}'.format(stat['success']))
    if 'steps_taken' in stat:
        lines.append('Steps-taken: {:.2f}'.format(stat['steps_taken']))
    if 'comm_action' in stat:
        lines.append('Comm-Action: {}'.format(stat['comm_action']))
    if 'enemy_comm' in stat:
        lines.append('Enemy-Comm: {}'.format(stat['enemy_comm']))
    return lines


def normalise_epoch(stat, log):
    """main.py:219-225: divide each logged stat by its `divide_by` field (num_episodes / num_steps) and append."""
    epoch = len(log['epoch'].data) + 1
    for k, v in log.items():
        if k == 'epoch':
            v.data.append(epoch)
        else:
            if k in stat and v.divide_by is not None and stat[v.divide_by] > 0:
                stat[k] = stat[k] / stat[v.divide_by]
            v.data.append(stat.get(k, 0))
    return epoch


def run(argv=None, out=print):
    argv = list(sys.argv if argv is None else ['main.py'] + list(argv))
    args = build_parser(argv).parse_args(argv[1:])
    if args.nprocesses != 1:
        out("note: --nprocesses %d ignored: environments are batched on the GPU (--nenvs), one process per GPU" %
            args.nprocesses)
        args.nprocesses = 1
    if args.plot:
        raise NotImplementedError("--plot (visdom) is outside the hot-path scope (SURVEY §2 row 14)")
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if args.device < 0:
        args.device = int(os.environ.get('LOCAL_RANK', 0))
    torch.cuda.set_device(args.device)
    if world > 1:
        import torch.distributed as dist
        if args.dist_backend == 'nccl':
            dist.init_process_group(backend='nccl', device_id=torch.device('cuda', args.device))
        else:
            dist.init_process_group(backend=args.dist_backend)
    derive_args(args)
    if args.seed == -1:                               # main.py:157-159
        args.seed = int(np.random.randint(0, 10000))
    if world > 1:
        # one seed for the whole job (the reference's workers inherit the master's args, main.py:177-178): the env /
        # sampling streams are keyed by (seed, global env id) and every replica must initialise the same policy
        args.seed = sharding.broadcast_seed(args.seed)
    args.env_id_offset = rank * args.nenvs            # shard-invariant env streams (global env ids)
    torch.manual_seed(args.seed)
    env = data.init(args.env_name, args, False)
    finish_args(args, env)
    policy_net = make_policy(args, args.num_inputs).to(torch.device('cuda', args.device)).float()
    trainer = Trainer(args, policy_net, env)
    log = checkpoint.new_log()
    if args.load != '':
        checkpoint.load(args.load, policy_net, log, trainer, map_location=torch.device('cuda', args.device))
    if world > 1:
        # the reference's workers share ONE parameter set (share_memory_, main.py:177-178); replicas get rank 0's
        # parameters (and RMSprop state after --load) so that identical all-reduced gradients keep them identical
        sharding.broadcast_parameters(policy_net, trainer.optimizer)
    if args.tune_gemm:     # TunableOp times the hipBLASLt/rocBLAS candidates for every GEMM shape of the first update
        import torch.cuda.tunable as tunable
        tunable.enable(True)
        tunable.tuning_enable(True)
        tunable.set_filename(os.path.join(os.environ.get('TMPDIR', '/tmp'), 'ic3_tunableop_%d.csv' % os.getpid()))
    gc.collect()
    gc.freeze()            # a full GC pass over the torch heap stalls the launch thread for 35-80 ms
    for ep in range(args.num_epochs):                 # main.py:206-258
        t0 = time.time()
        stat = dict()
        for n in range(args.epoch_size):
            if n == args.epoch_size - 1 and args.display:
                trainer.display = True
            merge_stat(trainer.train_batch(ep), stat)
            trainer.display = False
            if args.tune_gemm and ep == 0 and n == 0:
                torch.cuda.tunable.tuning_enable(False)      # keep the selections, stop tuning
        epoch_time = time.time() - t0
        epoch = normalise_epoch(stat, log)
        if rank == 0:
            for line in format_epoch(epoch, stat, epoch_time):
                out(line)
            if args.save_every and ep and args.save != '' and ep % args.save_every == 0:
                checkpoint.save(args.save + '_' + str(ep), policy_net, log, trainer)
            if args.save != '':
                checkpoint.save(args.save, policy_net, log, trainer)
    dump = os.environ.get('IC3_DUMP_PARAMS')           # tests: every rank's final parameters + last epoch's stats
    if dump:
        torch.save({'params': {k: v.detach().cpu() for k, v in policy_net.state_dict().items()},
                    'log': {k: list(v.data) for k, v in log.items()}}, '%s.rank%d.pt' % (dump, rank))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()
    return log


if __name__ == '__main__':
    run()
