"""GPU: the batched Trainer.get_episode / run_batch surface against golden vectors recorded from the
reference's own Trainer.get_episode (action tape + injected env RNG, tests/golden/trainer_*.npz)."""
import argparse
import ast

import numpy as np
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from golden_util import load  # noqa: E402

FIXTURES = [("trainer_pp_easy", "predator_prey"), ("trainer_pp_medium", "predator_prey"),
            ("trainer_tj_easy", "traffic_junction"), ("trainer_tj_medium_commnet", "traffic_junction"),
            ("trainer_pp_hard", "predator_prey"), ("trainer_tj_hard", "traffic_junction")]   # BASELINE shapes, 80 steps


def build_args(env_name, flags, N, T, nenv, seed):
    a = argparse.Namespace(
        batch_size=500, hid_size=64, recurrent=False, seed=seed, lrate=0.001, env_name=env_name, max_steps=T,
        display=False, commnet=False, ic3net=False, nagents=N, comm_mode='avg', comm_passes=1, comm_mask_zero=False,
        mean_ratio=1.0, rnn_type='MLP', detach_gap=10000, comm_init='uniform', hard_attn=False, comm_action_one=False,
        share_weights=False, nenvs=nenv, env_id_offset=300, store_states=False, gamma=1.0, normalize_rewards=False,
        entr=0, value_coeff=0.01, advantages_per_action=False)
    if env_name == 'predator_prey':
        a.__dict__.update(nenemies=1, dim=5, vision=2, moving_prey=False, no_stay=False, mode='mixed', enemy_comm=False)
    else:
        a.__dict__.update(dim=5, vision=1, add_rate_min=0.05, add_rate_max=0.2, curr_start=0, curr_end=0,
                          difficulty='easy', vocab_type='bool')
    a.__dict__.update(flags)
    if a.ic3net:                          # main.py:115-123
        a.commnet = 1
        a.hard_attn = 1
        a.mean_ratio = 0
        if a.env_name == 'traffic_junction':
            a.comm_action_one = True
    a.nfriendly = a.nagents
    return a


@pytest.mark.parametrize("name,env_name", FIXTURES)
def test_get_episode_matches_reference(name, env_name):
    from ic3net_amd import data, trainer as trmod
    from ic3net_amd.action_utils import parse_action_args
    from ic3net_amd.comm import CommNetMLP
    fx = load(name)
    N, T, nenv, nep, nh, seed = [int(x) for x in fx["cfg"]]
    flags = dict(ast.literal_eval(str(fx["flags"])))
    a = build_args(env_name, flags, N, T, nenv, seed)
    env = data.init(env_name, a, False)
    a.num_actions = [env.num_actions]     # main.py:134-152
    a.dim_actions = env.dim_actions
    a.num_inputs = env.observation_dim
    if a.hard_attn and a.commnet:
        a.num_actions = [*a.num_actions, 2]
        a.dim_actions = env.dim_actions + 1
    if a.commnet and (a.recurrent or a.rnn_type == 'LSTM'):
        a.recurrent, a.rnn_type = True, 'LSTM'
    parse_action_args(a)
    assert len(a.naction_heads) == nh
    torch.manual_seed(seed)
    net = CommNetMLP(a, a.num_inputs).cuda()
    tr = trmod.Trainer(a, net, env)
    tape = fx["tape"]                     # (nenv, nep, T, heads, N)

    def taped(args, action_out, clock, out=None):
        assert [tuple(x.shape) for x in action_out] == [(nenv, N, A) for A in a.naction_heads]
        act = torch.from_numpy(tape[:, clock.episode, clock.t]).permute(1, 0, 2).contiguous().int().cuda()
        if out is not None:
            out.copy_(act)
            return out
        return act
    orig = trmod.select_action
    trmod.select_action = taped
    try:
        for ep in range(nep):
            episode, stat = tr.get_episode(ep)
            nsteps = fx["nsteps"][:, ep]
            assert len(episode) == T
            for t, trn in enumerate(episode):
                assert trn.value.shape == (nenv * N, 1)
                rew, em, emm = trn.reward.cpu().numpy(), trn.episode_mask.cpu().numpy(), \
                    trn.episode_mini_mask.cpu().numpy()
                am, live = trn.misc['alive_mask'].cpu().numpy(), trn.misc['live'].cpu().numpy()
                act = trn.action.cpu().numpy()
                for e in range(nenv):
                    if t < nsteps[e]:
                        assert live[e] == 1
                        np.testing.assert_array_equal(act[:, e], fx["action"][e, ep, t])
                        np.testing.assert_array_equal(rew[e], fx["reward"][e, ep, t].astype(np.float32))
                        np.testing.assert_array_equal(em[e], fx["episode_mask"][e, ep, t])
                        np.testing.assert_array_equal(emm[e], fx["episode_mini_mask"][e, ep, t])
                        np.testing.assert_array_equal(am[e], fx["alive_mask"][e, ep, t])
                    else:
                        assert live[e] == 0 and not am[e].any() and not rew[e].any()
            # stats: the batched episode's stat is the sum of the E reference episodes' stats
            assert stat['num_steps'] == nsteps.sum() == stat['steps_taken']
            for k in [f[5:] for f in fx.files if f.startswith("stat:")]:
                want = fx["stat:" + k][:, ep].sum(0)
                np.testing.assert_allclose(np.asarray(stat[k], np.float64), want, rtol=1e-5, atol=1e-5, err_msg=k)
            assert set(stat) == set(f[5:] for f in fx.files if f.startswith("stat:"))
    finally:
        trmod.select_action = orig


def test_run_batch_surface():
    from ic3net_amd import data, trainer as trmod
    from ic3net_amd.action_utils import parse_action_args
    from ic3net_amd.comm import CommNetMLP
    a = build_args('predator_prey', dict(nagents=3, dim=5, vision=0, hid_size=32, ic3net=True, recurrent=True,
                                         detach_gap=10), 3, 20, 16, 1)
    a.env_id_offset = 0
    env = data.init('predator_prey', a, False)
    a.num_actions = [env.num_actions, 2]
    a.dim_actions = 2
    a.num_inputs = env.observation_dim
    a.recurrent, a.rnn_type = True, 'LSTM'
    parse_action_args(a)
    net = CommNetMLP(a, a.num_inputs).cuda()
    tr = trmod.Trainer(a, net, env)
    a.batch_size = 500
    batch, stats = tr.run_batch(0)
    assert stats['num_episodes'] % 16 == 0 and stats['num_steps'] >= 500
    assert stats['num_steps'] - 500 < 20 * 16                      # overshoot < one batched episode (quirk Q29)
    assert set(stats) >= {'num_episodes', 'num_steps', 'steps_taken', 'reward', 'comm_action', 'success'}
    assert isinstance(batch, trmod.Transition) and len(batch.reward) == len(batch.action)
    assert batch.action[0].shape == (2, 16, 3) and batch.action_out[0][0].shape == (16, 3, 5)
    # same seeds -> same rollout (counter-based streams)
    tr2 = trmod.Trainer(a, net, data.init('predator_prey', a, False))
    batch2, stats2 = tr2.run_batch(0)
    assert stats2['num_steps'] == stats['num_steps']
    assert all(torch.equal(x, y) for x, y in zip(batch.action, batch2.action))
    np.testing.assert_array_equal(stats['reward'], stats2['reward'])


@pytest.mark.parametrize("env_name,flags", [
    ("predator_prey", dict(nagents=5, dim=10, vision=1, hid_size=32, ic3net=True, recurrent=True, detach_gap=10)),
    ("traffic_junction", dict(nagents=10, dim=14, vision=1, hid_size=32, ic3net=True, recurrent=True, detach_gap=10,
                              difficulty='medium', add_rate_min=0.2, add_rate_max=0.2)),
    ("traffic_junction", dict(nagents=5, dim=6, vision=0, hid_size=32, commnet=True, recurrent=True, detach_gap=10,
                              difficulty='easy', add_rate_min=0.1, add_rate_max=0.3, curr_start=0, curr_end=4)),
    # the one-launch kernel, two communication passes (two launches per captured step)
    ("predator_prey", dict(nagents=5, dim=10, vision=1, hid_size=64, ic3net=True, recurrent=True, detach_gap=10,
                           comm_passes=2)),
])
def test_hip_graph_replay_equals_eager(env_name, flags):
    """args.hip_graph: episode 0 eager, episode 1 captured, episodes 2.. replayed — every episode must be
    identical to the all-eager run (counter-based streams; the curriculum case changes add_rate between episodes)."""
    from ic3net_amd import data, trainer as trmod
    from ic3net_amd.action_utils import parse_action_args
    from ic3net_amd.comm import CommNetMLP

    def run(graph, overlap=False):
        a = build_args(env_name, dict(flags), flags['nagents'], 12, 64, 5)
        a.env_id_offset = 0
        a.hip_graph = graph
        a.overlap_obs = overlap
        env = data.init(env_name, a, False)
        a.num_actions = [env.num_actions]
        a.dim_actions = env.dim_actions
        a.num_inputs = env.observation_dim
        if a.hard_attn and a.commnet:
            a.num_actions = [*a.num_actions, 2]
            a.dim_actions = env.dim_actions + 1
        a.recurrent, a.rnn_type = True, 'LSTM'
        parse_action_args(a)
        torch.manual_seed(11)
        net = CommNetMLP(a, a.num_inputs).cuda()
        tr = trmod.Trainer(a, net, env)
        out = []
        for ep in range(5):
            episode, stat = tr.get_episode(ep)
            out.append(([t.action.clone() for t in episode], [t.reward.clone() for t in episode],
                        [t.value.clone() for t in episode], [t.misc['alive_mask'].clone() for t in episode],
                        {k: np.asarray(v).copy() for k, v in stat.items()}))
        return out, tr

    eager, _ = run(False)
    graphed, tr = run('step')                                   # one graph per step index
    assert len(tr._graphs) == 12
    whole, tr3 = run(True)                                      # the default: ONE graph for the episode's 12 step launches
    assert list(tr3._graphs) == ['episode']
    # args.overlap_obs: observation assembled on a second stream from a state snapshot (ic3_env_observe_at)
    lapped, tr2 = run(True, overlap=True)                       # (keeps one graph per step: the obs launch sits between them)
    assert tr2._side is not None and 'episode' not in tr2._graphs
    raw = tr2.env.env
    torch.cuda.synchronize()
    last_obs = raw._obs.clone()
    assert torch.equal(last_obs, raw.observe())            # the side stream produced the observation of the final state
    for other in (graphed, whole, lapped):
        for ep, (e, g) in enumerate(zip(eager, other)):
            for i in range(4):
                for x, y in zip(e[i], g[i]):
                    assert torch.equal(x, y), (ep, i)
            assert set(e[4]) == set(g[4])
            for k in e[4]:
                np.testing.assert_array_equal(e[4][k], g[4][k], err_msg=k)


@pytest.mark.parametrize("mode", [True, 'step'], ids=["episode-graph", "step-graphs"])
@pytest.mark.parametrize("hid", [64, 32])
def test_graph_replay_follows_weight_updates(hid, mode):
    """Replayed step graphs hold the ADDRESSES of the derived weight tensors (packed layouts; for hid 32 the zero-padded
    twin's parameters and ITS packed layouts): after the parameters change in place (optimizer.step, checkpoint load) the
    next episode must play the new weights — begin_episode refreshes the derived tensors in place."""
    from ic3net_amd import data, trainer as trmod
    from ic3net_amd.action_utils import parse_action_args
    from ic3net_amd.comm import CommNetMLP
    flags = dict(nagents=5, dim=10, vision=1, hid_size=hid, ic3net=True, recurrent=True, detach_gap=10)

    def run(graph):
        a = build_args("predator_prey", dict(flags), 5, 12, 64, 5)
        a.env_id_offset = 0
        a.hip_graph = graph
        env = data.init("predator_prey", a, False)
        a.num_actions = [env.num_actions, 2]
        a.dim_actions = env.dim_actions + 1
        a.num_inputs = env.observation_dim
        a.recurrent, a.rnn_type = True, 'LSTM'
        parse_action_args(a)
        torch.manual_seed(11)
        net = CommNetMLP(a, a.num_inputs).cuda()
        tr = trmod.Trainer(a, net, env)
        out = []
        for ep in range(6):
            if ep == 3:
                with torch.no_grad():
                    for q in net.parameters():
                        q.mul_(0.5).add_(0.01)
            episode, _ = tr.get_episode(ep)
            out.append(([t.action.clone() for t in episode], [t.value.clone() for t in episode]))
        return out, tr

    eager, _ = run(False)
    graphed, tr = run(mode)
    assert len(tr._graphs) == (12 if mode == 'step' else 1) and getattr(tr.policy_net, 'mega_steps', 0) > 0
    assert not all(torch.equal(x, y) for x, y in zip(eager[2][1], eager[3][1]))     # the update does change the episode
    for ep, (e, g) in enumerate(zip(eager, graphed)):
        for i in range(2):
            for x, y in zip(e[i], g[i]):
                assert torch.equal(x, y), (ep, i)


GRAD_FIXTURES = [("grad_pp_easy_ic3net", "predator_prey"), ("grad_pp_medium_commnet_norm", "predator_prey"),
                 ("grad_tj_easy_ic3net", "traffic_junction"), ("grad_tj_medium_perhead", "traffic_junction"),
                 # BASELINE shapes (hid 128, 80 steps, detach_gap 10), closed-form weights: configs[1] and configs[3]
                 ("grad_pp_hard_ic3net", "predator_prey"), ("grad_tj_hard_ic3net", "traffic_junction"),
                 # config 5's hidden size and agent count (hid 256, 32 agents, vision 2): the recomputing backward at <256>
                 ("grad_pp_scaled_h256_ic3net", "predator_prey"),
                 # the NON-recurrent module (comm.py:127-129,220-224): CommNet with two passes; gated, shared weights
                 ("grad_pp_medium_commnet_mlp2", "predator_prey"), ("grad_tj_easy_ic3net_mlp2share", "traffic_junction"),
                 # the IC / IRIC baselines (models.py:8-97; round 5): models.MLP, models.RNN with the tanh recurrence / the LSTM cell
                 ("grad_pp_medium_ic_mlp", "predator_prey"), ("grad_pp_medium_iric_rnn", "predator_prey"),
                 ("grad_tj_easy_iric_lstm", "traffic_junction")]


@pytest.mark.parametrize("native", [False, True])
@pytest.mark.parametrize("name,env_name", GRAD_FIXTURES)
def test_compute_grad_matches_reference(name, env_name, native):
    """run_batch + compute_grad against the reference's own trainer.py:128-225 on the same weights, action tape and env
    draws: losses and every parameter gradient (fp32 on GPU vs the reference's fp64).  native=False: the rollout keeps
    the autograd graph (like the reference); native=True: a no-grad rollout + the explicit backward through time of
    ic3net_amd.bptt (recompute per step from the recorded (h, c) and env snapshots) — the default of train_batch."""
    from ic3net_amd import data, trainer as trmod
    from ic3net_amd.action_utils import parse_action_args
    from ic3net_amd.comm import CommNetMLP
    fx = load(name)
    N, T, nenv, nep, nh, seed = [int(x) for x in fx["cfg"]]
    flags = dict(ast.literal_eval(str(fx["flags"])))
    a = build_args(env_name, flags, N, T, nenv, seed)
    a.env_id_offset = 400
    env = data.init(env_name, a, False)
    a.num_actions = [env.num_actions]
    a.dim_actions = env.dim_actions
    a.num_inputs = env.observation_dim
    if a.hard_attn and a.commnet:
        a.num_actions = [*a.num_actions, 2]
        a.dim_actions = env.dim_actions + 1
    if a.commnet and (a.recurrent or a.rnn_type == 'LSTM'):
        a.recurrent, a.rnn_type = True, 'LSTM'
    parse_action_args(a)
    model = str(fx["model"]) if "model" in fx.files else None
    if model:                                         # main.py:161-168: the baselines instead of CommNetMLP
        from ic3net_amd import models
        a.continuous = False
        net = (models.RNN if model == 'rnn' else models.MLP)(a, a.num_inputs)
    else:
        net = CommNetMLP(a, a.num_inputs)
    if "param_names" in fx.files:                     # full-size fixture: weights from the index alone
        from policy_util import closed_form_weights
        shapes = {str(n): eval(str(sh)) for n, sh in zip(fx["param_names"], fx["param_shapes"])}
        net.load_state_dict({k: torch.from_numpy(v).float() for k, v in closed_form_weights(shapes).items()})
    else:
        net.load_state_dict({k[2:]: torch.from_numpy(fx[k]).float() for k in fx.files if k.startswith("w:")})
    net = net.cuda()
    tr = trmod.Trainer(a, net, env)
    tape = fx["tape"]

    def taped(args, action_out, clock, out=None):
        act = torch.from_numpy(tape[:, clock.episode, clock.t]).permute(1, 0, 2).contiguous().int().cuda()
        out.copy_(act)
        return out
    orig = trmod.select_action
    trmod.select_action = taped
    try:
        a.rollout_grad = not native
        a.batch_size = int(fx["num_steps"])          # exactly nep batched episodes
        if native:
            assert tr._native_update()
            tr._records = []
        batch, stats = tr.run_batch(0)
        assert stats['num_steps'] == int(fx["num_steps"]) and stats['num_episodes'] == nenv * nep
        tr.optimizer.zero_grad()
        if native:
            assert len(tr._records) == nep and not batch.value[0].requires_grad
            s = tr.compute_grad_native(batch, tr._records)
        else:
            s = tr.compute_grad(batch)
    finally:
        trmod.select_action = orig
        tr._records = None
    _check_against_reference(name + ("/native" if native else "/autograd"), fx, s, net)


def _check_against_reference(label, fx, s, net):
    """Losses and every parameter gradient against the reference's (fp64), each at <= 4 x the error MEASURED for the fixture
    (profiles/r06/grad_errors.txt, written by this function under IC3_GRAD_ERRORS=<file>; round-5 verdict item 2b: the blanket
    3e-4 of rounds 3-5 would have passed an error 100 x the stale-plane defect's).  Gradients: max |g - g_ref| over the tensor,
    relative to the tensor's max |g_ref|; losses: relative."""
    loss_err = max(abs(s[k] - float(fx[k])) / max(abs(float(fx[k])), 1.0) for k in ("action_loss", "value_loss", "entropy"))
    worst, worst_name = 0.0, None
    for pname, p in net.named_parameters():
        g = fx["g:" + pname]
        if g.size == 0:
            assert p.grad is None, pname            # unused hidd_encoder (quirk Q18)
            continue
        scale = max(np.abs(g).max(), 1e-6)
        err = float(np.abs(p.grad.cpu().numpy().astype(np.float64) - g).max() / scale)
        if err >= worst:
            worst, worst_name = err, pname
    out = os.environ.get("IC3_GRAD_ERRORS")
    if out:
        with open(out, "a") as f:
            f.write("%-44s gradients %.3e (%s)   losses %.3e\n" % (label, worst, worst_name, loss_err))
    fixture = label.split("/")[0]
    gtol, ltol = GRAD_TOL.get(fixture, GRAD_TOL_DEFAULT)
    assert loss_err <= ltol, (label, "losses", loss_err, ltol)
    assert worst <= gtol, (label, worst_name, worst, gtol)


# (gradient, loss) bars per fixture = 4 x the worst error measured over the paths that run it (profiles/r06/grad_errors.txt),
# floored at 4e-6 / 2e-6 — fp32 on the GPU against the reference's fp64.  A fixture without an entry fails: measure it first.
GRAD_TOL_DEFAULT = (0.0, 0.0)
GRAD_TOL = {
    "grad_pp_easy_ic3net":             (5.6e-06, 2.0e-06),
    "grad_pp_hard_ic3net":             (4.3e-05, 4.2e-06),
    "grad_pp_scaled_h256_ic3net":      (8.6e-06, 2.0e-06),
    "grad_pp_medium_commnet_mlp2":     (4.0e-06, 2.0e-06),
    "grad_pp_medium_commnet_norm":     (4.0e-06, 1.0e-04),
    "grad_pp_medium_ic_mlp":           (4.0e-06, 2.0e-06),
    "grad_pp_medium_iric_rnn":         (4.0e-06, 1.1e-04),
    "grad_tj_easy_ic3net":             (4.0e-06, 2.0e-06),
    "grad_tj_easy_ic3net_mlp2share":   (4.0e-06, 2.0e-06),
    "grad_tj_easy_iric_lstm":          (4.0e-06, 2.0e-06),
    "grad_tj_hard_ic3net":             (7.0e-05, 2.0e-06),
    "grad_tj_medium_perhead":          (4.0e-06, 2.0e-06),
    "gradstream_pp_small_h128":        (7.0e-06, 2.0e-06),
    "gradstream_pp_tiny_commnet_mlp2": (6.1e-06, 2.0e-06),
    "gradstream_pp_tiny_ic3net_p2":    (2.3e-05, 2.0e-06),
    "gradstream_pp_tiny_ic_mlp":       (4.0e-06, 2.0e-06),
    "gradstream_pp_tiny_iric_lstm":    (6.0e-06, 2.0e-06),
    "gradstream_pp_tiny_iric_rnn":     (4.0e-06, 2.0e-06),
    "gradstream_tj_easy_ic3net_mlp":   (9.3e-06, 2.0e-06),
    "gradstream_pp_tiny_commnet":      (4.0e-06, 6.5e-06),
    "gradstream_pp_tiny_ic3net":       (4.0e-06, 2.0e-06),
    "gradstream_tj_easy_h128":         (3.8e-05, 2.0e-06),
    "gradstream_tj_easy_ic3net":       (5.3e-06, 2.0e-06),
}


STREAM_FIXTURES = [("gradstream_pp_tiny_ic3net", "predator_prey"), ("gradstream_pp_tiny_commnet", "predator_prey"),
                   ("gradstream_tj_easy_ic3net", "traffic_junction"),
                   # hid 128 (round 6): the kernels of the BASELINE updates with cuts INSIDE the windows — 10 of 48 Predator-Prey
                   # episodes end early, 7 of 16 streams run out of phase with the windows; TJ: detach points every 3 steps
                   ("gradstream_pp_small_h128", "predator_prey"), ("gradstream_tj_easy_h128", "traffic_junction"),
                   # the other policy families in collection mode (round-5 verdict item 6), hid 64: the non-recurrent CommNet
                   # module (two passes; gated on TJ), IC (models.MLP), IRIC (models.RNN, LSTM cell), IC3Net with two passes
                   ("gradstream_pp_tiny_commnet_mlp2", "predator_prey"), ("gradstream_tj_easy_ic3net_mlp", "traffic_junction"),
                   ("gradstream_pp_tiny_ic_mlp", "predator_prey"), ("gradstream_pp_tiny_iric_lstm", "predator_prey"),
                   ("gradstream_pp_tiny_ic3net_p2", "predator_prey"),
                   # IRIC with the tanh recurrence: its rollout step is one launch too (ic3_commnet_step with h_in, round 6)
                   ("gradstream_pp_tiny_iric_rnn", "predator_prey")]


@pytest.mark.parametrize("name,env_name", STREAM_FIXTURES)
def test_collection_mode_grad_matches_reference(name, env_name):
    """train_batch in COLLECTION MODE (args.auto_reset; round-4 verdict item 5) against the reference in one hop: the
    fixture is the reference's run_batch + compute_grad (trainer.py:227-242,128-225) over, per env, the consecutive whole
    episodes that fit in nwin * max_steps slots — episodes of different lengths (Predator-Prey 'mixed' ends when every
    predator sits on the prey), a different number per env — with the actions drawn from the reference's own log-probs by
    the Philox inverse-CDF draw of ic3_policy_step.  Here: ONE auto-reset rollout of nwin windows on the one-launch kernel
    (envs restart inside the launch), the unfinished tails discarded, and the native backward with the recurrence cut at
    every episode end and at every detach point of an env's own step counter."""
    from ic3net_amd import data, trainer as trmod
    from ic3net_amd.action_utils import parse_action_args
    from ic3net_amd.comm import CommNetMLP
    fx = load(name)
    N, T, nenv, nwin, nh, seed = [int(x) for x in fx["cfg"]]
    flags = dict(ast.literal_eval(str(fx["flags"])))
    a = build_args(env_name, flags, N, T, nenv, seed)
    a.env_id_offset = 400
    a.auto_reset = True
    env = data.init(env_name, a, False)
    a.num_actions = [env.num_actions]
    a.dim_actions = env.dim_actions
    a.num_inputs = env.observation_dim
    if a.hard_attn and a.commnet:
        a.num_actions = [*a.num_actions, 2]
        a.dim_actions = env.dim_actions + 1
    if a.commnet and (a.recurrent or a.rnn_type == 'LSTM'):
        a.recurrent, a.rnn_type = True, 'LSTM'
    parse_action_args(a)
    model = str(fx["model"]) if "model" in fx.files else None
    if model:                                         # main.py:161-168: the baselines instead of CommNetMLP
        from ic3net_amd import models
        a.continuous = False
        net = (models.RNN if model == 'rnn' else models.MLP)(a, a.num_inputs)
    else:
        net = CommNetMLP(a, a.num_inputs)
    if "param_names" in fx.files:                     # hid-128 fixtures: weights from the index alone
        from policy_util import closed_form_weights
        shapes = {str(n): eval(str(sh)) for n, sh in zip(fx["param_names"], fx["param_shapes"])}
        net.load_state_dict({k: torch.from_numpy(v).float() for k, v in closed_form_weights(shapes).items()})
    else:
        net.load_state_dict({k[2:]: torch.from_numpy(fx[k]).float() for k in fx.files if k.startswith("w:")})
    net = net.cuda()
    tr = trmod.Trainer(a, net, env)
    a.batch_size = nenv * T * nwin                      # slots: exactly nwin windows
    assert tr._native_update(), "collection mode must take the native update"
    tr._records = []
    try:
        batch, stats = tr.run_batch(0)
        assert len(tr._records) == nwin and len(batch.reward) == nwin * T
        # the slots that count are the reference's batch: the same episodes, step for step
        ep_len = fx["ep_len"]
        act = torch.stack(batch.action).cpu().numpy()               # (slots, heads, E, N)
        live = torch.stack([m['live'] for m in batch.misc]).cpu().numpy()
        for e in range(nenv):
            n_ref = int(ep_len[e].sum())
            assert live[:, e].sum() == n_ref and live[:n_ref, e].all(), (e, live[:, e], ep_len[e])
            np.testing.assert_array_equal(act[:n_ref, :, e], fx["actions"][e, :n_ref], err_msg="env %d: a draw diverged" % e)
        assert stats['num_steps'] == int(fx["num_steps"]) and stats['num_episodes'] == int(fx["num_episodes"])
        np.testing.assert_allclose(stats['reward'], fx["reward"], rtol=1e-5, atol=1e-5)
        if env_name != 'predator_prey' or a.mode != 'competitive':
            assert stats['success'] == pytest.approx(float(fx["success"]))
        tr.optimizer.zero_grad()
        s = tr.compute_grad_native(batch, tr._records)
    finally:
        tr._records = None
    _check_against_reference(name + "/collection", fx, s, net)


def test_collection_mode_train_batch_runs_at_a_baseline_shape():
    """train_batch under args.auto_reset end to end (PP-hard shape, fewer envs, two windows per update): runs, updates the
    parameters, counts only whole episodes."""
    import bench
    tr, a = bench.build_trainer("pp_hard", 48, 3, 0, 0, max_steps=20)
    a.auto_reset = True
    a.__dict__.update(gamma=1.0, normalize_rewards=False, entr=0.01, value_coeff=0.01, advantages_per_action=False,
                      batch_size=2 * 48 * 20)
    before = tr.policy_net.encoder.weight.detach().clone()
    st = tr.train_batch(0)
    assert st['num_steps'] == 2 * 48 * 20 and st['num_episodes'] == 2 * 48      # a random-init PP-hard policy never ends early
    assert np.isfinite(st['action_loss']) and not torch.equal(before, tr.policy_net.encoder.weight)
    st = tr.train_batch(1)
    assert np.isfinite(st['value_loss'])


def test_train_batch_updates_parameters():
    from ic3net_amd import data, trainer as trmod
    from ic3net_amd.action_utils import parse_action_args
    from ic3net_amd.comm import CommNetMLP
    a = build_args('traffic_junction', dict(nagents=5, dim=6, vision=1, hid_size=32, ic3net=True, recurrent=True,
                                            detach_gap=10, difficulty='easy', add_rate_min=0.3, add_rate_max=0.3), 5, 20,
                   32, 2)
    a.env_id_offset = 0
    env = data.init('traffic_junction', a, False)
    a.num_actions = [env.num_actions, 2]
    a.dim_actions = 2
    a.num_inputs = env.observation_dim
    a.recurrent, a.rnn_type = True, 'LSTM'
    parse_action_args(a)
    torch.manual_seed(0)
    net = CommNetMLP(a, a.num_inputs).cuda()
    tr = trmod.Trainer(a, net, env)
    before = {k: v.detach().clone() for k, v in net.named_parameters()}
    st = tr.train_batch(0)
    assert set(st) >= {'num_steps', 'num_episodes', 'reward', 'success', 'add_rate', 'action_loss', 'value_loss',
                       'entropy', 'comm_action'}
    changed = [k for k, v in net.named_parameters() if not torch.equal(v, before[k])]
    assert 'encoder.weight' in changed and 'f_module.weight_hh' in changed and 'heads.0.weight' in changed
    assert not any(k.startswith('hidd_encoder') for k in changed)           # no grad -> untouched (quirk Q18)
    st2 = tr.train_batch(1)                                                  # a second update runs (graph freed)
    assert np.isfinite(st2['action_loss'])


@pytest.mark.parametrize("kind,rnn_type", [("mlp", "MLP"), ("rnn", "MLP"), ("rnn", "LSTM")])
def test_baseline_policies_through_trainer(kind, rnn_type):
    """IC / IRIC baselines (no communication) run through the same batched Trainer, incl. one update."""
    from ic3net_amd import data, models, trainer as trmod
    from ic3net_amd.action_utils import parse_action_args
    a = build_args('predator_prey', dict(nagents=3, dim=5, vision=1, hid_size=32, recurrent=(kind == "rnn"),
                                         rnn_type=rnn_type, detach_gap=10, mean_ratio=0.0), 3, 20, 32, 7)
    a.env_id_offset = 0
    env = data.init('predator_prey', a, False)
    a.num_actions, a.dim_actions, a.num_inputs = [env.num_actions], env.dim_actions, env.observation_dim
    a.continuous = False
    parse_action_args(a)
    torch.manual_seed(0)
    net = (models.RNN if kind == "rnn" else models.MLP)(a, a.num_inputs).cuda()
    tr = trmod.Trainer(a, net, env)
    episode, stat = tr.get_episode(0)
    assert len(episode) == 20 and episode[0].action.shape == (1, 32, 3) and episode[0].value.numel() == 96
    assert 'comm_action' not in stat and stat['num_steps'] <= 32 * 20
    before = net.affine1.weight.detach().clone()
    a.batch_size = 32 * 20
    st = tr.train_batch(0)
    assert np.isfinite(st['action_loss']) and not torch.equal(before, net.affine1.weight)


@pytest.mark.parametrize("kind", ["mlp", "lstm", "tanh"])
def test_baselines_roll_out_in_one_launch_like_the_launch_chain(kind):
    """IC / IRIC baselines (models.MLP; models.RNN with the LSTM cell; round 6: models.RNN with the tanh recurrence, through
    ic3_commnet_step's h_in) on the one-launch kernels through their stand-ins
    (ic3net_amd/models.py: the non-recurrent / recurrent CommNet module with the communication block off): the episode equals
    the launch chain's (same draws off CDF edges, values / log-probs to 1e-5), the native update's gradients too, and the
    update moves the baseline's own parameters."""
    import bench
    wl = {'mlp': 'pp_hard_ic', 'lstm': 'pp_hard_iric', 'tanh': 'pp_hard_iric_tanh'}[kind]

    def play(mega):
        tr, a = bench.build_trainer(wl, 24, 3, 0, 0, max_steps=12)
        a.mega_policy = mega
        ep, stat = tr.get_episode(0)
        act = torch.stack([t.action for t in ep]).cpu()
        val = torch.stack([t.value.reshape(24, 10) for t in ep]).cpu()
        lp = torch.stack([t.action_out[0] for t in ep]).cpu()
        rew = torch.stack([t.reward for t in ep]).cpu()
        return tr, a, act, val, lp, rew

    tr1, a1, act1, val1, lp1, rew1 = play(True)
    used = getattr(tr1.policy_net, 'mega_steps' if kind == 'lstm' else 'commnet_steps', 0)
    assert used == 12, "the one-launch path did not run"
    tr0, a0, act0, val0, lp0, rew0 = play(False)
    assert getattr(tr0.policy_net, 'commnet_steps', 0) == 0 and getattr(tr0.policy_net, 'mega_steps', 0) == 0
    # step 0 sees the same state: its outputs agree to rounding; the episodes then agree as long as no draw sits on a CDF edge
    torch.testing.assert_close(lp1[0], lp0[0], atol=1e-5, rtol=0)
    torch.testing.assert_close(val1[0], val0[0], atol=1e-5, rtol=0)
    assert torch.equal(act1, act0) and torch.equal(rew1, rew0)
    torch.testing.assert_close(lp1, lp0, atol=2e-5, rtol=0)
    torch.testing.assert_close(val1, val0, atol=2e-5, rtol=0)
    grads = []
    for tr, a in ((tr1, a1), (tr0, a0)):
        a.__dict__.update(gamma=1.0, normalize_rewards=False, entr=0.01, value_coeff=0.01, advantages_per_action=False,
                          batch_size=24 * 12, lrate=0.0)
        before = tr.policy_net.affine1.weight.detach().clone()
        st = tr.train_batch(1)
        assert np.isfinite(st['action_loss'])
        grads.append({k: p.grad.clone() for k, p in tr.policy_net.named_parameters() if p.grad is not None})
    for k in grads[0]:
        g1, g0 = grads[0][k], grads[1][k]
        assert float((g1 - g0).abs().max()) <= 2e-4 * max(1e-6, float(g0.abs().max())), k


def test_enemy_comm_through_trainer():
    """main.py:125-130: with --enemy_comm the policy sees nagents = nfriendly + nenemies; stats gain
    enemy_reward / enemy_comm (trainer.py:73-75,87-88)."""
    from ic3net_amd import data, trainer as trmod
    from ic3net_amd.action_utils import parse_action_args
    from ic3net_amd.comm import CommNetMLP
    a = build_args('predator_prey', dict(nagents=3, dim=5, vision=1, hid_size=32, ic3net=True, recurrent=True,
                                         detach_gap=10, enemy_comm=True), 3, 20, 16, 9)
    a.env_id_offset = 0
    a.nagents += a.nenemies               # main.py:126-128
    env = data.init('predator_prey', a, False)
    a.num_actions, a.dim_actions, a.num_inputs = [env.num_actions, 2], 2, env.observation_dim
    a.recurrent, a.rnn_type = True, 'LSTM'
    parse_action_args(a)
    torch.manual_seed(0)
    net = CommNetMLP(a, a.num_inputs).cuda()
    tr = trmod.Trainer(a, net, env)
    episode, stat = tr.get_episode(0)
    assert episode[0].action.shape == (2, 16, 4) and episode[0].reward.shape == (16, 4)
    assert stat['reward'].shape == (3,) and stat['enemy_reward'].shape == (1,) and stat['enemy_comm'].shape == (1,)
    # the prey earns 0.05 per live step without a predator on it, 0 otherwise (predator_prey_env.py:276-281)
    assert 0.0 <= stat['enemy_reward'][0] <= 0.05 * stat['num_steps'] + 1e-6
    a.batch_size = 16 * 20
    st = tr.train_batch(0)
    assert np.isfinite(st['action_loss'])


@pytest.mark.parametrize("workload,E", [("pp_hard", 96), ("tj_hard", 48), ("tj_medium", 64)])
def test_free_running_graph_rollout_replays_through_oracle(workload, E):
    """BASELINE shapes (PP-hard / TJ-hard / TJ-medium policies and envs, fewer envs), hipGraph mode: the third episode is
    a pure graph replay.  Its sampled env actions are replayed through the CPU oracle envs from the same reset: rewards,
    alive masks, done flags and the episode statistics must agree — and the sampled actions themselves must be the
    oracle's inverse-CDF draws from the recorded log-probs."""
    import bench
    import oracle
    from oracle import philox
    tr, a = bench.build_trainer(workload, E, 3, 1000, 0)
    a.hip_graph = True
    for ep in range(3):
        episode, stat = tr.get_episode(ep)
    env_name, f = bench.WORKLOADS[workload]
    N, T = a.nagents, a.max_steps
    act = torch.stack([t.action for t in episode]).cpu().numpy()            # (T, heads, E, N)
    rew = torch.stack([t.reward for t in episode]).cpu().numpy()
    alive = torch.stack([t.misc['alive_mask'] for t in episode]).cpu().numpy()
    live = torch.stack([t.misc['live'] for t in episode]).cpu().numpy()
    logp0 = torch.stack([t.action_out[0] for t in episode]).cpu().numpy()   # (T, E, N, A)
    total_reward = np.zeros(N)
    steps = 0
    for e in range(E):
        gid = 1000 + e
        if env_name == 'predator_prey':
            o = oracle.PPOracle(N, f['dim'], f['vision'], f['mode'], seed=3, env_gid=gid)
        else:
            o = oracle.TJOracle(N, f['dim'], f['vision'], f['difficulty'], add_rate_min=f['add_rate_min'],
                                add_rate_max=f['add_rate_max'], seed=3, env_gid=gid)
        for ep in range(3):                                                 # episode counters advance per reset
            o.reset() if env_name == 'predator_prey' else o.reset(ep)
        over = False
        for t in range(T):
            if over:
                assert live[t, e] == 0 and not rew[t, e].any()
                continue
            assert live[t, e] == 1
            if e < 8:   # sampling: the device draw equals the oracle's inverse-CDF on the same stream position
                for n in range(N):
                    want = oracle.sample_one(logp0[t, e, n], philox.x24(3, gid, philox.DOMAIN_SAMPLE, 2, t, 0 * N + n))
                    if want != act[t, 0, e, n]:
                        u = philox.x24(3, gid, philox.DOMAIN_SAMPLE, 2, t, n) / 2.0 ** 24
                        assert np.abs(np.cumsum(np.exp(logp0[t, e, n].astype(np.float64))) - u).min() < 1e-6
            oo, orew, od = o.step(act[t, 0, e])
            np.testing.assert_array_equal(rew[t, e], orew.astype(np.float32))
            if env_name == 'traffic_junction':
                np.testing.assert_array_equal(alive[t, e], o.alive.astype(np.float32))
            total_reward += orew
            steps += 1
            over = bool(od)
    assert stat['num_steps'] == steps
    np.testing.assert_allclose(stat['reward'], total_reward, rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("workload,E", [("pp_hard", 7), ("tj_medium", 5)])
def test_store_states_holds_state_and_next_state(workload, E):
    """args.store_states (trainer.py:49,67,104: Transition.state is the observation the policy saw, next_state the one
    env.step returned).  ic3_policy_step writes the rows of the state it ACTS ON, so a storing rollout must not take
    them for next_state: checked against the oracle env on the rollout's own actions, on the default (one-launch) path
    and on the launch chain — and a later non-storing, autograd rollout after a one-launch episode still starts from a
    freshly assembled reset observation (the sticky `_mega_last` flag of round 2)."""
    import bench
    import oracle
    T = 6
    for mega in (True, False):
        tr, a = bench.build_trainer(workload, E, 2, 50, 0)
        a.max_steps, a.store_states, a.mega_policy = T, True, mega
        episode, _ = tr.get_episode(0)
        assert (getattr(tr.policy_net, 'mega_steps', 0) > 0) == mega
        for e in range(E):
            if a.env_name == 'predator_prey':
                o = oracle.PPOracle(a.nagents, a.dim, a.vision, a.mode, seed=2, env_gid=50 + e)
                obs = o.reset()
            else:
                o = oracle.TJOracle(a.nagents, a.dim, a.vision, a.difficulty, a.add_rate_min, a.add_rate_max,
                                    a.curr_start, a.curr_end, seed=2, env_gid=50 + e, vocab_type=a.vocab_type)
                obs = o.reset(0)
            for t in range(T):
                tn = episode[t]
                np.testing.assert_array_equal(tn.state[e].cpu().numpy(), obs, err_msg="state e=%d t=%d mega=%s" % (e, t, mega))
                obs, _, _ = o.step(tn.action[0, e].cpu().numpy())
                np.testing.assert_array_equal(tn.next_state[e].cpu().numpy(), obs,
                                              err_msg="next_state e=%d t=%d mega=%s" % (e, t, mega))
    # a one-launch episode followed by an episode on the autograd path: its first forward reads the env's obs buffer
    tr, a = bench.build_trainer(workload, E, 2, 50, 0)
    a.max_steps = T
    tr.get_episode(0)
    assert getattr(tr.policy_net, 'mega_steps', 0) == T
    a.rollout_grad, a.sparse_encoder_grad = True, False
    tr.policy_net.obs_env = None
    tr.begin_episode(1)
    got = tr._state.clone()                                          # (the env's own buffer: copy before re-assembling it)
    fresh = tr.env.env.observe().clone()
    assert torch.equal(got.reshape(fresh.shape), fresh)              # reset() assembled the observation of the new state


@pytest.mark.parametrize("workload,E", [("pp_hard", 9), ("tj_medium", 7), ("pp_hard_p2", 9), ("tj_medium_p3share", 7),
                                        ("tj_medium_commnet_mlp", 7), ("pp_hard_mlp", 9),
                                        ("pp_hard_h100", 9), ("tj_medium_h48_p2", 7), ("tj_medium_commnet_mlp_h100", 7),
                                        ("pp_hard", 1024)])     # 10 240 rows: the weight gradients as batched row-block products
def test_native_update_on_the_one_launch_rollout_matches_autograd(workload, E):
    """train_batch's default path at a BASELINE shape: the rollout is the one-launch kernel (ic3_policy_step, hid 128) and
    the gradients come from ic3net_amd.bptt — compared with loss.backward() through the autograd rollout replaying the
    same actions (trainer.py:128-225 both ways; detach_gap cuts inside the episode, entropy term on).
    `_p2` / `_p3share`: comm_passes 2 (own C per pass) / 3 (shared C) — one launch per pass, multi-pass backward.
    `tj_medium_commnet_mlp` / `pp_hard_mlp`: the NON-recurrent module (round 4): rollout = ic3_commnet_step, gradients =
    bptt._backward_episode_commnet."""
    import bench
    from ic3net_amd import trainer as trmod
    bench.WORKLOADS.setdefault("pp_hard_p2", ("predator_prey", dict(bench.WORKLOADS["pp_hard"][1], comm_passes=2)))
    bench.WORKLOADS.setdefault("tj_medium_p3share", ("traffic_junction", dict(bench.WORKLOADS["tj_medium"][1], comm_passes=3,
                                                                               share_weights=True)))
    bench.WORKLOADS.setdefault("pp_hard_mlp", ("predator_prey", dict(bench.WORKLOADS["pp_hard"][1], recurrent=False)))
    # hidden sizes the kernels are not built for (main.py:34: any int): the policy's zero-padded twin at 128 / 64 runs the
    # one-launch rollout AND the explicit backward; its gradients, cut back, against autograd through the policy itself
    bench.WORKLOADS.setdefault("pp_hard_h100", ("predator_prey", dict(bench.WORKLOADS["pp_hard"][1], hid_size=100)))
    bench.WORKLOADS.setdefault("tj_medium_h48_p2", ("traffic_junction", dict(bench.WORKLOADS["tj_medium"][1], hid_size=48,
                                                                              comm_passes=2)))
    bench.WORKLOADS.setdefault("tj_medium_commnet_mlp_h100", ("traffic_junction",
                                                              dict(bench.WORKLOADS["tj_medium_commnet_mlp"][1], hid_size=100)))
    T = 12
    extra = dict(gamma=0.95, normalize_rewards=True, entr=0.01, value_coeff=0.01, advantages_per_action=False,
                 batch_size=E * T, detach_gap=5)
    tr, a = bench.build_trainer(workload, E, 3, 70, 0)
    a.__dict__.update(extra)
    a.max_steps = T
    assert tr._native_update()
    tr._records = []
    batch, _ = tr.run_batch(0)
    assert getattr(tr.policy_net, 'mega_steps' if a.recurrent else 'commnet_steps', 0) == T, "the one-launch rollout did not run"
    tr.optimizer.zero_grad()
    s1 = tr.compute_grad_native(batch, tr._records)
    tr._records = None
    g1 = {k: p.grad.clone() for k, p in tr.policy_net.named_parameters() if p.grad is not None}
    tape = torch.stack(batch.action).clone()                       # (T, heads, E, N)

    tr2, a2 = bench.build_trainer(workload, E, 3, 70, 0)           # same seed: same weights, same env streams
    a2.__dict__.update(extra)
    a2.max_steps = T

    def taped(args, action_out, clock, out=None):
        out.copy_(tape[clock.t])
        return out
    orig = trmod.select_action
    trmod.select_action = taped
    try:
        a2.rollout_grad = True
        batch2, _ = tr2.run_batch(0)
        tr2.optimizer.zero_grad()
        s2 = tr2.compute_grad(batch2)
    finally:
        trmod.select_action = orig
    for k in ("action_loss", "value_loss", "entropy"):
        np.testing.assert_allclose(s1[k], s2[k], rtol=2e-4, atol=1e-4, err_msg=k)
    g2 = {k: p.grad for k, p in tr2.policy_net.named_parameters() if p.grad is not None}
    assert set(g1) == set(g2)
    for k in g1:
        scale = max(float(g2[k].abs().max()), 1e-6)
        np.testing.assert_allclose(g1[k].cpu().numpy() / scale, g2[k].cpu().numpy() / scale, rtol=0, atol=5e-4, err_msg=k)


@pytest.mark.parametrize("env_name,flags", [
    ("predator_prey", dict(nagents=5, dim=8, vision=1, hid_size=64, commnet=True, comm_passes=2)),
    ("traffic_junction", dict(nagents=6, dim=6, vision=1, hid_size=128, ic3net=True, add_rate_min=0.3, add_rate_max=0.3)),
])
def test_nonrecurrent_commnet_rollout_runs_on_the_one_launch_module(env_name, flags):
    """recurrent = False at hid 64 / 128: one rollout iteration is ONE launch (ic3_commnet_step, round 4; round 3: the policy
    in one ic3_commnet_forward launch, five launches per step); same transitions as the generic module path (args.fused_policy = False) on a twin env — log-probs and
    values within fp32 rounding, hence (away from CDF edges) the same draws, rewards and masks; and one update runs."""
    from ic3net_amd import data, trainer as trmod
    from ic3net_amd.action_utils import parse_action_args
    from ic3net_amd.comm import CommNetMLP
    import copy
    T, E = 12, 24
    trs = []
    for fused in (True, False):
        a = build_args(env_name, dict(flags), flags['nagents'], T, E, 5)
        a.env_id_offset = 0
        env = data.init(env_name, a, False)
        a.num_actions, a.dim_actions, a.num_inputs = [env.num_actions], env.dim_actions, env.observation_dim
        if a.hard_attn and a.commnet:
            a.num_actions, a.dim_actions = [env.num_actions, 2], env.dim_actions + 1
        a.continuous = False
        a.fused_policy = fused
        parse_action_args(a)
        torch.manual_seed(3)
        net = CommNetMLP(a, a.num_inputs).cuda().float()
        trs.append((trmod.Trainer(a, net, env), net, a))
    (trA, netA, aA), (trB, netB, aB) = trs
    epA, statA = trA.get_episode(0)
    epB, statB = trB.get_episode(0)
    # ONE launch per step: ic3_commnet_step (sparse encoder, passes, heads, draws, env.step, obs rows) — and nothing else
    assert getattr(netA, 'commnet_steps', 0) == T and getattr(netA, 'commnet_forwards', 0) == 0
    assert getattr(netB, 'commnet_steps', 0) == 0 and getattr(netB, 'commnet_forwards', 0) == 0
    assert len(epA) == len(epB) == T
    same = 0
    for ta, tb in zip(epA, epB):
        for k in range(len(ta.action_out)):
            assert float((ta.action_out[k] - tb.action_out[k]).abs().max()) < 2e-5
        assert float((ta.value - tb.value).abs().max()) < 2e-5
        if not torch.equal(ta.action, tb.action):
            break                                  # a draw on a CDF edge: the trajectories part ways (legitimately)
        assert torch.equal(ta.reward, tb.reward) and torch.equal(ta.episode_mask, tb.episode_mask)
        same += 1
    assert same >= 3
    aA.batch_size = E * T
    before = netA.encoder.weight.detach().clone()
    st = trA.train_batch(0)
    assert np.isfinite(st['action_loss']) and not torch.equal(before, netA.encoder.weight)


def test_native_update_records_the_recurrent_state_in_place():
    """train_batch on the one-launch path: the step launch reads (h, c) from slot t of the episode record and writes slot
    t + 1 (ic3_env_set_hidden_out) — the record equals the one a copying rollout makes, bit for bit, and so do the grads."""
    import bench
    from ic3net_amd import bptt
    grads = []
    for inplace in (True, False):
        tr, a = bench.build_trainer('pp_hard', 12, 3, 0, 0)
        a.max_steps, a.batch_size = 12, 12 * 12
        a.entr, a.value_coeff, a.gamma, a.normalize_rewards, a.advantages_per_action = 0.01, 0.01, 1.0, False, False
        if not inplace:
            tr._rec_inplace = lambda: False
        assert tr._native_update()
        tr._records = []
        batch, stats = tr.run_batch(0)
        rec = tr._records[0]
        assert rec.n == 12 and (rec.h_last.data_ptr() == rec.hs[12].data_ptr()) == inplace
        tr.optimizer.zero_grad()
        tr.compute_grad_native(batch, tr._records)
        tr._records = None
        grads.append(({k: p.grad.clone() for k, p in tr.policy_net.named_parameters() if p.grad is not None},
                      rec.hs[:12].clone(), rec.cs[:12].clone(), rec.h_last.clone()))
    for x, y in zip(grads[0][1:], grads[1][1:]):
        assert torch.equal(x, y)
    for k in grads[0][0]:      # (the heads' gradient is one pass over the episode in place, two passes otherwise: summation order)
        g0, g1 = grads[0][0][k], grads[1][0][k]
        assert float((g0 - g1).abs().max()) <= 1e-5 * max(1.0, float(g1.abs().max())), k


@pytest.mark.parametrize("wl,collect,hid", [("pp_hard", False, 128), ("tj_medium", False, 128), ("pp_hard", True, 128),
                                            ("pp_hard_iric", False, 128), ("tj_hard", False, 64), ("pp_easy", False, 32)])
def test_native_update_reads_the_gates_the_rollout_recorded(wl, collect, hid):
    """Round 5: every step launch of a recorded rollout stores its cell's activated gates in the episode record
    (ic3_env_set_record_out) and the backward reads them (ic3_lstm_gates_backward_given) instead of running the gate product
    again: same rollout bit for bit, the recorded gates reproduce the recorded next state, and the gradients equal the
    recomputing backward's at 1e-5 (args.record_gates=False) — episodes and collection mode; the IRIC baseline (models.RNN
    with the LSTM cell) takes the same path through its kernel stand-in (bptt.standin_for_backward)."""
    import bench
    out = []
    for record in (True, False):
        tr, a = bench.build_trainer(wl, 12, 3, 0, 0, hid_size=hid)    # (hid 32: the zero-padded twin at 64 keeps the record)
        a.max_steps, a.batch_size = 10, 12 * 10
        a.entr, a.value_coeff, a.gamma, a.normalize_rewards, a.advantages_per_action = 0.01, 0.01, 1.0, False, False
        a.record_gates = record
        a.auto_reset = collect
        assert tr._native_update()
        tr._records = []
        batch, stats = tr.run_batch(0)
        recs = tr._records
        rec = recs[0]
        assert (rec.gates is not None and rec.gates_n == rec.n) == record
        if record:
            H = rec.hs.shape[2]
            g = rec.gates[:rec.n - 1].double()
            c1 = g[..., H:2 * H] * rec.cs[:rec.n - 1].double() + g[..., :H] * g[..., 2 * H:3 * H]
            if not collect:        # (collection mode: an env that starts an episode at slot t had zero state, not the record's)
                assert float((c1 - rec.cs[1:rec.n].double()).abs().max()) <= 1e-6
                assert float((g[..., 3 * H:] * torch.tanh(c1) - rec.hs[1:rec.n].double()).abs().max()) <= 2e-6
                # the recorded inp rows: the gates follow from [inp | h] . [W_ih | W_hh]^T + b
                knet = tr._kernel_net()
                fm = knet.f_module if hasattr(knet, 'f_module') else knet.lstm_unit     # (IRIC: models.RNN's own cell)
                W = torch.cat([fm.weight_ih, fm.weight_hh], 1).detach().double()
                xh = torch.cat([rec.xh[:rec.n - 1, :, :H], rec.hs[:rec.n - 1]], 2).double()
                pre = xh @ W.t() + (fm.bias_ih + fm.bias_hh).detach().double()
                assert float((torch.sigmoid(pre[..., :H]) - g[..., :H]).abs().max()) <= 2e-6
                assert float((torch.tanh(pre[..., 2 * H:3 * H]) - g[..., 2 * H:3 * H]).abs().max()) <= 2e-6
        hs_rolled, cs_rolled = rec.hs[:rec.n].clone(), rec.cs[:rec.n].clone()    # (the recomputing backward of collection mode
        tr.optimizer.zero_grad()                                                  #  zeroes the rows of fresh envs in the record)
        tr.compute_grad_native(batch, recs)
        tr._records = None
        out.append(({k: p.grad.clone() for k, p in tr.policy_net.named_parameters() if p.grad is not None}, hs_rolled, cs_rolled))
    assert torch.equal(out[0][1], out[1][1]) and torch.equal(out[0][2], out[1][2])
    for k in out[0][0]:
        g0, g1 = out[0][0][k], out[1][0][k]
        assert float((g0 - g1).abs().max()) <= 1e-5 * max(1.0, float(g1.abs().max())), k


@pytest.mark.parametrize("wl,collect,hid", [("pp_hard", False, 128), ("tj_hard", False, 128), ("pp_hard", True, 128),
                                            ("tj_medium", True, 64), ("pp_hard_iric", False, 128), ("pp_easy", False, 32)])
def test_backward_window_as_one_host_call_equals_the_per_step_loop(wl, collect, hid):
    """Round 6: ic3_bptt_backward (the whole window in one host call: gate launch in place with the heads' share folded in,
    ic3_comm_backward, encoder stage 1; the weight gradient of the window in one launch) against the per-step Python loop of
    round 5 on the same recorded gates (args.bptt_native_loop=False: library products, two masked means per step) — the same
    rollout bit for bit, gradients at 1e-5 of each tensor's scale (fp32 sums in another order; measured <= 3.2e-6)."""
    import bench
    out = []
    for loop in (True, False):
        tr, a = bench.build_trainer(wl, 12, 3, 0, 0, hid_size=hid)
        a.max_steps, a.batch_size = 10, 12 * 10 * (3 if collect else 1)
        a.detach_gap = 4
        a.entr, a.value_coeff, a.gamma, a.normalize_rewards, a.advantages_per_action = 0.01, 0.01, 0.9, False, False
        a.bptt_native_loop = loop
        a.auto_reset = collect
        assert tr._native_update()
        tr._records = []
        batch, stats = tr.run_batch(0)
        recs = tr._records
        assert recs[0].gates is not None and recs[0].gates_n == recs[0].n
        hs_rolled = torch.cat([r.hs[:r.n] for r in recs]).clone()
        tr.optimizer.zero_grad()
        tr.compute_grad_native(batch, recs)
        tr._records = None
        out.append(({k: p.grad.clone() for k, p in tr.policy_net.named_parameters() if p.grad is not None}, hs_rolled))
    assert torch.equal(out[0][1], out[1][1])
    assert out[0][0].keys() == out[1][0].keys()
    for k in out[0][0]:
        g0, g1 = out[0][0][k], out[1][0][k]
        assert float((g0 - g1).abs().max()) <= 1e-5 * max(1e-3, float(g1.abs().max())), (k, float((g0 - g1).abs().max()), float(g1.abs().max()))


@pytest.mark.parametrize("wl,collect,hid", [("pp_hard", False, 128), ("tj_medium", True, 64), ("pp_easy", True, 128),
                                            ("tj_hard", False, 128)])
def test_backward_as_two_chains_and_with_the_encoder_window_form(wl, collect, hid):
    """Round 6, later: ic3_bptt.two_chains (envs [0, E1) and [E1, E) launched on two streams) and ic3_bptt.dxh_step (the per-step
    input gradients kept in a ring, the encoder's first stage ONCE over the window: ic3_env_encode_backward_window) against the
    single chain with the encoder's per-step accumulate, on 192 envs (E1 = 64): the chains touch disjoint rows, so only the order
    of the partial sums changes — 1e-5 of each gradient's scale (measured <= 2e-6)."""
    import bench
    from ic3net_amd import ops
    assert ops.first_chain_envs(192, 10) == 64 and ops.first_chain_envs(100, 10) == 100
    out = []
    for two, win in ((False, False), (True, True), (False, True)):
        tr, a = bench.build_trainer(wl, 192, 3, 0, 0, hid_size=hid)
        a.max_steps, a.batch_size = 9, 192 * 9 * (2 if collect else 1)
        a.detach_gap = 4
        a.entr, a.value_coeff, a.gamma, a.normalize_rewards, a.advantages_per_action = 0.01, 0.01, 0.9, False, False
        a.bptt_two_chains, a.enc_window = two, win
        a.auto_reset = collect
        assert tr._native_update()
        tr._records = []
        batch, stats = tr.run_batch(0)
        recs = tr._records
        tr.optimizer.zero_grad()
        tr.compute_grad_native(batch, recs)
        tr._records = None
        torch.cuda.synchronize()
        out.append({k: p.grad.clone() for k, p in tr.policy_net.named_parameters() if p.grad is not None})
    for o in out[1:]:
        assert o.keys() == out[0].keys()
        for k in o:
            g0, g1 = o[k], out[0][k]
            err = float((g0 - g1).abs().max())
            assert err <= 1e-5 * max(1e-3, float(g1.abs().max())), (k, err, float(g1.abs().max()))


@pytest.mark.parametrize("wl,norm,entr", [("pp_hard", False, 0.0), ("tj_medium", True, 0.01), ("pp_hard_ic", False, 0.01)])
def test_losses_and_their_gradients_in_one_launch_equal_the_tensor_program(wl, norm, entr):
    """ic3_loss_gradients (round 6: compute_grad's losses, trainer.py:173-218, and dL/d[logits | value] of every transition in ONE
    launch on the rows the step launches wrote into the episode record) against the tensor program it replaces
    (args.fused_loss = False) on the same batch: the three losses to 1e-6 relative, d_out to 1e-6 of its scale."""
    import bench
    from ic3net_amd import bptt
    tr, a = bench.build_trainer(wl, 40, 3, 0, 0, max_steps=12)
    a.__dict__.update(gamma=0.95, normalize_rewards=norm, entr=entr, value_coeff=0.02, advantages_per_action=False,
                      batch_size=40 * 12)
    assert tr._native_update()
    tr._records = []
    try:
        batch, stats = tr.run_batch(0)
        recs = tr._records
        assert recs[0].out is not None and recs[0].out_n == recs[0].n == 12
        with torch.no_grad():
            s1, d1 = bptt.loss_gradients(a, batch, recs)
            a.fused_loss = False
            s0, d0 = bptt.loss_gradients(a, batch, recs)
    finally:
        tr._records = None
    for k in ("action_loss", "value_loss", "entropy"):
        assert abs(s1[k] - s0[k]) <= 1e-6 * max(1.0, abs(s0[k])), (k, s1[k], s0[k])
    assert d1.shape == d0.shape
    assert float((d1 - d0).abs().max()) <= 1e-6 * max(1.0, float(d0.abs().max()))


def test_native_update_is_not_taken_where_it_does_not_apply():
    """Round-3 advisor findings: with args.auto_reset the recorded (h, c) / masks of a restarted env belong to the previous
    episode — since round 5 the explicit backward CUTS there (collection mode, test_collection_mode_grad_matches_reference)
    for every policy family whose rollout step is ONE launch (round 6; the launch restarts finished envs itself); a tanh-recurrence
    RNN baseline at a hidden size the kernels do not take has no such launch and train_batch must not take the native path (the
    rollout then raises its explicit NotImplementedError); and hid
    sizes ic3_lstm_cell_backward does not take (H / 4 not a power of two <= 64) keep the autograd update instead of failing
    inside it."""
    import bench
    from ic3net_amd import bptt
    tr, a = bench.build_trainer('pp_hard', 8, 1, 0, 0)
    assert tr._native_update()
    a.auto_reset = True
    assert tr._native_update()
    trp, ap = bench.build_trainer('pp_hard', 8, 1, 0, 0, comm_passes=2)
    assert trp._native_update()
    ap.auto_reset = True
    assert trp._native_update()          # (round 6: every family with a one-launch rollout step — test_collection_mode_grad_...)
    trk, ak = bench.build_trainer('pp_hard_iric', 8, 1, 0, 0, rnn_type='MLP')     # the tanh-recurrence RNN: one launch too (round 6, later)
    ak.auto_reset = True
    assert trk._native_update()
    trr, ar = bench.build_trainer('pp_hard_iric', 8, 1, 0, 0, rnn_type='MLP', hid_size=96)   # ... at the kernels' own sizes only
    assert trr._native_update()
    ar.auto_reset = True
    assert not trr._native_update()
    ar.batch_size = 8 * ar.max_steps
    with pytest.raises(NotImplementedError):
        trr.train_batch(0)
    for hid, ok in ((96, False), (100, False), (128, True), (32, True)):
        tr2, a2 = bench.build_trainer('pp_easy', 8, 1, 0, 0, hid_size=hid)
        assert bptt.supported(a2, tr2.policy_net, tr2.env.env) == ok, hid
        # (round 4, later: such a policy runs as its zero-padded twin at 128 / 64 — comm.CommNetMLP._twin — and the twin's
        #  size IS one the backward takes; without the twin the update stays on autograd)
        assert tr2._native_update()
        a2.pad_hidden = False
        assert tr2._native_update() == ok, hid
    for pad in (False, True):                                             # ... and the update itself runs either way
        tr3, a3 = bench.build_trainer('pp_easy', 8, 1, 0, 0, hid_size=96)
        a3.pad_hidden = pad
        a3.batch_size = 8 * a3.max_steps
        a3.__dict__.update(gamma=1.0, normalize_rewards=False, entr=0, value_coeff=0.01, advantages_per_action=False)
        st = tr3.train_batch(0)
        assert np.isfinite(st['action_loss'])
        assert (getattr(tr3.policy_net, 'mega_steps', 0) > 0) == pad     # the one-launch rollout ran for the twin only


@pytest.mark.parametrize("kind,rnn_type,env_name,hid", [("mlp", "MLP", "predator_prey", 32), ("rnn", "MLP", "traffic_junction", 32),
                                                        ("rnn", "LSTM", "predator_prey", 32),
                                                        # hid 64: the LSTM baseline's backward runs on its kernel stand-in (fused launches)
                                                        ("rnn", "LSTM", "predator_prey", 64), ("rnn", "LSTM", "traffic_junction", 64)])
def test_native_update_of_the_baselines_matches_autograd(kind, rnn_type, env_name, hid):
    """IC / IRIC baselines (models.py:8-97): the graph-free update (bptt._backward_episode_baseline, round 4) against
    loss.backward() through the autograd rollout replaying the same actions — detach_gap cuts inside the episode, entropy
    term and reward normalisation on."""
    from ic3net_amd import bptt, data, models, trainer as trmod
    from ic3net_amd.action_utils import parse_action_args
    T, E = 12, 9
    flags = dict(nagents=3, dim=5, vision=1, hid_size=hid, recurrent=(kind == "rnn"), rnn_type=rnn_type, detach_gap=5,
                 mean_ratio=0.5, gamma=0.95, normalize_rewards=True, entr=0.01, value_coeff=0.01)
    if env_name == "traffic_junction":
        flags.update(nagents=5, dim=6, difficulty='easy', add_rate_min=0.4, add_rate_max=0.4)

    def make():
        a = build_args(env_name, dict(flags), flags['nagents'], T, E, 7)
        a.env_id_offset = 0
        env = data.init(env_name, a, False)
        a.num_actions, a.dim_actions, a.num_inputs = [env.num_actions], env.dim_actions, env.observation_dim
        a.continuous = False
        a.batch_size = E * T
        parse_action_args(a)
        torch.manual_seed(0)
        net = (models.RNN if kind == "rnn" else models.MLP)(a, a.num_inputs).cuda()
        return trmod.Trainer(a, net, env), a
    tr, a = make()
    assert bptt.supported(a, tr.policy_net, tr.env.env) and tr._native_update()
    tr._records = []
    batch, _ = tr.run_batch(0)
    tr.optimizer.zero_grad()
    s1 = tr.compute_grad_native(batch, tr._records)
    tr._records = None
    g1 = {k: p.grad.clone() for k, p in tr.policy_net.named_parameters() if p.grad is not None}
    tape = torch.stack(batch.action).clone()
    tr2, a2 = make()

    def taped(args, action_out, clock, out=None):
        out.copy_(tape[clock.t])
        return out
    orig = trmod.select_action
    trmod.select_action = taped
    try:
        a2.rollout_grad = True
        batch2, _ = tr2.run_batch(0)
        tr2.optimizer.zero_grad()
        s2 = tr2.compute_grad(batch2)
    finally:
        trmod.select_action = orig
    for k in ("action_loss", "value_loss", "entropy"):
        np.testing.assert_allclose(s1[k], s2[k], rtol=2e-4, atol=1e-4, err_msg=k)
    g2 = {k: p.grad for k, p in tr2.policy_net.named_parameters() if p.grad is not None}
    assert set(g1) == set(g2)
    for k in g1:
        scale = max(float(g2[k].abs().max()), 1e-6)
        np.testing.assert_allclose(g1[k].cpu().numpy() / scale, g2[k].cpu().numpy() / scale, rtol=0, atol=5e-4, err_msg=k)


def test_hip_graph_replay_of_the_tanh_rnn_baseline():
    """Round-5 advisor finding: models.RNN with the tanh recurrence (rnn_type 'MLP') under args.hip_graph — step 0's captured
    launches read the initial hidden state on every replay, so it lives in the Trainer's static buffers (zeroed in front of each
    replay) instead of being a fresh torch.zeros per episode; episodes are identical to the eager run in both graph modes, also
    with allocator traffic between the episodes."""
    import bench
    out = {}
    for mode in (False, True, 'step'):
        tr, a = bench.build_trainer('pp_hard_iric', 16, 5, 0, 0, rnn_type='MLP', max_steps=8, hid_size=64)
        a.hip_graph = mode
        eps = []
        junk = []
        for ep in range(5):
            episode, stat = tr.get_episode(ep)
            eps.append(([t.action.clone() for t in episode], [t.reward.clone() for t in episode], [t.value.clone() for t in episode]))
            junk.append(torch.full((16 * 10 * 64 * (ep + 1),), float('nan'), device='cuda'))   # (a freed block would be reused here)
            if ep % 2:
                junk.clear()
        out[mode] = eps
        if mode:
            assert tr._graphs, "the rollout did not go through a graph"
    for mode in (True, 'step'):
        for ep, (e, g) in enumerate(zip(out[False], out[mode])):
            for i in range(3):
                for x, y in zip(e[i], g[i]):
                    assert torch.equal(x, y), (mode, ep, i)


@pytest.mark.parametrize("wl", ["pp_easy", "tj_medium"])
def test_collection_mode_run_batch_is_the_same_under_hip_graph(wl):
    """Round-5 advisor finding: run_batch in collection mode (args.auto_reset) with args.hip_graph on and no records (a direct
    run_batch call): the windows step eagerly on buffers of their own (Trainer._use_graph), so the batch — actions, rewards,
    masks, statistics — equals the one collected with hip_graph off, window after window."""
    import bench
    res = []
    for graph in (False, True):
        tr, a = bench.build_trainer(wl, 24, 7, 0, 0, max_steps=10, hid_size=64)
        a.auto_reset = True
        a.hip_graph = graph
        a.batch_size = 3 * 24 * 10                      # three windows
        tr.get_episode(0)                               # (graph mode: an episode played, so that captures would be taken)
        tr.get_episode(0)
        batch, stats = tr.run_batch(0)
        res.append(([x.clone() for x in batch.action], [x.clone() for x in batch.reward], [m['live'].clone() for m in batch.misc],
                    [m['alive_mask'].clone() for m in batch.misc], [x.clone() for x in batch.episode_mask],
                    {k: np.asarray(v).copy() for k, v in stats.items()}))
    for i in range(5):
        assert len(res[0][i]) == len(res[1][i]) == 30
        for x, y in zip(res[0][i], res[1][i]):
            assert torch.equal(x, y), i
    for k in res[0][5]:
        np.testing.assert_array_equal(res[0][5][k], res[1][5][k], err_msg=k)
