"""GPU: auto-reset collection (ic3_env_set_auto_reset / args.auto_reset) — an env whose episode ends restarts inside
the step launch and keeps producing real transitions, like the reference's `while ...: get_episode()` loop
(trainer.py:107-108,227-242).  The per-env stream, cut at `done`, must equal consecutive episodes of the CPU oracle
env under the same actions, and the policy outputs must equal the fp64 numpy policy restarted (h = c = 0, gate 0, no
alive mask) at every episode start."""
import argparse

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from test_env_parity_gpu import make_pp, make_tj  # noqa: E402


def test_pp_env_restarts_inside_step_like_consecutive_oracle_episodes():
    import oracle
    E, N, dim, v, cap = 40, 2, 3, 1, 5
    env = make_pp(N, dim, v, "mixed", E, seed=21, offset=90)
    env.set_auto_reset(cap)
    env.reset()
    orcs = [oracle.PPOracle(N, dim, v, "mixed", seed=21, env_gid=90 + e) for e in range(E)]
    for o in orcs:
        o.reset()
    tcount = np.zeros(E, int)
    rs = np.random.RandomState(3)
    ends = succ = steps = 0
    for t in range(23):
        act = rs.randint(0, 5, size=(E, N)).astype(np.int32)
        obs, rew, done, _ = env.step(torch.from_numpy(act).cuda())
        obs, rew, done = obs.cpu().numpy(), rew.cpu().numpy(), done.cpu().numpy()
        st = env.get_state()
        for e, o in enumerate(orcs):
            oo, orew, od = o.step(act[e])
            tcount[e] += 1
            np.testing.assert_array_equal(rew[e], orew.astype(np.float32))
            end = bool(od) or tcount[e] == cap
            assert int(done[e]) == int(end), (t, e)
            if end:
                ends += 1
                succ += int(o.success.value)
                steps += tcount[e]
                oo = o.reset()                                    # next episode: same draws as the in-kernel restart
                tcount[e] = 0
            np.testing.assert_array_equal(obs[e], oo)            # the observation after the step is the new episode's
            np.testing.assert_array_equal(np.stack([st['loc_r'][e], st['loc_c'][e]], -1), o.loc)
            assert st['t'][e] == tcount[e] and st['episode'][e] == o.episode
    s = env.device_stats()
    assert ends > E and s.auto_episodes == ends and s.auto_success_sum == succ and s.auto_env_steps == steps
    env.set_auto_reset(0)                                         # back to lock-step: finished envs freeze again
    env.reset()
    assert env.device_stats().auto_episodes == 0


def test_tj_env_restarts_at_the_step_cap():
    import oracle
    E, N, cap = 12, 5, 4
    env = make_tj(N, 6, 1, "easy", E, seed=5, offset=7, add_rate_min=0.6, add_rate_max=0.6)
    env.set_auto_reset(cap)
    env.reset(0)
    orcs = [oracle.TJOracle(N, 6, 1, "easy", add_rate_min=0.6, add_rate_max=0.6, seed=5, env_gid=7 + e) for e in range(E)]
    for o in orcs:
        o.reset(0)
    rs = np.random.RandomState(1)
    succ = 0
    for t in range(11):
        act = (rs.rand(E, N) < 0.3).astype(np.int32)
        obs, rew, done, info = env.step(torch.from_numpy(act).cuda())
        obs, rew, done, alive = obs.cpu().numpy(), rew.cpu().numpy(), done.cpu().numpy(), info['alive_mask'].cpu().numpy()
        for e, o in enumerate(orcs):
            oo, orew, _ = o.step(act[e])
            np.testing.assert_array_equal(rew[e], orew.astype(np.float32))
            np.testing.assert_array_equal(alive[e], o.alive)      # info of the finishing step belongs to the old episode
            end = (t + 1) % cap == 0
            assert int(done[e]) == int(end)
            if end:
                succ += 1 - int(o.has_failed.value)
                oo = o.reset(0)
            np.testing.assert_array_equal(obs[e], oo)
    s = env.device_stats()
    assert s.auto_episodes == 2 * E and s.auto_success_sum == succ and s.auto_env_steps == 2 * E * cap


def _trainer(E, T, seed, auto, passes=1):
    from ic3net_amd import data
    from ic3net_amd.action_utils import parse_action_args
    from ic3net_amd.comm import CommNetMLP
    from ic3net_amd.trainer import Trainer
    a = argparse.Namespace(
        batch_size=500, hid_size=64, recurrent=True, seed=seed, lrate=0.001, env_name='predator_prey', max_steps=T,
        display=False, commnet=1, ic3net=True, nagents=2, comm_mode='avg', comm_passes=passes, comm_mask_zero=False,
        mean_ratio=0, rnn_type='LSTM', detach_gap=10, comm_init='uniform', hard_attn=1, comm_action_one=False,
        share_weights=False, nenvs=E, env_id_offset=400, store_states=False, gamma=1.0, normalize_rewards=False,
        entr=0, value_coeff=0.01, advantages_per_action=False, nenemies=1, dim=3, vision=1, moving_prey=False,
        no_stay=False, mode='mixed', enemy_comm=False, nfriendly=2, auto_reset=auto, hip_graph=False)
    env = data.init('predator_prey', a, False)
    a.num_actions = [env.num_actions, 2]
    a.dim_actions = env.dim_actions + 1
    a.num_inputs = env.observation_dim
    parse_action_args(a)
    torch.manual_seed(seed)
    net = CommNetMLP(a, a.num_inputs).cuda().float()
    with torch.no_grad():
        net.heads[0].weight.mul_(3.0)
    return Trainer(a, net, env), a, net


@pytest.mark.parametrize("graph,passes", [(False, 1), (True, 1), (False, 2)])
def test_auto_reset_stream_equals_consecutive_reference_style_episodes(graph, passes):
    """passes = 2: comm_passes > 1 — ic3_policy_step once per communication pass; an env restarted inside the launch must
    start from a zero state in the FIRST pass and keep the first pass's state in the second."""
    import oracle
    from oracle import philox, policy_ref
    E, T, seed, N = 30, 8, 17, 2
    tr, a, net = _trainer(E, T, seed, True, passes)
    a.hip_graph = graph
    params = {k: v.detach().cpu().double().numpy() for k, v in net.state_dict().items()}
    orcs = [oracle.PPOracle(N, 3, 1, "mixed", seed=seed, env_gid=400 + e) for e in range(E)]
    total_eps = 0
    for window in range(3):
        episode, stat = tr.get_episode(window)
        assert len(episode) == T and stat['num_steps'] == T * E          # every slot of every env is a real transition
        act = torch.stack([t.action for t in episode]).cpu().numpy()       # (T, heads, E, N)
        rew = torch.stack([t.reward for t in episode]).cpu().numpy()
        done = torch.stack([t.misc['done'] for t in episode]).cpu().numpy()
        emask = torch.stack([t.episode_mask for t in episode]).cpu().numpy()
        lp = [torch.stack([t.action_out[k] for t in episode]).cpu().numpy() for k in range(2)]
        val = torch.stack([t.value.reshape(E, N) for t in episode]).cpu().numpy()
        n_eps = succ = 0
        for e, o in enumerate(orcs):
            obs = o.reset()                                                # begin_episode resets every env of the window
            hc = (np.zeros((N, 64)), np.zeros((N, 64)))
            gate, tt = np.zeros(N), 0
            for t in range(T):
                logp, value, hc = policy_ref.forward(params, obs[None].astype(np.float64), hc, None, gate,
                                                     recurrent=True, hard_attn=True, nheads=2, comm_passes=passes)
                if e < 6:                                                  # fp64 numpy policy on 6 envs: 1e-5 (north_star)
                    for k in range(2):
                        assert np.abs(logp[k][0] - lp[k][t, e]).max() < 1e-5, (window, e, t, k)
                    assert np.abs(value.reshape(-1) - val[t, e]).max() < 1e-5
                    for k in range(2):                                     # the draw sits at (episode, t-in-episode)
                        for n in range(N):
                            x = philox.x24(seed, 400 + e, philox.DOMAIN_SAMPLE, o.episode, tt, k * N + n)
                            want = oracle.sample_one(lp[k][t, e, n], x)
                            if want != act[t, k, e, n]:
                                cdf = np.cumsum(np.exp(lp[k][t, e, n].astype(np.float64)))
                                assert np.abs(cdf - x / 2.0 ** 24).min() < 1e-6
                obs, orew, od = o.step(act[t, 0, e])
                tt += 1
                np.testing.assert_array_equal(rew[t, e], orew.astype(np.float32))
                end = bool(od) or tt == T
                assert bool(done[t, e]) == (end or t == T - 1)
                np.testing.assert_array_equal(emask[t, e], np.full(N, 0.0 if (end or t == T - 1) else 1.0))
                if end or t == T - 1:
                    n_eps += 1
                    succ += int(o.success.value)
                if end and t < T - 1:
                    obs = o.reset()
                    hc = (np.zeros((N, 64)), np.zeros((N, 64)))
                    gate, tt = np.zeros(N), 0
                else:
                    gate = act[t, 1, e].astype(np.float64)
                    if end:                                                # restarted on the last slot: zero-length episode
                        o.reset()
        assert stat['success'] == succ
        total_eps += n_eps
        assert n_eps > E                                                   # episodes did end early and restart
    # run_batch (collection mode, round 5): the batch holds WHOLE episodes only — a slot counts iff its episode ends inside the
    # batch — and counts the episodes that ended (stat normalisation main.py:219-225 divides by them)
    tr2, a2, _ = _trainer(E, T, seed, True, passes)
    a2.batch_size = 2 * T * E                                              # two windows of T slots: the streams run on across them
    batch, st = tr2.run_batch(0)
    done = torch.stack([m['done'] for m in batch.misc]).cpu().numpy()      # (2T, E)
    live = torch.stack([m['live'] for m in batch.misc]).cpu().numpy()
    assert done.shape == (2 * T, E)
    for e in range(E):
        ends = np.nonzero(done[:, e])[0]
        n_live = 0 if len(ends) == 0 else ends[-1] + 1                     # everything up to the env's last episode end
        assert live[:, e].sum() == n_live and live[:n_live, e].all()
        assert len(ends) >= 2 and (np.diff(np.concatenate([[-1], ends])) <= T).all()   # no episode longer than max_steps
    assert st['num_episodes'] == done.sum() > 2 * E and st['num_steps'] == live.sum() <= 2 * T * E
