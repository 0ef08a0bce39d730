"""ic3_commnet_step — one rollout iteration of the NON-recurrent CommNet module as ONE launch (trainer.py:43-108 through
comm.py:127-129,179-205,220-224) — pinned in one hop: free-running episodes through the Trainer against
oracle.policy_ref (numpy float64, recurrent = False; pinned by the non-recurrent policy fixtures recorded from the reference)
driven by the C oracle env on the kernel's own actions at the north_star's 1e-5; rewards and the dense observation rows
written by the same launch bit for bit; the draws bit-identical to ic3_env_sample_actions on the kernel's log-probs."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _oracle_env(a, seed, gid):
    import oracle
    if a.env_name == 'predator_prey':
        return oracle.PPOracle(a.nagents, a.dim, a.vision, a.mode, seed=seed, env_gid=gid)
    return oracle.TJOracle(a.nagents, a.dim, a.vision, a.difficulty, a.add_rate_min, a.add_rate_max, a.curr_start,
                           a.curr_end, seed=seed, env_gid=gid, vocab_type=a.vocab_type)


@pytest.mark.parametrize("workload,over,E,T", [
    ("tj_medium_commnet_mlp", dict(), 13, 40),
    ("tj_hard", dict(recurrent=False, comm_passes=1), 7, 30),             # IC3Net gating (hard_attn, comm_action_one)
    ("pp_hard", dict(recurrent=False, comm_passes=2), 13, 30),            # gated IC3Net-style, two passes
    ("pp_easy", dict(recurrent=False, commnet=True, ic3net=False, comm_passes=3, share_weights=True, hid_size=64), 9, 20),
])
def test_commnet_step_full_episode_vs_fp64_reference_policy(workload, over, E, T):
    import bench
    from ic3net_amd import ops
    from oracle import policy_ref
    seed, offset = 5, 300
    tr, a = bench.build_trainer(workload, E, seed, offset, 0, **over)
    a.max_steps = T
    assert not a.recurrent
    tr.begin_episode(0)
    raw = tr.env.env
    N, nheads = a.nagents, len(a.naction_heads)
    params = {k: v.detach().cpu().double().numpy() for k, v in tr.policy_net.state_dict().items()}
    rec = []
    for t in range(T):
        tr.step_episode(t)
        _, action_out, value, _ = tr._step_out[t]
        rec.append(dict(logp=[ao.cpu().numpy() for ao in action_out], value=value.reshape(E, N).cpu().numpy(),
                        act=tr._buf['action'][t].cpu().numpy(), rew=tr._buf['reward'][t].cpu().numpy(),
                        obs=raw._obs.cpu().numpy()))
    assert getattr(tr.policy_net, 'commnet_steps', 0) == T, "the one-launch path did not run"
    assert getattr(tr.policy_net, 'commnet_forwards', 0) == 0
    import oracle
    from oracle import philox
    tj = a.env_name == 'traffic_junction'
    worst = 0.0
    for e in range(E):
        o = _oracle_env(a, seed, offset + e)
        obs = o.reset(0) if tj else o.reset()
        alive, gate = None, np.zeros(N)
        for t in range(T):
            r = rec[t]
            np.testing.assert_array_equal(r['obs'][e], obs, err_msg="obs rows env %d step %d" % (e, t))
            logp, val, _ = policy_ref.forward(params, obs[None].astype(np.float64), None, alive, gate if a.hard_attn else None,
                                              recurrent=False, comm_passes=a.comm_passes, comm_mode_avg=(a.comm_mode == 'avg'),
                                              hard_attn=bool(a.hard_attn), nheads=nheads)
            for hd in range(nheads):
                worst = max(worst, np.abs(logp[hd][0] - r['logp'][hd][e]).max())
                for n in range(N):   # the draw = the oracle's inverse-CDF on the kernel's own log-probs at this stream position
                    want = oracle.sample_one(r['logp'][hd][e, n], philox.x24(seed, offset + e, philox.DOMAIN_SAMPLE, 0, t, hd * N + n))
                    if want != r['act'][hd, e, n]:
                        u = philox.x24(seed, offset + e, philox.DOMAIN_SAMPLE, 0, t, hd * N + n) / 2.0 ** 24
                        assert np.abs(np.cumsum(np.exp(r['logp'][hd][e, n].astype(np.float64))) - u).min() < 1e-6
            worst = max(worst, np.abs(val.reshape(-1) - r['value'][e]).max())
            assert worst < TOL, (workload, e, t, worst)
            obs, orew, done = o.step(r['act'][0, e])
            np.testing.assert_array_equal(r['rew'][e], np.asarray(orew).astype(np.float32))
            if done:
                break
            if tj:
                alive = o.alive.astype(np.float64)
            if a.hard_attn:
                gate = np.ones(N) if a.comm_action_one else r['act'][nheads - 1, e].astype(np.float64)
    assert worst < TOL


@pytest.mark.parametrize("gated", [True, False], ids=["gated", "commnet"])
def test_commnet_step_on_an_auto_reset_handle(gated):
    """Round 5: ic3_commnet_step takes handles in auto-reset mode — an env that finishes (Predator-Prey 'mixed': every predator
    on the prey; or the step cap) restarts inside the launch, and at an episode's first step nobody is dead and a gated policy's
    gate is 0 (trainer.py:41-46, quirks Q21 / Q22; the plain CommNet talks from the first step).  One window through the
    Trainer against the fp64 policy + consecutive oracle episodes per env."""
    import bench
    import oracle
    from oracle import policy_ref
    seed, offset, E, T = 7, 500, 24, 12
    over = dict(recurrent=False, comm_passes=2, nagents=2, dim=3, vision=1, hid_size=64, max_steps=T)
    if not gated:
        over.update(ic3net=False, commnet=True)
    tr, a = bench.build_trainer("pp_hard", E, seed, offset, 0, **over)
    a.auto_reset = True
    with torch.no_grad():
        tr.policy_net.heads[0].weight.mul_(3.0)                     # peaked action distributions: episodes do end early
    tr.begin_episode(0)
    raw = tr.env.env
    N, nheads = a.nagents, len(a.naction_heads)
    params = {k: v.detach().cpu().double().numpy() for k, v in tr.policy_net.state_dict().items()}
    rec = []
    for t in range(T):
        tr.step_episode(t)
        _, action_out, value, _ = tr._step_out[t]
        rec.append(dict(logp=[ao.cpu().numpy() for ao in action_out], value=value.reshape(E, N).cpu().numpy(),
                        act=tr._buf['action'][t].cpu().numpy(), rew=tr._buf['reward'][t].cpu().numpy(),
                        done=tr._buf['done'][t].cpu().numpy(), obs=raw._obs.cpu().numpy()))
    assert getattr(tr.policy_net, 'commnet_steps', 0) == T, "the one-launch path did not run"
    worst, restarts = 0.0, 0
    for e in range(E):
        o = oracle.PPOracle(N, a.dim, a.vision, a.mode, seed=seed, env_gid=offset + e)
        obs = o.reset()
        gate, tt = np.zeros(N), 0
        for t in range(T):
            r = rec[t]
            np.testing.assert_array_equal(r['obs'][e], obs, err_msg="obs rows env %d slot %d" % (e, t))
            logp, val, _ = policy_ref.forward(params, obs[None].astype(np.float64), None, None, gate if a.hard_attn else None,
                                              recurrent=False, comm_passes=a.comm_passes, comm_mode_avg=(a.comm_mode == 'avg'),
                                              hard_attn=bool(a.hard_attn), nheads=nheads)
            for hd in range(nheads):
                worst = max(worst, np.abs(logp[hd][0] - r['logp'][hd][e]).max())
            worst = max(worst, np.abs(val.reshape(-1) - r['value'][e]).max())
            assert worst < TOL, (e, t, worst)
            obs, orew, od = o.step(r['act'][0, e])
            tt += 1
            np.testing.assert_array_equal(r['rew'][e], np.asarray(orew).astype(np.float32))
            end = bool(od) or tt == T
            assert bool(r['done'][e]) == end, (e, t)
            if end:
                restarts += 1
                obs = o.reset()
                gate, tt = np.zeros(N), 0
            elif a.hard_attn:
                gate = r['act'][nheads - 1, e].astype(np.float64)
    assert restarts > E // 4, "no episode ended early: the test would not see a restart"


def test_commnet_step_refuses_what_it_does_not_cover():
    import bench
    from ic3net_amd import ops
    tr, a = bench.build_trainer("tj_medium_commnet_mlp", 8, 1, 0, 0)
    raw = tr.env.env
    assert ops.commnet_step_supported(raw, 128) and not ops.commnet_step_supported(raw, 96)
    raw.set_auto_reset(5)
    assert ops.commnet_step_supported(raw, 128)          # (round 5: episode starts inside the launch are handled here too)


@pytest.mark.parametrize("wl", ["pp_hard_ic", "pp_hard_iric_tanh"])
def test_narrow_launch_does_not_depend_on_its_tile_plan(wl):
    """The narrow Predator-Prey launch picks its envs per tile from the env count (plan_store_bound_ept in csrc/commnet_fwd.hip:
    2 / 3 / 6 / 4 envs per tile at the counts below).  An env's trajectory is keyed by (seed, env id) alone: the first 24 envs of
    every run — obs rows, log-probs, values, draws, rewards — are the same bits whatever tile they sat in."""
    import bench
    runs = []
    for E in (24, 600, 1536, 2048):
        tr, a = bench.build_trainer(wl, E, 3, 40, 0, max_steps=6)
        tr.begin_episode(0)
        rec = []
        for t in range(6):
            tr.step_episode(t)
            _, action_out, value, _ = tr._step_out[t]
            rec.append((action_out[0].reshape(E, 10, -1)[:24].clone(), value.reshape(E, 10)[:24].clone(),
                        tr._buf['action'][t][:, :24].clone(), tr._buf['reward'][t][:24].clone(),
                        tr.env.env._obs[:24].clone()))
        assert getattr(tr.policy_net, 'commnet_steps', 0) == 6, "the one-launch path did not run"
        runs.append(rec)
    for other in runs[1:]:
        for t in range(6):
            for x, y in zip(runs[0][t], other[t]):
                assert torch.equal(x, y), t


def _play_first_steps(wl, E, offset, T, sl, **over):
    import bench
    tr, a = bench.build_trainer(wl, E, 3, offset, 0, max_steps=T, **over)
    tr.begin_episode(0)
    N = a.nagents
    rec = []
    for t in range(T):
        tr.step_episode(t)
        _, action_out, value, _ = tr._step_out[t]
        rec.append((action_out[0].reshape(E, N, -1)[sl].clone(), value.reshape(E, N)[sl].clone(),
                    tr._buf['action'][t][:, sl].clone(), tr._buf['reward'][t][sl].clone(), tr.env.env._obs[sl].clone()))
    assert getattr(tr.policy_net, 'commnet_steps', 0) == T, "the one-launch path did not run"
    return rec


@pytest.mark.parametrize("wl,over", [("tj_medium_commnet_mlp", dict()), ("pp_hard_iric_tanh", dict()),
                                     ("tj_medium_commnet_mlp", dict(hid_size=64)), ("tj_medium_commnet_mlp", dict(hid_size=256))],
                         ids=["tj_commnet", "pp_tanh", "tj_commnet_h64", "tj_commnet_h256"])
def test_full_and_short_tiles_give_the_same_bits(wl, over):
    """plan_commnet_tiles (csrc/commnet_fwd.hip): at 2048 envs of 10 agents the launch runs 256 full tiles of 6 envs and 171 short
    tiles of 3 envs (one 32-row MFMA tile) behind them; at 24 envs full tiles only.  Envs [0, 24) of the big launch sit in full
    tiles, envs [2024, 2048) in short ones: both equal the same env ids played by a 24-env launch, bit for bit.  (The narrow
    Predator-Prey launch with obs rows takes its own plan — test_narrow_launch_does_not_depend_on_its_tile_plan; without this
    plan's short tiles the second comparison still pins the env-id keying.)"""
    T = 6
    big_lo = _play_first_steps(wl, 2048, 40, T, slice(0, 24), **over)
    big_hi = _play_first_steps(wl, 2048, 40, T, slice(2024, 2048), **over)
    small_lo = _play_first_steps(wl, 24, 40, T, slice(0, 24), **over)
    small_hi = _play_first_steps(wl, 24, 40 + 2024, T, slice(0, 24), **over)
    for big, small in ((big_lo, small_lo), (big_hi, small_hi)):
        for t in range(T):
            for x, y in zip(big[t], small[t]):
                assert torch.equal(x, y), t
