"""GPU: ic3_policy_step — policy forward + action draws + env.step as ONE launch (csrc/policy_step.hip) — against the
chain of separately verified launches it replaces (sparse encoder, comm_masked_mean, library GEMMs, lstm_cell_heads,
sample_actions, pp/tj_step), step by step over free-running episodes:
  * log-probs, value, h', c' within fp32 rounding of the chain (both sides are pinned to the reference at 1e-5 by
    tests/test_policy_gpu.py; here 2e-5 between two fp32 summation orders),
  * the draws of every head bit-identical to ic3_sample_actions on the kernel's own log-probs at the env's own
    (episode, t) stream position,
  * the env transition (reward, done, alive, is_completed, full integer state) bit-identical to ic3_env_step fed with
    the same actions.
"""
import argparse
import os
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from test_env_parity_gpu import make_pp, make_tj  # noqa: E402


def policy_args(N, H, heads, hard_attn, mode='avg', mask_zero=False, passes=1):
    return argparse.Namespace(nagents=N, hid_size=H, comm_passes=passes, recurrent=True, continuous=False,
                              naction_heads=list(heads), comm_mask_zero=mask_zero, share_weights=False,
                              comm_init='uniform', hard_attn=hard_attn, comm_mode=mode, rnn_type='LSTM', init_std=0.2)


CASES = [
    # kind, env kwargs, H, hard_attn (talk head), E, T
    ("pp", dict(N=10, dim=20, vision=1, mode="mixed"), 128, True, 37, 12),            # PP-hard shape, ragged last tile
    ("pp", dict(N=3, dim=5, vision=0, mode="cooperative"), 128, True, 50, 8),         # PP-easy: 21 envs per tile
    ("pp", dict(N=32, dim=40, vision=2, mode="mixed"), 256, True, 5, 4),              # PP-scaled: H = 256, 2 envs per tile
    ("pp", dict(N=5, dim=10, vision=1, mode="competitive"), 64, False, 33, 8),        # CommNet (one head), H = 64
    ("pp", dict(N=64, dim=12, vision=1, mode="mixed"), 128, True, 3, 4),              # one env per tile, G = 64
    ("pp", dict(N=1, dim=5, vision=1, mode="mixed"), 64, True, 70, 5),                # single agent: n_alive - 1 == 0
    ("pp", dict(N=4, dim=6, vision=1, mode="mixed", enemy_comm=True), 128, True, 19, 8),   # prey rows in the policy
    ("tj", dict(N=10, dim=14, vision=1, difficulty="medium", add_rate_min=0.2, add_rate_max=0.2), 128, False, 40, 14),
    ("tj", dict(N=20, dim=18, vision=1, difficulty="hard", add_rate_min=0.3, add_rate_max=0.3), 128, True, 21, 14),
    ("tj", dict(N=5, dim=6, vision=0, difficulty="easy", add_rate_min=0.5, add_rate_max=0.5), 64, True, 30, 10),
    ("tj", dict(N=6, dim=6, vision=1, difficulty="easy", add_rate_min=0.4, add_rate_max=0.4, vocab_type='scalar'), 128,
     True, 11, 8),
    # more envs than one round of resident workgroups (512 x 6): full tiles + a tail of half tiles (3 envs, one MFMA row tile)
    ("pp", dict(N=10, dim=8, vision=1, mode="mixed"), 128, True, 512 * 6 + 25, 2),
    ("tj", dict(N=20, dim=18, vision=0, difficulty="hard", add_rate_min=0.3, add_rate_max=0.3), 128, True, 512 * 3 + 7, 2),
]


def make_env(kind, kw, E, seed, offset):
    kw = dict(kw)
    if kind == "pp":
        return make_pp(kw.pop('N'), kw.pop('dim'), kw.pop('vision'), kw.pop('mode'), E, seed=seed, offset=offset, **kw)
    return make_tj(kw.pop('N'), kw.pop('dim'), kw.pop('vision'), kw.pop('difficulty'), E, seed=seed, offset=offset, **kw)


@pytest.mark.parametrize("kind,kw,H,hard_attn,E,T", CASES)
@pytest.mark.parametrize("comm_mode", ["avg", "sum", "avg-2passes", "avg-fp32", "sum-fp32", "avg-2passes-fp32"])
def test_policy_step_equals_the_launch_chain(kind, kw, H, hard_attn, E, T, comm_mode):
    """("avg-2passes": comm_passes = 2 — the one-launch kernel once per communication pass against the generic
    forward() of the same module, which is pinned to the reference by the multi-pass policy fixtures.  "-fp32": the gate
    product on the fp32 matrix instruction (args.gate_split = False) instead of the default exact bf16 split products —
    both arithmetic modes run every shape at the same tolerances.)"""
    from ic3net_amd import ops
    from ic3net_amd.comm import CommNetMLP
    passes, split = 1, True
    if comm_mode.endswith("-fp32"):
        comm_mode, split = comm_mode[:-5], False
    if comm_mode == "avg-2passes":
        if not ((kind == "pp" and kw['N'] in (10, 32) and E < 100) or (kind == "tj" and kw['N'] == 10)):
            pytest.skip("comm_passes = 2 is covered on three shapes")
        comm_mode, passes = "avg", 2
    if comm_mode == "sum" and not (kind == "pp" and kw['N'] in (10, 5)):
        pytest.skip("comm_mode='sum' is covered on two shapes")
    seed, offset = 13, 700
    envA, envB = make_env(kind, kw, E, seed, offset), make_env(kind, kw, E, seed, offset)
    N = envA.nagents_env
    nact = envA.dims.naction
    heads = [nact, 2] if hard_attn else [nact]
    torch.manual_seed(H + N)
    netA = CommNetMLP(policy_args(N, H, heads, hard_attn, comm_mode, passes=passes), envA.obs_dim).cuda().float()
    with torch.no_grad():                    # livelier logits than the default init
        for hd in netA.heads:
            hd.weight.mul_(4.0)
    netB = copy.deepcopy(netA)
    netA.args.mega_policy = False
    netB.args = copy.copy(netA.args)
    netB.args.mega_policy = True
    netB.args.gate_split = split
    for net, env in ((netA, envA), (netB, envB)):
        net.obs_encoder, net.obs_table = env.encode, env.encode_table
    assert ops.policy_step_supported(envB, H)
    R = E * N
    dev = 'cuda'
    for ep in range(2):
        obsA = envA.reset(ep) if kind == "tj" else envA.reset()
        obsB = envB.reset(ep) if kind == "tj" else envB.reset()
        assert torch.equal(obsA, obsB)
        hid = netA.init_hidden(E)
        info = {}
        if hard_attn:
            info['comm_action'] = torch.zeros((E, N), dtype=torch.int32, device=dev)          # quirk Q22
        episode = int(envB.get_state()['episode'][0])
        for t in range(T):
            with torch.no_grad():
                hidA = (hid[0].detach().clone(), hid[1].detach().clone())
                logpA, valA, (hA, cA) = netA([obsA, hidA], info)
                assert netB.mega_ok(envB, [envB._obs, hid])
                act = torch.full((len(heads), E, N), -1, dtype=torch.int32, device=dev)
                rew = torch.full((E, N), 7.0, device=dev)
                done = torch.full((E,), -1, dtype=torch.int32, device=dev)
                alive = torch.full((E, N), -1, dtype=torch.int32, device=dev)
                comp = torch.full((E, N), -1, dtype=torch.int32, device=dev)
                with_obs = (t % 3 != 2)       # dense obs of the state acted on, from the same launch (2 steps of 3)
                obs_in = obsA.clone()
                if with_obs:
                    envB._obs.fill_(-5.0)
                logpB, valB, (hB, cB) = netB.step_env(envB, [envB._obs, hid], info, act, rew, done, alive, comp,
                                                      obs=envB._obs if with_obs else None)
            tol = 2e-5
            for k in range(len(heads)):
                assert float((logpA[k] - logpB[k]).abs().max()) < tol, (t, k)
            assert float((valA - valB).abs().max()) < tol * max(1.0, float(valA.abs().max()))
            assert float((hA - hB).abs().max()) < tol and float((cA - cB).abs().max()) < tol * max(1.0, float(cA.abs().max()))
            # draws: bit-identical to the stand-alone sampler on the kernel's own log-probs at the envs' own stream
            # positions (the twin env A has not stepped yet: same per-env (episode, t); a finished env keeps its t)
            for k in range(len(heads)):
                want = ops.sample_actions_env(envA, logpB[k], k)
                assert torch.equal(act[k], want), (t, k)
            # env transition: the stand-alone step kernel on the twin env, same actions
            obsA, rA, dA, infoA = envA.step(act[0])
            assert torch.equal(rA, rew) and torch.equal(dA, done)
            sa, sb = envA.get_state(), envB.get_state()
            for f in sa:
                np.testing.assert_array_equal(sa[f], sb[f], err_msg="%s t=%d" % (f, t))
            if kind == "tj":
                assert torch.equal(infoA['alive_mask'], alive) and torch.equal(infoA['is_completed'], comp)
                info = {'alive_mask': alive.clone()}
            else:
                assert bool((alive == 1).all()) and bool((comp == 0).all())
                info = {}
            if hard_attn:
                info['comm_action'] = act[-1].clone()
            if with_obs:        # the observation of the INPUT state, bit-identical to the stand-alone obs-assembly kernel
                assert torch.equal(envB._obs, obs_in), "obs t=%d: %d entries differ" % (t, int((envB._obs != obs_in).sum()))
            envB.observe()
            hid = (hB, cB)


def test_policy_step_masks_and_dead_envs():
    """alive masks with n_alive in {0, 1, N}, silent gates and comm_mask_zero through ic3_policy_step vs the chain."""
    from ic3net_amd.comm import CommNetMLP
    E, N, H = 23, 10, 128
    envA, envB = make_env("pp", dict(N=N, dim=8, vision=1, mode="mixed"), E, 3, 0), \
        make_env("pp", dict(N=N, dim=8, vision=1, mode="mixed"), E, 3, 0)
    for mask_zero in (False, True):
        torch.manual_seed(5)
        netA = CommNetMLP(policy_args(N, H, [5, 2], True, mask_zero=mask_zero), envA.obs_dim).cuda().float()
        netB = copy.deepcopy(netA)
        netA.args.mega_policy = False
        netB.args = copy.copy(netA.args)
        netB.args.mega_policy = True
        for net, env in ((netA, envA), (netB, envB)):
            net.obs_encoder, net.obs_table = env.encode, env.encode_table
        envA.reset()
        envB.reset()
        alive = (torch.rand(E, N, device='cuda') < 0.6).int()
        alive[0] = 0
        alive[1] = 0
        alive[1, 3] = 1
        alive[2] = 1
        gate = (torch.rand(E, N, device='cuda') < 0.5).int()
        gate[3] = 0
        info = {'alive_mask': alive, 'comm_action': gate}
        h0 = torch.randn(E * N, H, device='cuda') * 0.5
        c0 = torch.randn(E * N, H, device='cuda') * 0.5
        with torch.no_grad():
            logpA, valA, (hA, cA) = netA([envA._obs, (h0.clone(), c0.clone())], info)
            act = torch.zeros((2, E, N), dtype=torch.int32, device='cuda')
            logpB, valB, (hB, cB) = netB.step_env(envB, [envB._obs, (h0.clone(), c0.clone())], info, act,
                                                  torch.zeros(E, N, device='cuda'),
                                                  torch.zeros(E, dtype=torch.int32, device='cuda'))
        for x, y in ((logpA[0], logpB[0]), (logpA[1], logpB[1]), (valA, valB), (hA, hB), (cA, cB)):
            assert float((x - y).abs().max()) < 2e-5 * max(1.0, float(x.abs().max()))


def test_policy_step_is_deterministic_and_graph_replay_identical():
    """Same inputs -> bit-identical outputs, launch after launch and as a hipGraph replay (no data race between the
    phases of the kernel: every LDS hand-over sits behind a barrier)."""
    from ic3net_amd.comm import CommNetMLP
    E, N, H = 96, 10, 128
    env = make_env("pp", dict(N=N, dim=20, vision=1, mode="mixed"), E, 5, 0)
    torch.manual_seed(2)
    net = CommNetMLP(policy_args(N, H, [5, 2], True), env.obs_dim).cuda().float()
    net.obs_encoder, net.obs_table = env.encode, env.encode_table
    env.reset()
    st0 = env.get_state()
    h0 = torch.randn(E * N, H, device='cuda') * 0.5
    c0 = torch.randn(E * N, H, device='cuda') * 0.5
    gate = (torch.rand(E, N, device='cuda') < 0.5).int()
    info = {'comm_action': gate}
    bufs = dict(act=torch.zeros((2, E, N), dtype=torch.int32, device='cuda'), rew=torch.zeros(E, N, device='cuda'),
                done=torch.zeros(E, dtype=torch.int32, device='cuda'))

    def once():
        env.set_state(**st0)
        with torch.no_grad():
            logp, val, (h, c) = net.step_env(env, [env._obs, (h0.clone(), c0.clone())], info, bufs['act'], bufs['rew'],
                                             bufs['done'])
        return [x.clone() for x in (logp[0], logp[1], val, h, c, bufs['act'], bufs['rew'])]

    ref = once()
    for rep in range(10):
        got = once()
        for i, (x, y) in enumerate(zip(ref, got)):
            assert torch.equal(x, y), (rep, i)
    # the same launch captured and replayed
    env.set_state(**st0)
    hh, cc = h0.clone(), c0.clone()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.no_grad(), torch.cuda.graph(g):
        logp, val, (h, c) = net.step_env(env, [env._obs, (hh, cc)], info, bufs['act'], bufs['rew'], bufs['done'])
    for rep in range(3):
        env.set_state(**st0)
        net._mb['h'].copy_(h0)
        net._mb['c'].copy_(c0)
        hh.copy_(h0)
        cc.copy_(c0)
        g.replay()
        got = [logp[0], logp[1], val, h, c, bufs['act'], bufs['rew']]
        for i, (x, y) in enumerate(zip(ref, got)):
            assert torch.equal(x, y), ("graph", rep, i)


def test_dispatch_stamped_events_time_the_launch_and_leave_the_results_alone():
    """ic3_env_set_step_events: the next ic3_policy_step dispatch stamps the two events (one shot); outputs are the
    ones of an untimed launch."""
    from ic3net_amd.comm import CommNetMLP
    from ic3net_amd.envs import DispatchEvent
    E, N, H = 512, 10, 128
    env = make_env("pp", dict(N=N, dim=20, vision=1, mode="mixed"), E, 5, 0)
    torch.manual_seed(3)
    net = CommNetMLP(policy_args(N, H, [5, 2], True), env.obs_dim).cuda().float()
    net.obs_encoder, net.obs_table = env.encode, env.encode_table
    env.reset()
    st0 = env.get_state()
    h0 = torch.randn(E * N, H, device='cuda') * 0.5
    bufs = dict(act=torch.zeros((2, E, N), dtype=torch.int32, device='cuda'), rew=torch.zeros(E, N, device='cuda'),
                done=torch.zeros(E, dtype=torch.int32, device='cuda'))
    info = {'comm_action': torch.ones(E, N, dtype=torch.int32, device='cuda')}

    def once(events=None):
        env.set_state(**st0)
        if events is not None:
            env.set_step_events(*events)
        with torch.no_grad():
            logp, val, (h, c) = net.step_env(env, [env._obs, (h0.clone(), h0.clone())], info, bufs['act'], bufs['rew'],
                                             bufs['done'], obs=env._obs)
        return [x.clone() for x in (logp[0], logp[1], val, h, c, bufs['act'], bufs['rew'], env._obs)]

    ref = once()
    e0, e1 = DispatchEvent(), DispatchEvent()
    timed = once((e0, e1))
    untimed_again = once()                   # the events were one-shot: this launch must not touch them
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    assert 0.005 < ms < 5.0, ms
    for x, y, z in zip(ref, timed, untimed_again):
        assert torch.equal(x, y) and torch.equal(x, z)
    assert e0.elapsed_time(e1) == ms


def test_policy_step_rejects_what_it_cannot_run():
    from ic3net_amd import ops
    env = make_env("pp", dict(N=10, dim=20, vision=1, mode="mixed"), 4, 1, 0)
    assert ops.policy_step_supported(env, 128) and ops.policy_step_supported(env, 64)
    assert not ops.policy_step_supported(env, 32) and not ops.policy_step_supported(env, 96)
    big = make_env("pp", dict(N=10, dim=30, vision=8, mode="mixed"), 2, 1, 0)   # 6 envs x 10 x 17^2 descriptors: > 80 KB
    assert not ops.policy_step_supported(big, 128)


@pytest.mark.parametrize("workload", ["pp_hard", "tj_medium"])
def test_trainer_takes_the_one_launch_path_and_graph_replay_is_identical(workload):
    """BASELINE policies through the Trainer: with hid 128 the per-step launch sequence is ic3_policy_step + obs
    assembly; eager and hipGraph-replayed episodes are identical, and the policy numbers agree with the launch
    chain (args.mega_policy = False) on the first step of an episode (same state, same h = 0)."""
    import bench
    from ic3net_amd import ops
    calls = []
    orig = ops.policy_step

    def counting(*a, **k):
        calls.append(1)
        return orig(*a, **k)
    ops.policy_step = counting
    try:
        runs = {}
        for mode in ("eager", "graph", "chain"):
            tr, a = bench.build_trainer(workload, 48, 7, 200, 0)
            a.hip_graph = mode == "graph"
            a.mega_policy = mode != "chain"
            n0 = len(calls)
            runs[mode] = []
            for ep in range(3):       # copy out per episode: in graph mode action_out / value are the graphs' static outputs
                e, s = tr.get_episode(ep)
                runs[mode].append(([t.action.clone() for t in e], [t.reward.clone() for t in e],
                                   [t.value.clone() for t in e], [t.action_out[0].clone() for t in e], s))
            if mode == "eager":
                assert len(calls) - n0 == 3 * a.max_steps
            if mode == "chain":
                assert len(calls) == n0
    finally:
        ops.policy_step = orig
    for ep in range(3):
        for i in range(4):
            for t, (x, y) in enumerate(zip(runs["eager"][ep][i], runs["graph"][ep][i])):
                if not torch.equal(x, y):
                    d = (x.float() - y.float()).abs().reshape(-1)
                    raise AssertionError("episode %d field %d step %d: %d of %d entries differ, max |d| = %g at %d" %
                                         (ep, i, t, int((d != 0).sum()), d.numel(), float(d.max()), int(d.argmax())))
        assert runs["eager"][ep][4]['num_steps'] == runs["graph"][ep][4]['num_steps']
    v0, v1 = runs["eager"][0][2][0], runs["chain"][0][2][0]
    assert float((v0 - v1).abs().max()) < 2e-5
    l0, l1 = runs["eager"][0][3][0], runs["chain"][0][3][0]
    assert float((l0 - l1).abs().max()) < 2e-5


def test_result_changing_environment_knobs_are_gone():
    """Round 2 read IC3_PS_DEBUG / ZMODE / SKEW / WGS with getenv on the hot entry point ("results are wrong when set");
    rounds 3-5 kept the pacing overrides IC3_PS_ZS / ZF / ZFRAC / ... for sweeps.  Round 6 removed them all: the library
    reads ONE variable, IC3_PS_HALF (the tile plan, a test hook: same results either way), and produces the same bits with
    or without any of the old names in the environment."""
    import subprocess
    import sys as _sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def run(extra):
        env = dict(os.environ)
        env.update(extra)
        r = subprocess.run([_sys.executable, os.path.join(root, "tests", "ps_checksum_worker.py")], cwd=root, env=env,
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        return [l for l in r.stdout.splitlines() if l.startswith("PS_CRC")][0]
    base = run({})
    assert run({"IC3_PS_DEBUG": "63", "IC3_PS_ZMODE": "48", "IC3_PS_SKEW": "9", "IC3_PS_WGS": "1"}) == base
    assert run({"IC3_PS_ZS": "2", "IC3_PS_ZF": "9", "IC3_PS_ZFRAC": "40", "IC3_PS_HALF": "1", "IC3_PS_ZEPI": "0",
                "IC3_PS_Z0": "5", "IC3_PS_Z3": "5", "IC3_PS_ZC": "4", "IC3_PS_ZH": "6"}) == base


@pytest.mark.parametrize("workload,E", [("pp_hard", 29), ("tj_hard", 11), ("tj_medium", 16)])
def test_incremental_obs_rows_are_the_same_rows(workload, E):
    """EXPERIMENT ic3_env_set_incremental_obs: with it on, ic3_policy_step clears what the previous call painted into the
    obs buffer and paints the new entries instead of zero-filling the rows — after every step of two episodes (lock-step
    reset in between, which does not touch the buffer) the buffer equals the stand-alone obs kernel's rows of the state
    acted on, bit for bit; a foreign write through the handle (observe into that buffer) falls back to a full rewrite."""
    import bench
    tr, a = bench.build_trainer(workload, E, 6, 10, 0)
    a.max_steps = 12
    a.incremental_obs = True
    raw = tr.env.env
    for ep in range(2):
        tr.begin_episode(ep)
        for t in range(a.max_steps):
            from ic3net_amd import _lib
            from ic3net_amd._lib import check, ptr, stream
            want = torch.empty_like(raw._obs)                 # the stand-alone kernel's rows, into ANOTHER buffer
            check(_lib.lib().ic3_env_observe_at(raw._h, None, ptr(want), stream()))
            tr.step_episode(t)
            assert torch.equal(raw._obs, want), (workload, ep, t)
            if ep == 1 and t == 5:
                raw.observe()                      # another writer of the buffer: the next step rewrites every row
        tr.end_episode()
    assert getattr(tr.policy_net, 'mega_steps', 0) == 2 * a.max_steps
