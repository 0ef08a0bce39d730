"""GPU: the HIP env kernels (through the C ABI / ic3net_amd.envs) against
  (1) golden vectors captured from the reference itself, and
  (2) the CPU oracle on seeded random trajectories at sizes the oracle finishes in seconds.
Integer state is compared bit-exact; rewards as float32(reference float64) bit-exact."""
import argparse

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from golden_util import load, SparseObs, PP_FIXTURES, TJ_FIXTURES, MODES, DIFFS  # noqa: E402


def pp_args(N, dim, vision, mode, E, seed=0, offset=0, no_stay=False, enemy_comm=False):
    return argparse.Namespace(nfriendly=N, nenemies=1, dim=dim, vision=vision, moving_prey=False, mode=mode,
                              enemy_comm=enemy_comm, no_stay=no_stay, nenvs=E, seed=seed, env_id_offset=offset)


def tj_args(N, dim, vision, difficulty, E, seed=0, offset=0, add_rate_min=0.05, add_rate_max=0.05, curr_start=0,
            curr_end=0, vocab_type='bool'):
    return argparse.Namespace(nagents=N, dim=dim, vision=vision, difficulty=difficulty, vocab_type=vocab_type,
                              add_rate_min=add_rate_min, add_rate_max=add_rate_max, curr_start=curr_start,
                              curr_end=curr_end, nenvs=E, seed=seed, env_id_offset=offset)


def make_pp(*a, **k):
    from ic3net_amd.envs import PredatorPreyEnv
    env = PredatorPreyEnv()
    env.multi_agent_init(pp_args(*a, **k))
    return env


def make_tj(*a, **k):
    from ic3net_amd.envs import TrafficJunctionEnv
    env = TrafficJunctionEnv()
    env.multi_agent_init(tj_args(*a, **k))
    return env


@pytest.mark.parametrize("name", PP_FIXTURES)
def test_pp_hip_matches_reference_golden(name):
    fx = load(name)
    N, dim, vision, mode, T, no_stay = [int(x) for x in fx["cfg"]]
    nenv, nep = fx["nsteps"].shape
    ec = bool(int(fx["enemy_comm"]))
    sp = SparseObs(fx["obs_coo"], N + (1 if ec else 0), int(fx["obs_dim"]))
    env = make_pp(N, dim, vision, MODES[mode], nenv, seed=int(fx["seed"]), offset=int(fx["env_gid0"]),
                  no_stay=bool(no_stay), enemy_comm=ec)
    assert env.obs_dim == int(fx["obs_dim"]) and env.nagents_env == N + (1 if ec else 0)
    for ep in range(nep):
        obs = env.reset().cpu().numpy()
        st = env.get_state()
        loc = np.stack([st["loc_r"], st["loc_c"]], -1)
        np.testing.assert_array_equal(loc, fx["init_loc"][:, ep])
        for e in range(nenv):
            np.testing.assert_array_equal(obs[e], sp.dense(e, ep, 0))
        nsteps = fx["nsteps"][:, ep]
        last = {}
        for t in range(int(nsteps.max())):
            act = np.where((t < nsteps)[:, None], fx["actions"][:, ep, t], 0)
            live = [e for e in range(nenv) if t < nsteps[e]]
            obs, rew, done, info = env.step(act)
            obs, rew, done = obs.cpu().numpy(), rew.cpu().numpy(), done.cpu().numpy()
            st = env.get_state()
            loc = np.stack([st["loc_r"], st["loc_c"]], -1)
            for e in range(nenv):
                if e in live:
                    np.testing.assert_array_equal(loc[e], fx["loc"][e, ep, t])
                    np.testing.assert_array_equal(st["reached"][e], fx["reached"][e, ep, t])
                    np.testing.assert_array_equal(rew[e], fx["reward"][e, ep, t].astype(np.float32))
                    assert done[e] == fx["done"][e, ep, t]
                    if fx["success"][e, ep, t] >= 0:
                        assert st["success"][e] == fx["success"][e, ep, t]
                    np.testing.assert_array_equal(obs[e], sp.dense(e, ep, t + 1))
                    last[e] = (loc[e].copy(), st["reached"][e].copy())
                else:   # frozen after done (the reference raises RuntimeError here)
                    assert done[e] == 1 and not rew[e].any()
                    np.testing.assert_array_equal(loc[e], last[e][0])
                    np.testing.assert_array_equal(st["reached"][e], last[e][1])
        env.check_actions()


@pytest.mark.parametrize("name", TJ_FIXTURES)
def test_tj_hip_matches_reference_golden(name):
    fx = load(name)
    N, dim, vision, diff, T = [int(x) for x in fx["cfg"]]
    nenv, nep = fx["epochs"].shape
    sp = SparseObs(fx["obs_coo"], N, int(fx["obs_dim"]))
    cur = fx["curriculum"]
    has_curr = bool(cur[3] > cur[2])
    kw = dict(add_rate_min=float(fx["add_rate"]), add_rate_max=float(fx["add_rate"]))
    if has_curr:
        kw = dict(add_rate_min=cur[0], add_rate_max=cur[1], curr_start=cur[2], curr_end=cur[3])
    kw["vocab_type"] = 'scalar' if ("scalar" in fx.files and int(fx["scalar"])) else 'bool'
    # the curriculum fixture gives every env instance its own epoch sequence -> one handle per env there
    groups = [[e] for e in range(nenv)] if has_curr else [list(range(nenv))]
    for grp in groups:
        env = make_tj(N, dim, vision, DIFFS[diff], len(grp), seed=int(fx["seed"]),
                      offset=int(fx["env_gid0"]) + grp[0], **kw)
        assert env.obs_dim == int(fx["obs_dim"])
        for ep in range(nep):
            obs = env.reset(int(fx["epochs"][grp[0], ep]))
            assert not obs.any().item()
            for t in range(T):
                obs, rew, done, info = env.step(fx["actions"][grp, ep, t])
                obs, rew = obs.cpu().numpy(), rew.cpu().numpy()
                st = env.get_state()
                st["loc"] = np.stack([st["loc_r"], st["loc_c"]], -1)
                for k in ("alive", "wait", "loc", "last_act", "route_loc", "route_id", "is_completed",
                          "cars_in_sys", "has_failed"):
                    np.testing.assert_array_equal(st[k], fx[k][grp, ep, t], err_msg="%s ep=%d t=%d" % (k, ep, t))
                np.testing.assert_array_equal(info["alive_mask"].cpu().numpy(), fx["alive"][grp, ep, t])
                np.testing.assert_array_equal(info["is_completed"].cpu().numpy(), fx["is_completed"][grp, ep, t])
                np.testing.assert_array_equal(rew, fx["reward"][grp, ep, t].astype(np.float32))
                assert env.add_rate == fx["add_rate_seen"][grp[0], ep, t]
                assert not done.any().item()
                for i, e in enumerate(grp):
                    np.testing.assert_array_equal(obs[i], sp.dense(e, ep, t + 1))
        env.check_actions()


def test_tj_tables_through_handle():
    fx = load("tj_tables")
    for key in ("medium_14_v1", "hard_18_v0", "easy_6_v0"):
        diff, dim, v = key.split("_")
        env = make_tj(5, int(dim), int(v[1:]), diff, 2)
        grid, off, rc = env.tables()
        np.testing.assert_array_equal(grid, fx[key + "_grid"])
        np.testing.assert_array_equal(off, fx[key + "_off"])
        np.testing.assert_array_equal(rc, fx[key + "_rc"])


@pytest.mark.parametrize("cfg", [(10, 20, 1, "mixed", 96, 40), (3, 5, 0, "cooperative", 200, 20),
                                 (32, 40, 2, "mixed", 8, 12), (7, 9, 2, "competitive", 64, 25)])
def test_pp_hip_matches_oracle_random(cfg):
    import oracle
    N, dim, vision, mode, E, T = cfg
    env = make_pp(N, dim, vision, mode, E, seed=77, offset=1000)
    orcs = [oracle.PPOracle(N, dim, vision, mode, seed=77, env_gid=1000 + e) for e in range(E)]
    rs = np.random.RandomState(5)
    for ep in range(2):
        obs = env.reset().cpu().numpy()
        for e, o in enumerate(orcs):
            np.testing.assert_array_equal(obs[e], o.reset())
        for t in range(T):
            act = rs.randint(0, 5, size=(E, N))
            obs, rew, done, _ = env.step(act)
            obs, rew, done = obs.cpu().numpy(), rew.cpu().numpy(), done.cpu().numpy()
            st = env.get_state()
            for e, o in enumerate(orcs):
                if o.over.value:
                    assert done[e] == 1
                    continue
                oo, orew, od = o.step(act[e])
                np.testing.assert_array_equal(np.stack([st["loc_r"][e], st["loc_c"][e]], -1), o.loc)
                np.testing.assert_array_equal(st["reached"][e], o.reached)
                np.testing.assert_array_equal(rew[e], orew.astype(np.float32))
                assert done[e] == int(od)
                np.testing.assert_array_equal(obs[e], oo)


@pytest.mark.parametrize("cfg", [(10, 14, 1, "medium", 48, 40, 0.3), (20, 18, 0, "hard", 32, 60, 0.1),
                                 (5, 6, 0, "easy", 128, 20, 0.6), (20, 18, 1, "hard", 8, 30, 1.0)])
def test_tj_hip_matches_oracle_random(cfg):
    import oracle
    N, dim, vision, diff, E, T, rate = cfg
    env = make_tj(N, dim, vision, diff, E, seed=9, offset=500, add_rate_min=rate, add_rate_max=rate)
    orcs = [oracle.TJOracle(N, dim, vision, diff, add_rate_min=rate, add_rate_max=rate, seed=9, env_gid=500 + e)
            for e in range(E)]
    rs = np.random.RandomState(6)
    for ep in range(2):
        env.reset(ep)
        for o in orcs:
            o.reset(ep)
        for t in range(T):
            act = (rs.rand(E, N) < 0.35).astype(np.int32)
            obs, rew, done, info = env.step(act)
            obs, rew = obs.cpu().numpy(), rew.cpu().numpy()
            st = env.get_state()
            for e, o in enumerate(orcs):
                oo, orew, _ = o.step(act[e])
                for k in ("alive", "wait", "last_act", "route_loc", "route_id", "is_completed"):
                    np.testing.assert_array_equal(st[k][e], getattr(o, k), err_msg=k)
                np.testing.assert_array_equal(np.stack([st["loc_r"][e], st["loc_c"][e]], -1), o.loc)
                assert st["cars_in_sys"][e] == o.cars_in_sys.value and st["has_failed"][e] == o.has_failed.value
                np.testing.assert_array_equal(rew[e], orew.astype(np.float32))
                np.testing.assert_array_equal(obs[e], oo)


def test_pp_full_size_properties():
    """BASELINE config 2 at full size (E=8192): size-independent properties of the obs tensor."""
    E, N, dim, v = 8192, 10, 20, 1
    env = make_pp(N, dim, v, "mixed", E, seed=3)
    obs = env.reset()
    vocab = dim * dim + 4
    o = obs.view(E, N, 9, vocab)
    # exactly one location bit (cell id or OUTSIDE) per window cell
    assert torch.equal(o[..., :dim * dim + 2].sum(-1), torch.ones(E, N, 9, device=obs.device))
    # the centre cell holds the agent itself: predator count >= 1; total predator count over the 3x3 window
    assert (o[:, :, 4, vocab - 1] >= 1).all()
    st = env.get_state()
    r, c = torch.as_tensor(st["loc_r"][:, :N]).cuda(), torch.as_tensor(st["loc_c"][:, :N]).cuda()
    centre_id = (r * dim + c).long()
    assert torch.equal(o[:, :, 4, :dim * dim].argmax(-1), centre_id)
    # shard invariance: envs [4096, 8192) of this handle == a second handle with env_id_offset 4096
    env2 = make_pp(N, dim, v, "mixed", 4096, seed=3, offset=4096)
    obs2 = env2.reset()
    assert torch.equal(obs[4096:], obs2)


def test_bad_action_flag_and_errors():
    env = make_pp(3, 5, 0, "mixed", 4)
    env.reset()
    env.step(np.full((4, 3), 5))          # 5 is tolerated (<= naction, quirk Q2)
    env.check_actions()
    env.step(np.full((4, 3), 6))
    with pytest.raises(AssertionError):
        env.check_actions()
    from ic3net_amd.envs import PredatorPreyEnv
    bad = PredatorPreyEnv()
    a = pp_args(3, 5, 0, "mixed", 4)
    a.moving_prey = True
    with pytest.raises(NotImplementedError):
        bad.multi_agent_init(a)
    with pytest.raises(AssertionError):
        make_tj(5, 7, 0, "medium", 2)


def test_reference_style_single_env_calls():
    """E = 1 drop-in: list-of-arrays actions through GymWrapper like the reference's Trainer passes them."""
    from ic3net_amd import data
    import oracle
    a = pp_args(3, 5, 1, "mixed", 1, seed=4, offset=17)
    a.display = False
    w = data.init('predator_prey', a, False)
    assert (w.observation_dim, w.num_actions, w.dim_actions) == (9 * 29, 5, 1)
    o = oracle.PPOracle(3, 5, 1, seed=4, env_gid=17)
    obs = w.reset(0)
    assert tuple(obs.shape) == (1, 3, 9 * 29) and np.array_equal(obs[0].cpu().numpy(), o.reset())
    for t in range(6):
        act = [np.array([t % 5, (t + 1) % 5, 4]), np.array([0, 1, 0])]        # [env head, talk head]
        obs, r, done, info = w.step(act)
        oo, orew, od = o.step(act[0])
        assert np.array_equal(obs[0].cpu().numpy(), oo) and np.array_equal(r[0].cpu().numpy(), orew.astype(np.float32))
    assert not w.reward_terminal().any().item()
    st = w.get_stat()
    assert st['success'] in (0.0, 1.0)
    with pytest.raises(AssertionError):
        w.env.step(np.zeros(4))                                               # wrong number of agents


def test_set_state_injection_and_single_agent():
    import oracle
    env = make_pp(1, 3, 0, "mixed", 2, seed=0)
    env.reset()
    env.set_state(loc_r=[[0, 2], [2, 2]], loc_c=[[0, 2], [1, 2]], reached=[[0], [0]], over=[0, 0])
    obs, r, d, _ = env.step(np.array([[2], [1]]))          # env 0 moves down (not on prey), env 1 right -> on prey
    st = env.get_state()
    assert st["loc_r"].tolist() == [[1, 2], [2, 2]] and st["loc_c"].tolist() == [[0, 2], [2, 2]]
    assert r.cpu().numpy().tolist() == [[np.float32(-0.05)], [0.0]] and d.cpu().numpy().tolist() == [0, 1]
    assert st["success"].tolist() == [0, 1]


def test_tj_full_size_properties():
    """BASELINE config 4 per-GPU size (TJ-hard, 20 cars, E=8192): size-independent invariants after 30 steps."""
    E, N = 8192, 20
    env = make_tj(N, 18, 1, "hard", E, seed=2, add_rate_min=0.3, add_rate_max=0.3)
    env.reset(0)
    g = torch.Generator(device='cuda').manual_seed(0)
    for t in range(30):
        act = (torch.rand((E, N), device='cuda', generator=g) < 0.3).int()
        obs, rew, done, info = env.step(act)
    st = {k: torch.as_tensor(v).cuda() for k, v in env.get_state().items()}
    assert torch.equal(st["cars_in_sys"], st["alive"].sum(1).int())
    dead = st["alive"] == 0
    assert (st["loc_r"][dead] == 0).all() and (st["loc_c"][dead] == 0).all() and (st["wait"][dead] == 0).all()
    assert not obs[dead].any().item() and not rew[dead].any().item()
    rows = obs[~dead]
    win = rows[:, 2:].view(rows.shape[0], 9, -1)
    assert (win[:, :, :win.shape[2] - 1].sum(-1) == 1).all()
    assert torch.equal(info["alive_mask"], st["alive"])
    # crashes cost exactly -10 on top of -0.01*wait
    w = st["wait"].float()
    base = -0.01 * w.double()
    diff = (rew.double() - base.double())[~dead]
    assert (((diff.abs() < 1e-6) | ((diff + 10).abs() < 1e-5))).all()
    # shard invariance on the add_cars stream
    env2 = make_tj(N, 18, 1, "hard", 1024, seed=2, offset=4096, add_rate_min=0.3, add_rate_max=0.3)
    env2.reset(0)
    env3 = make_tj(N, 18, 1, "hard", 8192, seed=2, add_rate_min=0.3, add_rate_max=0.3)
    env3.reset(0)
    for t in range(5):
        z = torch.zeros((8192, N), dtype=torch.int32, device='cuda')
        o3, _, _, _ = env3.step(z)
        o2, _, _, _ = env2.step(z[:1024])
    assert torch.equal(o3[4096:5120], o2)


def test_pp_scaled_full_size_spotcheck():
    """BASELINE config 5 per-GPU size: 32 agents, dim 40, vision 2, E = 8192 -> a 42 GB observation tensor.
    Envs at the start, middle and far end of the tensor (64-bit offsets) are compared with the oracle."""
    import oracle
    E, N = 8192, 32
    env = make_pp(N, 40, 2, "mixed", E, seed=5)
    assert env.obs_dim == 40100
    obs = env.reset()
    assert obs.numel() == E * N * 40100
    picks = (0, 4097, 8191)
    orcs = {e: oracle.PPOracle(N, 40, 2, "mixed", seed=5, env_gid=e) for e in picks}
    for e, o in orcs.items():
        assert np.array_equal(obs[e].cpu().numpy(), o.reset()), e
    act = torch.randint(0, 5, (E, N), device="cuda", dtype=torch.int32)
    obs, r, d, _ = env.step(act)
    for e, o in orcs.items():
        oo, orew, od = o.step(act[e].cpu().numpy())
        assert np.array_equal(obs[e].cpu().numpy(), oo) and np.array_equal(r[e].cpu().numpy(), orew.astype(np.float32)), e
    # one location bit per window cell everywhere (checked on a slice to bound the temporary)
    o4 = obs[8000:8192].view(192, N, 25, 1604)
    assert (o4[..., :1602].sum(-1) == 1).all()
    del obs, env
    torch.cuda.empty_cache()


def test_pp_enemy_comm_vs_oracle_random():
    """enemy_comm (prey rows in obs / reward, N+1 action slots) at a larger size against the oracle."""
    import oracle
    N, dim, vision, E, T = 5, 8, 1, 64, 15
    env = make_pp(N, dim, vision, "cooperative", E, seed=21, offset=50, enemy_comm=True)
    orcs = [oracle.PPOracle(N, dim, vision, "cooperative", seed=21, env_gid=50 + e, enemy_comm=True) for e in range(E)]
    obs = env.reset().cpu().numpy()
    assert obs.shape == (E, N + 1, env.obs_dim)
    for e, o in enumerate(orcs):
        np.testing.assert_array_equal(obs[e], o.reset())
    rs = np.random.RandomState(3)
    lin = torch.nn.Linear(env.obs_dim, 32).cuda()
    for t in range(T):
        act = rs.randint(0, 5, size=(E, N + 1))
        obs, rew, done, _ = env.step(act)
        o_np, rew = obs.cpu().numpy(), rew.cpu().numpy()
        for e, o in enumerate(orcs):
            oo, orew, od = o.step(act[e])
            np.testing.assert_array_equal(o_np[e], oo)
            np.testing.assert_array_equal(rew[e], orew.astype(np.float32))
    with torch.no_grad():   # the sparse encoder covers the prey rows too
        dense = obs.double() @ lin.weight.double().t() + lin.bias.double()
        wt = lin.weight.detach().t().contiguous()
        sparse = env.encode(wt, lin.bias.detach())
        tabled = env.encode(wt, lin.bias.detach(), loc_table=env.encode_table(wt))
    torch.testing.assert_close(sparse.double(), dense, atol=2e-6, rtol=0)
    torch.testing.assert_close(tabled.double(), dense, atol=2e-6, rtol=0)


def test_tj_scalar_vocab_vs_oracle_and_encoder():
    """vocab_type='scalar' (traffic_junction_env.py:139-148): rows [last_act, route, r/(h-1), c/(w-1), (road, #cars)
    per window cell]; checked against the oracle on random steps, plus the sparse encoder on those rows."""
    import oracle
    N, dim, vision, diff, E, T, rate = 10, 14, 1, "medium", 48, 25, 0.4
    env = make_tj(N, dim, vision, diff, E, seed=4, offset=70, add_rate_min=rate, add_rate_max=rate, vocab_type='scalar')
    assert env.obs_dim == 4 + 9 * 2
    from ic3net_amd.env_wrappers import GymWrapper
    assert GymWrapper(env).observation_dim == env.obs_dim
    orcs = [oracle.TJOracle(N, dim, vision, diff, add_rate_min=rate, add_rate_max=rate, seed=4, env_gid=70 + e,
                            vocab_type='scalar') for e in range(E)]
    env.reset(0)
    for o in orcs:
        o.reset(0)
    rs = np.random.RandomState(8)
    lin = torch.nn.Linear(env.obs_dim, 64).cuda()
    for t in range(T):
        act = (rs.rand(E, N) < 0.35).astype(np.int32)
        obs, rew, done, info = env.step(act)
        o_np = obs.cpu().numpy()
        for e, o in enumerate(orcs):
            oo, orew, _ = o.step(act[e])
            np.testing.assert_array_equal(o_np[e], oo)
            np.testing.assert_array_equal(rew[e].cpu().numpy(), orew.astype(np.float32))
    with torch.no_grad():
        dense = obs.double() @ lin.weight.double().t() + lin.bias.double()
        wt = lin.weight.detach().t().contiguous()
        sparse = env.encode(wt, lin.bias.detach())
        tabled = env.encode(wt, lin.bias.detach(), loc_table=env.encode_table(wt))
    torch.testing.assert_close(sparse.double(), dense, atol=2e-6, rtol=0)
    torch.testing.assert_close(tabled.double(), dense, atol=2e-6, rtol=0)
    grid, off, rc = env.tables()
    assert set(np.unique(grid)) == {0, 1}                      # the reference's self.grid holds road flags here


def test_reset_to_a_given_state():
    """ic3_env_reset_to (SURVEY 8(b2) `init_state_or_null`): reset into a recorded state == reset + set_state + observe,
    and stepping from there reproduces the recorded trajectory."""
    envA = make_pp(5, 8, 1, "mixed", 6, seed=3)
    envA.reset()
    rs = np.random.RandomState(0)
    for _ in range(3):
        envA.step(rs.randint(0, 5, size=(6, 5)))
    snap = envA.get_state()
    obs_snap = envA.observe().clone()
    acts = rs.randint(0, 5, size=(4, 6, 5))
    want = [tuple(x.clone() for x in envA.step(a)[:3]) for a in acts]
    envB = make_pp(5, 8, 1, "mixed", 6, seed=3)
    envB.reset()                                              # a different state (fresh episode) ...
    obs = envB.reset_to(snap)                                 # ... replaced by the recorded one
    assert torch.equal(obs, obs_snap)
    for k in snap:
        np.testing.assert_array_equal(envB.get_state()[k], snap[k])
    for a, (o, r, d) in zip(acts, want):
        o2, r2, d2, _ = envB.step(a)
        assert torch.equal(o2, o) and torch.equal(r2, r) and torch.equal(d2, d)
    tj = make_tj(5, 6, 1, "easy", 3, seed=1, add_rate_min=0.5, add_rate_max=0.5)
    tj.reset(0)
    tj.step(np.zeros((3, 5), np.int32))
    st = tj.get_state()
    o1 = tj.observe().clone()
    tj.step(np.ones((3, 5), np.int32))
    assert torch.equal(tj.reset_to(st, epoch=0), o1)


def test_render_views_follow_the_reference_drawing():
    """env.render() of a state set through the handle == the draw calls the reference's curses code makes for that state
    (tests/golden/render_fixture.json, recorded from predator_prey_env.py:307-336 / traffic_junction_env.py:254-292)."""
    import json
    import os
    fx = json.load(open(os.path.join(os.path.dirname(__file__), 'golden', 'render_fixture.json')))
    pp = [o for o in fx if o['env'] == 'pp' and o['N'] == 3]
    env = make_pp(3, 5, 1, "mixed", len(pp), seed=0)
    env.reset()
    env.set_state(loc_r=[o['loc_r'] for o in pp], loc_c=[o['loc_c'] for o in pp])
    for k, o in enumerate(pp):
        assert env.render(mode='cells', env_index=k) == o['cells']
    assert 'XXXP' in env.render(mode='ansi', env_index=1)
    for o in [o for o in fx if o['env'] == 'tj' and o['difficulty'] in ('medium', 'hard')][::3]:
        tj = make_tj(o['N'], o['dim'], o['vision'], o['difficulty'], 1, seed=1)
        tj.reset(0)
        tj.set_state(alive=[o['alive']], loc_r=[o['loc_r']], loc_c=[o['loc_c']], last_act=[o['last_act']])
        assert tj.render(mode='cells') == o['cells'], (o['difficulty'], o['t'])


def test_max_agents_per_env_vs_oracle():
    """N = 64 (one env per wavefront, the full 64-bit ballot mask) for PP, and N = 48 cars for TJ-hard."""
    import oracle
    E, T = 6, 10
    env = make_pp(64, 10, 1, "cooperative", E, seed=8, offset=3)
    orcs = [oracle.PPOracle(64, 10, 1, "cooperative", seed=8, env_gid=3 + e) for e in range(E)]
    obs = env.reset().cpu().numpy()
    for e, o in enumerate(orcs):
        np.testing.assert_array_equal(obs[e], o.reset())
    rs = np.random.RandomState(1)
    for t in range(T):
        act = rs.randint(0, 5, size=(E, 64))
        obs, rew, done, _ = env.step(act)
        for e, o in enumerate(orcs):
            oo, orew, od = o.step(act[e])
            np.testing.assert_array_equal(obs[e].cpu().numpy(), oo)
            np.testing.assert_array_equal(rew[e].cpu().numpy(), orew.astype(np.float32))
    tj = make_tj(48, 18, 1, "hard", E, seed=8, offset=3, add_rate_min=0.9, add_rate_max=0.9)
    torcs = [oracle.TJOracle(48, 18, 1, "hard", add_rate_min=0.9, add_rate_max=0.9, seed=8, env_gid=3 + e)
             for e in range(E)]
    tj.reset(0)
    for o in torcs:
        o.reset(0)
    for t in range(25):
        act = (rs.rand(E, 48) < 0.5).astype(np.int32)
        obs, rew, done, info = tj.step(act)
        st = tj.get_state()
        for e, o in enumerate(torcs):
            oo, orew, _ = o.step(act[e])
            np.testing.assert_array_equal(st["alive"][e], o.alive)
            np.testing.assert_array_equal(st["route_id"][e], o.route_id)
            np.testing.assert_array_equal(obs[e].cpu().numpy(), oo)
            np.testing.assert_array_equal(rew[e].cpu().numpy(), orew.astype(np.float32))
    from ic3net_amd.envs import PredatorPreyEnv
    with pytest.raises(ValueError):
        PredatorPreyEnv().multi_agent_init(pp_args(65, 10, 1, "mixed", 2))       # N > 64 is rejected, not mis-simulated


def test_randomized_config_sweep_vs_oracle():
    """40 random Predator-Prey and 30 random Traffic-Junction configurations (odd/even dims -> both obs store paths,
    vision 0..2, every lane-group size, all modes / difficulties / vocab types, enemy_comm) for a few steps each,
    compared env by env with the oracle."""
    import oracle
    rs = np.random.RandomState(2024)
    for trial in range(40):
        dim = int(rs.randint(2, 13))
        N = int(rs.randint(1, min(12, dim * dim - 1) + 1))
        v = int(rs.randint(0, 3))
        mode = ["mixed", "cooperative", "competitive"][rs.randint(3)]
        ec = bool(rs.rand() < 0.3)
        no_stay = bool(rs.rand() < 0.2)
        E = int(rs.randint(1, 20))
        seed, off = int(rs.randint(1 << 30)), int(rs.randint(1 << 20))
        env = make_pp(N, dim, v, mode, E, seed=seed, offset=off, no_stay=no_stay, enemy_comm=ec)
        orcs = [oracle.PPOracle(N, dim, v, mode, stay=not no_stay, seed=seed, env_gid=off + e, enemy_comm=ec)
                for e in range(E)]
        R = N + (1 if ec else 0)
        obs = env.reset().cpu().numpy()
        for e, o in enumerate(orcs):
            np.testing.assert_array_equal(obs[e], o.reset(), err_msg="pp reset trial %d" % trial)
        for t in range(6):
            act = rs.randint(0, 4 if no_stay else 5, size=(E, R))
            obs, rew, done, _ = env.step(act)
            obs, rew, done = obs.cpu().numpy(), rew.cpu().numpy(), done.cpu().numpy()
            for e, o in enumerate(orcs):
                if o.over.value:
                    assert done[e] == 1
                    continue
                oo, orew, od = o.step(act[e])
                np.testing.assert_array_equal(obs[e], oo, err_msg="pp obs trial %d" % trial)
                np.testing.assert_array_equal(rew[e], orew.astype(np.float32), err_msg="pp reward trial %d" % trial)
                assert done[e] == int(od)
    tj_dims = {"easy": [6, 8, 10], "medium": [6, 8, 10, 14], "hard": [9, 12, 15, 18]}
    for trial in range(30):
        diff = ["easy", "medium", "hard"][rs.randint(3)]
        dim = int(tj_dims[diff][rs.randint(len(tj_dims[diff]))])
        v = int(rs.randint(0, 3))
        if diff != "hard" and dim < 4 + v:
            v = 0
        N = int(rs.randint(1, 25))
        rate = float([0.05, 0.3, 0.7, 1.0][rs.randint(4)])
        vt = "scalar" if rs.rand() < 0.3 else "bool"
        E = int(rs.randint(1, 12))
        seed, off = int(rs.randint(1 << 30)), int(rs.randint(1 << 20))
        env = make_tj(N, dim, v, diff, E, seed=seed, offset=off, add_rate_min=rate, add_rate_max=rate, vocab_type=vt)
        orcs = [oracle.TJOracle(N, dim, v, diff, add_rate_min=rate, add_rate_max=rate, seed=seed, env_gid=off + e,
                                vocab_type=vt) for e in range(E)]
        env.reset(0)
        for o in orcs:
            o.reset(0)
        for t in range(10):
            act = (rs.rand(E, N) < 0.4).astype(np.int32)
            obs, rew, done, info = env.step(act)
            obs, rew = obs.cpu().numpy(), rew.cpu().numpy()
            st = env.get_state()
            for e, o in enumerate(orcs):
                oo, orew, _ = o.step(act[e])
                np.testing.assert_array_equal(st["alive"][e], o.alive, err_msg="tj alive trial %d" % trial)
                np.testing.assert_array_equal(st["route_id"][e], o.route_id)
                np.testing.assert_array_equal(obs[e], oo, err_msg="tj obs trial %d (%s dim %d v %d N %d %s)" %
                                              (trial, diff, dim, v, N, vt))
                np.testing.assert_array_equal(rew[e], orew.astype(np.float32))


def test_hip_reproduces_reference_checksum_sweep():
    """The same 210-configuration checksum sweep recorded from the reference, through the HIP path (one handle per
    configuration, E = 1 like the reference)."""
    from golden_util import crc_of, SWEEP_RATES
    fx = load("sweep_checksums")
    seed = int(fx["seed"])
    for cfg, acts, crcs in zip(fx["pp_cfg"], fx["pp_act"], fx["pp_crc"]):
        N, dim, v, mode, ec, ns, gid = [int(x) for x in cfg]
        env = make_pp(N, dim, v, MODES[mode], 1, seed=seed, offset=gid, no_stay=bool(ns), enemy_comm=bool(ec))
        obs = env.reset().cpu().numpy()[0]
        st = env.get_state()
        loc = np.stack([st["loc_r"][0], st["loc_c"][0]], -1)
        assert crc_of(loc[:N], loc[N:], obs) == crcs[0], cfg
        over = False
        for t in range(acts.shape[0]):
            if over:
                assert crcs[t + 1] == 0
                continue
            obs, rew, done, _ = env.step(acts[t:t + 1, :N + ec])
            st = env.get_state()
            loc = np.stack([st["loc_r"][0], st["loc_c"][0]], -1)
            over = bool(done.cpu().numpy()[0])
            got = crc_of(loc[:N], st["reached"][0], rew.cpu().numpy()[0], obs.cpu().numpy()[0], np.int32(int(over)))
            assert got == crcs[t + 1], (cfg, t)
    for cfg, acts, crcs in zip(fx["tj_cfg"], fx["tj_act"], fx["tj_crc"]):
        N, dim, v, diff, rate_i, scalar, gid = [int(x) for x in cfg]
        r = SWEEP_RATES[rate_i]
        env = make_tj(N, dim, v, DIFFS[diff], 1, seed=seed, offset=gid, add_rate_min=r, add_rate_max=r,
                      vocab_type='scalar' if scalar else 'bool')
        env.reset(0)
        for t in range(acts.shape[0]):
            obs, rew, _, _ = env.step(acts[t:t + 1, :N])
            st = env.get_state()
            loc = np.stack([st["loc_r"][0], st["loc_c"][0]], -1)
            got = crc_of(st["alive"][0], st["wait"][0], loc, st["last_act"][0], st["route_loc"][0], st["route_id"][0],
                         rew.cpu().numpy()[0], obs.cpu().numpy()[0])
            assert got == crcs[t], (cfg, t)


@pytest.mark.parametrize("kind", ["pp", "tj_zero_fill_rows", "tj_vec4_rows"])
def test_observe_at_reads_the_snapshot_not_the_live_state(kind):
    """ic3_env_observe_at: the rows of the state in `snap` whatever the handle's live state has become since (every launch
    geometry of the observation kernels; found by tests/test_host_abi_cpu.py — the Traffic-Junction zero-fill geometry read
    the live state)."""
    if kind == "pp":
        env = make_pp(5, 10, 1, "mixed", 6, seed=3)
    elif kind == "tj_zero_fill_rows":
        env = make_tj(6, 8, 1, "medium", 6, seed=3, add_rate_min=0.5, add_rate_max=0.5)
        assert env.nagents_env * env.obs_dim < 8192
    else:
        env = make_tj(20, 18, 1, "hard", 4, seed=3, add_rate_min=0.5, add_rate_max=0.5)
        assert env.nagents_env * env.obs_dim >= 8192
    rng = np.random.default_rng(4)
    E, N, A = env.nenvs, env.nagents_env, env.dims.naction
    env.reset() if kind == "pp" else env.reset(0)
    for _ in range(6):
        env.step(rng.integers(0, A, (E, N)))
    want = env.observe().clone()
    snap = env.snapshot()
    for _ in range(3):
        env.step(rng.integers(0, A, (E, N)))
    assert not torch.equal(env.observe(), want)                        # the live state has moved on
    assert torch.equal(env.observe_timed(snap), want)
    assert not torch.equal(env.observe(), want)


def test_gpu_library_refuses_the_host_device():
    """device = -1 is the host build's (tests/host/libic3rollout_host.so); the product has no CPU path."""
    import ctypes as C
    from ic3net_amd import _lib
    cfg = _lib.PPCfg(2, 3, 1, 5, 0, 0, 1, 0, 0, 0, 0)
    h = C.c_void_p()
    assert _lib.lib().ic3_pp_create(C.byref(cfg), -1, C.byref(h)) < 0
    assert b"hipSetDevice" in _lib.lib().ic3_last_error()
    torch.cuda.synchronize()                                           # the refusal leaves no sticky runtime error behind
    assert torch.zeros(4, device='cuda').sum().item() == 0
