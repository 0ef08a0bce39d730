"""CPU: the product's own kernels behind the same C ABI on a CPU (`device = -1`).

tests/host/libic3rollout_host.so is every ic3net_amd/csrc/*.hip and tj_tables.cpp — the files libic3rollout.so is built
from, unmodified — compiled as C++ against a stand-in HIP runtime
(tests/host/shim: a launch = workgroups of cooperatively scheduled lane fibers, LDS poisoned before every workgroup, device
memory = malloc).
It exports the entry points of include/ic3_rollout.h ("identical entry points exist in the CPU build
(device = -1, hipStream_t ignored) so the same tests drive both"); the bodies below are the GPU parity tests' bodies
(tests/test_env_parity_gpu.py, test_encode_backward_gpu.py) on numpy buffers.  With IC3_HOST_ASAN=1 (tools/host_asan.sh) the
same tests run under AddressSanitizer + UndefinedBehaviorSanitizer: every index a kernel forms is bounds-checked.
Integer state bit-exact; rewards as float32(reference float64) bit-exact; observations bit-exact; fp32 sums 1e-5."""
import ctypes as C

import numpy as np
import pytest

from golden_util import load, SparseObs, PP_FIXTURES, TJ_FIXTURES, MODES, DIFFS
from host_abi_util import ASAN, HostEnv, host_lib, check, p


@pytest.mark.parametrize("name", PP_FIXTURES)
def test_pp_kernels_match_reference_golden(name):
    fx = load(name)
    N, dim, vision, mode, T, no_stay = [int(x) for x in fx["cfg"]]
    nenv, nep = fx["nsteps"].shape
    ec = bool(int(fx["enemy_comm"]))
    sp = SparseObs(fx["obs_coo"], N + (1 if ec else 0), int(fx["obs_dim"]))
    env = HostEnv.pp(N, dim, vision, MODES[mode], nenv, seed=int(fx["seed"]), offset=int(fx["env_gid0"]),
                     no_stay=bool(no_stay), enemy_comm=ec)
    assert env.obs_dim == int(fx["obs_dim"]) and env.N == N + (1 if ec else 0)
    for ep in range(nep):
        obs = env.reset()
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
    env.close()


@pytest.mark.parametrize("name", TJ_FIXTURES)
def test_tj_kernels_match_reference_golden(name):
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
    groups = [[e] for e in range(nenv)] if has_curr else [list(range(nenv))]
    for grp in groups:
        env = HostEnv.tj(N, dim, vision, DIFFS[diff], len(grp), seed=int(fx["seed"]), offset=int(fx["env_gid0"]) + grp[0],
                         **kw)
        assert env.obs_dim == int(fx["obs_dim"])
        for ep in range(nep):
            obs = env.reset(int(fx["epochs"][grp[0], ep]))
            assert not obs.any()
            for t in range(T):
                obs, rew, done, info = env.step(fx["actions"][grp, ep, t])
                st = env.get_state()
                st["loc"] = np.stack([st["loc_r"], st["loc_c"]], -1)
                for k in ("alive", "wait", "loc", "last_act", "route_loc", "route_id", "is_completed",
                          "cars_in_sys", "has_failed"):
                    np.testing.assert_array_equal(st[k], fx[k][grp, ep, t], err_msg="%s ep=%d t=%d" % (k, ep, t))
                np.testing.assert_array_equal(info["alive_mask"], fx["alive"][grp, ep, t])
                np.testing.assert_array_equal(info["is_completed"], fx["is_completed"][grp, ep, t])
                np.testing.assert_array_equal(rew, fx["reward"][grp, ep, t].astype(np.float32))
                assert env.add_rate == fx["add_rate_seen"][grp[0], ep, t]
                assert not done.any()
                for i, e in enumerate(grp):
                    np.testing.assert_array_equal(obs[i], sp.dense(e, ep, t + 1))
        env.check_actions()
        env.close()


def test_tj_tables_through_handle():
    fx = load("tj_tables")
    for key in ("medium_14_v1", "hard_18_v0", "easy_6_v0"):
        diff, dim, v = key.split("_")
        env = HostEnv.tj(5, int(dim), int(v[1:]), diff, 2)
        grid, off, rc = env.tables()
        np.testing.assert_array_equal(grid, fx[key + "_grid"])
        np.testing.assert_array_equal(off, fx[key + "_off"])
        np.testing.assert_array_equal(rc, fx[key + "_rc"])
        env.close()


def _play(env, steps, seed):
    rng = np.random.default_rng(seed)
    env.reset(0)
    for _ in range(steps):
        env.step(rng.integers(0, env.dims.naction, (env.E, env.N)), with_obs=False)


ENC_CASES = [('pp', (3, 5, 0, 'mixed', 5)), ('pp', (5, 10, 1, 'cooperative', 3)), ('pp', (4, 7, 2, 'mixed', 2)),
             ('pp', (3, 6, 1, 'mixed', 4, dict(enemy_comm=True))),
             ('tj', (5, 6, 0, 'easy', 4, dict(add_rate_min=0.6, add_rate_max=0.6))),
             ('tj', (6, 8, 1, 'medium', 3, dict(add_rate_min=0.5, add_rate_max=0.5))),
             ('tj', (6, 9, 1, 'hard', 2, dict(add_rate_min=0.5, add_rate_max=0.5, vocab_type='scalar')))]


def _make(kind, cfg):
    kw = cfg[5] if len(cfg) > 5 else {}
    return (HostEnv.pp if kind == 'pp' else HostEnv.tj)(*cfg[:5], seed=11, offset=40, **kw)


@pytest.mark.parametrize("kind,cfg", ENC_CASES)
def test_sparse_encoder_equals_dense_rows_times_weight(kind, cfg):
    """ic3_env_encode / _encode_at / _encode_table (comm.py:119 without the dense observation): obs @ Wt + bias."""
    env = _make(kind, cfg)
    _play(env, 6, 3)
    H = 16
    rng = np.random.default_rng(5)
    wt = rng.standard_normal((env.obs_dim, H)).astype(np.float32)
    bias = rng.standard_normal(H).astype(np.float32)
    obs = env.observe()
    want = obs.reshape(-1, env.obs_dim).astype(np.float64) @ wt.astype(np.float64) + bias
    got = env.encode(wt, bias)
    np.testing.assert_allclose(got.reshape(-1, H), want, rtol=0, atol=1e-5)
    table = env.encode_table(wt)
    got_t = env.encode(wt, bias, loc_table=table, ldo=2 * H)          # into the left half of a wider row, as the update does
    np.testing.assert_allclose(got_t.reshape(-1, H), want, rtol=0, atol=1e-5)
    snap = env.snapshot()
    env.step(np.zeros((env.E, env.N), np.int32), with_obs=False)
    np.testing.assert_array_equal(env.observe(snap), obs)             # ic3_env_observe_at: the rows of the snapshot
    np.testing.assert_array_equal(env.encode(wt, bias, snap=snap), got)
    env.close()


@pytest.mark.parametrize("kind,cfg", ENC_CASES)
def test_encoder_backward_equals_dense_transpose_product(kind, cfg):
    """ic3_env_encode_backward (row-parallel form and its fallback): dWt = obs^T @ g, dbias = column sums of g."""
    env = _make(kind, cfg)
    _play(env, 5, 9)
    H = 8
    rng = np.random.default_rng(6)
    g = rng.standard_normal((env.E * env.N, H)).astype(np.float32)
    obs = env.observe().reshape(-1, env.obs_dim).astype(np.float64)
    dwt, db = env.encode_backward(g)
    np.testing.assert_allclose(dwt, obs.T @ g.astype(np.float64), rtol=0, atol=2e-5)
    np.testing.assert_allclose(db, g.astype(np.float64).sum(0), rtol=0, atol=2e-5)
    snap = env.snapshot()
    env.step(np.ones((env.E, env.N), np.int32), with_obs=False)
    dwt2, _ = env.encode_backward(g, snap=snap, want_bias=False)
    np.testing.assert_allclose(dwt2, dwt, rtol=0, atol=2e-5)           # (atomic accumulation order is free)
    env.close()


@pytest.mark.parametrize("kind,cfg", ENC_CASES)
def test_encoder_backward_accumulated_over_states(kind, cfg):
    """ic3_env_encode_backward_accumulate / _finish on the host: three states (snapshots), one expansion = the sum of the
    three dense products; the work buffer starts as NaN (the `first` call must write, not add)."""
    from host_abi_util import check, host_lib, p
    env = _make(kind, cfg)
    lib = host_lib()
    H = 8
    rng = np.random.default_rng(8)
    n = int(lib.ic3_env_encode_backward_work(env._h, H))
    work = np.full((n,), np.nan, np.float32)
    want, wantb = 0.0, 0.0
    for k in range(3):
        _play(env, 3 + k, 20 + k)
        snap = env.snapshot()
        obs = env.observe().reshape(-1, env.obs_dim).astype(np.float64)
        wide = rng.standard_normal((env.E * env.N, 2 * H)).astype(np.float32)
        g = wide[:, H:]                                              # strided rows (ldg = 2H)
        env.step(np.zeros((env.E, env.N), np.int32), with_obs=False)
        rc = lib.ic3_env_encode_backward_accumulate(env._h, p(snap), C.c_void_p(wide.ctypes.data + 4 * H), 2 * H, H, p(work),
                                                    int(k == 0), None)
        if rc == -38:
            env.close()
            pytest.skip("this configuration's first stage has no partial-sums form (the caller keeps the per-state calls)")
        check(rc)
        want = want + obs.T @ g.astype(np.float64)
        wantb = wantb + g.astype(np.float64).sum(0)
    dwt = np.full((env.obs_dim, H), np.nan, np.float32)
    db = np.full((H,), np.nan, np.float32)
    check(lib.ic3_env_encode_backward_finish(env._h, H, p(dwt), p(db), p(work), None))
    np.testing.assert_allclose(dwt, want, rtol=0, atol=4e-5)
    np.testing.assert_allclose(db, wantb, rtol=0, atol=4e-5)
    env.close()


@pytest.mark.parametrize("kind,cfg", ENC_CASES)
def test_encoder_backward_over_a_window_of_states_in_one_launch(kind, cfg):
    """ic3_env_encode_backward_window / _window_finish on the host: the first stage over T recorded states in ONE launch (position
    sums as one-hot x gradient products, the shared columns' weights from a per-batch table) = the sum of the dense products
    obs_t^T @ g_t; called twice (first = 1, then 0: a second window adds), on strided rows of a [T][R][2H] ring; the work buffer
    starts as NaN."""
    from host_abi_util import check, host_lib, p
    env = _make(kind, cfg)
    lib = host_lib()
    H, T = 32, 3
    rng = np.random.default_rng(9)
    n = int(lib.ic3_env_encode_backward_window_work(env._h, H))
    assert n > 0
    assert int(lib.ic3_env_encode_backward_window_work(env._h, 24)) == 0       # hid_size must be a multiple of 32
    work = np.full((n,), np.nan, np.float32)
    R = env.E * env.N
    want, wantb = 0.0, 0.0
    for win in range(2):
        snaps, ring = [], rng.standard_normal((T, R, 2 * H)).astype(np.float32)
        for k in range(T):
            _play(env, 2 + k, 30 + 7 * win + k)
            snaps.append(env.snapshot())
            obs = env.observe().reshape(-1, env.obs_dim).astype(np.float64)
            g = ring[k, :, :H]
            want = want + obs.T @ g.astype(np.float64)
            wantb = wantb + g.astype(np.float64).sum(0)
        snaps = np.ascontiguousarray(np.stack(snaps))
        env.step(np.zeros((env.E, env.N), np.int32), with_obs=False)        # the live state moves on: the snapshots count
        check(lib.ic3_env_encode_backward_window(env._h, p(snaps), snaps.shape[1], T, p(ring), 2 * H, R * 2 * H, H, p(work),
                                                 int(win == 0), None))
    dwt = np.full((env.obs_dim, H), np.nan, np.float32)
    db = np.full((H,), np.nan, np.float32)
    check(lib.ic3_env_encode_backward_window_finish(env._h, H, p(dwt), p(db), p(work), None))
    np.testing.assert_allclose(dwt, want, rtol=0, atol=4e-5)
    np.testing.assert_allclose(db, wantb, rtol=0, atol=4e-5)
    env.close()


@pytest.mark.parametrize("T,E,N,heads,entr,norm", [(7, 9, 3, [5, 2], 0.01, False), (11, 30, 10, [5], 0.0, True), (3, 2, 20, [2, 2, 3], 0.1, False)])
def test_loss_gradients_in_one_launch_on_the_host(T, E, N, heads, entr, norm):
    """ic3_loss_gradients == the losses of /root/reference/trainer.py:173-218 and their gradients w.r.t. [logits | value] (float64
    numpy, the log-softmax folded in): rows of (slot, agent), 257+ rows so that several workgroups' partials are summed."""
    from host_abi_util import check, host_lib, p
    lib = host_lib()
    rng = np.random.default_rng(T * 100 + E)
    R, OT, nh = E * N, sum(heads) + 1, len(heads)
    logits = rng.standard_normal((T, R, OT))
    out = np.empty((T, R, OT), np.float32)
    off = 0
    for A in heads:
        z = logits[:, :, off:off + A]
        out[:, :, off:off + A] = z - np.log(np.exp(z).sum(2, keepdims=True))
        off += A
    out[:, :, off] = logits[:, :, off]
    action = np.stack([rng.integers(0, A, size=(T, R)) for A in heads], 1).astype(np.int32)        # (T, heads, R)
    returns = rng.standard_normal((T, R)).astype(np.float32)
    live = (rng.random((T, E)) < 0.8).astype(np.float32)
    alive = ((rng.random((T, R)) < 0.9) * np.repeat(live, N, axis=1)).astype(np.float32)
    vc = 0.03
    o64, ret, al, lv = out.astype(np.float64), returns.astype(np.float64), alive.astype(np.float64), np.repeat(live, N, axis=1).astype(np.float64)
    adv = ret - o64[:, :, -1]
    shift, scale = 0.0, 1.0
    if norm:
        cnt = lv.sum()
        shift = float((adv * lv).sum() / cnt)
        scale = float(1.0 / np.sqrt((((adv - shift) ** 2) * lv).sum() / (cnt - 1)))
    adv = (adv - shift) * scale
    want = np.zeros((T, R, OT))
    act_loss = ent = 0.0
    off = 0
    for k, A in enumerate(heads):
        lp = o64[:, :, off:off + A]
        onehot = np.eye(A)[action[:, k]]
        dlp = onehot * (-adv * al)[:, :, None] + entr * lv[:, :, None] * np.exp(lp) * (lp + 1.0) * (entr > 0)
        want[:, :, off:off + A] = dlp - np.exp(lp) * dlp.sum(2, keepdims=True)
        act_loss += (-adv * (lp * onehot).sum(2) * al).sum()
        ent -= (lp * np.exp(lp) * lv[:, :, None]).sum()
        off += A
    want[:, :, off] = 2.0 * vc * (o64[:, :, -1] - ret) * al
    val_loss = (((o64[:, :, -1] - ret) ** 2) * al).sum()
    nparts = int(lib.ic3_loss_gradients_partials(T, R))
    assert nparts == (T * R + 255) // 256
    d_out = np.full((T, R, OT), np.nan, np.float32)
    sums = np.full((nparts, 3), np.nan, np.float64)
    sizes = np.array(heads, np.int32)
    check(lib.ic3_loss_gradients(p(out), p(action), p(returns), p(alive), p(live), p(sizes), nh, C.c_float(shift), C.c_float(scale),
                                 C.c_float(entr), C.c_float(vc), p(d_out), p(sums), T, E, N, None))
    np.testing.assert_allclose(d_out, want, rtol=0, atol=2e-6 * max(1.0, np.abs(want).max()))
    got = sums.sum(0)
    for g, w_ in zip(got, (act_loss, val_loss, ent)):
        assert abs(g - w_) <= 2e-6 * max(1.0, abs(w_)), (got, act_loss, val_loss, ent)
    assert lib.ic3_loss_gradients(p(out), p(action), p(returns), p(alive), p(live), p(sizes), 5, C.c_float(0), C.c_float(1),
                                  C.c_float(0), C.c_float(vc), p(d_out), p(sums), T, E, N, None) == -22      # at most four heads


@pytest.mark.parametrize("T,E,N,gamma,ratio", [(9, 7, 3, 1.0, 0.0), (12, 30, 10, 0.9, 0.5), (5, 3, 64, 1.0, 1.0)])
def test_returns_scan_on_the_host(T, E, N, gamma, ratio):
    """ic3_returns_scan == the loop of /root/reference/trainer.py:162-171 (float64)."""
    from host_abi_util import check, host_lib, p
    rng = np.random.default_rng(T)
    rew = rng.standard_normal((T, E, N)).astype(np.float32)
    em = (rng.random((T, E)) < 0.8).astype(np.float32)
    mm = (rng.random((T, E, N)) < 0.8).astype(np.float32)
    out = np.full((T, E, N), np.nan, np.float32)
    check(host_lib().ic3_returns_scan(p(rew), p(em), p(mm), gamma, ratio, p(out), T, E, N, None))
    coop = np.zeros((T, E, N))
    ncoop = np.zeros((T, E, N))
    pc = np.zeros((E, N))
    pn = np.zeros((E, N))
    for i in reversed(range(T)):
        coop[i] = rew[i] + gamma * pc * em[i][:, None]
        ncoop[i] = rew[i] + gamma * pn * em[i][:, None] * mm[i]
        pc, pn = coop[i], ncoop[i]
    ref = ratio * coop.mean(2, keepdims=True) + (1 - ratio) * ncoop
    np.testing.assert_allclose(out, ref, rtol=0, atol=2e-5 * max(1.0, np.abs(ref).max()))
    assert host_lib().ic3_returns_scan(p(rew), p(em), p(mm), gamma, ratio, None, T, E, N, None) == -22


def test_comm_masked_mean_add_on_the_host():
    """ic3_comm_masked_mean_add: the block's output + an addend read through a row stride."""
    from host_abi_util import check, host_lib, p
    E, N, H = 5, 7, 16
    rng = np.random.default_rng(2)
    h = rng.standard_normal((E, N, H)).astype(np.float32)
    alive = (rng.random((E, N)) < 0.8).astype(np.int32)
    gate = (rng.random((E, N)) < 0.6).astype(np.int32)
    wide = rng.standard_normal((E * N, 2 * H)).astype(np.float32)
    lib = host_lib()
    base = np.full((E, N, H), np.nan, np.float32)
    check(lib.ic3_comm_masked_mean(p(h), 0, p(alive), p(gate), p(base), E, N, H, 1, 1, None))
    out = np.full((E, N, H), np.nan, np.float32)
    check(lib.ic3_comm_masked_mean_add(p(h), 0, p(alive), p(gate), C.c_void_p(wide.ctypes.data + 4 * H), 2 * H, None, p(out), E, N, H, 1,
                                       1, None))
    np.testing.assert_array_equal(out.reshape(E * N, H), wide[:, H:] + base.reshape(E * N, H))
    scale = (rng.random(E * N) < 0.7).astype(np.float32)           # per-row factor (collection mode: 0 across an episode boundary)
    out2 = np.full((E, N, H), np.nan, np.float32)
    check(lib.ic3_comm_masked_mean_add(p(h), 0, p(alive), p(gate), C.c_void_p(wide.ctypes.data + 4 * H), 2 * H, p(scale), p(out2), E, N,
                                       H, 1, 1, None))
    np.testing.assert_array_equal(out2.reshape(E * N, H), out.reshape(E * N, H) * scale[:, None])
    check(lib.ic3_comm_masked_mean_add(p(h), 0, p(alive), p(gate), C.c_void_p(wide.ctypes.data + 4 * H), 2 * H, p(scale), p(out2), E, N,
                                       H, 1, 0, None))                    # comm_mask_zero: the scaled addend alone
    np.testing.assert_array_equal(out2.reshape(E * N, H), wide[:, H:] * scale[:, None])
    assert lib.ic3_comm_masked_mean_add(p(h), 0, p(alive), p(gate), None, 0, None, p(out), E, N, H, 1, 1, None) == -22


def test_stats_and_state_round_trip():
    env = HostEnv.pp(3, 5, 0, 'mixed', 6, seed=2)                    # (episode_over: mixed mode only, PP:285-286)
    env.reset()
    st = env.get_state()
    # put everyone on the prey in env 0 and 3: the next step ends those episodes successfully (PP:254-290)
    for e in (0, 3):
        st["loc_r"][e, :] = st["loc_r"][e, -1]
        st["loc_c"][e, :] = st["loc_c"][e, -1]
    env.set_state(loc_r=st["loc_r"], loc_c=st["loc_c"])
    np.testing.assert_array_equal(env.get_state()["loc_r"], st["loc_r"])
    _, rew, done, _ = env.step(np.full((6, 3), 4, np.int32))          # STAY
    np.testing.assert_array_equal(done, [1, 0, 0, 1, 0, 0])
    s = env.stats()
    assert s.success_sum == 2.0 and s.episodes == 6 and s.live_env_steps == 6
    env.step(np.full((6, 3), 7, np.int32))                            # out of range: sticky flag, PP:137
    with pytest.raises(AssertionError, match="range"):
        env.check_actions()
    env.close()


def test_comm_mean_cell_heads_and_draws_on_the_host():
    """policy_ops.hip's pointwise kernels (comm.py:131-170,194-203; torch.nn.LSTMCell; comm.py:228-239; action_utils.py:32-36)."""
    lib = host_lib()
    rng = np.random.default_rng(1)
    E, N, H = 5, 6, 32
    h = rng.standard_normal((E, N, H)).astype(np.float32)
    alive = (rng.random((E, N)) < 0.7).astype(np.int32)
    gate = (rng.random((E, N)) < 0.6).astype(np.int32)
    out = np.full((E, N, H), np.nan, np.float32)
    check(lib.ic3_comm_masked_mean(p(h), H, p(alive), p(gate), p(out), E, N, H, 1, 1, None))
    m = (alive * gate).astype(np.float64)
    n_alive = alive.sum(1)
    want = np.zeros((E, N, H))
    for e in range(E):
        tot = (m[e][:, None] * h[e]).sum(0)
        for j in range(N):
            c = (tot - m[e, j] * h[e, j]) * m[e, j]
            want[e, j] = c / (n_alive[e] - 1) if n_alive[e] > 1 else c
    np.testing.assert_allclose(out, want, rtol=0, atol=1e-5)

    R = E * N
    gates = rng.standard_normal((R, 4 * H)).astype(np.float32)
    c = rng.standard_normal((R, H)).astype(np.float32)
    c_in = c.copy()
    h_out = np.full((R, H), np.nan, np.float32)
    check(lib.ic3_lstm_cell(p(gates), p(c), p(h_out), H, R, H, None))
    sig = lambda x: 1 / (1 + np.exp(-x.astype(np.float64)))
    i, f, g, o = [gates[:, k * H:(k + 1) * H] for k in range(4)]
    c_want = sig(f) * c_in + sig(i) * np.tanh(g.astype(np.float64))
    np.testing.assert_allclose(c, c_want, rtol=0, atol=1e-5)
    np.testing.assert_allclose(h_out, sig(o) * np.tanh(c_want), rtol=0, atol=1e-5)

    heads = np.array([5, 2], np.int32)
    OT = int(heads.sum()) + 1
    W = (rng.standard_normal((OT, H)) * 0.3).astype(np.float32)
    b = rng.standard_normal(OT).astype(np.float32)
    logp = np.full((R, OT), np.nan, np.float32)
    check(lib.ic3_policy_heads(p(h_out), H, p(W), p(b), p(heads), 2, p(logp), R, H, None))
    z = h_out.astype(np.float64) @ W.T.astype(np.float64) + b
    off = 0
    for A in heads:
        zz = z[:, off:off + A]
        np.testing.assert_allclose(logp[:, off:off + A], zz - np.log(np.exp(zz).sum(1, keepdims=True)), rtol=0, atol=1e-5)
        off += A
    np.testing.assert_allclose(logp[:, off], z[:, off], rtol=0, atol=1e-5)

    # inverse-CDF draw on the Philox stream (DESIGN.md RNG contract), against the oracle's sampler
    import oracle
    from oracle import philox
    act = np.full((E, N), -1, np.int32)
    chosen = np.full((E, N), np.nan, np.float32)
    check(lib.ic3_sample_actions(p(logp), OT, 5, 1, 123, 7, 2, 9, p(act), p(chosen), E, N, None))
    lp3 = logp.reshape(E, N, OT)
    for e in range(E):
        for n in range(N):
            x = philox.x24(123, 7 + e, philox.DOMAIN_SAMPLE, 2, 9, 1 * N + n)
            assert act[e, n] == oracle.sample_one(lp3[e, n, :5], x)
    np.testing.assert_array_equal(chosen, np.take_along_axis(lp3, act[..., None], 2)[..., 0])
    ra = np.full((E, N), -1, np.int32)
    check(lib.ic3_random_actions(p(ra), 5, 11, 100, 2, 3, E, N, None))
    for e in range(E):
        for n in range(N):
            assert ra[e, n] == (philox.x24(11, 100 + e, philox.DOMAIN_BENCH, 2, 3, n) * 5) >> 24


def test_host_build_is_device_minus_one_only():
    from ic3net_amd import _lib as binding
    lib = host_lib()
    assert lib.ic3_version() >= 1
    cfg = binding.PPCfg(2, 3, 1, 5, 0, 0, 1, 0, 0, 0, 0)
    h = C.c_void_p()
    assert lib.ic3_pp_create(C.byref(cfg), 0, C.byref(h)) < 0          # a GPU ordinal: not in this library
    assert b"hipSetDevice" in lib.ic3_last_error()
    env = HostEnv.pp(3, 5, 0, 'mixed', 2)
    # the matrix-core kernels are in the host build too (tests/test_host_policy_step_cpu.py); same shape rules as on the GPU
    assert lib.ic3_policy_step_supported(env._h, 128) > 0 and lib.ic3_policy_step_supported(env._h, 96) == 0
    assert lib.ic3_lstm_gates_backward_supported(128) == 1 and lib.ic3_commnet_forward_supported(128, 3) == 1
    env.close()


FIN_CASES = [  # n, E, N, info (alive / is_completed present), gate: None | 'ones' | 'head', auto_reset, forced_last
    (20, 300, 10, False, 'head', False, True), (20, 300, 10, False, 'head', True, True), (7, 37, 3, True, None, False, False),
    (1, 5, 1, False, 'head', False, False), (3, 1, 5, True, 'head', True, False), (12, 70, 64, True, 'ones', False, True)]


@pytest.mark.parametrize("n,E,N,info,gate,auto,forced", FIN_CASES)
def test_episode_finalize_equals_tensor_ops(n, E, N, info, gate, auto, forced):
    """ic3_episode_finalize (episode_kernels.hip; trainer.py:70-105,109-110) on the host against the same derivations as
    tensor ops (Trainer._finalize_torch, which the GPU test compares with too): bit-equal masks, equal fp64 sums."""
    import torch
    from ic3net_amd import _lib as binding
    from ic3net_amd.trainer import Trainer
    lib = host_lib()
    g = torch.Generator(device='cpu').manual_seed(1000 * n + E + N)
    done = (torch.rand((n, E), generator=g) < 0.08).to(torch.int32)
    done[0, ::7] = -1
    reward = torch.randint(-40, 20, (n, E, N), generator=g).float() * 0.05
    alive = (torch.rand((n, E, N), generator=g) < 0.7).to(torch.int32) if info else None
    comp = (torch.rand((n, E, N), generator=g) < 0.3).to(torch.int32) if info else None
    gate_t, gate_stride = None, 0
    if gate == 'head':
        action = torch.randint(0, 2, (n, 2, E, N), generator=g, dtype=torch.int32)
        gate_t = action[:, -1]
        gate_stride = gate_t.stride(0) if n > 1 else E * N
    tp = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    nbytes = int(lib.ic3_episode_scratch_bytes(E, N))
    assert nbytes > 0
    scratch = torch.full((nbytes // 8,), float('nan'), dtype=torch.float64)
    counter = torch.zeros(1, dtype=torch.int32)
    want = Trainer._finalize_torch(n, done, reward, alive, comp, gate_t, gate == 'ones', auto, forced)
    for rep in range(2):                                    # second call reuses the scratch / counter
        f = lambda *shape: torch.full(shape, float('nan'), dtype=torch.float32)
        got = dict(live=f(n, E), alive_mask=f(n, E, N), episode_mask=f(n, E), episode_mini_mask=f(n, E, N), live_after=f(E),
                   stats=torch.full((2 + 2 * N,), float('nan'), dtype=torch.float64))
        ep = binding.Episode(n, E, N, int(auto), int(forced), int(gate == 'ones'), tp(done), tp(alive), tp(comp), tp(reward),
                             tp(gate_t), gate_stride, tp(got['live']), tp(got['alive_mask']), tp(got['episode_mask']),
                             tp(got['episode_mini_mask']), tp(got['live_after']), tp(got['stats']), tp(scratch), tp(counter))
        check(lib.ic3_episode_finalize(C.byref(ep), None))
        for k in ('live', 'alive_mask', 'episode_mask', 'episode_mini_mask', 'live_after'):
            assert got[k].shape == want[k].shape, k
            assert torch.equal(got[k], want[k]), k
        np.testing.assert_allclose(got['stats'].numpy(), want['stats'].numpy(), rtol=1e-13, atol=1e-9)
        assert int(counter.item()) == 0


@pytest.mark.skipif(not ASAN, reason="negative control of the sanitizer run (IC3_HOST_ASAN=1, tools/host_asan.sh)")
def test_sanitizer_sees_a_kernel_writing_past_its_buffer():
    """The sanitizers watch the kernels' own stores through the fiber switches: an observation buffer one row short makes
    the obs kernel's last stores land in the heap red zone, and the run must die with an AddressSanitizer report."""
    import os
    import subprocess
    import sys
    code = ("import numpy as np, ctypes as C\n"
            "from host_abi_util import HostEnv, p\n"
            "env = HostEnv.pp(3, 5, 0, 'mixed', 2)\n"
            "short = np.zeros((2 * 3 - 1) * env.obs_dim, np.float32)\n"
            "env.lib.ic3_env_reset(env._h, -1, p(short), None)\n"
            "print('survived')\n")
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.dirname(os.path.abspath(__file__)),
                                                       os.path.dirname(os.path.dirname(os.path.abspath(__file__)))]))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env)
    assert r.returncode != 0 and "survived" not in r.stdout
    assert "AddressSanitizer" in r.stderr and "heap-buffer-overflow" in r.stderr, r.stderr[-2000:]


def test_randomized_config_sweep_vs_oracle():
    """The GPU suite's randomized sweep (tests/test_env_parity_gpu.py, same seed, same 40 + 30 configurations) through the
    host ABI: random Predator-Prey and Traffic-Junction configurations (odd / even dims -> both obs store paths, vision 0..2, every lane-group size, all modes
    / difficulties / vocab types, enemy_comm, big rows -> the 1024-lane obs geometries) for a few steps each, env by env
    against the oracle.  Under the sanitizers this is the bounds check of every launch geometry of the env kernels."""
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
        env = HostEnv.pp(N, dim, v, mode, E, seed=seed, offset=off, no_stay=no_stay, enemy_comm=ec)
        orcs = [oracle.PPOracle(N, dim, v, mode, stay=not no_stay, seed=seed, env_gid=off + e, enemy_comm=ec)
                for e in range(E)]
        R = N + (1 if ec else 0)
        obs = env.reset()
        for e, o in enumerate(orcs):
            np.testing.assert_array_equal(obs[e], o.reset(), err_msg="pp reset trial %d" % trial)
        for t in range(6):
            act = rs.randint(0, 4 if no_stay else 5, size=(E, R))
            obs, rew, done, _ = env.step(act)
            for e, o in enumerate(orcs):
                if o.over.value:
                    assert done[e] == 1
                    continue
                oo, orew, od = o.step(act[e])
                np.testing.assert_array_equal(obs[e], oo, err_msg="pp obs trial %d" % trial)
                np.testing.assert_array_equal(rew[e], orew.astype(np.float32), err_msg="pp reward trial %d" % trial)
                assert done[e] == int(od)
        env.close()
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
        env = HostEnv.tj(N, dim, v, diff, E, seed=seed, offset=off, add_rate_min=rate, add_rate_max=rate, vocab_type=vt)
        orcs = [oracle.TJOracle(N, dim, v, diff, add_rate_min=rate, add_rate_max=rate, seed=seed, env_gid=off + e,
                                vocab_type=vt) for e in range(E)]
        env.reset(0)
        for o in orcs:
            o.reset(0)
        for t in range(10):
            act = (rs.rand(E, N) < 0.4).astype(np.int32)
            obs, rew, done, info = env.step(act)
            st = env.get_state()
            for e, o in enumerate(orcs):
                oo, orew, _ = o.step(act[e])
                np.testing.assert_array_equal(st["alive"][e], o.alive, err_msg="tj alive trial %d" % trial)
                np.testing.assert_array_equal(st["route_id"][e], o.route_id)
                np.testing.assert_array_equal(obs[e], oo, err_msg="tj obs trial %d (%s dim %d v %d N %d %s)" %
                                              (trial, diff, dim, v, N, vt))
                np.testing.assert_array_equal(rew[e], orew.astype(np.float32))
        env.close()


def test_reset_to_a_given_state():
    """ic3_env_reset_to (SURVEY 8(b2) `init_state_or_null`): reset into a recorded state, and stepping from there reproduces
    the recorded trajectory."""
    envA = HostEnv.pp(5, 8, 1, "mixed", 6, seed=3)
    envA.reset()
    rs = np.random.RandomState(0)
    for _ in range(3):
        envA.step(rs.randint(0, 5, size=(6, 5)))
    snap = envA.raw_state()
    obs_snap = envA.observe()
    acts = rs.randint(0, 5, size=(4, 6, 5))
    want = [envA.step(a)[:3] for a in acts]
    envB = HostEnv.pp(5, 8, 1, "mixed", 6, seed=3)
    envB.reset()                                              # a different state (fresh episode) ...
    obs = envB.reset_to(snap)                                 # ... replaced by the recorded one
    np.testing.assert_array_equal(obs, obs_snap)
    np.testing.assert_array_equal(envB.raw_state(), snap)
    for a, (o, r, d) in zip(acts, want):
        o2, r2, d2, _ = envB.step(a)
        np.testing.assert_array_equal(o2, o)
        np.testing.assert_array_equal(r2, r)
        np.testing.assert_array_equal(d2, d)
    with pytest.raises(ValueError, match="size mismatch"):
        envB.reset_to(snap[:-1].copy())
    tj = HostEnv.tj(5, 6, 1, "easy", 3, seed=1, add_rate_min=0.5, add_rate_max=0.5)
    tj.reset(0)
    tj.step(np.zeros((3, 5), np.int32))
    st = tj.raw_state()
    o1 = tj.observe()
    tj.step(np.ones((3, 5), np.int32))
    np.testing.assert_array_equal(tj.reset_to(st, epoch=0), o1)


def test_max_agents_per_env_vs_oracle():
    """N = 64 (one env per wavefront, the full 64-bit ballot mask) for PP, and N = 48 cars for TJ-hard; N = 65 is refused."""
    import oracle
    E = 3
    env = HostEnv.pp(64, 10, 1, "cooperative", E, seed=8, offset=3)
    orcs = [oracle.PPOracle(64, 10, 1, "cooperative", seed=8, env_gid=3 + e) for e in range(E)]
    obs = env.reset()
    for e, o in enumerate(orcs):
        np.testing.assert_array_equal(obs[e], o.reset())
    rs = np.random.RandomState(1)
    for t in range(6):
        act = rs.randint(0, 5, size=(E, 64))
        obs, rew, done, _ = env.step(act)
        for e, o in enumerate(orcs):
            oo, orew, od = o.step(act[e])
            np.testing.assert_array_equal(obs[e], oo)
            np.testing.assert_array_equal(rew[e], orew.astype(np.float32))
    tj = HostEnv.tj(48, 18, 1, "hard", E, seed=8, offset=3, add_rate_min=0.9, add_rate_max=0.9)
    torcs = [oracle.TJOracle(48, 18, 1, "hard", add_rate_min=0.9, add_rate_max=0.9, seed=8, env_gid=3 + e)
             for e in range(E)]
    tj.reset(0)
    for o in torcs:
        o.reset(0)
    for t in range(15):
        act = (rs.rand(E, 48) < 0.5).astype(np.int32)
        obs, rew, done, info = tj.step(act)
        st = tj.get_state()
        for e, o in enumerate(torcs):
            oo, orew, _ = o.step(act[e])
            np.testing.assert_array_equal(st["alive"][e], o.alive)
            np.testing.assert_array_equal(st["route_id"][e], o.route_id)
            np.testing.assert_array_equal(obs[e], oo)
            np.testing.assert_array_equal(rew[e], orew.astype(np.float32))
    with pytest.raises(ValueError):
        HostEnv.pp(65, 10, 1, "mixed", 2)                    # N > 64 is rejected, not mis-simulated
    with pytest.raises(NotImplementedError):                  # predator_prey_env.py:84-85
        from ic3net_amd import _lib as binding
        cfg = binding.PPCfg(2, 3, 1, 5, 0, 0, 1, 1, 0, 0, 0)
        check(host_lib().ic3_pp_create(C.byref(cfg), -1, C.byref(C.c_void_p())))


def test_kernels_reproduce_reference_checksum_sweep():
    """The 210-configuration checksum sweep recorded from the reference (tests/golden/make_golden_sweep.py), through the
    product's kernels on the host (one handle per configuration, E = 1 like the reference)."""
    from golden_util import crc_of, SWEEP_RATES
    fx = load("sweep_checksums")
    seed = int(fx["seed"])
    for cfg, acts, crcs in zip(fx["pp_cfg"], fx["pp_act"], fx["pp_crc"]):
        N, dim, v, mode, ec, ns, gid = [int(x) for x in cfg]
        env = HostEnv.pp(N, dim, v, MODES[mode], 1, seed=seed, offset=gid, no_stay=bool(ns), enemy_comm=bool(ec))
        obs = env.reset()[0]
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
            over = bool(done[0])
            got = crc_of(loc[:N], st["reached"][0], rew[0], obs[0], np.int32(int(over)))
            assert got == crcs[t + 1], (cfg, t)
        env.close()
    for cfg, acts, crcs in zip(fx["tj_cfg"], fx["tj_act"], fx["tj_crc"]):
        N, dim, v, diff, rate_i, scalar, gid = [int(x) for x in cfg]
        r = SWEEP_RATES[rate_i]
        env = HostEnv.tj(N, dim, v, DIFFS[diff], 1, seed=seed, offset=gid, add_rate_min=r, add_rate_max=r,
                         vocab_type='scalar' if scalar else 'bool')
        env.reset(0)
        for t in range(acts.shape[0]):
            obs, rew, _, _ = env.step(acts[t:t + 1, :N])
            st = env.get_state()
            loc = np.stack([st["loc_r"][0], st["loc_c"][0]], -1)
            got = crc_of(st["alive"][0], st["wait"][0], loc, st["last_act"][0], st["route_loc"][0], st["route_id"][0],
                         rew[0], obs[0])
            assert got == crcs[t], (cfg, t)
        env.close()


@pytest.mark.parametrize("H", [4, 16, 64, 256])
def test_cell_backward_and_fused_cell_heads_draws(H):
    """ic3_lstm_cell_backward (torch.nn.LSTMCell's derivative + per-workgroup bias partials) against the closed form, and
    ic3_lstm_cell_heads (cell + heads + log_softmax + the draws of every head in one launch, comm.py:215-239 +
    action_utils.py:32-36) against ic3_lstm_cell -> ic3_policy_heads -> ic3_env_sample_actions on the same inputs."""
    lib = host_lib()
    rng = np.random.default_rng(H)
    env = HostEnv.pp(5, 8, 1, "mixed", 7, seed=21, offset=5)
    env.reset()
    env.step(rng.integers(0, 5, (7, 5)), with_obs=False)               # t = 1, episode 0: the draw counters of the env
    R = env.E * env.N
    gates = rng.standard_normal((R, 4 * H)).astype(np.float32)
    c_prev = rng.standard_normal((R, H)).astype(np.float32)
    dh = rng.standard_normal((R, H)).astype(np.float32)
    dc = rng.standard_normal((R, H)).astype(np.float32)
    dgates = np.full((R, 4 * H), np.nan, np.float32)
    dc_prev = np.full((R, H), np.nan, np.float32)
    parts = np.full((2048, 4 * H), np.nan, np.float32)
    nb = check(lib.ic3_lstm_cell_backward(p(gates), p(c_prev), p(dh), p(dc), p(dgates), p(dc_prev), p(parts), R, H, None))
    g64 = gates.astype(np.float64)
    sig = lambda x: 1 / (1 + np.exp(-x))
    i, f, g, o = sig(g64[:, :H]), sig(g64[:, H:2 * H]), np.tanh(g64[:, 2 * H:3 * H]), sig(g64[:, 3 * H:])
    c = f * c_prev + i * g
    tc = np.tanh(c)
    dct = dc + dh * o * (1 - tc * tc)
    want = np.concatenate([dct * g * i * (1 - i), dct * c_prev * f * (1 - f), dct * i * (1 - g * g), dh * tc * o * (1 - o)], 1)
    np.testing.assert_allclose(dgates, want, rtol=0, atol=2e-5)
    np.testing.assert_allclose(dc_prev, dct * f, rtol=0, atol=2e-5)
    np.testing.assert_allclose(parts[:nb].astype(np.float64).sum(0), want.sum(0), rtol=0, atol=1e-4)

    heads = np.array([5, 2], np.int32)
    OT = 8
    W = (rng.standard_normal((OT, H)) * 0.3).astype(np.float32)
    b = rng.standard_normal(OT).astype(np.float32)
    c1, c2 = c_prev.copy(), c_prev.copy()
    h1 = np.full((R, 2 * H), np.nan, np.float32)                       # h written into the right half of an [x | h] row
    h2 = np.full((R, H), np.nan, np.float32)
    out1 = np.full((R, OT), np.nan, np.float32)
    out2 = np.full((R, OT), np.nan, np.float32)
    act1 = np.full((2, env.E, env.N), -1, np.int32)
    act2 = np.full((2, env.E, env.N), -1, np.int32)
    check(lib.ic3_lstm_cell_heads(p(gates), p(c1), C.c_void_p(h1.ctypes.data + 4 * H), 2 * H, R, H, p(W), p(b), p(heads), 2,
                                  p(out1), env._h, p(act1), None))
    check(lib.ic3_lstm_cell(p(gates), p(c2), p(h2), H, R, H, None))
    check(lib.ic3_policy_heads(p(h2), H, p(W), p(b), p(heads), 2, p(out2), R, H, None))
    np.testing.assert_array_equal(c1, c2)
    np.testing.assert_array_equal(h1[:, H:], h2)
    assert np.isnan(h1[:, :H]).all()
    np.testing.assert_allclose(out1, out2, rtol=0, atol=2e-6)
    off = 0
    for k, A in enumerate(heads):
        lp = np.ascontiguousarray(out1[:, off:off + A])
        check(lib.ic3_env_sample_actions(env._h, p(lp), int(A), int(A), k, p(act2[k]), None, None))
        off += A
    np.testing.assert_array_equal(act1, act2)
    env.close()


def test_auto_reset_env_restarts_inside_the_step_launch():
    """ic3_env_set_auto_reset through ic3_env_step (tests/test_auto_reset_gpu.py): a Predator-Prey env restarts when its
    episode is over or the step cap is reached, a Traffic-Junction env at the cap; the per-env stream, cut at `done`, equals
    consecutive oracle episodes, and the finished episodes' statistics accumulate in ic3_stats.auto_*."""
    import oracle
    E, N, dim, v, cap = 20, 2, 3, 1, 5
    env = HostEnv.pp(N, dim, v, "mixed", E, seed=21, offset=90)
    check(env.lib.ic3_env_set_auto_reset(env._h, cap))
    env.reset()
    orcs = [oracle.PPOracle(N, dim, v, "mixed", seed=21, env_gid=90 + e) for e in range(E)]
    for o in orcs:
        o.reset()
    tcount = np.zeros(E, int)
    rs = np.random.RandomState(3)
    ends = succ = steps = 0
    for t in range(23):
        act = rs.randint(0, 5, size=(E, N)).astype(np.int32)
        obs, rew, done, _ = env.step(act)
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
    s = env.stats()
    assert ends > E and s.auto_episodes == ends and s.auto_success_sum == succ and s.auto_env_steps == steps
    check(env.lib.ic3_env_set_auto_reset(env._h, 0))              # back to lock-step: finished envs freeze again
    env.reset()
    assert env.stats().auto_episodes == 0
    env.close()

    E, N, cap = 8, 5, 4
    tj = HostEnv.tj(N, 6, 1, "easy", E, seed=5, offset=7, add_rate_min=0.6, add_rate_max=0.6)
    check(tj.lib.ic3_env_set_auto_reset(tj._h, cap))
    tj.reset(0)
    torcs = [oracle.TJOracle(N, 6, 1, "easy", add_rate_min=0.6, add_rate_max=0.6, seed=5, env_gid=7 + e) for e in range(E)]
    for o in torcs:
        o.reset(0)
    rs = np.random.RandomState(1)
    succ = 0
    for t in range(11):
        act = (rs.rand(E, N) < 0.3).astype(np.int32)
        obs, rew, done, info = tj.step(act)
        for e, o in enumerate(torcs):
            oo, orew, _ = o.step(act[e])
            np.testing.assert_array_equal(rew[e], orew.astype(np.float32))
            np.testing.assert_array_equal(info['alive_mask'][e], o.alive)   # info of the finishing step: the old episode's
            end = (t + 1) % cap == 0
            assert int(done[e]) == int(end)
            if end:
                succ += 1 - int(o.has_failed.value)
                oo = o.reset(0)
            np.testing.assert_array_equal(obs[e], oo)
    s = tj.stats()
    assert s.auto_episodes == 2 * E and s.auto_success_sum == succ and s.auto_env_steps == 2 * E * cap
    tj.close()


@pytest.mark.parametrize("which,message", [("ic3_host_selftest_lds_overrun", "wrote past the dynamic LDS"),
                                           ("ic3_host_selftest_stuck_barrier", "no lane can make progress")])
def test_runtime_guards_fire(which, message):
    """Negative controls of the stand-in runtime (kernels in tests/host/ic3_host_abi.cpp): a workgroup that writes behind the
    dynamic LDS its launch asked for, and a barrier that not every live lane reaches, end the process with a message
    instead of corrupting memory / hanging."""
    import os
    import subprocess
    import sys
    code = ("from host_abi_util import host_lib\n"
            "host_lib().%s()\n"
            "print('survived')\n" % which)
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([here, os.path.dirname(here)]))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0 and "survived" not in r.stdout
    assert message in r.stderr or "AddressSanitizer" in r.stderr, r.stderr[-2000:]
