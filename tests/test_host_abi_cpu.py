"""CPU: the product's own kernels behind the same C ABI on a CPU (`device = -1`).

tests/host/libic3rollout_host.so is ic3net_amd/csrc/{ic3_api, pp_kernels, tj_kernels, policy_ops, episode_kernels}.hip and
tj_tables.cpp — the files libic3rollout.so is built from, unmodified — compiled as C++ against a stand-in HIP runtime
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


def test_host_build_is_device_minus_one_only_and_has_no_matrix_core_kernels():
    from ic3net_amd import _lib as binding
    lib = host_lib()
    assert lib.ic3_version() >= 1
    cfg = binding.PPCfg(2, 3, 1, 5, 0, 0, 1, 0, 0, 0, 0)
    h = C.c_void_p()
    assert lib.ic3_pp_create(C.byref(cfg), 0, C.byref(h)) < 0          # a GPU ordinal: not in this library
    assert b"hipSetDevice" in lib.ic3_last_error()
    env = HostEnv.pp(3, 5, 0, 'mixed', 2)
    assert lib.ic3_policy_step_supported(env._h, 128) == 0
    assert lib.ic3_lstm_gates_backward_supported(128) == 0 and lib.ic3_commnet_forward_supported(128, 3) == 0
    pol = binding.Policy()
    rc = lib.ic3_policy_step(env._h, C.byref(pol), *([None] * 12))
    assert rc == -38 and b"matrix cores" in lib.ic3_last_error()
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
