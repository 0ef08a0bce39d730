"""CPU: `ic3_policy_step` — the one-launch rollout kernel itself — on the host, against the fp64 reference policy + oracle env.

tests/host/libic3rollout_host.so also holds ic3net_amd/csrc/{policy_step, gates_bwd, commnet_fwd}.hip, unmodified, compiled
as C++: the stand-in runtime (tests/host/shim) executes `__builtin_amdgcn_mfma_*` as cross-lane operations with the
hardware's operand / result layouts, raw buffer loads / stores with the descriptor's range check (the kernel's zero stores
are issued unconditionally and DROPPED by that check past the tile's slice), and the kernel-argument re-read.  The bodies
are tests/test_policy_step_onehop_gpu.py's: free-running episodes at the BASELINE shapes, every per-step output (log-probs of
every head, value, h, c) against oracle.policy_ref (numpy float64, /root/reference/comm.py:134-244) driven by the C oracle
env on the kernel's own actions at the north_star's 1e-5, rewards and the dense observation rows of the same launch bit for
bit (/root/reference/trainer.py:43-108).  With IC3_HOST_ASAN=1 (tools/host_asan.sh) every LDS / global index the kernel
forms is bounds-checked.  What this does NOT see: the hardware's own behaviour (waitcnt, hazards, occupancy) — that is the
GPU suite's."""
import ctypes as C

import os

import numpy as np
import pytest

from host_abi_util import ASAN, HostEnv, HostPolicy, host_lib, check, p

TOL = 1e-5     # north_star: policy forward within 1e-5 fp32

# the BASELINE shapes; E / T sized for a CPU
WORKLOADS = {
    'pp_easy': dict(env='pp', N=3, dim=5, vision=0, H=64, heads=[5, 2], hard_attn=True, E=9, T=12),
    'pp_hard': dict(env='pp', N=10, dim=20, vision=1, H=128, heads=[5, 2], hard_attn=True, E=7, T=8),
    'tj_medium': dict(env='tj', N=10, dim=14, vision=0, difficulty='medium', H=128, heads=[2], hard_attn=False, rate=0.3, E=7,
                      T=10),
    'tj_hard': dict(env='tj', N=20, dim=18, vision=1, difficulty='hard', H=128, heads=[2, 2], hard_attn=True, rate=0.3, E=4,
                    T=8),
    'pp_scaled': dict(env='pp', N=32, dim=40, vision=2, H=256, heads=[5, 2], hard_attn=True, E=3, T=3),
}


def make_params(obs_dim, H, heads, seed, comm_passes=1):
    """state_dict-shaped float64 arrays holding float32-representable values (both sides read the same numbers)."""
    rng = np.random.default_rng(seed)
    r = lambda *s, sc=0.1: (rng.standard_normal(s) * sc).astype(np.float32).astype(np.float64)
    P = {'encoder.weight': r(H, obs_dim, sc=0.2), 'encoder.bias': r(H), 'f_module.weight_ih': r(4 * H, H),
         'f_module.weight_hh': r(4 * H, H), 'f_module.bias_ih': r(4 * H), 'f_module.bias_hh': r(4 * H),
         'value_head.weight': r(1, H, sc=0.2), 'value_head.bias': r(1)}
    for i in range(comm_passes):
        P['C_modules.%d.weight' % i] = r(H, H)
        P['C_modules.%d.bias' % i] = r(H)
    for k, A in enumerate(heads):
        P['heads.%d.weight' % k] = r(A, H, sc=0.2)
        P['heads.%d.bias' % k] = r(A)
    return P


def make_env(w, E, seed, offset):
    if w['env'] == 'pp':
        return HostEnv.pp(w['N'], w['dim'], w['vision'], 'mixed', E, seed=seed, offset=offset)
    return HostEnv.tj(w['N'], w['dim'], w['vision'], w['difficulty'], E, seed=seed, offset=offset, add_rate_min=w['rate'],
                      add_rate_max=w['rate'])


def make_oracle(w, seed, gid):
    import oracle
    if w['env'] == 'pp':
        return oracle.PPOracle(w['N'], w['dim'], w['vision'], 'mixed', seed=seed, env_gid=gid)
    return oracle.TJOracle(w['N'], w['dim'], w['vision'], w['difficulty'], w['rate'], w['rate'], 0, 0, seed=seed, env_gid=gid)


def free_run(name, seed=5, offset=300, gate_split=False, use_table=True, E=None, T=None, mode_avg=True, w=None):
    """T lock-step iterations of ic3_policy_step on E envs, replayed env by env through the fp64 policy + the oracle env on
    the kernel's actions (tests/test_policy_step_onehop_gpu.py::_free_run).  Returns the worst policy error."""
    from oracle import policy_ref
    w = w or WORKLOADS[name]
    E, T = E or w['E'], T or w['T']
    N, H, heads = w['N'], w['H'], w['heads']
    nheads = len(heads)
    env = make_env(w, E, seed, offset)
    P = make_params(env.obs_dim, H, heads, seed=seed + 1)
    gate_split = gate_split or os.environ.get("IC3_HOST_FORCE_SPLIT") == "1"     # (the wave-specialised kernel: split products only)
    pol = HostPolicy(env, P, H, heads, mode_avg=mode_avg, gate_split=gate_split, use_table=use_table)
    tj = w['env'] == 'tj'
    env.reset(0) if tj else env.reset()
    h = np.zeros((E * N, H), np.float32)
    c = np.zeros((E * N, H), np.float32)
    alive_in = None                                            # trainer.py:41-46: info is empty at t = 0 (quirk Q21)
    gate = np.zeros((E, N), np.int32) if w['hard_attn'] else None   # quirk Q22
    rec = []
    for t in range(T):
        out, act, obs, rew, done, alive, comp = pol.step(env, h, c, alive_in, gate)
        rec.append(dict(out=out.reshape(E, N, -1).copy(), h=h.reshape(E, N, H).copy(), c=c.reshape(E, N, H).copy(), act=act, obs=obs,
                        rew=rew))
        alive_in = alive if tj else None                       # info['alive_mask'] of this step feeds the next (TJ:244-247)
        if w['hard_attn']:                                     # trainer.py:70-71
            gate = np.ascontiguousarray(act[nheads - 1])
    worst = 0.0
    for e in range(E):
        o = make_oracle(w, seed, offset + e)
        obs = o.reset(0) if tj else o.reset()
        hc = (np.zeros((N, H)), np.zeros((N, H)))
        alive, g = None, np.zeros(N)
        for t in range(T):
            r = rec[t]
            np.testing.assert_array_equal(r['obs'][e], obs, err_msg="obs rows env %d step %d" % (e, t))
            logp, val, hc = policy_ref.forward(P, obs[None].astype(np.float64), hc, alive, g if w['hard_attn'] else None,
                                               recurrent=True, comm_mode_avg=mode_avg, hard_attn=w['hard_attn'], nheads=nheads)
            off = 0
            for hd, A in enumerate(heads):
                worst = max(worst, np.abs(logp[hd][0] - r['out'][e][:, off:off + A]).max())
                off += A
            worst = max(worst, np.abs(val.reshape(-1) - r['out'][e][:, off]).max())
            worst = max(worst, np.abs(hc[0] - r['h'][e]).max(), np.abs(hc[1] - r['c'][e]).max())
            assert worst < TOL, (name, e, t, worst)
            obs, orew, _ = o.step(r['act'][0, e])
            np.testing.assert_array_equal(r['rew'][e], np.asarray(orew).astype(np.float32))
            if tj:
                alive = o.alive.astype(np.float64)
            if w['hard_attn']:
                g = r['act'][nheads - 1, e].astype(np.float64)
    env.close()
    return worst


@pytest.mark.parametrize("name", ["pp_easy", "pp_hard", "tj_medium", "tj_hard", "pp_scaled"])
def test_policy_step_free_run_vs_fp64_reference_policy(name):
    assert free_run(name) < TOL


@pytest.mark.parametrize("name", ["pp_hard", "tj_hard", "pp_scaled"])
def test_gate_split_experiment_free_run(name):
    """The opt-in gate_split mode (ic3_policy.gate_split: nine exact bf16 x bf16 products per fp32 product on
    v_mfma_f32_32x32x16_bf16, fp32 accumulation) against the same fp64 policy at the same 1e-5."""
    assert free_run(name, gate_split=True, T=4) < TOL


def test_variants_without_location_table_and_with_sum_mode():
    """loc_table = NULL (the encoder gathers every window cell's row) and comm_mode = 'sum' (comm.py:194-196 skipped)."""
    assert free_run("pp_easy", use_table=False, T=5) < TOL
    assert free_run("tj_medium", use_table=False, mode_avg=False, T=5, E=4) < TOL


@pytest.mark.parametrize("name,passes,auto", [("pp_easy", 2, 0), ("pp_hard", 3, 0), ("tj_medium", 2, 0), ("pp_easy", 2, 4)])
def test_communication_passes_inside_one_launch(name, passes, auto):
    """ic3_policy.npasses (round 5): comm_passes > 1 as a loop INSIDE ic3_policy_step — h stays in the A tile, c in registers
    between the passes — gives exactly what one launch per pass gives (the same split products in the same order), on plain
    and on auto-reset handles (an env restarted in the launch starts from zeros in the FIRST pass only)."""
    w = WORKLOADS[name]
    E, N, H, heads, T = min(w['E'], 4), w['N'], w['H'], w['heads'], 3
    res = []
    for one_launch in (False, True):
        env = make_env(w, E, 8, 40)
        P = make_params(env.obs_dim, H, heads, seed=2, comm_passes=passes)
        if auto:
            check(env.lib.ic3_env_set_auto_reset(env._h, auto))
        env.reset(0) if w['env'] == 'tj' else env.reset()
        if one_launch:
            pol = HostPolicy(env, P, H, heads, gate_split=True, passes=passes)
        else:
            pols = [HostPolicy(env, P, H, heads, gate_split=True, pass_index=i, inner=(i + 1 < passes)) for i in range(passes)]
        h = np.zeros((E * N, H), np.float32)
        c = np.zeros((E * N, H), np.float32)
        gate = np.zeros((E, N), np.int32) if w['hard_attn'] else None
        alive_in = None
        rec = []
        for t in range(T):
            if one_launch:
                out, act, obs, rew, done, alive, comp = pol.step(env, h, c, alive_in, gate)
            else:
                for q in pols[:-1]:
                    q.inner_pass(env, h, c, alive_in, gate)
                out, act, obs, rew, done, alive, comp = pols[-1].step(env, h, c, alive_in, gate)
            rec.append([x.copy() for x in (out, act, obs, rew, done, h, c)])
            alive_in = alive if w['env'] == 'tj' else None
            if w['hard_attn']:
                gate = np.ascontiguousarray(act[len(heads) - 1])
        env.close()
        res.append(rec)
    for t, (a, b) in enumerate(zip(*res)):
        for k, (x, y) in enumerate(zip(a, b)):
            np.testing.assert_array_equal(x, y, err_msg="step %d field %d" % (t, k))


def test_two_communication_passes_one_launch_per_pass():
    """comm_passes = 2 on the recurrent policy (comm.py:179-218): pass 0 is an inner pass (ic3_policy.inner_pass: sparse
    encoder, communication block, C_modules[0], LSTMCell — h, c only), pass 1 the ordinary call with C_modules[1]."""
    from oracle import policy_ref
    import oracle
    w = WORKLOADS['pp_easy']
    E, N, H, heads, T = 5, w['N'], w['H'], w['heads'], 4
    env = make_env(w, E, 8, 40)
    P = make_params(env.obs_dim, H, heads, seed=2, comm_passes=2)
    pols = [HostPolicy(env, P, H, heads, pass_index=0, inner=True), HostPolicy(env, P, H, heads, pass_index=1)]
    env.reset()
    h = np.zeros((E * N, H), np.float32)
    c = np.zeros((E * N, H), np.float32)
    gate = np.zeros((E, N), np.int32)
    orcs = [oracle.PPOracle(N, w['dim'], w['vision'], 'mixed', seed=8, env_gid=40 + e) for e in range(E)]
    obs_o = [o.reset() for o in orcs]
    hcs = [(np.zeros((N, H)), np.zeros((N, H))) for _ in range(E)]
    for t in range(T):
        pols[0].inner_pass(env, h, c, None, gate)
        out, act, obs, rew, done, alive, comp = pols[1].step(env, h, c, None, gate)
        for e, o in enumerate(orcs):
            np.testing.assert_array_equal(obs[e], obs_o[e])
            logp, val, hcs[e] = policy_ref.forward(P, obs_o[e][None].astype(np.float64), hcs[e], None, gate[e].astype(np.float64),
                                                   recurrent=True, comm_passes=2, hard_attn=True, nheads=2)
            o3 = out.reshape(E, N, -1)[e]
            err = max(np.abs(logp[0][0] - o3[:, :5]).max(), np.abs(logp[1][0] - o3[:, 5:7]).max(),
                      np.abs(val.reshape(-1) - o3[:, 7]).max(), np.abs(hcs[e][0] - h.reshape(E, N, H)[e]).max(),
                      np.abs(hcs[e][1] - c.reshape(E, N, H)[e]).max())
            assert err < TOL, (t, e, err)
            obs_o[e], orew, _ = o.step(act[0, e])
            np.testing.assert_array_equal(rew[e], orew.astype(np.float32))
        gate = np.ascontiguousarray(act[1])
    env.close()


@pytest.mark.parametrize("H,N,E", [(64, 3, 11), (128, 10, 7), (256, 20, 2)])
def test_policy_forward_on_a_caller_supplied_encoder_output(H, N, E):
    """ic3_policy_forward (the kernel's KIND 0: no env, enc = encoder(x) + C.bias from the caller) with random alive / talk
    masks, incl. envs with 0 and 1 agents alive (comm.py:194: the division only when more than one is alive)."""
    from oracle import policy_ref
    heads = [2, 2]
    obs_dim = 12
    P = make_params(obs_dim, H, heads, seed=H)
    pol = HostPolicy(None, P, H, heads, use_table=False)
    rng = np.random.default_rng(3)
    x = rng.standard_normal((E, N, obs_dim))
    enc = np.ascontiguousarray((x @ P['encoder.weight'].T + P['encoder.bias'] + P['C_modules.0.bias']).reshape(E * N, H), np.float32)
    h0 = (rng.standard_normal((E * N, H)) * 0.5).astype(np.float32)
    c0 = (rng.standard_normal((E * N, H)) * 0.5).astype(np.float32)
    alive = (rng.random((E, N)) < 0.7).astype(np.int32)
    alive[0] = 0
    if E > 1:
        alive[1] = 0
        alive[1, N // 2] = 1
    gate = (rng.random((E, N)) < 0.6).astype(np.int32)
    h, c = h0.copy(), c0.copy()
    out = pol.forward(enc, E, N, h, c, alive, gate).reshape(E, N, -1)
    for e in range(E):
        Pe = dict(P)
        # the caller's enc already holds encoder(x) + C.bias: feed the reference the same numbers through its own encoder
        logp, val, hc = policy_ref.forward(Pe, x[e][None], (h0.reshape(E, N, H)[e].astype(np.float64),
                                                            c0.reshape(E, N, H)[e].astype(np.float64)),
                                           alive[e].astype(np.float64), gate[e].astype(np.float64), recurrent=True, hard_attn=True,
                                           nheads=2)
        err = max(np.abs(logp[0][0] - out[e][:, :2]).max(), np.abs(logp[1][0] - out[e][:, 2:4]).max(),
                  np.abs(val.reshape(-1) - out[e][:, 4]).max(), np.abs(hc[0] - h.reshape(E, N, H)[e]).max(),
                  np.abs(hc[1] - c.reshape(E, N, H)[e]).max())
        assert err < 2e-5, (e, err)     # (enc itself is rounded to fp32 here before the kernel sees it)


def test_incremental_obs_rows_are_the_same_rows():
    """EXPERIMENT ic3_env_set_incremental_obs: the launch clears what it painted last step instead of zero-filling; the
    rows must be bit-identical to the default's, step after step, on the same buffer."""
    w = WORKLOADS['pp_hard']
    E, N, H, heads = 5, w['N'], w['H'], w['heads']
    outs = []
    for incr in (0, 1):
        env = make_env(w, E, 3, 70)
        check(env.lib.ic3_env_set_incremental_obs(env._h, incr))
        P = make_params(env.obs_dim, H, heads, seed=4)
        pol = HostPolicy(env, P, H, heads)
        env.reset()
        h = np.zeros((E * N, H), np.float32)
        c = np.zeros((E * N, H), np.float32)
        gate = np.zeros((E, N), np.int32)
        obs = np.full((E, N, env.obs_dim), np.nan, np.float32)       # ONE buffer for the whole run (the caller's promise)
        rows = []
        for t in range(4):
            out = np.full((E * N, pol.OT), np.nan, np.float32)
            act = np.full((2, E, N), -1, np.int32)
            rew, done = np.zeros((E, N), np.float32), np.zeros((E,), np.int32)
            check(env.lib.ic3_policy_step(env._h, C.byref(pol.struct), p(h), p(c), None, p(gate), p(out), p(act), p(obs), p(rew),
                                          p(done), None, None, None))
            rows.append((obs.copy(), out.copy(), act.copy(), rew.copy()))
            gate = np.ascontiguousarray(act[1])
        outs.append(rows)
        env.close()
    for a, b in zip(*outs):
        for x, y in zip(a, b):
            np.testing.assert_array_equal(x, y)


def test_auto_reset_stream_restarts_policy_and_env_inside_the_launch():
    """ic3_env_set_auto_reset: an env whose episode ends (episode_over, or the step cap) starts its next episode inside the
    same ic3_policy_step launch, and the launch treats an env at t = 0 as an episode start (h = c = 0, no alive mask, gate 0:
    trainer.py:38-46, quirks Q21 / Q22).  Per env the stream must equal consecutive oracle episodes under the kernel's
    actions with the fp64 policy restarted at every episode start (tests/test_auto_reset_gpu.py)."""
    import oracle
    from oracle import policy_ref
    E, N, dim, v, cap, H, heads, T = 9, 2, 3, 1, 5, 64, [5, 2], 14
    env = HostEnv.pp(N, dim, v, "mixed", E, seed=21, offset=90)
    check(env.lib.ic3_env_set_auto_reset(env._h, cap))
    P = make_params(env.obs_dim, H, heads, seed=9)
    pol = HostPolicy(env, P, H, heads)
    env.reset()
    orcs = [oracle.PPOracle(N, dim, v, "mixed", seed=21, env_gid=90 + e) for e in range(E)]
    obs_o = [o.reset() for o in orcs]
    hcs = [(np.zeros((N, H)), np.zeros((N, H))) for _ in range(E)]
    gates = [np.zeros(N) for _ in range(E)]
    tcount = np.zeros(E, int)
    h = np.zeros((E * N, H), np.float32)
    c = np.zeros((E * N, H), np.float32)
    gate = np.zeros((E, N), np.int32)
    ends = 0
    for t in range(T):
        out, act, obs, rew, done, alive, comp = pol.step(env, h, c, None, gate)
        for e, o in enumerate(orcs):
            np.testing.assert_array_equal(obs[e], obs_o[e])
            logp, val, hcs[e] = policy_ref.forward(P, obs_o[e][None].astype(np.float64), hcs[e], None, gates[e], recurrent=True,
                                                   hard_attn=True, nheads=2)
            o3 = out.reshape(E, N, -1)[e]
            err = max(np.abs(logp[0][0] - o3[:, :5]).max(), np.abs(logp[1][0] - o3[:, 5:7]).max(),
                      np.abs(val.reshape(-1) - o3[:, 7]).max(), np.abs(hcs[e][0] - h.reshape(E, N, H)[e]).max())
            assert err < TOL, (t, e, err)
            oo, orew, od = o.step(act[0, e])
            tcount[e] += 1
            np.testing.assert_array_equal(rew[e], orew.astype(np.float32))
            end = bool(od) or tcount[e] == cap
            assert int(done[e]) == int(end), (t, e)
            gates[e] = act[1, e].astype(np.float64)
            if end:                                               # the next launch starts this env's next episode
                ends += 1
                oo = o.reset()
                tcount[e] = 0
                hcs[e] = (np.zeros((N, H)), np.zeros((N, H)))
                gates[e] = np.zeros(N)
            obs_o[e] = oo
        gate = np.ascontiguousarray(act[1])                       # (the launch itself ignores it for envs at t = 0)
    s = env.stats()
    assert ends > E and s.auto_episodes == ends
    env.close()


PLAN_WORKER = r"""
import sys, zlib
import numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[2])
from test_host_policy_step_cpu import WORKLOADS, make_env, make_params
from host_abi_util import HostPolicy
w = WORKLOADS['pp_hard']
E, N, H, heads = 13, w['N'], w['H'], w['heads']
env = make_env(w, E, 6, 10)
pol = HostPolicy(env, make_params(env.obs_dim, H, heads, seed=6), H, heads)
env.reset()
h = np.zeros((E * N, H), np.float32); c = np.zeros((E * N, H), np.float32)
gate = np.zeros((E, N), np.int32)
crc = 0
for t in range(3):
    res = pol.step(env, h, c, None, gate)
    for a in res + (h, c):
        crc = zlib.crc32(np.ascontiguousarray(a).tobytes(), crc)
    gate = np.ascontiguousarray(res[1][1])
print("CRC", crc)
"""


def test_results_do_not_depend_on_the_tile_plan():
    """IC3_PS_HALF = 0 / 1 forces plan A (ceil(E / EPT) tiles of up to two 32-row MFMA tiles) / plan B (full tiles + half
    tiles): 13 PP-hard envs are 3 tiles one way, 2 full + 1 half the other (the host runtime reports 2 CUs).  Everything the
    launches produce must be bit-identical (tests/test_policy_step_plans_gpu.py)."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    crcs = []
    for plan in ("0", "1"):
        r = subprocess.run([sys.executable, "-c", PLAN_WORKER, here, os.path.dirname(here)], capture_output=True, text=True,
                           env=dict(os.environ, IC3_PS_HALF=plan))
        assert r.returncode == 0, r.stderr[-3000:]
        crcs.append([l for l in r.stdout.splitlines() if l.startswith("CRC")][-1])
    assert crcs[0] == crcs[1], crcs


@pytest.mark.parametrize("split", [False, True], ids=["fp32", "bf16x9"])
@pytest.mark.parametrize("H,N,E,passes", [(64, 3, 9, 1), (128, 10, 5, 2), (256, 5, 3, 3)])
def test_commnet_forward_nonrecurrent_module(H, N, E, passes, split):
    """ic3_commnet_forward: the non-recurrent CommNet module behind the encoder, every communication pass in one launch
    (comm.py:127-129,179-205,220-224,228-239), against oracle.policy_ref with recurrent = False; the [comm | h] product on the
    fp32 matrix instruction (wp3 = NULL) and as exact bf16 split products (ic3_commnet_pack_split)."""
    from oracle import policy_ref
    lib = host_lib()
    heads = [2]
    obs_dim = 9
    P = make_params(obs_dim, H, heads, seed=H + passes, comm_passes=passes)
    rng = np.random.default_rng(passes)
    for i in range(passes):
        P['f_modules.%d.weight' % i] = (rng.standard_normal((H, H)) * 0.1).astype(np.float32).astype(np.float64)
        P['f_modules.%d.bias' % i] = (rng.standard_normal(H) * 0.1).astype(np.float32).astype(np.float64)
    f32 = lambda a: np.ascontiguousarray(a, np.float32)
    wp = np.full((passes, 2 * H * H), np.nan, np.float32)
    for i in range(passes):
        cw, fw = f32(P['C_modules.%d.weight' % i]), f32(P['f_modules.%d.weight' % i])
        check(lib.ic3_commnet_pack(p(cw), p(fw), C.c_void_p(wp.ctypes.data + i * 2 * H * H * 4), H, None))
    wp3 = None
    if split:
        wp3 = np.full((passes, 3 * H * H), np.nan, np.float32)
        for i in range(passes):
            cw, fw = f32(P['C_modules.%d.weight' % i]), f32(P['f_modules.%d.weight' % i])
            check(lib.ic3_commnet_pack_split(p(cw), p(fw), C.c_void_p(wp3.ctypes.data + i * 3 * H * H * 4), H, None))
    bias = f32(np.stack([P['C_modules.%d.bias' % i] + P['f_modules.%d.bias' % i] for i in range(passes)]))
    head_w = f32(np.concatenate([P['heads.0.weight'], P['value_head.weight']], 0))
    head_b = f32(np.concatenate([P['heads.0.bias'], P['value_head.bias']], 0))
    x = rng.standard_normal((E, N, obs_dim))
    enc = f32((x @ P['encoder.weight'].T + P['encoder.bias']).reshape(E * N, H))
    alive = (rng.random((E, N)) < 0.8).astype(np.int32)
    alive[0] = 0
    out = np.full((E * N, 3), np.nan, np.float32)
    h_out = np.full((E * N, H), np.nan, np.float32)
    sizes = np.array(heads, np.int32)
    check(lib.ic3_commnet_forward(p(enc), E, N, H, passes, p(wp), p(wp3) if split else None, p(bias), p(head_w), p(head_b), p(sizes),
                                  1, 1, 0, p(alive), None, p(out), p(h_out), None))
    for e in range(E):
        logp, val, hh = policy_ref.forward(P, x[e][None], None, alive[e].astype(np.float64), None, recurrent=False,
                                           comm_passes=passes, hard_attn=False, nheads=1)
        o3 = out.reshape(E, N, -1)[e]
        err = max(np.abs(logp[0][0] - o3[:, :2]).max(), np.abs(val.reshape(-1) - o3[:, 2]).max(),
                  np.abs(hh.reshape(N, H) - h_out.reshape(E, N, H)[e]).max())
        assert err < 2e-5, (e, err)


@pytest.mark.parametrize("H,R,split", [(64, 100, False), (64, 100, True), (128, 70, False), (128, 70, True), (256, 65, True)])
def test_gates_backward_recompute_and_cell_derivative(H, R, split):
    """ic3_lstm_gates_backward (gates_bwd.hip): gates = [inp | h_prev] . [W_ih | W_hh]^T + bias re-computed on the matrix cores
    and torch.nn.LSTMCell's derivative applied in the epilogue, against the closed form in float64; bias partials per 64 rows,
    written then accumulated; h_prev given separately fills the h half of xh (tests/test_gates_backward_gpu.py)."""
    lib = host_lib()
    rng = np.random.default_rng(H + R)
    f32 = lambda a: np.ascontiguousarray(a, np.float32)
    w_ih, w_hh = f32(rng.standard_normal((4 * H, H)) / H ** 0.5), f32(rng.standard_normal((4 * H, H)) / H ** 0.5)
    c_w = f32(rng.standard_normal((H, H)))
    b = f32(rng.standard_normal(4 * H))
    pad = 8
    wide = f32(rng.standard_normal((R, 2 * H + pad)))
    c_prev, dh, dc = [f32(rng.standard_normal((R, H))) for _ in range(3)]
    c_wp = np.empty(H * H, np.float32)
    l_wp = np.empty(4 * H * 2 * H, np.float32)
    check(lib.ic3_policy_pack(p(c_w), p(w_ih), p(w_hh), p(c_wp), p(l_wp), H, None))
    wp3 = None
    if split:
        wp3 = np.zeros(3 * 2 * H * 4 * H, np.uint16)
        check(lib.ic3_policy_pack_split(p(w_ih), p(w_hh), p(wp3), H, None))
    tiles = (R + 63) // 64
    xh = wide[:, :2 * H].astype(np.float64)
    g = xh @ np.concatenate([w_ih, w_hh], 1).T.astype(np.float64) + b
    sig = lambda z: 1 / (1 + np.exp(-z))
    i, f, gg, o = sig(g[:, :H]), sig(g[:, H:2 * H]), np.tanh(g[:, 2 * H:3 * H]), sig(g[:, 3 * H:])
    cn = f * c_prev + i * gg
    tc = np.tanh(cn)
    for with_dc in (True, False):
        dct = (dc if with_dc else 0) + dh * o * (1 - tc * tc)
        want = np.concatenate([dct * gg * i * (1 - i), dct * c_prev * f * (1 - f), dct * i * (1 - gg * gg), dh * tc * o * (1 - o)], 1)
        dgates = np.full((R, 4 * H), np.nan, np.float32)
        dcp = np.full((R, H), np.nan, np.float32)
        parts = np.full((tiles, 4 * H), np.nan, np.float32)
        n = check(lib.ic3_lstm_gates_backward(p(wide), 2 * H + pad, None, p(l_wp), p(wp3), p(b), p(c_prev), p(dh),
                                              p(dc) if with_dc else None, p(dgates), p(dcp), p(parts), 0, R, H, None))
        assert n == tiles
        # (4e-6, the GPU test has 2e-6: the emulated MFMA adds its k terms to the accumulator one at a time in fp32)
        assert np.abs(dgates - want).max() <= 4e-6 * max(1.0, np.abs(want).max())
        assert np.abs(dcp - dct * f).max() <= 4e-6 * max(1.0, np.abs(dct * f).max())
        np.testing.assert_allclose(parts.astype(np.float64).sum(0), want.sum(0), rtol=1e-5, atol=1e-4)
        before = parts.copy()
        dc_io = (dc if with_dc else np.zeros_like(dh)).copy()
        check(lib.ic3_lstm_gates_backward(p(wide), 2 * H + pad, None, p(l_wp), p(wp3), p(b), p(c_prev), p(dh), p(dc_io), p(dgates),
                                          p(dc_io), p(parts), 1, R, H, None))
        np.testing.assert_allclose(parts, 2 * before, rtol=1e-6, atol=1e-6)
        np.testing.assert_array_equal(dc_io, dcp)                  # dc_prev written over dc (in place)
    xh2 = np.ascontiguousarray(np.concatenate([wide[:, :H], np.full((R, H), np.nan, np.float32)], 1))
    h_prev = np.ascontiguousarray(wide[:, H:2 * H])
    dg3 = np.full((R, 4 * H), np.nan, np.float32)
    dcp3 = np.full((R, H), np.nan, np.float32)
    check(lib.ic3_lstm_gates_backward(p(xh2), 2 * H, p(h_prev), p(l_wp), p(wp3), p(b), p(c_prev), p(dh), None, p(dg3), p(dcp3), None,
                                      0, R, H, None))
    np.testing.assert_array_equal(xh2, wide[:, :2 * H])
    np.testing.assert_array_equal(dg3, dgates)
    if split and H in (64, 128):
        # ic3_lstm_gates_backward_dx (round 5): the same outputs + [d inp | d h_prev] = dgates . [W_ih | W_hh] in the launch
        wb3 = np.zeros(3 * 4 * H * 2 * H, np.uint16)
        check(lib.ic3_policy_pack_split_bwd(p(w_ih), p(w_hh), p(wb3), H, None))
        dg5 = np.full((R, 4 * H), np.nan, np.float32)
        dcp5 = np.full((R, H), np.nan, np.float32)
        dxh = np.full((R, 2 * H), np.nan, np.float32)
        check(lib.ic3_lstm_gates_backward_dx(p(xh2), 2 * H, p(h_prev), p(l_wp), p(wp3), p(wb3), p(b), p(c_prev), p(dh), None, p(dg5),
                                             p(dcp5), None, 0, p(dxh), R, H, None))
        np.testing.assert_array_equal(dg5, dg3)
        np.testing.assert_array_equal(dcp5, dcp3)
        want_dx = dg5.astype(np.float64) @ np.concatenate([w_ih, w_hh], 1).astype(np.float64)
        assert np.abs(dxh - want_dx).max() <= 6e-6 * max(1.0, np.abs(want_dx).max())
        assert lib.ic3_lstm_gates_backward_dx(p(xh2), 2 * H, p(h_prev), p(l_wp), None, p(wb3), p(b), p(c_prev), p(dh), None, p(dg5),
                                              p(dcp5), None, 0, p(dxh), R, H, None) == -22
        # ic3_lstm_gates_backward_given (round 5): the activated gates handed in (here: float32 roundings of the closed form)
        # instead of the gate product — the cell's derivative of exactly those values, h_prev copied into xh, dx in the launch
        acts = f32(np.concatenate([i, f, gg, o], 1))
        a64 = acts.astype(np.float64)
        ai, af, ag, ao = a64[:, :H], a64[:, H:2 * H], a64[:, 2 * H:3 * H], a64[:, 3 * H:]
        tcg = np.tanh(af * c_prev + ai * ag)
        dct = dc + dh * ao * (1 - tcg * tcg)
        want_g = np.concatenate([dct * ag * ai * (1 - ai), dct * c_prev * af * (1 - af), dct * ai * (1 - ag * ag), dh * tcg * ao * (1 - ao)], 1)
        xh6 = np.ascontiguousarray(np.concatenate([wide[:, :H], np.full((R, H), np.nan, np.float32)], 1))
        dg6, dcp6 = np.full((R, 4 * H), np.nan, np.float32), np.full((R, H), np.nan, np.float32)
        dxh6, parts6 = np.full((R, 2 * H), np.nan, np.float32), np.full((tiles, 4 * H), np.nan, np.float32)
        n = check(lib.ic3_lstm_gates_backward_given(p(acts), p(xh6), 2 * H, p(h_prev), p(wb3), p(c_prev), p(dh), p(dc), p(dg6), p(dcp6),
                                                    p(parts6), 0, p(dxh6), None, None, None, None, 0, R, H, None))
        assert n == tiles
        np.testing.assert_array_equal(xh6, wide[:, :2 * H])
        assert np.abs(dg6 - want_g).max() <= 2e-6 * max(1.0, np.abs(want_g).max())
        assert np.abs(dcp6 - dct * af).max() <= 2e-6 * max(1.0, np.abs(dct * af).max())
        np.testing.assert_allclose(parts6.astype(np.float64).sum(0), want_g.sum(0), rtol=1e-5, atol=1e-4)
        want_dx6 = dg6.astype(np.float64) @ np.concatenate([w_ih, w_hh], 1).astype(np.float64)
        assert np.abs(dxh6 - want_dx6).max() <= 6e-6 * max(1.0, np.abs(want_dx6).max())
        dg7, dcp7 = np.full((R, 4 * H), np.nan, np.float32), np.full((R, H), np.nan, np.float32)
        check(lib.ic3_lstm_gates_backward_given(p(acts), None, 0, None, None, p(c_prev), p(dh), p(dc), p(dg7), p(dcp7), None, 0, None,
                                                None, None, None, None, 0, R, H, None))   # (pointwise only: no copy, no dx)
        np.testing.assert_array_equal(dg7, dg6)
        np.testing.assert_array_equal(dcp7, dcp6)
        assert lib.ic3_lstm_gates_backward_given(p(acts), p(xh6), 2 * H, None, None, p(c_prev), p(dh), p(dc), p(dg7), p(dcp7), None, 0,
                                                 None, None, None, None, None, 0, R, H, None) == -22
        # collection mode's per-row cuts: the same launch on pre-multiplied inputs
        live = (rng.random(R) < 0.7).astype(np.float32)
        keep = (rng.random(R) < 0.6).astype(np.float32)
        xh8, xh9 = xh6.copy(), xh6.copy()
        xh8[:, H:] = np.nan
        xh9[:, H:] = np.nan
        outs = []
        for cut in (True, False):
            cp = c_prev if cut else f32(c_prev * live[:, None])
            hp = h_prev if cut else f32(h_prev * live[:, None])
            dcx = dc if cut else f32(dc * keep[:, None])
            dg8, dcp8 = np.full((R, 4 * H), np.nan, np.float32), np.full((R, H), np.nan, np.float32)
            dx8 = np.full((R, 2 * H), np.nan, np.float32)
            xx = xh8 if cut else xh9
            check(lib.ic3_lstm_gates_backward_given(p(acts), p(xx), 2 * H, p(hp), p(wb3), p(cp), p(dh), p(dcx), p(dg8), p(dcp8), None, 0,
                                                    p(dx8), p(live) if cut else None, p(keep) if cut else None, None, None, 0, R, H,
                                                    None))
            outs.append((dg8, dcp8, dx8, xx.copy()))
        for u, v in zip(*outs):
            np.testing.assert_array_equal(u, v)
        # round 6: the heads' share of dL/dh folded in (dh + dhead . w_heads is what the cell sees), and IN PLACE on the record
        for OT in (3, 8, 16):
            dhead, w_heads = f32(rng.standard_normal((R, OT))), f32(rng.standard_normal((OT, H)) / H ** 0.5)
            dh_full = f32(dh.astype(np.float64) + dhead.astype(np.float64) @ w_heads.astype(np.float64))
            ref = [np.full((R, 4 * H), np.nan, np.float32), np.full((R, H), np.nan, np.float32), np.full((R, 2 * H), np.nan, np.float32)]
            check(lib.ic3_lstm_gates_backward_given(p(acts), None, 0, None, p(wb3), p(c_prev), p(dh_full), p(dc), p(ref[0]), p(ref[1]),
                                                    None, 0, p(ref[2]), None, None, None, None, 0, R, H, None))
            inplace = acts.copy()
            got = [inplace, np.full((R, H), np.nan, np.float32), np.full((R, 2 * H), np.nan, np.float32)]
            check(lib.ic3_lstm_gates_backward_given(p(inplace), None, 0, None, p(wb3), p(c_prev), p(dh), p(dc), p(inplace), p(got[1]),
                                                    None, 0, p(got[2]), None, None, p(dhead), p(w_heads), OT, R, H, None))
            for u, v in zip(got, ref):     # (the fold adds its terms one at a time: the last ulp of dh differs from the fp64 sum's rounding)
                assert np.abs(u - v).max() <= 3e-6 * max(1.0, np.abs(v).max())
        assert lib.ic3_lstm_gates_backward_given(p(acts), None, 0, None, None, p(c_prev), p(dh), p(dc), p(dg7), p(dcp7), None, 0, None,
                                                 None, None, p(dhead), None, 3, R, H, None) == -22
        assert lib.ic3_lstm_gates_backward_given(p(acts), None, 0, None, None, p(c_prev), p(dh), p(dc), p(dg7), p(dcp7), None, 0, None,
                                                 None, None, p(dhead), p(w_heads), 17, R, H, None) == -22


def _mix(x, alive, gate, mode_avg):
    """comm.py:181-205 in closed form on (E, N, H) float64 (ic3_comm_masked_mean)."""
    E, N, H = x.shape
    al = np.ones((E, N)) if alive is None else alive.astype(np.float64)
    g = al * (np.ones((E, N)) if gate is None else gate.astype(np.float64))
    S = (g[:, :, None] * x).sum(1, keepdims=True)
    n_alive = al.sum(1)
    scale = np.where(n_alive > 1, 1.0 / np.maximum(n_alive - 1, 1), 1.0) if mode_avg else np.ones(E)
    return g[:, :, None] * (S - g[:, :, None] * x) * scale[:, None, None]


@pytest.mark.parametrize("H,N,E,avg,masks", [(128, 10, 13, True, 'both'), (64, 3, 50, True, 'gate'), (128, 20, 7, False, 'alive'),
                                             (64, 32, 3, True, None), (128, 64, 2, True, 'both'), (64, 1, 5, True, None)])
def test_comm_backward_one_launch(H, N, E, avg, masks):
    """ic3_comm_backward (bptt_kernels.hip): dh_out = (d h_direct + (M d inp) . C) * out_scale and the partials of
    (M d inp)^T . h_prev — against d h_direct + M (d inp . C) and d inp^T . (M h_prev) in float64 (the two forms the Python
    loop of rounds 3-5 computed: the mixing matrix is symmetric), tiles of 64 / N whole envs, a ragged last tile."""
    lib = host_lib()
    rng = np.random.default_rng(H + N + E)
    f32 = lambda a: np.ascontiguousarray(a, np.float32)
    R = E * N
    dxh, hp = f32(rng.standard_normal((R, 2 * H))), f32(rng.standard_normal((R, H)))
    cw = f32(rng.standard_normal((H, H)) / H ** 0.5)
    alive = (rng.random((E, N)) < 0.8).astype(np.int32) if masks in ('alive', 'both') else None
    gate = (rng.random((E, N)) < 0.6).astype(np.int32) if masks in ('gate', 'both') else None
    scale = f32(rng.random(R) < 0.7)
    dinp, dhd = dxh[:, :H].astype(np.float64), dxh[:, H:].astype(np.float64)
    want_dh = dhd + _mix((dinp @ cw.astype(np.float64)).reshape(E, N, H), alive, gate, avg).reshape(R, H)
    want_dc = dinp.T @ _mix(hp.astype(np.float64).reshape(E, N, H), alive, gate, avg).reshape(R, H)
    nparts = lib.ic3_comm_backward_partials(E, N)
    assert nparts == min(512, -(-E // (64 // N)))
    for with_scale in (False, True):
        dh = np.full((R, H), np.nan, np.float32)
        parts = np.full((nparts, H, H), np.nan, np.float32)
        n = check(lib.ic3_comm_backward(p(dxh), 2 * H, p(hp), p(alive), p(gate), p(cw), p(scale) if with_scale else None, p(dh), p(parts),
                                        0, E, N, H, int(avg), 0, None))
        assert n == nparts
        w = want_dh * (scale[:, None] if with_scale else 1.0)
        assert np.abs(dh - w).max() <= 4e-6 * max(1.0, np.abs(w).max())
        got = parts.astype(np.float64).sum(0)
        assert np.abs(got - want_dc).max() <= 1e-5 * max(1.0, np.abs(want_dc).max())
        before = parts.copy()
        check(lib.ic3_comm_backward(p(dxh), 2 * H, p(hp), p(alive), p(gate), p(cw), None, p(dh), p(parts), 1, E, N, H, int(avg), 0, None))
        np.testing.assert_allclose(parts, 2 * before, rtol=1e-6, atol=1e-6)
    # comm_mask_zero: the gate product's share alone (nothing else is read)
    dh = np.full((R, H), np.nan, np.float32)
    assert check(lib.ic3_comm_backward(p(dxh), 2 * H, None, None, None, None, p(scale), p(dh), None, 0, E, N, H, int(avg), 1, None)) == 0
    np.testing.assert_array_equal(dh, dxh[:, H:] * scale[:, None])
    assert lib.ic3_comm_backward(p(dxh), 2 * H, p(hp), None, None, p(cw), None, p(dh), None, 0, E, N, H, 1, 0, None) == -22


@pytest.mark.parametrize("H,Q,ldi", [(128, 150, 256), (64, 200, 128), (64, 37, 64)])   # (two emulated CUs: a slice per CU at H = 64)
def test_weight_gradient_of_a_window_in_one_launch(H, Q, ldi):
    """ic3_lstm_weight_grad: dW (2H, 4H) (+)= [inp | h_prev]^T . dgates over Q rows — inp read from rows of stride ldi (the
    record's [inp | h] rows), h_prev from its own array, row_live scaling the h rows; written, then accumulated."""
    lib = host_lib()
    rng = np.random.default_rng(H + Q)
    f32 = lambda a: np.ascontiguousarray(a, np.float32)
    inp, h, dg = f32(rng.standard_normal((Q, ldi))), f32(rng.standard_normal((Q, H))), f32(rng.standard_normal((Q, 4 * H)))
    live = f32(rng.random(Q) < 0.8)
    n = lib.ic3_lstm_weight_grad_scratch_floats(Q, H)
    assert n > 0 and lib.ic3_lstm_weight_grad_scratch_floats(Q, 256) == 0
    scratch = np.full(n, np.nan, np.float32)
    for lv, split in ((None, 0), (live, 0), (None, 1), (live, 1)):    # split: nine exact bf16 x bf16 products per fp32 product
        x = np.concatenate([inp[:, :H].astype(np.float64), h.astype(np.float64) * (1.0 if lv is None else lv[:, None])], 1)
        want = x.T @ dg.astype(np.float64)
        dW = np.full((2 * H, 4 * H), np.nan, np.float32)
        check(lib.ic3_lstm_weight_grad(p(inp), ldi, p(h), p(dg), p(lv), Q, H, p(dW), 0, split, p(scratch), None))
        assert np.abs(dW - want).max() <= 2e-5 * max(1.0, np.abs(want).max())
        once = dW.copy()
        check(lib.ic3_lstm_weight_grad(p(inp), ldi, p(h), p(dg), p(lv), Q, H, p(dW), 1, split, p(scratch), None))
        np.testing.assert_allclose(dW, 2 * once, rtol=1e-6, atol=1e-6)
    assert lib.ic3_lstm_weight_grad(p(inp), H - 4, p(h), p(dg), None, Q, H, p(dW), 0, 0, p(scratch), None) == -22


def commnet_weights(lib, P, H, heads, passes):
    """The derived weights ic3_commnet_forward / ic3_commnet_step stream (ic3net_amd.comm._commnet_cache on numpy buffers)."""
    f32 = lambda a: np.ascontiguousarray(a, np.float32)
    wp = np.full((passes, 2 * H * H), np.nan, np.float32)
    for i in range(passes):
        cw, fw = f32(P['C_modules.%d.weight' % i]), f32(P['f_modules.%d.weight' % i])
        check(lib.ic3_commnet_pack(p(cw), p(fw), C.c_void_p(wp.ctypes.data + i * 2 * H * H * 4), H, None))
    bias = f32(np.stack([P['C_modules.%d.bias' % i] + P['f_modules.%d.bias' % i] for i in range(passes)]))
    head_w = f32(np.concatenate([P['heads.%d.weight' % k] for k in range(len(heads))] + [P['value_head.weight']], 0))
    head_b = f32(np.concatenate([P['heads.%d.bias' % k] for k in range(len(heads))] + [P['value_head.bias']], 0))
    wp3 = np.full((passes, 3 * H * H), np.nan, np.float32)       # the same weights as three exact bf16 planes
    for i in range(passes):
        cw, fw = f32(P['C_modules.%d.weight' % i]), f32(P['f_modules.%d.weight' % i])
        check(lib.ic3_commnet_pack_split(p(cw), p(fw), C.c_void_p(wp3.ctypes.data + i * 3 * H * H * 4), H, None))
    return dict(wp=wp, wp3=wp3, bias=bias, head_w=head_w, head_b=head_b, wt=f32(P['encoder.weight'].T),
                enc_bias=f32(P['encoder.bias']))


@pytest.mark.parametrize("name,passes,use_table", [("pp_easy", 1, True), ("pp_hard", 2, True), ("tj_medium", 2, False),
                                                   ("tj_hard", 1, True)])
def test_commnet_step_free_run_vs_fp64_reference_policy(name, passes, use_table):
    """ic3_commnet_step — the whole rollout iteration of the NON-recurrent module as one launch (trainer.py:43-108 through
    comm.py:127-129,179-205,220-224): free-running steps against oracle.policy_ref (recurrent = False) driven by the oracle env
    on the kernel's own actions at 1e-5; rewards, alive masks and the dense obs rows of the same launch bit for bit."""
    from oracle import policy_ref
    lib = host_lib()
    w = WORKLOADS[name]
    E, T = min(w['E'], 5), min(w['T'], 6)
    N, H, heads = w['N'], w['H'], w['heads']
    nheads = len(heads)
    tj = w['env'] == 'tj'
    env = make_env(w, E, 5, 300)
    P = make_params(env.obs_dim, H, heads, seed=6, comm_passes=passes)
    rng = np.random.default_rng(passes + H)
    for i in range(passes):
        P['f_modules.%d.weight' % i] = (rng.standard_normal((H, H)) * 0.1).astype(np.float32).astype(np.float64)
        P['f_modules.%d.bias' % i] = (rng.standard_normal(H) * 0.1).astype(np.float32).astype(np.float64)
    cw = commnet_weights(lib, P, H, heads, passes)
    table = None
    if use_table:
        table = np.empty((env.dims.grid_h * env.dims.grid_w, H), np.float32)
        check(lib.ic3_env_encode_table(env._h, p(cw['wt']), H, p(table), None))
    assert lib.ic3_commnet_step_supported(env._h, H) > 0
    env.reset(0) if tj else env.reset()
    sizes = np.array(heads, np.int32)
    OT = sum(heads) + 1
    alive_in = None
    gate = np.zeros((E, N), np.int32) if w['hard_attn'] else None
    rec = []
    for t in range(T):
        out = np.full((E * N, OT), np.nan, np.float32)
        act = np.full((nheads, E, N), -1, np.int32)
        obs = np.full((E, N, env.obs_dim), np.nan, np.float32)
        rew, done = np.zeros((E, N), np.float32), np.zeros((E,), np.int32)
        alive, comp = np.zeros((E, N), np.int32), np.zeros((E, N), np.int32)
        check(lib.ic3_commnet_step(env._h, p(cw['wt']), p(cw['enc_bias']), p(table), H, passes, p(cw['wp']),
                                   p(cw['wp3']) if passes > 1 else None, p(cw['bias']),
                                   p(cw['head_w']), p(cw['head_b']), p(sizes), nheads, 1, 0, p(alive_in), p(gate), None, None, p(out),
                                   p(act), p(obs), p(rew), p(done), p(alive), p(comp), None))
        rec.append(dict(out=out.reshape(E, N, -1).copy(), act=act, obs=obs, rew=rew, alive=alive))
        alive_in = alive if tj else None
        if w['hard_attn']:
            gate = np.ascontiguousarray(act[nheads - 1])
    worst = 0.0
    for e in range(E):
        o = make_oracle(w, 5, 300 + e)
        obs = o.reset(0) if tj else o.reset()
        alive, g = None, np.zeros(N)
        for t in range(T):
            r = rec[t]
            np.testing.assert_array_equal(r['obs'][e], obs, err_msg="obs rows env %d step %d" % (e, t))
            logp, val, _ = policy_ref.forward(P, obs[None].astype(np.float64), None, alive, g if w['hard_attn'] else None,
                                              recurrent=False, comm_passes=passes, hard_attn=w['hard_attn'], nheads=nheads)
            off = 0
            for hd, A in enumerate(heads):
                worst = max(worst, np.abs(logp[hd][0] - r['out'][e][:, off:off + A]).max())
                off += A
            worst = max(worst, np.abs(val.reshape(-1) - r['out'][e][:, off]).max())
            assert worst < TOL, (name, e, t, worst)
            obs, orew, _ = o.step(r['act'][0, e])
            np.testing.assert_array_equal(r['rew'][e], np.asarray(orew).astype(np.float32))
            if tj:
                np.testing.assert_array_equal(r['alive'][e], o.alive)
                alive = o.alive.astype(np.float64)
            if w['hard_attn']:
                g = r['act'][nheads - 1, e].astype(np.float64)
    env.close()


@pytest.mark.parametrize("name,split,auto", [("pp_easy", False, False), ("pp_hard", True, True), ("tj_medium", True, False)])
def test_commnet_step_as_the_tanh_recurrence_of_the_iric_baseline(name, split, auto):
    """ic3_commnet_step with h_in (round 6): models.RNN with the tanh recurrence (models.py:68-92) as ONE launch per step —
    h_t = tanh(affine1(obs) + affine2(h_{t-1})), heads on h_t — free-running against float64 numpy driven by the oracle env on the
    kernel's own actions; h_t ping-pongs between two buffers; in auto-reset mode an env that starts an episode reads h = 0
    (its entering rows are poisoned with a large value to prove it); a call with h_out == h_in, two passes or the communication
    block on is refused."""
    lib = host_lib()
    w = WORKLOADS[name]
    E, T = min(w['E'], 4), 5
    N, H, heads = w['N'], w['H'], w['heads'][:1]
    tj = w['env'] == 'tj'
    env = make_env(w, E, 5, 300)
    P = make_params(env.obs_dim, H, heads, seed=9, comm_passes=1)
    rng = np.random.default_rng(H)
    A2 = (rng.standard_normal((H, H)) * 0.1).astype(np.float32).astype(np.float64)
    b2 = (rng.standard_normal(H) * 0.1).astype(np.float32).astype(np.float64)
    P['C_modules.0.weight'], P['C_modules.0.bias'] = np.zeros((H, H)), np.zeros(H)
    P['f_modules.0.weight'], P['f_modules.0.bias'] = A2, b2
    cw = commnet_weights(lib, P, H, heads, 1)
    if auto:
        check(lib.ic3_env_set_auto_reset(env._h, 3))          # episodes of 3 steps: restarts inside the launches
    env.reset(0) if tj else env.reset()
    sizes = np.array(heads, np.int32)
    OT = sum(heads) + 1
    hbuf = np.zeros((2, E * N, H), np.float32)
    alive_in = None
    W1, b1 = P['encoder.weight'], P['encoder.bias']
    Wh = np.concatenate([P['heads.0.weight'], P['value_head.weight']], 0)
    bh = np.concatenate([P['heads.0.bias'], P['value_head.bias']], 0)
    h_ref = np.zeros((E * N, H))
    worst = 0.0
    for t in range(T):
        out = np.full((E * N, OT), np.nan, np.float32)
        act = np.full((1, E, N), -1, np.int32)
        obs = np.full((E, N, env.obs_dim), np.nan, np.float32)
        rew, done = np.zeros((E, N), np.float32), np.zeros((E,), np.int32)
        alive, comp = np.zeros((E, N), np.int32), np.zeros((E, N), np.int32)
        h_in, h_out = hbuf[t & 1], hbuf[(t + 1) & 1]
        fresh = auto and t % 3 == 0 and t > 0
        if fresh:
            h_in[:] = 1e3                                      # what the launch must NOT read
            h_ref[:] = 0.0
        args = [env._h, p(cw['wt']), p(cw['enc_bias']), None, H, 1, p(cw['wp']), p(cw['wp3']) if split else None, p(cw['bias']),
                p(cw['head_w']), p(cw['head_b']), p(sizes), 1, 1, 1, p(alive_in), None]
        tail = [p(out), p(act), p(obs), p(rew), p(done), p(alive), p(comp), None]
        if t == 0:
            assert lib.ic3_commnet_step(*(args + [p(h_in), p(h_in)] + tail)) == -22          # h_out == h_in
            assert lib.ic3_commnet_step(*(args[:14] + [0] + args[15:] + [p(h_in), p(h_out)] + tail)) == -22   # communication on
        check(lib.ic3_commnet_step(*(args + [p(h_in), p(h_out)] + tail)))
        x = obs.reshape(E * N, -1).astype(np.float64) @ W1.T + b1
        h_ref = np.tanh(x + h_ref @ A2.T + b2)
        z = h_ref @ Wh.T + bh
        A = heads[0]
        zl = z[:, :A] - z[:, :A].max(1, keepdims=True)
        logp = zl - np.log(np.exp(zl).sum(1, keepdims=True))
        worst = max(worst, np.abs(h_out - h_ref).max(), np.abs(out[:, :A] - logp).max(), np.abs(out[:, A] - z[:, A]).max())
        assert worst < TOL, (name, t, worst)
        h_ref = h_out.astype(np.float64)                       # (free run: follow the kernel's own state)
        alive_in = alive if tj else None
    env.close()


@pytest.mark.parametrize("H,R,OT", [(64, 100, 3), (128, 70, 8)])
def test_heads_grad_over_an_episode(H, R, OT):
    """ic3_heads_grad accumulates d^T h and the column sums of d over all rows (float64 check)."""
    lib = host_lib()
    rng = np.random.default_rng(H + R + OT)
    f32 = lambda a: np.ascontiguousarray(a, np.float32)
    d = f32(rng.standard_normal((R, OT)))
    dW, db = np.ones((OT, H), np.float32), np.full(OT, 2.0, np.float32)
    scratch = np.empty(int(lib.ic3_heads_grad_scratch_floats(H)), np.float32)
    hrows = f32(rng.standard_normal((R, H)))
    check(lib.ic3_heads_grad(p(d), p(hrows), R, H, OT, p(dW), p(db), p(scratch), None))
    np.testing.assert_allclose(dW, 1.0 + d.astype(np.float64).T @ hrows.astype(np.float64), rtol=0, atol=1e-4)
    np.testing.assert_allclose(db, 2.0 + d.astype(np.float64).sum(0), rtol=0, atol=1e-4)


def test_hidden_out_writes_the_next_slot_and_leaves_the_input_alone():
    """ic3_env_set_hidden_out: the next ic3_policy_step reads (h, c) from its arguments and writes h', c' to the given buffers —
    same values as the in-place call, inputs untouched, one-shot."""
    w = WORKLOADS['pp_easy']
    E, N, H, heads = 7, w['N'], w['H'], w['heads']
    res = []
    for use_out in (False, True):
        env = make_env(w, E, 3, 70)
        P = make_params(env.obs_dim, H, heads, seed=4)
        pol = HostPolicy(env, P, H, heads)
        env.reset()
        rng = np.random.default_rng(1)
        h = (rng.standard_normal((E * N, H)) * 0.3).astype(np.float32)
        c = (rng.standard_normal((E * N, H)) * 0.3).astype(np.float32)
        h0, c0 = h.copy(), c.copy()
        ho, co = np.full_like(h, np.nan), np.full_like(c, np.nan)
        gate = np.ones((E, N), np.int32)
        if use_out:
            check(env.lib.ic3_env_set_hidden_out(env._h, p(ho), p(co)))
        out, act, obs, rew, done, alive, comp = pol.step(env, h, c, None, gate)
        if use_out:
            np.testing.assert_array_equal(h, h0)
            np.testing.assert_array_equal(c, c0)
            res.append((ho.copy(), co.copy(), out.copy()))
            out2, *_ = pol.step(env, h, c, None, gate)            # one-shot: this call updates h, c in place again
            assert not np.array_equal(h, h0)
        else:
            res.append((h.copy(), c.copy(), out.copy()))
        env.close()
    for x, y in zip(*res):
        np.testing.assert_array_equal(x, y)


def test_gates_out_records_the_cell_s_activated_gates():
    """ic3_env_set_record_out (round 5): the next ic3_policy_step also stores sigmoid(i) | sigmoid(f) | tanh(g) | sigmoid(o) of its
    LSTM cell [E*N][4H] — every other output unchanged bit for bit, the new cell / hidden state reproduced from the stored gates
    exactly (c' = fma(f, c, i g), h' = o tanh(c') up to the kernel's tanh), one-shot; refused (-38) without gate_split."""
    w = WORKLOADS['pp_easy']
    E, N, H, heads = 7, w['N'], w['H'], w['heads']
    res = []
    for armed in (False, True):
        env = make_env(w, E, 3, 70)
        P = make_params(env.obs_dim, H, heads, seed=4)
        pol = HostPolicy(env, P, H, heads, gate_split=True)
        env.reset()
        rng = np.random.default_rng(1)
        h = (rng.standard_normal((E * N, H)) * 0.3).astype(np.float32)
        c = (rng.standard_normal((E * N, H)) * 0.3).astype(np.float32)
        h0, c0 = h.copy(), c.copy()
        gate = np.ones((E, N), np.int32)
        gates = np.full((E * N, 4 * H), np.nan, np.float32)
        xrows = np.full((E * N, 2 * H), np.nan, np.float32)
        if armed:
            check(env.lib.ic3_env_set_record_out(env._h, p(gates), p(xrows)))
        out, act, obs, rew, done, alive, comp = pol.step(env, h, c, None, gate)
        res.append((h.copy(), c.copy(), out.copy(), act.copy(), obs.copy(), rew.copy()))
        if armed:
            assert np.isfinite(gates).all()
            g64 = gates.astype(np.float64)
            gi, gf, gg, go = g64[:, :H], g64[:, H:2 * H], g64[:, 2 * H:3 * H], g64[:, 3 * H:]
            assert (gi > 0).all() and (gi < 1).all() and (np.abs(gg) <= 1).all()
            c1 = gf * c0 + gi * gg
            assert np.abs(c1 - c).max() <= 1e-6
            assert np.abs(go * np.tanh(c1) - h).max() <= 2e-6
            # the inp half of xh: the gate pre-activations follow from it and the state that entered (float64 product)
            assert np.isnan(xrows[:, H:]).all() and np.isfinite(xrows[:, :H]).all()
            pre = np.concatenate([xrows[:, :H], h0], 1).astype(np.float64) @ np.concatenate([pol.w_ih, pol.w_hh], 1).T.astype(np.float64) \
                + (P['f_module.bias_ih'] + P['f_module.bias_hh'])
            assert np.abs(1 / (1 + np.exp(-pre[:, :H])) - gi).max() <= 2e-6 and np.abs(np.tanh(pre[:, 2 * H:3 * H]) - gg).max() <= 2e-6
            keep = gates.copy()
            pol.step(env, h, c, None, gate)                          # one-shot: this call stores no gates
            np.testing.assert_array_equal(gates, keep)
            plain = HostPolicy(env, P, H, heads)                      # the fp32-instruction gate product: no gate record
            check(env.lib.ic3_env_set_record_out(env._h, p(gates), p(xrows)))
            with pytest.raises(NotImplementedError):
                plain.step(env, h, c, None, gate)
        env.close()
    for x, y in zip(*res):
        np.testing.assert_array_equal(x, y)


def test_auto_reset_stream_traffic_junction():
    """The same for Traffic-Junction: every env restarts at the step cap (quirk Q12: TJ never sets episode_over); the alive
    mask of the previous step is ignored for an env at t = 0."""
    import oracle
    from oracle import policy_ref
    E, N, cap, H, heads, T = 5, 5, 4, 64, [2, 2], 10
    env = HostEnv.tj(N, 6, 1, "easy", E, seed=5, offset=7, add_rate_min=0.6, add_rate_max=0.6)
    check(env.lib.ic3_env_set_auto_reset(env._h, cap))
    P = make_params(env.obs_dim, H, heads, seed=12)
    pol = HostPolicy(env, P, H, heads)
    env.reset(0)
    orcs = [oracle.TJOracle(N, 6, 1, "easy", add_rate_min=0.6, add_rate_max=0.6, seed=5, env_gid=7 + e) for e in range(E)]
    obs_o = [o.reset(0) for o in orcs]
    hcs = [(np.zeros((N, H)), np.zeros((N, H))) for _ in range(E)]
    gates = [np.zeros(N) for _ in range(E)]
    alives = [None] * E
    h = np.zeros((E * N, H), np.float32)
    c = np.zeros((E * N, H), np.float32)
    gate = np.zeros((E, N), np.int32)
    alive_in = None
    for t in range(T):
        out, act, obs, rew, done, alive, comp = pol.step(env, h, c, alive_in, gate)
        end = (t + 1) % cap == 0
        for e, o in enumerate(orcs):
            np.testing.assert_array_equal(obs[e], obs_o[e])
            logp, val, hcs[e] = policy_ref.forward(P, obs_o[e][None].astype(np.float64), hcs[e], alives[e], gates[e],
                                                   recurrent=True, hard_attn=True, nheads=2)
            o3 = out.reshape(E, N, -1)[e]
            err = max(np.abs(logp[0][0] - o3[:, :2]).max(), np.abs(logp[1][0] - o3[:, 2:4]).max(),
                      np.abs(val.reshape(-1) - o3[:, 4]).max(), np.abs(hcs[e][0] - h.reshape(E, N, H)[e]).max())
            assert err < TOL, (t, e, err)
            oo, orew, _ = o.step(act[0, e])
            np.testing.assert_array_equal(rew[e], orew.astype(np.float32))
            np.testing.assert_array_equal(alive[e], o.alive)
            assert int(done[e]) == int(end)
            alives[e], gates[e] = o.alive.astype(np.float64), act[1, e].astype(np.float64)
            if end:
                oo = o.reset(0)
                hcs[e] = (np.zeros((N, H)), np.zeros((N, H)))
                alives[e], gates[e] = None, np.zeros(N)
            obs_o[e] = oo
        alive_in, gate = alive, np.ascontiguousarray(act[1])
    assert env.stats().auto_episodes == 2 * E
    env.close()


def test_randomized_shapes_sweep():
    """Random small configurations of both envs through the one-launch kernel (every lane-group size G = 1 .. 32, envs per
    tile from 2 to 64, partial last tiles, full + half tile plans, odd and even grids, vision 0 .. 2, one or two heads, IC3Net
    and CommNet gating, hid 64 / 128) for a few steps each against the fp64 policy + oracle env."""
    rs = np.random.RandomState(77)
    tj_dims = {"easy": [6, 8], "medium": [6, 8, 10], "hard": [9, 12]}
    ran = 0
    for trial in range(14):
        H = int(rs.choice([64, 64, 128]))
        hard = bool(rs.rand() < 0.6)
        if rs.rand() < 0.5:
            dim = int(rs.randint(3, 9))
            w = dict(env='pp', N=int(rs.randint(1, min(20, dim * dim - 1) + 1)), dim=dim, vision=int(rs.randint(0, 3)), H=H,
                     heads=[5, 2] if hard else [5], hard_attn=hard)
        else:
            diff = ["easy", "medium", "hard"][rs.randint(3)]
            w = dict(env='tj', N=int(rs.randint(1, 21)), dim=int(rs.choice(tj_dims[diff])), vision=int(rs.randint(0, 2)),
                     difficulty=diff, H=H, heads=[2, 2] if hard else [2], hard_attn=hard, rate=float(rs.choice([0.3, 0.7])))
        E = int(rs.randint(1, 1 + max(2, 150 // w['N'])))
        seed, offset = int(rs.randint(1 << 20)), int(rs.randint(1 << 16))
        try:
            worst = free_run('trial %d' % trial, seed=seed, offset=offset, E=E, T=3, w=w)
        except NotImplementedError:           # an env tile that does not fit in LDS: the launch chain's case, not this kernel's
            continue
        ran += 1
        assert worst < TOL, (trial, w, E, worst)
    assert ran >= 10, ran


def test_step_events_are_stamped_by_the_launch():
    """ic3_event_create / ic3_env_set_step_events / ic3_event_elapsed_ms (bench.py's per-launch timing): the armed pair is
    consumed by exactly the next ic3_policy_step (one shot)."""
    lib = host_lib()
    w = WORKLOADS['pp_easy']
    env = make_env(w, 3, 1, 0)
    pol = HostPolicy(env, make_params(env.obs_dim, w['H'], w['heads'], seed=1), w['H'], w['heads'])
    env.reset()
    h = np.zeros((3 * w['N'], w['H']), np.float32)
    c = np.zeros_like(h)
    gate = np.zeros((3, w['N']), np.int32)
    e0, e1 = C.c_void_p(), C.c_void_p()
    check(lib.ic3_event_create(C.byref(e0)))
    check(lib.ic3_event_create(C.byref(e1)))
    check(lib.ic3_env_set_step_events(env._h, e0, e1))
    pol.step(env, h, c, None, gate)
    ms = C.c_float(-1.0)
    check(lib.ic3_event_elapsed_ms(e0, e1, C.byref(ms)))
    first = ms.value
    assert first > 0.0
    pol.step(env, h, c, None, gate)                            # not armed again: the pair keeps the first launch's stamps
    check(lib.ic3_event_elapsed_ms(e0, e1, C.byref(ms)))
    assert ms.value == first
    check(lib.ic3_event_destroy(e0))
    check(lib.ic3_event_destroy(e1))
    env.close()


def test_results_do_not_depend_on_the_lane_schedule():
    """The stand-in runtime schedules the lanes of a workgroup round-robin from lane 0 up; IC3_HOST_SCHED=reverse walks them
    from the top down, =shuffle in a new pseudo-random order every round.  Device code whose result depends on which lane runs first between two cross-lane operations — a
    wave-lockstep assumption, like the one found in pp_step_lanes in round 3 — behaves differently under these orders; a
    selection of the parity tests must pass under each of them (the whole host suite does: DESIGN.md section 7)."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sel = ("auto_reset or (free_run_vs and (pp_easy or tj_medium)) or two_communication or commnet_forward or "
           "(golden and (pp_easy_mixed or pp_enemycomm_mixed or tj_easy_v1_full or tj_medium_v0)) or encoder_backward or "
           "cell_backward or finalize")
    host_lib()                                                 # (built once, here: the two runs below only load it)
    procs = [(sched, subprocess.Popen([sys.executable, "-m", "pytest", os.path.join(here, "test_host_policy_step_cpu.py"),
                                       os.path.join(here, "test_host_abi_cpu.py"), "-q", "-x", "-p", "no:cacheprovider", "-k", sel],
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                                      env=dict(os.environ, IC3_HOST_SCHED=sched), cwd=os.path.dirname(here)))
             for sched in ("reverse", "shuffle")]              # shuffle: another pseudo-random order in every round; side by side
    for sched, pr in procs:
        out, err = pr.communicate()
        assert pr.returncode == 0, sched + out[-3000:] + err[-2000:]
        assert " passed" in out and "failed" not in out
