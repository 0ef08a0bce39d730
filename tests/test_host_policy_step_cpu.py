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


def free_run(name, seed=5, offset=300, gate_split=False, use_table=True, E=None, T=None, mode_avg=True):
    """T lock-step iterations of ic3_policy_step on E envs, replayed env by env through the fp64 policy + the oracle env on
    the kernel's actions (tests/test_policy_step_onehop_gpu.py::_free_run).  Returns the worst policy error."""
    from oracle import policy_ref
    w = WORKLOADS[name]
    E, T = E or w['E'], T or w['T']
    N, H, heads = w['N'], w['H'], w['heads']
    nheads = len(heads)
    env = make_env(w, E, seed, offset)
    P = make_params(env.obs_dim, H, heads, seed=seed + 1)
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
