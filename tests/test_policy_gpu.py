"""GPU: the product policy (ic3net_amd.comm.CommNetMLP, fp32, comm_masked_mean HIP op) against outputs of
the reference's own CommNetMLP (fp64 golden vectors) — tolerance 1e-5 absolute (north_star), over a
free-running recurrence; plus the two custom ops against the oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from policy_util import POLICY_FIXTURES, PolicyCase  # noqa: E402

TOL = 1e-5   # north_star: policy forward within 1e-5 (fp32 vs the reference's fp64)


def build(pc):
    from ic3net_amd.comm import CommNetMLP
    net = CommNetMLP(pc.args(), pc.obs_dim)
    sd = {k: torch.from_numpy(np.asarray(v)).float() for k, v in pc.params.items()}
    net.load_state_dict(sd)           # strict: same keys and shapes as the reference's state_dict
    return net.cuda().float()


@pytest.mark.parametrize("fused", ["mega", "mega-fp32", "chain", False])
@pytest.mark.parametrize("name", POLICY_FIXTURES)
def test_policy_matches_reference(name, fused):
    """"mega" / "mega-fp32": the two arithmetic modes of the gate product (the default exact bf16 split products /
    args.gate_split = False: the fp32 matrix instruction) — both run every fixture.
    "mega": the no-grad rollout fast path with everything after the encoder as ONE launch (ic3_policy_forward, the
    policy half of ic3_policy_step: communication block, C, LSTMCell, heads, log_softmax) where it applies (recurrent,
    H in {64,128,256} — the hid-16 fixtures run as their zero-padded twin at 64, comm.CommNetMLP._twin; comm_passes > 1: one launch per communication pass; the non-recurrent module: ic3_commnet_forward,
    every pass in one launch); "chain": the same path as separate launches (one [inp|h] buffer, library GEMMs,
    lstm_cell / policy_heads HIP kernels); False: the generic torch path + comm_masked_mean op."""
    pc = PolicyCase(name)
    fx = pc.fx
    net = build(pc)
    net.args.fused_policy = bool(fused)
    net.args.gate_split = fused != "mega-fp32"
    if fused == "mega-fp32":
        fused = "mega"
    net.args.mega_policy = (fused == "mega")
    if fused == "mega":
        from ic3net_amd import ops
        if pc.H not in ops.POLICY_STEP_SIZES and ops.padded_hidden(pc.H) is None:
            pytest.skip("the one-launch policy kernels need H <= 256")       # (other sizes: the zero-padded twin)
    hid = net.init_hidden(pc.B) if pc.recurrent else None
    worst = 0.0
    with torch.no_grad():
        for t in range(pc.steps):
            info = {}
            if pc.alive(t) is not None:
                info['alive_mask'] = torch.from_numpy(pc.alive(t)).int().cuda()
            if pc.hard_attn:
                info['comm_action'] = torch.from_numpy(pc.comm_action(t)).int().cuda()
            x = torch.from_numpy(pc.x_step(t)).float().cuda()
            if pc.recurrent:
                logp, val, hid = net([x, hid], info)
            else:
                logp, val = net(x, info)
            for k in range(pc.nheads):
                worst = max(worst, np.abs(logp[k].cpu().numpy() - fx["logp%d" % k][t]).max())
            worst = max(worst, np.abs(val.reshape(-1, 1).cpu().numpy() - fx["value"][t]).max())
            if pc.recurrent:
                worst = max(worst, np.abs(hid[0].cpu().numpy() - fx["h"][t]).max())
                worst = max(worst, np.abs(hid[1].cpu().numpy() - fx["c"][t]).max())
    assert worst < TOL, worst
    if fused == "mega":                                  # ... and it was the one-launch kernel that produced them
        assert getattr(net, 'mega_forwards' if pc.recurrent else 'commnet_forwards', 0) == pc.steps


@pytest.mark.parametrize("E,N,H", [(64, 10, 128), (7, 3, 16), (33, 20, 128), (5, 32, 256), (3, 1, 64)])
def test_comm_masked_mean_vs_oracle(E, N, H):
    from oracle import policy_ref
    from ic3net_amd import ops
    rs = np.random.RandomState(E + N)
    h = rs.randn(E, N, H).astype(np.float32)
    alive = (rs.rand(E, N) < 0.7).astype(np.int32)
    ca = (rs.rand(E, N) < 0.6).astype(np.int32)
    alive[0] = 0
    if E > 1:
        alive[1] = 0
        alive[1, 0] = 1
    ht = torch.from_numpy(h).cuda()
    for avg in (True, False):
        for use_alive, use_ca in ((True, True), (False, True), (True, False), (False, False)):
            out = ops.comm_masked_mean(ht, torch.from_numpy(alive).cuda() if use_alive else None,
                                       torch.from_numpy(ca).cuda() if use_ca else None, avg, True).cpu().numpy()
            for e in range(E):
                lit = policy_ref.comm_block(h[e:e + 1].astype(np.float64), alive[e] if use_alive else None,
                                            ca[e] if use_ca else np.ones(N), avg, False, True)
                np.testing.assert_allclose(out[e], lit[0], rtol=1e-5, atol=1e-5)   # unbounded N(0,1) inputs, N up to 32
    z = ops.comm_masked_mean(ht, None, None, True, False)
    assert not z.any().item()


def test_comm_masked_mean_backward():
    from ic3net_amd import ops
    torch.manual_seed(0)
    E, N, H = 6, 5, 32
    h = torch.randn(E, N, H, device='cuda', requires_grad=True)
    alive = (torch.rand(E, N, device='cuda') < 0.7).int()
    ca = (torch.rand(E, N, device='cuda') < 0.7).int()
    out = ops.comm_masked_mean(h, alive, ca, True, True)
    w = torch.randn_like(out)
    (out * w).sum().backward()
    # dense torch reference of the same linear map
    m = (alive * ca).float()
    n_alive = alive.sum(1, keepdim=True).float()
    scale = torch.where(n_alive > 1, 1.0 / (n_alive - 1).clamp(min=1), torch.ones_like(n_alive))
    M = m.unsqueeze(2) * m.unsqueeze(1) * (1 - torch.eye(N, device='cuda')) * scale.unsqueeze(2)   # [e, j, i]
    h2 = h.detach().clone().requires_grad_(True)
    out2 = torch.einsum('eji,eih->ejh', M, h2)
    (out2 * w).sum().backward()
    torch.testing.assert_close(out, out2, atol=1e-5, rtol=0)
    torch.testing.assert_close(h.grad, h2.grad, atol=1e-5, rtol=0)


@pytest.mark.parametrize("E,N,H,avg", [(37, 10, 128, True), (9, 20, 64, False), (5, 40, 32, True)])
def test_comm_masked_mean_add_is_the_block_plus_the_addend(E, N, H, avg):
    """ic3_comm_masked_mean_add: out = addend + comm(h) with the addend a strided column slice (bptt's [d inp | d h] buffer)."""
    from ic3net_amd import ops
    g = torch.Generator(device='cuda').manual_seed(1)
    h = torch.randn(E, N, H, device='cuda', generator=g)
    alive = (torch.rand(E, N, device='cuda', generator=g) < 0.8).int()
    gate = (torch.rand(E, N, device='cuda', generator=g) < 0.6).int()
    wide = torch.randn(E * N, 2 * H, device='cuda', generator=g)
    for al, ga in ((alive, gate), (None, gate), (alive, None)):
        base = ops.comm_masked_mean_raw(h, al, ga, avg, True)
        out = ops.comm_masked_mean_raw(h, al, ga, avg, True, addend=wide[:, H:])
        # (the kernel's last multiply and the addition are one fused multiply-add: a last-ulp difference to the two ops)
        torch.testing.assert_close(out.view(E * N, H), wide[:, H:] + base.view(E * N, H), rtol=1e-6, atol=1e-6)
    out = ops.comm_masked_mean_raw(h, alive, gate, avg, False, addend=wide[:, H:])     # comm_mask_zero: the addend alone
    assert torch.equal(out.view(E * N, H), wide[:, H:])
    scale = (torch.rand(E * N, device='cuda', generator=g) < 0.7).float()             # per-row factor (collection mode)
    plain = ops.comm_masked_mean_raw(h, alive, gate, avg, True, addend=wide[:, H:])
    cut = ops.comm_masked_mean_raw(h, alive, gate, avg, True, addend=wide[:, H:], row_scale=scale)
    assert torch.equal(cut.view(E * N, H), plain.view(E * N, H) * scale[:, None])


@pytest.mark.parametrize("T,E,N,gamma,ratio", [(20, 33, 3, 1.0, 0.0), (40, 17, 10, 0.95, 0.5), (7, 300, 5, 0.9, 1.0),
                                                (13, 5, 64, 1.0, 0.3)])
def test_returns_scan_matches_the_reference_loop(T, E, N, gamma, ratio):
    """ic3_returns_scan against the loop of /root/reference/trainer.py:162-171 in float64 (episode cuts inside the batch,
    per-agent mini masks)."""
    from ic3net_amd import ops
    g = torch.Generator(device='cuda').manual_seed(T)
    rew = torch.randn(T, E, N, device='cuda', generator=g)
    em = (torch.rand(T, E, device='cuda', generator=g) < 0.85).float()
    mm = (torch.rand(T, E, N, device='cuda', generator=g) < 0.8).float()
    out = ops.returns_scan(rew, em, mm, gamma, ratio)
    r, e, m = rew.double().cpu(), em.double().cpu().unsqueeze(2), mm.double().cpu()
    coop = torch.zeros(T, E, N, dtype=torch.float64)
    ncoop = torch.zeros(T, E, N, dtype=torch.float64)
    pc = torch.zeros(E, N, dtype=torch.float64)
    pn = torch.zeros(E, N, dtype=torch.float64)
    for i in reversed(range(T)):
        coop[i] = r[i] + gamma * pc * e[i]
        ncoop[i] = r[i] + gamma * pn * e[i] * m[i]
        pc, pn = coop[i], ncoop[i]
    ref = ratio * coop.mean(2, keepdim=True) + (1 - ratio) * ncoop
    torch.testing.assert_close(out.double().cpu(), ref, atol=2e-5 * float(ref.abs().max()), rtol=0)
    out3 = ops.returns_scan(rew, em.unsqueeze(2).expand(T, E, N), mm, gamma, ratio)       # the expanded mask of a Transition
    assert torch.equal(out, out3)


def test_sample_actions_vs_oracle_and_distribution():
    import oracle
    from oracle import philox
    from ic3net_amd import ops
    E, N, A = 512, 10, 5
    rs = np.random.RandomState(1)
    logits = rs.randn(E, N, A).astype(np.float32) * 1.5
    logp = torch.log_softmax(torch.from_numpy(logits), -1)
    act, chosen = ops.sample_actions(logp.cuda(), 1, 42, 7000, 3, 9, want_logp=True)
    act, chosen = act.cpu().numpy(), chosen.cpu().numpy()
    lp = logp.numpy()
    mism = 0
    for e in range(E):
        for n in range(N):
            x = philox.x24(42, 7000 + e, philox.DOMAIN_SAMPLE, 3, 9, 1 * N + n)
            want = oracle.sample_one(lp[e, n], x)
            if want != act[e, n]:      # only possible when u sits within an ulp of a cdf edge (expf rounding)
                mism += 1
                u = x / 2.0 ** 24
                cdf = np.cumsum(np.exp(lp[e, n].astype(np.float64)))
                assert np.abs(cdf - u).min() < 1e-6
            assert chosen[e, n] == lp[e, n, act[e, n]]
    assert mism <= 2
    # distribution: many draws (different t) from one fixed row -> chi^2 against exp(logp)
    p = np.array([0.05, 0.15, 0.4, 0.3, 0.1], np.float32)
    row = torch.log(torch.from_numpy(p)).view(1, 1, A).expand(4096, 8, A).contiguous().cuda()
    counts = np.zeros(A)
    for t in range(8):
        a = ops.sample_actions(row, 0, 5, 0, 0, t).cpu().numpy()
        counts += np.bincount(a.ravel(), minlength=A)
    n = counts.sum()
    chi2 = ((counts - n * p) ** 2 / (n * p)).sum()
    assert chi2 < 30.0, (chi2, counts)     # 4 dof; 30 is far beyond the 1e-5 quantile


def test_random_actions_stream():
    from oracle import philox
    from ic3net_amd import ops
    a = ops.random_actions(16, 6, 5, 11, 100, 2, 3).cpu().numpy()
    for e in range(16):
        for n in range(6):
            assert a[e, n] == (philox.x24(11, 100 + e, philox.DOMAIN_BENCH, 2, 3, n) * 5) >> 24


@pytest.mark.parametrize("kind,cfg", [("pp", (10, 20, 1, 128, 96)), ("pp", (3, 5, 0, 64, 33)), ("pp", (32, 40, 2, 256, 4)),
                                      ("tj", (10, 14, 1, "medium", 128, 64)), ("tj", (20, 18, 0, "hard", 128, 32)),
                                      ("tj", (5, 6, 1, "easy", 32, 17))])
def test_sparse_encoder_equals_dense(kind, cfg):
    """ic3_env_encode == observe() @ W^T + b on states reached after some random steps (incl. dead TJ cars)."""
    from test_env_parity_gpu import make_pp, make_tj
    torch.manual_seed(0)
    if kind == "pp":
        N, dim, v, H, E = cfg
        env = make_pp(N, dim, v, "mixed", E, seed=1)
        env.reset()
        nact = 5
    else:
        N, dim, v, diff, H, E = cfg
        env = make_tj(N, dim, v, diff, E, seed=1, add_rate_min=0.4, add_rate_max=0.4)
        env.reset(0)
        nact = 2
    lin = torch.nn.Linear(env.obs_dim, H).cuda()
    for t in range(12):
        env.step(torch.randint(0, nact, (E, N), device='cuda', dtype=torch.int32))
        if t in (0, 5, 11):
            obs = env.observe()
            with torch.no_grad():
                dense = lin(obs.double().cpu().cuda()) if False else (obs.double() @ lin.weight.double().t() + lin.bias.double())
                wt = lin.weight.detach().t().contiguous()
                sparse = env.encode(wt, lin.bias.detach())
                tabled = env.encode(wt, lin.bias.detach(), loc_table=env.encode_table(wt))   # pre-summed location rows
            torch.testing.assert_close(sparse.double(), dense, atol=2e-6, rtol=0)
            torch.testing.assert_close(tabled.double(), dense, atol=2e-6, rtol=0)


def test_policy_sparse_encoder_hook_matches_dense():
    from test_env_parity_gpu import make_pp
    from policy_util import PolicyCase
    from ic3net_amd.comm import CommNetMLP
    import argparse
    N, dim, v, H, E = 10, 20, 1, 128, 64
    env = make_pp(N, dim, v, "mixed", E, seed=2)
    obs = env.reset()
    a = argparse.Namespace(nagents=N, hid_size=H, comm_passes=1, recurrent=True, continuous=False,
                           naction_heads=[5, 2], comm_mask_zero=False, share_weights=False, comm_init='uniform',
                           hard_attn=True, comm_mode='avg', rnn_type='LSTM')
    torch.manual_seed(1)
    net = CommNetMLP(a, env.obs_dim).cuda()
    info = {'comm_action': torch.ones(E, N, dtype=torch.int32, device='cuda')}
    with torch.no_grad():
        hid = net.init_hidden(E)
        d_logp, d_val, d_h = net([obs, hid], info)
        net.obs_encoder = env.encode
        s_logp, s_val, s_h = net([obs, hid], info)
    for x, y in zip(d_logp + [d_val, d_h[0], d_h[1]], s_logp + [s_val, s_h[0], s_h[1]]):
        torch.testing.assert_close(x, y, atol=2e-6, rtol=0)


def test_lstm_cell_and_heads_kernels_vs_torch():
    from ic3net_amd import ops
    torch.manual_seed(3)
    R, H = 777, 128
    cell = torch.nn.LSTMCell(H, H).cuda()
    x, h, c = torch.randn(R, H, device='cuda'), torch.randn(R, H, device='cuda') * 0.5, torch.randn(R, H, device='cuda')
    with torch.no_grad():
        h_ref, c_ref = cell(x, (h, c))
        gates = x @ cell.weight_ih.t() + cell.bias_ih + h @ cell.weight_hh.t() + cell.bias_hh
        xh = torch.zeros(R, 2 * H, device='cuda')
        c2 = c.clone()
        ops.lstm_cell_(gates.contiguous(), c2, xh[:, H:])
        torch.testing.assert_close(xh[:, H:], h_ref, atol=2e-6, rtol=0)
        torch.testing.assert_close(c2, c_ref, atol=2e-6, rtol=0)
        assert not xh[:, :H].any().item()
        for sizes in ([5, 2], [2], [5, 2, 3, 4], [9]):
            OT = sum(sizes) + 1
            W, b = torch.randn(OT, H, device='cuda') * 0.2, torch.randn(OT, device='cuda')
            out = ops.policy_heads(xh[:, H:], W, b, sizes)
            z = h_ref @ W.t() + b
            off = 0
            for A in sizes:
                torch.testing.assert_close(out[:, off:off + A], torch.log_softmax(z[:, off:off + A], -1), atol=3e-6,
                                           rtol=0)
                off += A
            torch.testing.assert_close(out[:, off], z[:, off], atol=3e-6, rtol=0)


@pytest.mark.parametrize("N,H,sizes", [(10, 128, [5, 2]), (5, 64, [2]), (3, 256, [5, 2]), (7, 32, [5, 2, 3, 4]), (4, 16, [9]),
                                         (6, 8, [5, 2]), (5, 4, [3]), (3, 16, [2, 2, 2, 2])])
def test_lstm_cell_heads_sample_fused_kernel(N, H, sizes):
    """ic3_lstm_cell_heads == lstm_cell_ + policy_heads (+ sample_actions_env per head) — cell outputs bit-identical,
    heads within fp32 rounding (different reduction order), draws bit-identical to the separate sampling kernel run on
    the fused kernel's own log-probabilities."""
    from ic3net_amd import ops
    from test_env_parity_gpu import make_pp
    E = 37
    env = make_pp(N, 8, 1, "mixed", E, seed=9, offset=4)
    env.reset()
    for _ in range(3):                                  # move the device-side (episode, t) counters off zero
        env.step(torch.randint(0, 5, (E, N), device='cuda', dtype=torch.int32))
    torch.manual_seed(N + H)
    R, OT = E * N, sum(sizes) + 1
    gates = torch.randn(R, 4 * H, device='cuda')
    c0 = torch.randn(R, H, device='cuda')
    W, b = torch.randn(OT, H, device='cuda') * 0.3, torch.randn(OT, device='cuda')
    xh_a, xh_b = torch.zeros(R, 2 * H, device='cuda'), torch.zeros(R, 2 * H, device='cuda')
    c_a, c_b = c0.clone(), c0.clone()
    ops.lstm_cell_(gates, c_a, xh_a[:, H:])
    out_a = ops.policy_heads(xh_a[:, H:], W, b, sizes)
    act = torch.full((len(sizes), E, N), -1, dtype=torch.int32, device='cuda')
    assert ops.lstm_cell_heads_ok(H)
    out_b = ops.lstm_cell_heads_(gates, c_b, xh_b[:, H:], W, b, sizes, env=env, action=act)
    assert torch.equal(c_a, c_b) and torch.equal(xh_a, xh_b)
    torch.testing.assert_close(out_b, out_a, atol=3e-6, rtol=0)
    off = 0
    for k, A in enumerate(sizes):
        lp = out_b.view(E, N, OT)[:, :, off:off + A]
        assert torch.equal(act[k], ops.sample_actions_env(env, lp, k))
        off += A
    # without an env handle: no draws, same outputs
    c_c, xh_c = c0.clone(), torch.zeros(R, 2 * H, device='cuda')
    out_c = ops.lstm_cell_heads_(gates, c_c, xh_c[:, H:], W, b, sizes)
    assert torch.equal(out_c, out_b) and torch.equal(c_c, c_b)


@pytest.mark.parametrize("name", ["baseline_mlp", "baseline_rnn", "baseline_rnn_lstm"])
def test_baseline_models_match_reference(name):
    import argparse
    from golden_util import load
    from ic3net_amd import models
    fx = load(name)
    N, obs_dim, H, steps, B, rec, lstm = [int(v) for v in fx["cfg"]]
    a = argparse.Namespace(nagents=N, hid_size=H, continuous=False, naction_heads=[5], rnn_type='LSTM' if lstm else 'MLP')
    net = (models.RNN if rec else models.MLP)(a, obs_dim)
    net.load_state_dict({k[2:]: torch.from_numpy(fx[k]).float() for k in fx.files if k.startswith("w:")})
    net = net.cuda()
    hid = None
    if rec:
        hid = net.init_hidden(B) if lstm else torch.zeros(B, N, H, device='cuda')
    with torch.no_grad():
        for t in range(steps):
            x = torch.from_numpy(fx["x"][t]).float().cuda()
            if rec:
                logp, v, hid = net([x, hid])
                h = hid[0] if lstm else hid
                assert np.abs(h.cpu().numpy() - fx["h"][t]).max() < TOL
            else:
                logp, v = net(x)
            assert np.abs(logp[0].cpu().numpy() - fx["logp0"][t]).max() < TOL
            assert np.abs(v.cpu().numpy() - fx["value"][t]).max() < TOL
