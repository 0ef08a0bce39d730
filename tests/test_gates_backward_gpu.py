"""GPU: ic3_lstm_gates_backward (gate recompute + LSTM cell backward in one launch, update half) against fp64 autograd
through torch.nn.LSTMCell's formula (comm.py:215) — and against the two-launch path it replaces (library GEMM +
ic3_lstm_cell_backward)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def reference(xh, w_ih, w_hh, b, c_prev, dh, dc):
    H = c_prev.shape[1]
    xh, c_prev = xh.double(), c_prev.double()
    gates = (xh @ torch.cat([w_ih, w_hh], 1).double().t() + b.double()).requires_grad_(True)
    c0 = c_prev.clone().requires_grad_(True)
    i, f, g, o = gates[:, :H].sigmoid(), gates[:, H:2 * H].sigmoid(), gates[:, 2 * H:3 * H].tanh(), gates[:, 3 * H:].sigmoid()
    c1 = f * c0 + i * g
    h1 = o * c1.tanh()
    loss = (h1 * dh.double()).sum() + ((c1 * dc.double()).sum() if dc is not None else 0)
    loss.backward()
    return gates.grad, c0.grad


@pytest.mark.parametrize("H,R,pad", [(128, 640, 0), (128, 1000, 0), (64, 77, 8), (256, 333, 0), (128, 64 * 300 + 5, 4)])
def test_gates_backward_matches_autograd(H, R, pad):
    from ic3net_amd import ops
    gen = torch.Generator(device='cuda').manual_seed(H + R)
    rn = lambda *s: torch.randn(*s, device='cuda', generator=gen)
    w_ih, w_hh, c_w = rn(4 * H, H) / H ** 0.5, rn(4 * H, H) / H ** 0.5, rn(H, H)
    b = rn(4 * H)
    wide = rn(R, 2 * H + pad)
    xh = wide[:, :2 * H]                                      # row stride 2H + pad
    c_prev, dh, dc = rn(R, H), rn(R, H), rn(R, H)
    wp = ops.policy_step_pack(c_w, w_ih, w_hh)['ps_l_wp']
    tiles = (R + 63) // 64
    for with_dc in (True, False):
        dgates = torch.full((R, 4 * H), float('nan'), device='cuda')
        dcp = torch.full((R, H), float('nan'), device='cuda')
        parts = torch.full((tiles, 4 * H), float('nan'), device='cuda')
        n = ops.lstm_gates_backward(xh, wp, b, c_prev, dh, dc if with_dc else None, dgates, dcp, parts, False)
        assert n == tiles
        ref_dg, ref_dc = reference(xh, w_ih, w_hh, b, c_prev, dh, dc if with_dc else None)
        assert float((dgates.double() - ref_dg).abs().max()) <= 2e-6 * max(1.0, float(ref_dg.abs().max()))
        assert float((dcp.double() - ref_dc).abs().max()) <= 2e-6 * max(1.0, float(ref_dc.abs().max()))
        torch.testing.assert_close(parts.double().sum(0), dgates.double().sum(0), atol=1e-4, rtol=1e-5)
        # accumulate: the partial rows grow by the same sums; dc_prev written over dc (in place)
        dc_io = (dc if with_dc else torch.zeros_like(dh)).clone()
        before = parts.clone()
        ops.lstm_gates_backward(xh, wp, b, c_prev, dh, dc_io, dgates, dc_io, parts, True)
        torch.testing.assert_close(parts, 2 * before, atol=1e-6, rtol=1e-6)
        torch.testing.assert_close(dc_io, dcp, atol=0, rtol=0)
    # EXPERIMENT gate_split: the recompute as nine exact bf16 split products per fp32 product — same bars
    wp3 = ops.policy_pack_split(w_ih, w_hh)
    dg4, dcp4 = torch.full_like(dgates, float('nan')), torch.full_like(dcp, float('nan'))
    parts4 = torch.full((tiles, 4 * H), float('nan'), device='cuda')
    ops.lstm_gates_backward(xh, wp, b, c_prev, dh, None, dg4, dcp4, parts4, False, lstm_wp3=wp3)
    ref_dg, ref_dc = reference(xh, w_ih, w_hh, b, c_prev, dh, None)
    assert float((dg4.double() - ref_dg).abs().max()) <= 2e-6 * max(1.0, float(ref_dg.abs().max()))
    assert float((dcp4.double() - ref_dc).abs().max()) <= 2e-6 * max(1.0, float(ref_dc.abs().max()))
    torch.testing.assert_close(parts4.double().sum(0), dg4.double().sum(0), atol=1e-4, rtol=1e-5)
    # h_prev given separately: same results, and the launch fills the h half of xh
    xh2 = torch.cat([xh[:, :H], torch.full((R, H), float('nan'), device='cuda')], 1)
    dg3, dcp3 = torch.empty_like(dgates), torch.empty_like(dcp)
    ops.lstm_gates_backward(xh2, wp, b, c_prev, dh, None, dg3, dcp3, None, False, h_prev=xh[:, H:].contiguous())
    torch.testing.assert_close(xh2, xh.contiguous(), atol=0, rtol=0)
    torch.testing.assert_close(dg3, dgates, atol=0, rtol=0)
    # the two-launch path it replaces
    gates = torch.addmm(b, xh, torch.cat([w_ih, w_hh], 1).t())
    dg2, dcp2 = torch.empty_like(dgates), torch.empty_like(dcp)
    ops.lstm_cell_backward(gates.contiguous(), c_prev, dh, None, dg2, dcp2, None)
    assert float((dgates - dg2).abs().max()) <= 2e-5 * max(1.0, float(dg2.abs().max()))


@pytest.mark.parametrize("H,M,OT", [(128, 5000, 8), (64, 333, 3), (256, 70000, 16), (128, 64 * 1024 * 3 + 17, 5)])
def test_heads_grad_over_an_episode(H, M, OT):
    """ic3_heads_grad: dW += d^T h, db += column sums of d over all (step, row) pairs, against a float64 product."""
    from ic3net_amd import ops
    gen = torch.Generator(device='cuda').manual_seed(H + M)
    d = torch.randn(M, OT, device='cuda', generator=gen)
    h = torch.randn(M, H, device='cuda', generator=gen)
    dW = torch.ones((OT, H), device='cuda')                      # (accumulates on top of what is there)
    db = torch.full((OT,), 2.0, device='cuda')
    ops.heads_grad(d, h, dW, db)
    refW = 1.0 + d.double().t() @ h.double()
    refb = 2.0 + d.double().sum(0)
    tol = 2e-6 * M ** 0.5 * 8
    assert float((dW.double() - refW).abs().max()) <= tol and float((db.double() - refb).abs().max()) <= tol
    dW2, db2 = torch.ones_like(dW), torch.full_like(db, 2.0)
    ops.heads_grad(d, h, dW2, db2)
    assert torch.equal(dW, dW2) and torch.equal(db, db2)         # fixed-order reduction: reproducible


def test_gates_backward_rejects_other_sizes():
    from ic3net_amd import ops
    assert ops.lstm_gates_backward_supported(128) and not ops.lstm_gates_backward_supported(96)
    z = torch.zeros(8, 192, device='cuda')
    with pytest.raises(Exception):
        ops.lstm_gates_backward(z, torch.zeros(8 * 96 * 96, device='cuda'), torch.zeros(384, device='cuda'),
                                torch.zeros(8, 96, device='cuda'), torch.zeros(8, 96, device='cuda'), None,
                                torch.zeros(8, 384, device='cuda'), torch.zeros(8, 96, device='cuda'))


@pytest.mark.parametrize("H,R", [(128, 640), (128, 64 * 200 + 37), (64, 150), (64, 128)])
def test_gates_backward_with_the_input_gradient_in_the_same_launch(H, R):
    """ic3_lstm_gates_backward_dx (round 5): dgates / dc_prev / bias partials exactly as without it (the same launch), and
    dxh = dgates . [W_ih | W_hh] against the float64 product at the gate product's own bar (exact bf16 split products)."""
    from ic3net_amd import ops
    gen = torch.Generator(device='cuda').manual_seed(3 * H + R)
    rn = lambda *s: torch.randn(*s, device='cuda', generator=gen)
    w_ih, w_hh, c_w = rn(4 * H, H) / H ** 0.5, rn(4 * H, H) / H ** 0.5, rn(H, H)
    b = rn(4 * H)
    xh, h_prev = rn(R, 2 * H), rn(R, H)
    c_prev, dh, dc = rn(R, H), rn(R, H), rn(R, H)
    wp = ops.policy_step_pack(c_w, w_ih, w_hh)['ps_l_wp']
    wp3, wb3 = ops.policy_pack_split(w_ih, w_hh), ops.policy_pack_split_bwd(w_ih, w_hh)
    tiles = (R + 63) // 64
    out = []
    for fused in (False, True, True):    # (twice: a cold and a warm launch agree bit for bit — see gb_settle in csrc/gates_bwd.hip)
        x = xh.clone()
        dgates = torch.full((R, 4 * H), float('nan'), device='cuda')
        dcp = torch.full((R, H), float('nan'), device='cuda')
        parts = torch.zeros((tiles, 4 * H), device='cuda')
        dxh = torch.full((R, 2 * H), float('nan'), device='cuda') if fused else None
        ops.lstm_gates_backward(x, wp, b, c_prev, dh, dc, dgates, dcp, parts, True, h_prev=h_prev, lstm_wp3=wp3,
                                lstm_wp3_bwd=wb3 if fused else None, dxh=dxh)
        out.append((dgates, dcp, parts, x, dxh))
    for k in range(4):
        assert torch.equal(out[0][k], out[1][k]), k
    ref = out[1][0].double() @ torch.cat([w_ih, w_hh], 1).double()
    err = float((out[1][4].double() - ref).abs().max())
    assert err <= 2e-6 * max(1.0, float(ref.abs().max())), err
    assert torch.equal(out[1][4], out[2][4]) and torch.equal(out[1][0], out[2][0])


@pytest.mark.parametrize("H,R", [(128, 64 * 150 + 37), (64, 150)])
def test_gates_backward_from_recorded_gates(H, R):
    """ic3_lstm_gates_backward_given (round 5): the cell's derivative of the activated gates handed in (what the rollout's step
    launch stored), h_prev copied into the h half of xh, dx in the same launch — against the closed form in float64; and a
    policy step's recorded gates give the gradient the recomputing launch gives (1e-6)."""
    from ic3net_amd import ops
    gen = torch.Generator(device='cuda').manual_seed(5 * H + R)
    rn = lambda *s: torch.randn(*s, device='cuda', generator=gen)
    w_ih, w_hh, c_w = rn(4 * H, H) / H ** 0.5, rn(4 * H, H) / H ** 0.5, rn(H, H)
    b = rn(4 * H)
    xh, h_prev = rn(R, 2 * H), rn(R, H)
    c_prev, dh, dc = rn(R, H), rn(R, H), rn(R, H)
    wb3 = ops.policy_pack_split_bwd(w_ih, w_hh)
    W = torch.cat([w_ih, w_hh], 1).double()
    xfull = torch.cat([xh[:, :H], h_prev], 1)
    pre = xfull.double() @ W.t() + b.double()
    acts = torch.cat([torch.sigmoid(pre[:, :H]), torch.sigmoid(pre[:, H:2 * H]), torch.tanh(pre[:, 2 * H:3 * H]),
                      torch.sigmoid(pre[:, 3 * H:])], 1).float().contiguous()
    a = acts.double()
    i, f, g, o = a[:, :H], a[:, H:2 * H], a[:, 2 * H:3 * H], a[:, 3 * H:]
    tc = torch.tanh(f * c_prev.double() + i * g)
    dct = dc.double() + dh.double() * o * (1 - tc * tc)
    want = torch.cat([dct * g * i * (1 - i), dct * c_prev.double() * f * (1 - f), dct * i * (1 - g * g), dh.double() * tc * o * (1 - o)], 1)
    tiles = (R + 63) // 64
    x = xh.clone()
    dgates = torch.full((R, 4 * H), float('nan'), device='cuda')
    dcp = torch.full((R, H), float('nan'), device='cuda')
    parts = torch.zeros((tiles, 4 * H), device='cuda')
    dxh = torch.full((R, 2 * H), float('nan'), device='cuda')
    n = ops.lstm_gates_backward_given(acts, c_prev, dh, dc, dgates, dcp, parts, True, xh=x, h_prev=h_prev, lstm_wp3_bwd=wb3, dxh=dxh)
    assert n == tiles and torch.equal(x, xfull)
    assert float((dgates.double() - want).abs().max()) <= 2e-6 * max(1.0, float(want.abs().max()))
    assert float((dcp.double() - dct * f).abs().max()) <= 2e-6 * max(1.0, float((dct * f).abs().max()))
    assert float((parts.double().sum(0) - want.sum(0)).abs().max()) <= 1e-4 * max(1.0, float(want.sum(0).abs().max()))
    ref = dgates.double() @ W
    assert float((dxh.double() - ref).abs().max()) <= 2e-6 * max(1.0, float(ref.abs().max()))
    # the recomputing launch on the same step: its own activations differ from float64's by rounding only
    wp = ops.policy_step_pack(c_w, w_ih, w_hh)['ps_l_wp']
    wp3 = ops.policy_pack_split(w_ih, w_hh)
    dg2, dcp2, dx2 = torch.empty_like(dgates), torch.empty_like(dcp), torch.empty_like(dxh)
    ops.lstm_gates_backward(xh.clone(), wp, b, c_prev, dh, dc, dg2, dcp2, None, False, h_prev=h_prev, lstm_wp3=wp3, lstm_wp3_bwd=wb3, dxh=dx2)
    assert float((dg2 - dgates).abs().max()) <= 4e-6 * max(1.0, float(want.abs().max()))
    assert float((dx2 - dxh).abs().max()) <= 6e-6 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("H,R,OT", [(128, 64 * 150 + 37, 8), (64, 150, 5), (128, 640, 16)])
def test_gates_backward_given_in_place_with_the_heads_share(H, R, OT):
    """Round 6: ic3_lstm_gates_backward_given with dhead / w_heads (dh + dhead . w_heads is what the cell sees) and with
    dgates written over the gates (in place) = the same launch on a dh that already holds the heads' share, out of place;
    cold and warm launches agree bit for bit (builtin MFMAs: the compiler places the wait states)."""
    from ic3net_amd import ops
    gen = torch.Generator(device='cuda').manual_seed(7 * H + R)
    rn = lambda *s: torch.randn(*s, device='cuda', generator=gen)
    w_ih, w_hh = rn(4 * H, H) / H ** 0.5, rn(4 * H, H) / H ** 0.5
    wb3 = ops.policy_pack_split_bwd(w_ih, w_hh)
    acts = torch.rand((R, 4 * H), device='cuda', generator=gen)
    acts[:, 2 * H:3 * H] = acts[:, 2 * H:3 * H] * 2 - 1
    c_prev, dh, dc = rn(R, H), rn(R, H), rn(R, H)
    dhead, w_heads = rn(R, OT), rn(OT, H) / H ** 0.5
    dh_full = (dh.double() + dhead.double() @ w_heads.double()).float()
    tiles = (R + 63) // 64
    ref = [torch.empty((R, 4 * H), device='cuda'), torch.empty((R, H), device='cuda'), torch.empty((R, 2 * H), device='cuda'),
           torch.zeros((tiles, 4 * H), device='cuda')]
    ops.lstm_gates_backward_given(acts, c_prev, dh_full, dc, ref[0], ref[1], ref[3], True, lstm_wp3_bwd=wb3, dxh=ref[2])
    runs = []
    for _ in range(3):
        g = acts.clone()
        got = [g, torch.empty((R, H), device='cuda'), torch.empty((R, 2 * H), device='cuda'), torch.zeros((tiles, 4 * H), device='cuda')]
        ops.lstm_gates_backward_given(g, c_prev, dh, dc, g, got[1], got[3], True, lstm_wp3_bwd=wb3, dxh=got[2], dhead=dhead,
                                      w_heads=w_heads)
        runs.append(got)
    for u, v in zip(runs[0], ref):      # (the fold adds its OT terms one at a time in fp32: the last ulp of dh may differ)
        assert float((u - v).abs().max()) <= 3e-6 * max(1.0, float(v.abs().max()))
    for k in range(4):
        assert torch.equal(runs[0][k], runs[1][k]) and torch.equal(runs[1][k], runs[2][k]), k


def _mix(x, alive, gate, avg):
    """comm.py:181-205 in closed form, float64 (E, N, H)"""
    E, N, H = x.shape
    al = torch.ones((E, N), device=x.device, dtype=torch.float64) if alive is None else alive.double()
    g = al * (1.0 if gate is None else gate.double())
    S = (g.unsqueeze(2) * x).sum(1, keepdim=True)
    n_alive = al.sum(1)
    scale = torch.where(n_alive > 1, 1.0 / (n_alive - 1).clamp(min=1), torch.ones_like(n_alive)) if avg else torch.ones_like(n_alive)
    return g.unsqueeze(2) * (S - g.unsqueeze(2) * x) * scale.view(E, 1, 1)


@pytest.mark.parametrize("H,N,E,avg,masks", [(128, 10, 8192, True, 'both'), (128, 20, 1000, True, 'alive'), (64, 3, 5000, False, 'gate'),
                                             (128, 32, 333, True, None), (64, 64, 17, True, 'both'), (128, 10, 5, True, None)])
def test_comm_backward_one_launch(H, N, E, avg, masks):
    """ic3_comm_backward: dh_out = (d h_direct + (M d inp) . C) * out_scale, dC partials = (M d inp)^T . h_prev — against
    d h_direct + M (d inp . C) and d inp^T . (M h_prev) in float64; launched twice: reproducible."""
    from ic3net_amd import ops
    gen = torch.Generator(device='cuda').manual_seed(H + N + E)
    rn = lambda *s: torch.randn(*s, device='cuda', generator=gen)
    R = E * N
    dxh, hp, cw = rn(R, 2 * H), rn(R, H), rn(H, H) / H ** 0.5
    alive = (torch.rand((E, N), device='cuda', generator=gen) < 0.8).to(torch.int32) if masks in ('alive', 'both') else None
    gate = (torch.rand((E, N), device='cuda', generator=gen) < 0.6).to(torch.int32) if masks in ('gate', 'both') else None
    scale = (torch.rand(R, device='cuda', generator=gen) < 0.7).float()
    dinp, dhd = dxh[:, :H].double(), dxh[:, H:].double()
    want_dh = (dhd + _mix((dinp @ cw.double()).view(E, N, H), alive, gate, avg).view(R, H)) * scale.double().unsqueeze(1)
    want_dc = dinp.t() @ _mix(hp.double().view(E, N, H), alive, gate, avg).view(R, H)
    outs = []
    for _ in range(2):
        dh = torch.full((R, H), float('nan'), device='cuda')
        parts = torch.zeros((ops.comm_backward_partials(E, N), H, H), device='cuda')
        ops.comm_backward(dxh, hp, alive, gate, cw, dh, parts, E, N, mode_avg=avg, out_scale=scale)
        outs.append((dh, parts))
    dh, parts = outs[0]
    assert float((dh.double() - want_dh).abs().max()) <= 2e-6 * max(1.0, float(want_dh.abs().max()))
    got = parts.double().sum(0)
    assert float((got - want_dc).abs().max()) <= 2e-6 * (R ** 0.5) * max(1.0, float(hp.abs().max()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    dz = torch.empty((R, H), device='cuda')
    ops.comm_backward(dxh, None, None, None, None, dz, None, E, N, comm_zero=True, out_scale=scale)
    assert torch.equal(dz, dxh[:, H:] * scale.unsqueeze(1))


@pytest.mark.parametrize("split", [True, False], ids=["bf16x9", "fp32"])
@pytest.mark.parametrize("H,T,R,live", [(128, 5, 81920, False), (128, 3, 1000, True), (64, 7, 333, True), (128, 1, 17, False),
                                        (64, 2, 30000, False)])
def test_weight_gradient_of_a_window_in_one_launch(H, T, R, live, split):
    """ic3_lstm_weight_grad over T x R rows (K slices across the CUs) against the float64 product of [inp | h * live]^T . dgates;
    added on top of what dW holds; reproducible.  Both arithmetic modes: nine exact bf16 x bf16 products per fp32 product (the
    default, the rollout's gate_split arithmetic) and the fp32 matrix instruction."""
    from ic3net_amd import ops
    gen = torch.Generator(device='cuda').manual_seed(H + T + R)
    rn = lambda *s: torch.randn(*s, device='cuda', generator=gen)
    xh, hs, dg = rn(T, R, 2 * H), rn(T, R, H), rn(T, R, 4 * H)
    lv = (torch.rand((T, R), device='cuda', generator=gen) < 0.8).float() if live else None
    x = torch.cat([xh[:, :, :H].double(), hs.double() * (lv.double().unsqueeze(2) if live else 1.0)], 2).view(T * R, 2 * H)
    want = 1.0 + x.t() @ dg.double().view(T * R, 4 * H)
    outs = []
    for _ in range(2):
        dW = torch.ones((2 * H, 4 * H), device='cuda')
        ops.lstm_weight_grad(xh, hs, dg, dW, row_live=lv, split=split)
        outs.append(dW)
    tol = 2e-6 * (T * R) ** 0.5 * 16
    assert float((outs[0].double() - want).abs().max()) <= tol
    assert torch.equal(outs[0], outs[1])
