"""The gate product's two arithmetic modes on operands no rollout produces (round-3 verdict, conditions of the ruling that
made the split products the default): ic3_gate_product_probe runs the gate product ALONE — gates = [inp | h] . [W_ih | W_hh]^T
(/root/reference/comm.py:215, torch.nn.LSTMCell) — through the operand layouts, the activation split and the instruction
order of ic3_policy_step's gate loops, once on the fp32 matrix instruction and once as nine exact bf16 x bf16 products per
fp32 product, and both are compared with a float64 product of the same float32 operands.

Bars: wherever the float64 result is finite and inside the float32 range, the split products' error (relative to
sum_k |x_k w_k|, the scale both roundings live on) is no larger than the fp32 matrix instruction's in the mean square
(5 % slack) and in its worst entry (25 % slack: an order statistic of ~10^5 entries); wherever it is not finite (inf / nan operands, products beyond the float32 range), both
modes return a non-finite value."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _products(x, w_ih, w_hh, H):
    """x (R, 2H) float32, weights (4H, H) float32 -> (fp32 MFMA gates, split gates) as (R, 4H) float32 arrays."""
    from ic3net_amd import _lib, ops
    from ic3net_amd._lib import check, ptr, stream
    dev = 'cuda'
    xt = torch.from_numpy(x).to(dev)
    wi, wh = torch.from_numpy(w_ih).to(dev), torch.from_numpy(w_hh).to(dev)
    packed = ops.policy_step_pack(torch.zeros((H, H), device=dev), wi, wh)
    wp3 = ops.policy_pack_split(wi, wh)
    R = x.shape[0]
    out = []
    for split in (False, True):
        g = torch.full((R, 4 * H), float('nan'), dtype=torch.float32, device=dev)
        check(_lib.lib().ic3_gate_product_probe(ptr(xt), ptr(packed['ps_l_wp']), ptr(wp3) if split else None, ptr(g), R, H,
                                                stream()))
        out.append(g.cpu().numpy())
    return out


def _errors(x, w_ih, w_hh, got):
    w = np.concatenate([w_ih, w_hh], 1).astype(np.float64)          # (4H, 2H)
    x64 = x.astype(np.float64)
    with np.errstate(all='ignore'):
        ref = x64 @ w.T
        scale = np.abs(x64) @ np.abs(w).T
    ok = np.isfinite(ref) & (np.abs(ref) < 3.0e38) & np.isfinite(scale) & (scale < 3.0e38) & (scale > 0)
    with np.errstate(all='ignore'):
        err = np.abs(got.astype(np.float64) - ref) / scale
    return ref, ok, err


@pytest.mark.parametrize("H", [64, 128, 256])
def test_split_products_are_no_less_accurate_than_the_fp32_matrix_instruction(H):
    rng = np.random.default_rng(H)
    R = 192
    x = rng.standard_normal((R, 2 * H)).astype(np.float32)
    w_ih = (rng.standard_normal((4 * H, H)) * 0.1).astype(np.float32)
    w_hh = (rng.standard_normal((4 * H, H)) * 0.1).astype(np.float32)
    f32, spl = _products(x, w_ih, w_hh, H)
    _, ok, e32 = _errors(x, w_ih, w_hh, f32)
    _, _, esp = _errors(x, w_ih, w_hh, spl)
    assert ok.all()
    # (the worst of 98 304 entries is an order statistic: it moves by +-10 % with the summation order alone; the mean
    #  square is the stable comparison)
    assert np.sqrt((esp ** 2).mean()) <= np.sqrt((e32 ** 2).mean()) * 1.05, (np.sqrt((esp ** 2).mean()), np.sqrt((e32 ** 2).mean()))
    assert esp.max() <= max(e32.max(), 2.0 ** -24) * 1.25, (esp.max(), e32.max())
    assert e32.max() < 2.0 ** -20 and esp.max() < 2.0 ** -20


def test_edge_magnitudes_1e_minus_30_to_1e_plus_30():
    """Row blocks of activations at magnitudes 1e-30 ... 1e+30 against weight columns at magnitudes that keep the products
    inside the float32 range (and some that do not), signs mixed."""
    H = 128
    rng = np.random.default_rng(7)
    mags = [1e-30, 1e-20, 1e-10, 1e-3, 1.0, 1e3, 1e10, 1e20, 1e30]
    R = 16 * len(mags)
    x = rng.standard_normal((R, 2 * H)).astype(np.float64)
    for i, m in enumerate(mags):
        x[16 * i:16 * i + 16] *= m
    x = x.astype(np.float32)
    w = rng.standard_normal((4 * H, 2 * H)).astype(np.float64)
    wm = [1e-30, 1e-15, 1e-5, 1.0, 1e5, 1e15, 1e30, 1e-8]
    for c in range(4 * H):
        w[c] *= wm[c % len(wm)]
    w = w.astype(np.float32)
    w_ih, w_hh = np.ascontiguousarray(w[:, :H]), np.ascontiguousarray(w[:, H:])
    f32, spl = _products(x, w_ih, w_hh, H)
    ref, ok, e32 = _errors(x, w_ih, w_hh, f32)
    _, _, esp = _errors(x, w_ih, w_hh, spl)
    assert ok.sum() > 0.5 * ok.size                         # most (activation, weight) magnitude pairs stay in range
    assert np.sqrt((esp[ok] ** 2).mean()) <= np.sqrt((e32[ok] ** 2).mean()) * 1.05
    assert esp[ok].max() <= max(e32[ok].max(), 2.0 ** -24) * 1.25, (esp[ok].max(), e32[ok].max())
    # beyond the float32 range both modes overflow (inf, or nan where +inf and -inf partial sums meet)
    with np.errstate(all='ignore'):
        over = ~np.isfinite(ref.astype(np.float32)) | (np.abs(ref) > 3.4e38)
    if over.any():
        assert (~np.isfinite(f32[over])).all() and (~np.isfinite(spl[over])).all()


def test_low_terms_that_go_subnormal_in_bf16():
    """|x| below ~1e-33: the second and third bf16 terms of the split fall under bf16's smallest normal number.  The
    products' ABSOLUTE error is then bounded by what a flush of those terms costs — 2^-8 |x w| per product — which for
    operands of that size is far below one ulp of any result a policy computes; measured against float64 on weights of
    ordinary size (the fp32 matrix instruction is held to the same bound)."""
    H = 64
    rng = np.random.default_rng(3)
    R = 64
    x = (rng.standard_normal((R, 2 * H)) * 1e-36).astype(np.float32)
    w_ih = rng.standard_normal((4 * H, H)).astype(np.float32)
    w_hh = rng.standard_normal((4 * H, H)).astype(np.float32)
    f32, spl = _products(x, w_ih, w_hh, H)
    w = np.concatenate([w_ih, w_hh], 1).astype(np.float64)
    ref = x.astype(np.float64) @ w.T
    bound = (np.abs(x.astype(np.float64)) @ np.abs(w).T) * 2.0 ** -8 + 1e-44
    assert (np.abs(f32 - ref) <= bound).all() and (np.abs(spl - ref) <= bound).all()


def test_inf_and_nan_operands_stay_non_finite():
    H = 64
    rng = np.random.default_rng(5)
    R = 64
    x = rng.standard_normal((R, 2 * H)).astype(np.float32)
    x[3, 5] = np.inf
    x[9, 70] = -np.inf
    x[20, 1] = np.nan
    w_ih = rng.standard_normal((4 * H, H)).astype(np.float32)
    w_hh = rng.standard_normal((4 * H, H)).astype(np.float32)
    w_ih[7, 9] = np.inf
    f32, spl = _products(x, w_ih, w_hh, H)
    for g in (f32, spl):
        assert not np.isfinite(g[3]).any() and not np.isfinite(g[9]).any() and not np.isfinite(g[20]).any()
        assert not np.isfinite(g[:, 7]).any()
        clean = np.ones(g.shape, bool)
        clean[[3, 9, 20]] = False
        clean[:, 7] = False
        assert np.isfinite(g[clean]).all()
    np.testing.assert_allclose(spl[np.isfinite(spl)], f32[np.isfinite(f32)], rtol=0, atol=1e-4)
