"""CPU: the numpy policy oracle (oracle/policy_ref.py, literal N x N x H restatement of comm.py) against
outputs of the reference's own CommNetMLP (fp64) — free-running recurrence over the fixture's steps."""
import numpy as np
import pytest

from oracle import policy_ref
from policy_util import POLICY_FIXTURES, PolicyCase


@pytest.mark.parametrize("name", POLICY_FIXTURES)
def test_policy_oracle_matches_reference(name):
    pc = PolicyCase(name)
    fx = pc.fx
    hc = (np.zeros((pc.B * pc.N, pc.H)), np.zeros((pc.B * pc.N, pc.H))) if pc.recurrent else None
    for t in range(pc.steps):
        logp, value, hc2 = policy_ref.forward(pc.params, pc.x_step(t), hc, pc.alive(t), pc.comm_action(t),
                                              recurrent=pc.recurrent, comm_passes=pc.comm_passes,
                                              comm_mode_avg=pc.mode_avg, comm_mask_zero=pc.mask_zero,
                                              hard_attn=pc.hard_attn, nheads=pc.nheads)
        for k in range(pc.nheads):
            np.testing.assert_allclose(logp[k], fx["logp%d" % k][t], rtol=0, atol=1e-12)
        np.testing.assert_allclose(value.reshape(-1, 1), fx["value"][t], rtol=0, atol=1e-12)
        if pc.recurrent:
            tol = 2e-7 if pc.state_f32 else 1e-12          # h, c of the full-size fixtures are stored as float32
            np.testing.assert_allclose(hc2[0], fx["h"][t], rtol=0, atol=tol)
            np.testing.assert_allclose(hc2[1], fx["c"][t], rtol=tol, atol=tol)
            hc = hc2


def test_closed_form_comm_equals_literal():
    """The closed form the HIP op implements (SURVEY B.5 i) against the literal expand/mask chain."""
    rs = np.random.RandomState(0)
    for trial in range(50):
        N, H, B = rs.randint(1, 9), 8, 2
        h = rs.randn(B, N, H)
        alive = (rs.rand(N) < 0.7).astype(np.float64)
        ca = (rs.rand(N) < 0.6).astype(np.float64)
        for avg in (True, False):
            lit = policy_ref.comm_block(h, alive, ca, avg, False, True)
            m = alive * ca
            S = (m[None, :, None] * h).sum(1, keepdims=True)
            n_alive = alive.sum()
            scale = 1.0 / (n_alive - 1) if (avg and n_alive > 1) else 1.0
            closed = m[None, :, None] * (S - m[None, :, None] * h) * scale
            np.testing.assert_allclose(closed, lit, atol=1e-13)


def test_sampling_oracle_known_answers():
    import oracle
    lp = np.log(np.array([0.1, 0.2, 0.3, 0.4], np.float32))
    # u = x24 / 2^24 against cdf 0.1, 0.3, 0.6
    for u, want in ((0.0, 0), (0.05, 0), (0.15, 1), (0.299, 1), (0.31, 2), (0.59, 2), (0.61, 3), (0.9999, 3)):
        assert oracle.sample_one(lp, int(u * 2 ** 24)) == want
    assert oracle.sample_one(np.log(np.array([1.0], np.float32)), 12345) == 0


BASELINES = ["baseline_mlp", "baseline_rnn", "baseline_rnn_lstm"]


def _baseline(name):
    from golden_util import load
    fx = load(name)
    N, obs_dim, H, steps, B, rec, lstm = [int(v) for v in fx["cfg"]]
    params = {k[2:]: fx[k] for k in fx.files if k.startswith("w:")}
    return fx, params, (N, obs_dim, H, steps, B, bool(rec), bool(lstm))


@pytest.mark.parametrize("name", BASELINES)
def test_baseline_oracle_and_state_dict(name):
    """IC/IRIC baselines (models.py:8-97): numpy oracle vs the reference's outputs; the product modules expose the
    reference's parameter names/shapes (checked on CPU; their GPU numerics in tests/test_policy_gpu.py)."""
    import argparse
    from ic3net_amd import models
    fx, params, (N, obs_dim, H, steps, B, rec, lstm) = _baseline(name)
    hid = None
    if rec:
        hid = (np.zeros((B * N, H)), np.zeros((B * N, H))) if lstm else np.zeros((B, N, H))
    for t in range(steps):
        if rec:
            logp, v, hid = policy_ref.rnn_forward(params, fx["x"][t], hid, lstm=lstm)
            np.testing.assert_allclose(hid[0] if lstm else hid, fx["h"][t], atol=1e-12)
        else:
            logp, v = policy_ref.mlp_forward(params, fx["x"][t])
        np.testing.assert_allclose(logp[0], fx["logp0"][t], atol=1e-12)
        np.testing.assert_allclose(v, fx["value"][t], atol=1e-12)
    a = argparse.Namespace(nagents=N, hid_size=H, continuous=False, naction_heads=[5], rnn_type='LSTM' if lstm else 'MLP')
    net = (models.RNN if rec else models.MLP)(a, obs_dim)
    sd = net.state_dict()
    assert sorted(sd) == [str(n) for n in fx["param_names"]]
    assert [str(tuple(sd[k].shape)) for k in sorted(sd)] == [str(s) for s in fx["param_shapes"]]
