"""Hidden sizes the one-launch kernels are not built for run as a ZERO-PADDED TWIN of the policy at the next size they are
(ic3net_amd/comm.py: CommNetMLP._twin; /root/reference/main.py:34 takes any hid_size).  On the CPU: the twin's parameters
are the policy's, padded; the generic forward of both agrees to rounding (extra exact zeros in every sum) over a free-running
recurrence with the padded state staying exactly zero; and the twin's gradients, cut back by unpad_grads(), are the
policy's own (what Trainer.compute_grad_native relies on).  The GPU side — the twin on the one-launch kernels against the
REFERENCE's fixtures (hid 16) — is tests/test_policy_gpu.py / test_trainer_gpu.py."""
import argparse

import numpy as np
import pytest
import torch


def _args(H, recurrent, passes=1, share=False, N=4):
    a = argparse.Namespace(
        hid_size=H, recurrent=recurrent, rnn_type='LSTM' if recurrent else 'MLP', nagents=N, comm_mode='avg',
        comm_passes=passes, comm_mask_zero=False, comm_init='uniform', hard_attn=True, share_weights=share,
        continuous=False, naction_heads=[5, 2], commnet=True, mean_ratio=0)
    return a


def _comm_masked_mean_torch(h, alive, comm_action, mode_avg, enabled):
    """The communication block (comm.py:181-205) in plain torch — a stand-in for the HIP op, which has no CPU form, so
    that the generic forward runs here; same closed form: comm_j = m_j (sum_i m_i h_i - m_j h_j) scale_e."""
    E, N, _ = h.shape
    if not enabled:
        return torch.zeros_like(h)
    al = torch.ones(E, N, dtype=h.dtype) if alive is None else alive.to(h.dtype)
    m = al if comm_action is None else al * comm_action.to(h.dtype)
    n_alive = al.sum(1, keepdim=True)
    scale = torch.where(n_alive > 1, 1.0 / (n_alive - 1).clamp(min=1), torch.ones_like(n_alive)) if mode_avg \
        else torch.ones_like(n_alive)
    S = (m.unsqueeze(2) * h).sum(1, keepdim=True)
    return m.unsqueeze(2) * (S - m.unsqueeze(2) * h) * scale.unsqueeze(2)


def test_padded_hidden_sizes():
    from ic3net_amd import ops
    assert [ops.padded_hidden(h) for h in (16, 60, 64, 100, 128, 200, 256, 300)] == [64, 64, None, 128, None, 256, None, None]


@pytest.mark.parametrize("H,recurrent,passes,share", [(20, True, 1, False), (100, True, 2, True), (36, False, 2, False),
                                                       (12, False, 3, True)])
def test_twin_forward_and_gradients_are_the_policys(H, recurrent, passes, share, monkeypatch):
    from ic3net_amd import ops
    monkeypatch.setattr(ops, 'comm_masked_mean', _comm_masked_mean_torch)
    from ic3net_amd.comm import CommNetMLP
    torch.manual_seed(H)
    N, B, obs = 4, 3, 29
    net = CommNetMLP(_args(H, recurrent, passes, share, N), obs).double()
    Hp = ops.padded_hidden(H)
    tw = net._padded_twin(Hp)
    assert tw.hid_size == Hp and 'f_module.weight_ih' in dict(tw.named_parameters()) or not recurrent
    assert not any(k.startswith('_twin') for k in net.state_dict())          # derived data, not part of the checkpoint
    for q in tw.parameters():
        q.requires_grad_(True)
    hid = net.init_hidden(B) if recurrent else None
    hid_t = tuple(torch.zeros(B * N, Hp, dtype=torch.float64) for _ in range(2)) if recurrent else None
    loss = loss_t = 0.0
    for t in range(4):
        x = torch.randn(B, N, obs, dtype=torch.float64)
        info = {'comm_action': (torch.rand(B, N) < 0.7).int()}
        if t:
            info['alive_mask'] = (torch.rand(B, N) < 0.8).int()
        if recurrent:
            lp, v, hid = net([x, hid], info)
            lpt, vt, hid_t = tw([x, hid_t], info)
            for k in range(2):
                np.testing.assert_allclose(hid_t[k][:, :H].detach().numpy(), hid[k].detach().numpy(), rtol=0, atol=1e-13)
                assert not hid_t[k][:, H:].any().item()                         # the padded units stay exactly zero
        else:
            lp, v = net(x, info)
            lpt, vt = tw(x, info)
        w = [torch.randn_like(l) for l in lp] + [torch.randn_like(v)]
        for a_, b_ in zip(lp + [v], lpt + [vt]):
            np.testing.assert_allclose(b_.detach().numpy(), a_.detach().numpy(), rtol=0, atol=1e-12)
        loss = loss + sum((l * ww).sum() for l, ww in zip(lp + [v], w))
        loss_t = loss_t + sum((l * ww).sum() for l, ww in zip(lpt + [vt], w))
    loss.backward()
    want = {k: (None if p.grad is None else p.grad.clone()) for k, p in net.named_parameters()}
    for p in net.parameters():
        p.grad = None
    loss_t.backward()
    for k, q in tw.named_parameters():                                          # nothing leaks into the padding
        if q.grad is not None and not k.startswith('hidd_encoder.'):
            mask = torch.ones_like(q.grad, dtype=torch.bool)
            ref = dict(net.named_parameters())[k]
            if recurrent and k.startswith('f_module.'):
                (mask.view(4, Hp, Hp)[:, :H, :H] if q.dim() == 2 else mask.view(4, Hp)[:, :H]).fill_(False)
            else:
                mask[tuple(slice(0, s) for s in ref.shape)] = False
            assert not q.grad[mask].any().item(), k
    net.unpad_grads()
    for k, p in net.named_parameters():
        if want[k] is None:
            assert p.grad is None, k
        else:
            np.testing.assert_allclose(p.grad.numpy(), want[k].numpy(), rtol=0, atol=1e-11, err_msg=k)


def test_twin_follows_parameter_updates_in_place():
    from ic3net_amd.comm import CommNetMLP
    torch.manual_seed(1)
    net = CommNetMLP(_args(20, True), 11)
    tw = net._padded_twin(64)
    ptr = tw.f_module.weight_ih.data_ptr()
    with torch.no_grad():
        net.f_module.weight_ih.add_(1.0)
        net.args.comm_mask_zero = True                                           # args are read live through the proxy
    tw2 = net._padded_twin(64)
    assert tw2 is tw and tw.f_module.weight_ih.data_ptr() == ptr                 # refreshed in place (graphs hold addresses)
    np.testing.assert_array_equal(tw.f_module.weight_ih.view(4, 64, 64)[:, :20, :20].numpy(),
                                  net.f_module.weight_ih.detach().view(4, 20, 20).numpy())
    assert tw.args.comm_mask_zero and tw.args.hid_size == 64 and net.args.hid_size == 20
