"""TEST INFRASTRUCTURE ONLY (oracle).  numpy float64 restatement of the reference policy forward,
/root/reference/comm.py:99-244 (CommNetMLP), written the way the reference computes it — the
(B,N,N,H) expand / mask chain, not the closed form the HIP op uses — so that it can pin the op.
Pinned by tests/golden/policy_*.npz (outputs of the reference's own CommNetMLP).

params: dict of numpy arrays keyed like the reference state_dict:
  encoder.weight/bias, f_module.weight_ih/weight_hh/bias_ih/bias_hh (recurrent) or f_modules.{i}.weight/bias,
  C_modules.{i}.weight/bias, heads.{k}.weight/bias, value_head.weight/bias
"""
import numpy as np


def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def _linear(p, name, x):
    return x @ p[name + '.weight'].T + p[name + '.bias']


def log_softmax(z):
    z = z - z.max(-1, keepdims=True)
    return z - np.log(np.exp(z).sum(-1, keepdims=True))


def lstm_cell(p, x, h, c):
    """torch.nn.LSTMCell, gate order i,f,g,o"""
    g = x @ p['f_module.weight_ih'].T + p['f_module.bias_ih'] + h @ p['f_module.weight_hh'].T + p['f_module.bias_hh']
    H = h.shape[-1]
    i, f, gg, o = g[:, :H], g[:, H:2 * H], g[:, 2 * H:3 * H], g[:, 3 * H:]
    c2 = _sigmoid(f) * c + _sigmoid(i) * np.tanh(gg)
    h2 = _sigmoid(o) * np.tanh(c2)
    return h2, c2


def comm_block(h, alive, comm_action, comm_mode_avg=True, comm_mask_zero=False, hard_attn=True):
    """comm.py:168-205 for one pass.  h (B,N,H); alive (N,) or None; comm_action (N,) -> comm_sum (B,N,H)."""
    B, N, H = h.shape
    if alive is not None:                                   # comm.py:102-107
        agent_mask = np.asarray(alive, np.float64)
        num_alive = agent_mask.sum()
    else:
        agent_mask = np.ones(N)
        num_alive = N
    agent_mask = np.broadcast_to(agent_mask.reshape(1, 1, N), (B, N, N))[..., None].copy()   # [b,i,j] = alive[j]
    if hard_attn:                                           # comm.py:171-175
        ca = np.asarray(comm_action, np.float64)
        agent_mask = agent_mask * np.broadcast_to(ca.reshape(1, 1, N), (B, N, N))[..., None]
    agent_mask_t = agent_mask.transpose(0, 2, 1, 3)         # comm.py:177
    comm = np.broadcast_to(h[:, :, None, :], (B, N, N, H))  # [b,i,j,:] = h[b,i,:]  comm.py:181-184
    mask = np.zeros((N, N)) if comm_mask_zero else (np.ones((N, N)) - np.eye(N))   # comm.py:40-44
    comm = comm * mask.reshape(1, N, N, 1)                  # comm.py:187-192
    if comm_mode_avg and num_alive > 1:                     # comm.py:194-196
        comm = comm / (num_alive - 1)
    comm = comm * agent_mask                                # comm.py:200
    comm = comm * agent_mask_t                              # comm.py:202
    return comm.sum(axis=1)                                 # comm.py:205


def forward(p, x, hc=None, alive=None, comm_action=None, recurrent=True, comm_passes=1, comm_mode_avg=True,
            comm_mask_zero=False, hard_attn=True, nheads=2):
    """x (B,N,obs). recurrent: hc = (h, c) each (B*N,H).  Returns (list of logp (B,N,A_k), value, (h,c) or h)."""
    B, N, _ = x.shape
    enc = _linear(p, 'encoder', x)                          # comm.py:119 / :127
    H = enc.shape[-1]
    if recurrent:
        h, c = hc
        xe = enc
    else:
        xe = np.tanh(enc)                                   # comm.py:128
        h = xe
        c = None
    for i in range(comm_passes):
        hv = h.reshape(B, N, H)
        comm_sum = comm_block(hv, alive, comm_action, comm_mode_avg, comm_mask_zero, hard_attn)
        cvec = _linear(p, 'C_modules.%d' % i, comm_sum)     # comm.py:206
        if recurrent:
            inp = (xe + cvec).reshape(B * N, H)             # comm.py:209-213
            h, c = lstm_cell(p, inp, h, c)
        else:
            h = np.tanh(xe + _linear(p, 'f_modules.%d' % i, hv) + cvec)   # comm.py:220-224
    value = _linear(p, 'value_head', h)                     # comm.py:228
    hv = h.reshape(B, N, H)
    logp = [log_softmax(_linear(p, 'heads.%d' % k, hv)) for k in range(nheads)]   # comm.py:239
    return logp, value, ((h, c) if recurrent else h)


def mlp_forward(p, x, nheads=1):
    """models.py:23-34 (MLP): tanh(affine1) -> tanh(affine2(x) + x) -> heads, value."""
    x1 = np.tanh(_linear(p, 'affine1', x))
    h = np.tanh(_linear(p, 'affine2', x1) + x1)
    return [log_softmax(_linear(p, 'heads.%d' % k, h)) for k in range(nheads)], _linear(p, 'value_head', h)


def rnn_forward(p, x, prev_hid, lstm=False, nheads=1):
    """models.py:68-92 (RNN).  lstm: prev_hid = (h, c) each (B*N,H); else prev_hid (B,N,H)."""
    B, N, _ = x.shape
    enc = _linear(p, 'affine1', x)
    H = enc.shape[-1]
    if lstm:
        q = {('f_module.' + k[len('lstm_unit.'):]): v for k, v in p.items() if k.startswith('lstm_unit.')}
        h, c = lstm_cell(q, enc.reshape(B * N, H), prev_hid[0], prev_hid[1])
        ret, nh = (h, c), h.reshape(B, N, H)
    else:
        nh = np.tanh(_linear(p, 'affine2', prev_hid) + enc)
        ret = nh
    return [log_softmax(_linear(p, 'heads.%d' % k, nh)) for k in range(nheads)], _linear(p, 'value_head', nh), ret
