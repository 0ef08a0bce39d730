"""Non-communicating baselines of the reference (IC / IRIC: /root/reference/models.py:8-97), batched over E envs.
Same constructors, `forward(x, info)` contracts and parameter names as `models.MLP` / `models.RNN`; `Random`
(:37-56) draws from torch's global generator and is provided for API completeness only.

  MLP.forward(x)                 x (E,N,obs)                       -> ([logp_k (E,N,A_k)], value (E,N,1))
  RNN.forward([x, prev_hid])     rnn_type 'MLP': prev_hid (E,N,H)  -> (..., value (E,N,1), next_hid (E,N,H))
                                 rnn_type 'LSTM': (h, c) (E*N,H)   -> (..., value (E,N,1), (h, c))
"""
import torch
import torch.nn.functional as F
from torch import nn


class MLP(nn.Module):
    def __init__(self, args, num_inputs):
        super(MLP, self).__init__()
        self.args = args
        self.affine1 = nn.Linear(num_inputs, args.hid_size)
        self.affine2 = nn.Linear(args.hid_size, args.hid_size)
        self.continuous = args.continuous
        if self.continuous:
            raise NotImplementedError("continuous actions are outside the hot-path scope (neither PP nor TJ)")
        self.heads = nn.ModuleList([nn.Linear(args.hid_size, o) for o in args.naction_heads])
        self.value_head = nn.Linear(args.hid_size, 1)
        self.tanh = nn.Tanh()
        self.obs_encoder = None          # optional sparse-gather evaluation of affine1 (see comm.CommNetMLP)
        self._wt_cache = (None, None)

    def _affine1(self, x):
        if self.obs_encoder is not None and not torch.is_grad_enabled() and self.args.hid_size % 4 == 0:
            w = self.affine1.weight
            key = (w._version, w.data_ptr())
            if self._wt_cache[0] != key:
                self._wt_cache = (key, w.detach().t().contiguous())
            return self.obs_encoder(self._wt_cache[1], self.affine1.bias.detach())
        return self.affine1(x)

    def _outputs(self, h):
        return [F.log_softmax(head(h), dim=-1) for head in self.heads], self.value_head(h)

    def forward(self, x, info={}):                         # models.py:23-34
        x = self.tanh(self._affine1(x))
        h = self.tanh(self.affine2(x) + x)
        return self._outputs(h)


class Random(nn.Module):                                   # models.py:37-56
    def __init__(self, args, num_inputs):
        super(Random, self).__init__()
        self.naction_heads = args.naction_heads
        self.parameter = nn.Parameter(torch.randn(3))

    def forward(self, x, info={}):
        sizes = x.size()[:-1]
        v = torch.rand(sizes + (1,), device=x.device, requires_grad=True)
        out = [F.log_softmax(torch.randn(sizes + (o,), device=x.device, requires_grad=True), dim=-1)
               for o in self.naction_heads]
        return out, v


class _TanhRecurrence(object):
    """rnn_type 'MLP' (models.py:86-88): next = tanh(affine2(prev) + enc), hidden state (E, N, H)."""

    def __init__(self, owner):
        self.owner = owner

    def step(self, enc, state):
        nxt = torch.tanh(self.owner.affine2(state) + enc)
        return nxt, nxt

    def blank(self, envs):
        raise AttributeError("init_hidden() belongs to the LSTM variant; the Trainer starts this one from zeros (trainer.py:41)")


class _LSTMRecurrence(object):
    """rnn_type 'LSTM' (models.py:75-84): one LSTMCell over the E*N agent rows; the state handed back is a fresh pair
    (the Trainer detaches / stores it), the features keep the (E, N, H) view."""

    def __init__(self, owner):
        self.owner = owner

    def step(self, enc, state):
        o = self.owner
        envs = enc.shape[0]
        h, c = o.lstm_unit(enc.reshape(envs * o.nagents, o.hid_size), state)
        return h.view(envs, o.nagents, o.hid_size), (h.clone(), c.clone())

    def blank(self, envs):
        w = self.owner.affine1.weight
        rows = envs * self.owner.nagents
        return tuple(torch.zeros(rows, self.owner.hid_size, requires_grad=True, device=w.device, dtype=w.dtype)
                     for _ in range(2))


class RNN(MLP):
    """IRIC baseline (models.py:59-97).  Parameter names follow the reference's checkpoints: `affine1`, `heads.k`,
    `value_head`, and `affine2` (rnn_type 'MLP') or `lstm_unit` (rnn_type 'LSTM').  The recurrence itself is a small
    strategy object chosen once, so forward() has one shape: encode -> recur -> heads."""

    def __init__(self, args, num_inputs):
        super(RNN, self).__init__(args, num_inputs)      # (creates affine2 too: the reference's constructor draws its
        self.nagents = args.nagents                      #  weights from the global generator before the LSTM's, so a
        self.hid_size = args.hid_size                    #  seeded default init comes out the same)
        if args.rnn_type == 'LSTM':
            self._modules.pop('affine2')                 # not part of this variant's state_dict (models.py:64-66)
            self.lstm_unit = nn.LSTMCell(self.hid_size, self.hid_size)
            self._recur = _LSTMRecurrence(self)
        else:
            self._recur = _TanhRecurrence(self)

    def forward(self, x, info={}):
        obs, state = x
        feat, state = self._recur.step(self._affine1(obs), state)
        return self._outputs(feat) + (state,)

    def init_hidden(self, batch_size):
        return self._recur.blank(batch_size)
