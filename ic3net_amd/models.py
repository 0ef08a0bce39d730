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


class RNN(MLP):
    def __init__(self, args, num_inputs):
        super(RNN, self).__init__(args, num_inputs)
        self.nagents = self.args.nagents
        self.hid_size = self.args.hid_size
        if self.args.rnn_type == 'LSTM':                   # models.py:64-66
            del self.affine2
            self.lstm_unit = nn.LSTMCell(self.hid_size, self.hid_size)

    def forward(self, x, info={}):                         # models.py:68-92
        x, prev_hid = x
        encoded_x = self._affine1(x)
        if self.args.rnn_type == 'LSTM':
            batch_size = encoded_x.size(0)
            encoded_x = encoded_x.reshape(batch_size * self.nagents, self.hid_size)
            next_hid, cell_state = self.lstm_unit(encoded_x, prev_hid)
            ret = (next_hid.clone(), cell_state.clone())
            next_hid = next_hid.view(batch_size, self.nagents, self.hid_size)
        else:
            next_hid = torch.tanh(self.affine2(prev_hid) + encoded_x)
            ret = next_hid
        action, v = self._outputs(next_hid)
        return action, v, ret

    def init_hidden(self, batch_size):                     # models.py:94-97
        p = self.affine1.weight
        return tuple((torch.zeros(batch_size * self.nagents, self.hid_size, requires_grad=True, device=p.device,
                                  dtype=p.dtype),
                      torch.zeros(batch_size * self.nagents, self.hid_size, requires_grad=True, device=p.device,
                                  dtype=p.dtype)))
