"""CommNet / IC3Net policy with the reference's interface (/root/reference/comm.py:8-254): same
constructor, same `forward(x, info)` contract, same state_dict keys and shapes (so reference
checkpoints load), but batched over E environments (B = E) in fp32 on the GPU, with the O(N^2 H)
communication block replaced by the `comm_masked_mean` HIP op.

forward(x, info):
  recurrent:  x = [state (E,N,obs), (h, c) each (E*N, H)]  -> ([logp_k (E,N,A_k)], value (E*N,1), (h, c))
  otherwise:  x = state (E,N,obs)                           -> ([logp_k (E,N,A_k)], value (E,N,1))
  info['alive_mask']  (E,N) or (N,) — absent at t = 0 (all alive, quirk Q21)
  info['comm_action'] (E,N) or (N,) — gate sampled at t-1 (quirk Q22); only read when args.hard_attn
"""
import torch
import torch.nn.functional as F
from torch import nn

from . import ops


class CommNetMLP(nn.Module):
    def __init__(self, args, num_inputs):
        super(CommNetMLP, self).__init__()
        self.args = args
        self.nagents = args.nagents
        self.hid_size = args.hid_size
        self.comm_passes = args.comm_passes
        self.recurrent = args.recurrent
        self.continuous = args.continuous
        if self.continuous:
            raise NotImplementedError("continuous actions are outside the hot-path scope (neither PP nor TJ)")
        self.heads = nn.ModuleList([nn.Linear(args.hid_size, o) for o in args.naction_heads])      # comm.py:35-36
        self.init_std = args.init_std if hasattr(args, 'comm_init_std') else 0.2                    # quirk Q19
        self.encoder = nn.Linear(num_inputs, args.hid_size)                                         # comm.py:51
        if args.recurrent:
            self.hidd_encoder = nn.Linear(args.hid_size, args.hid_size)                             # unused, Q18
            self.f_module = nn.LSTMCell(args.hid_size, args.hid_size)                               # comm.py:61
        else:
            if args.share_weights:                                                                  # comm.py:64-67
                self.f_module = nn.Linear(args.hid_size, args.hid_size)
                self.f_modules = nn.ModuleList([self.f_module for _ in range(self.comm_passes)])
            else:
                self.f_modules = nn.ModuleList([nn.Linear(args.hid_size, args.hid_size)
                                                for _ in range(self.comm_passes)])
        if args.share_weights:                                                                      # comm.py:76-79
            self.C_module = nn.Linear(args.hid_size, args.hid_size)
            self.C_modules = nn.ModuleList([self.C_module for _ in range(self.comm_passes)])
        else:
            self.C_modules = nn.ModuleList([nn.Linear(args.hid_size, args.hid_size)
                                            for _ in range(self.comm_passes)])
        if args.comm_init == 'zeros':                                                               # comm.py:86-88
            for i in range(self.comm_passes):
                self.C_modules[i].weight.data.zero_()
        self.tanh = nn.Tanh()
        self.value_head = nn.Linear(self.hid_size, 1)
        # Optional fast path for the encoder during rollouts: a callable (weight_t, bias) -> (E,N,H) that evaluates
        # encoder(current observation) straight from env state (envs.encode, the sparse-gather HIP kernel).  Set by
        # Trainer when args.sparse_encoder; only used under torch.no_grad() (no backward through the gather yet).
        self.obs_encoder = None
        self._wt_cache = (None, None)

    # ------------------------------------------------------------------------------------------
    def _mask(self, info, key, batch, device):
        m = info.get(key) if isinstance(info, dict) else None
        if m is None:
            return None
        if not torch.is_tensor(m):
            m = torch.as_tensor(m)
        m = m.to(device=device, dtype=torch.int32)
        if m.dim() == 1:
            m = m.unsqueeze(0).expand(batch, -1)
        return m.reshape(batch, self.nagents).contiguous()

    def forward(self, x, info={}):
        n, H = self.nagents, self.hid_size
        if self.args.recurrent:                                   # comm.py:117-122 (no tanh on this branch)
            x, (hidden_state, cell_state) = x
            x = self._encode(x)
        else:                                                     # comm.py:127-129
            x = self.tanh(self._encode(x))
            hidden_state, cell_state = x, None
        batch = x.size(0)
        alive = self._mask(info, 'alive_mask', batch, x.device)
        comm_action = self._mask(info, 'comm_action', batch, x.device) if self.args.hard_attn else None
        mode_avg = hasattr(self.args, 'comm_mode') and self.args.comm_mode == 'avg'
        for i in range(self.comm_passes):
            h = hidden_state.view(batch, n, H)
            comm_sum = ops.comm_masked_mean(h, alive, comm_action, mode_avg, not self.args.comm_mask_zero)
            c = self.C_modules[i](comm_sum)                       # comm.py:206 (bias even with zero comm, Q24)
            if self.args.recurrent:
                inp = (x + c).view(batch * n, H)                  # comm.py:209-213
                hidden_state, cell_state = self.f_module(inp, (hidden_state, cell_state))
            else:
                hidden_state = self.tanh(x + self.f_modules[i](hidden_state) + c)   # comm.py:222-224
        value_head = self.value_head(hidden_state)                # comm.py:228 (shape quirk Q25)
        h = hidden_state.view(batch, n, H)
        action = [F.log_softmax(head(h), dim=-1) for head in self.heads]            # comm.py:239
        if self.args.recurrent:
            return action, value_head, (hidden_state.clone(), cell_state.clone())
        return action, value_head

    def _encode(self, x):
        """self.encoder(x) (comm.py:51,119); during no-grad rollouts optionally via the env's sparse gather."""
        if self.obs_encoder is not None and not torch.is_grad_enabled() and self.hid_size % 4 == 0:
            w = self.encoder.weight
            key = (w._version, w.data_ptr())
            if self._wt_cache[0] != key:
                self._wt_cache = (key, w.detach().t().contiguous())
            return self.obs_encoder(self._wt_cache[1], self.encoder.bias.detach())
        return self.encoder(x)

    def init_hidden(self, batch_size):                            # comm.py:250-253
        p = self.encoder.weight
        return tuple((torch.zeros(batch_size * self.nagents, self.hid_size, requires_grad=True, device=p.device,
                                  dtype=p.dtype),
                      torch.zeros(batch_size * self.nagents, self.hid_size, requires_grad=True, device=p.device,
                                  dtype=p.dtype)))
