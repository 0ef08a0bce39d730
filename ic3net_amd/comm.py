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


DEFAULT_GATE_SPLIT = True     # (tests run the policy parity set in both modes by flipping this)


class _PaddedArgs(object):
    """The args of a policy, read live, with another hid_size (CommNetMLP._twin)."""

    def __init__(self, base, hid_size):
        object.__setattr__(self, '_base', base)
        object.__setattr__(self, '_hid_size', int(hid_size))

    def __getattr__(self, k):                                    # (only reached for names not set on the proxy itself)
        return getattr(object.__getattribute__(self, '_base'), k)

    def __setattr__(self, k, v):
        setattr(object.__getattribute__(self, '_base'), k, v)

    @property
    def hid_size(self):
        return object.__getattribute__(self, '_hid_size')


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
        # Trainer when args.sparse_encoder; under autograd the differentiable variant ops.env_encode is used when obs_env is set.
        self.obs_encoder = None
        # (raw env, (heads, E, N) int32 buffer): set by the Trainer for one forward call when the fused policy path may
        # draw the actions itself (ic3_lstm_cell_heads); `sampled` reports whether it did
        self.sample_into = None
        self.sampled = False
        self.obs_table = None       # envs.encode_table: per-position pre-sums of the location rows (per weight version)
        self.obs_env = None         # env handle for the differentiable variant (ops.env_encode); set by Trainer
        self._wt_cache = (None, None, None)

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
        tw = self._twin_for(x)
        if tw is not None:
            return self._forward_twin(tw, x, info)
        if self._fused_ok(x):
            return self._forward_fused(x, info)
        if self._commnet_ok(x):
            return self._forward_commnet(x, info)
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

    # ------------------------------------------------------------------------------------------
    # Hidden sizes the one-launch kernels are not built for (they exist for 64 / 128 / 256; main.py:34 takes any int):
    # during no-grad rollouts the policy runs as its ZERO-PADDED TWIN at the next such size.  Exact, not approximate: a
    # padded hidden unit has zero encoder / C / LSTM weights and biases, so its gate pre-activations are 0, its cell
    # state stays 0.5 * 0 + 0.5 * tanh(0) = 0, its output o * tanh(0) = 0, and the zero columns of the weights that read
    # it contribute nothing to any real unit, head or value.  The twin is a CommNetMLP of its own (not a submodule: its
    # tensors are derived data like the packed weights of _fused_cache, refreshed in place when a parameter's version
    # changes); callers keep seeing (R, hid_size) hidden states — views of the twin's (R, padded) buffers.  The update
    # half runs on the twin as well (Trainer._kernel_net: the no-grad rollout records the twin's state, bptt differentiates
    # the twin, unpad_grads() cuts its gradients back to this policy's parameters — the padded entries are exactly 0).
    # ------------------------------------------------------------------------------------------
    def _twin(self):
        a = self.args
        Hp = ops.padded_hidden(self.hid_size)
        if Hp is None or not (getattr(a, 'pad_hidden', True) and getattr(a, 'fused_policy', True)
                              and getattr(a, 'mega_policy', True)) or len(self.heads) > 4:
            return None
        if a.recurrent and getattr(a, 'rnn_type', '') != 'LSTM':
            return None
        if not self.encoder.weight.is_cuda or self.encoder.weight.dtype != torch.float32:
            return None
        tw = self._padded_twin(Hp)
        tw.obs_encoder, tw.obs_table, tw.obs_env, tw.sample_into = self.obs_encoder, self.obs_table, self.obs_env, self.sample_into
        return tw

    def _padded_twin(self, Hp):
        """The twin at hidden size Hp, its parameters up to date with this policy's (any device)."""
        a = self.args
        box = self.__dict__.get('_twin_box')
        dev, dt = self.encoder.weight.device, self.encoder.weight.dtype
        if box is None or box[0].encoder.weight.device != dev or box[0].encoder.weight.dtype != dt or box[0].hid_size != Hp:
            tw = CommNetMLP(_PaddedArgs(a, Hp), self.encoder.in_features).to(device=dev, dtype=dt)
            for q in tw.parameters():
                q.requires_grad_(False)
                q.zero_()
            box = self.__dict__['_twin_box'] = [tw, None]        # (a list: not registered as a submodule)
        tw = box[0]
        key = tuple((q._version, q.data_ptr()) for q in self.parameters())
        if box[1] != key:
            H = self.hid_size
            with torch.no_grad():
                mine, theirs = dict(self.named_parameters()), dict(tw.named_parameters())
                assert mine.keys() == theirs.keys()
                for name, src in mine.items():
                    dst = theirs[name]
                    if name.startswith('hidd_encoder.'):
                        continue                                  # unused by forward (quirk Q18)
                    if a.recurrent and name.startswith('f_module.'):       # LSTMCell: four gate blocks of H rows
                        if src.dim() == 2:
                            dst.view(4, Hp, Hp)[:, :H, :H].copy_(src.view(4, H, H))
                        else:
                            dst.view(4, Hp)[:, :H].copy_(src.view(4, H))
                    elif name.startswith('heads.') or name.startswith('value_head.'):
                        if src.dim() == 2:
                            dst[:, :H].copy_(src)
                        else:
                            dst.copy_(src)
                    elif name.startswith('encoder.'):
                        dst[:H].copy_(src)
                    elif src.dim() == 2:                          # C_module(s), the non-recurrent f_module(s): H x H
                        dst[:H, :H].copy_(src)
                    else:
                        dst[:H].copy_(src)
            box[1] = key
        return tw

    def _twin_for(self, x):
        if torch.is_grad_enabled():
            return None
        x0 = x[0] if isinstance(x, (list, tuple)) else x
        if not (torch.is_tensor(x0) and x0.is_cuda):
            return None
        return self._twin()

    def _twin_done(self, tw):
        """mirror what callers read off the policy after a call"""
        for k in ('mega_steps', 'mega_forwards', 'commnet_steps', 'commnet_forwards', 'cache_generation'):
            if k in tw.__dict__:
                self.__dict__[k] = tw.__dict__[k]
        self.sampled = tw.sampled
        if getattr(tw, '_fc', None) is not None:
            self.__dict__['_fc'] = tw._fc

    def _twin_hidden(self, tw, hidden_state, cell_state, R):
        """The caller's hidden state as the twin takes it: (h, c, wide).  A state that already has the twin's width (the
        native update's episode record, trainer.py) passes through (wide = True); an (R, hid_size) state goes into the
        twin's (R, padded) buffers — without a copy when it is the views a previous call returned."""
        H, Hp = self.hid_size, tw.hid_size
        if hidden_state.shape[-1] == Hp:
            return hidden_state, cell_state, True
        mb = getattr(tw, '_mb', None)
        dev = hidden_state.device
        if mb is None or mb['h'].shape[0] != R or mb['h'].device != dev:
            mb = tw._mb = dict(h=torch.zeros((R, Hp), dtype=torch.float32, device=dev),
                               c=torch.zeros((R, Hp), dtype=torch.float32, device=dev))
        for src, k in ((hidden_state, 'h'), (cell_state, 'c')):
            buf = mb[k]
            if src.data_ptr() == buf.data_ptr() and tuple(src.shape) == (R, H) and src.stride() == (Hp, 1):
                continue
            buf.zero_()
            buf[:, :H].copy_(src.detach().reshape(R, H))
        return mb['h'], mb['c'], False

    def _twin_state_out(self, hc, wide):
        return tuple(hc) if wide else (hc[0][:, :self.hid_size], hc[1][:, :self.hid_size])

    def kernel_module(self):
        """The module the one-launch kernels and the native update (bptt) run: this policy, or its zero-padded twin."""
        tw = self._twin()
        return self if tw is None else tw

    def unpad_grads(self):
        """After a native update on the twin: p.grad of every parameter <- its region of the twin's gradient (the padded
        rows / columns of the twin's gradients are exactly zero)."""
        box = self.__dict__.get('_twin_box')
        if not box or box[0] is None:
            raise RuntimeError("unpad_grads(): this policy has no zero-padded twin (nothing ran on one: hid_size %d is a size "
                               "the kernels take as it is, or args.pad_hidden is off)" % self.hid_size)
        tw = box[0]
        H, Hp = self.hid_size, tw.hid_size
        theirs = dict(tw.named_parameters())
        for name, p in self.named_parameters():
            g = theirs[name].grad
            if g is None or name.startswith('hidd_encoder.'):
                p.grad = None
            elif self.args.recurrent and name.startswith('f_module.'):
                p.grad = (g.view(4, Hp, Hp)[:, :H, :H] if g.dim() == 2 else g.view(4, Hp)[:, :H]).reshape(p.shape).clone()
            elif name.startswith('heads.') or name.startswith('value_head.'):
                p.grad = (g[:, :H] if g.dim() == 2 else g).clone()
            elif name.startswith('encoder.'):
                p.grad = g[:H].clone()
            else:
                p.grad = (g[:H, :H] if g.dim() == 2 else g[:H]).clone()
            theirs[name].grad = None

    def _forward_twin(self, tw, x, info):
        if self.args.recurrent:
            x0, (hidden_state, cell_state) = x
            hp, cp, wide = self._twin_hidden(tw, hidden_state, cell_state, x0.size(0) * self.nagents)
            action, value, hc = tw([x0, (hp, cp)], info)
            self._twin_done(tw)
            return action, value, self._twin_state_out(hc, wide)
        out = tw(x, info)
        self._twin_done(tw)
        return out

    # ------------------------------------------------------------------------------------------
    # Fused rollout path (no autograd): the same math as forward() for the recurrent LSTM policy with one
    # communication pass, restructured around one persistent [inp | h] buffer XH (R, 2H) so that a step is
    #   encode -> XH[:, :H]      (sparse gather, bias = encoder.bias + C.bias)          1 kernel
    #   comm_masked_mean(XH[:, H:])                                                     1 kernel
    #   XH[:, :H] += comm_sum @ C^T                       (fp32 MFMA GEMM, hipBLASLt)   1 kernel
    #   gates = XH @ [W_ih | W_hh]^T + (b_ih + b_hh)      (fp32 MFMA GEMM, hipBLASLt)   1 kernel
    #   lstm_cell: (gates, c) -> c (in place), h' -> XH[:, H:]                          1 kernel
    #   policy_heads: XH[:, H:] -> [log_softmax heads | value]                          1 kernel
    # The returned (h, c) are views of internal buffers, valid until the next forward: one rollout at a time per
    # policy instance in this mode (set args.fused_policy = False to get fresh tensors from the generic path).
    # ------------------------------------------------------------------------------------------
    def _fused_ok(self, x):
        a = self.args
        if torch.is_grad_enabled() or not getattr(a, 'fused_policy', True):
            return False
        if not (a.recurrent and self.hid_size % 4 == 0 and len(self.heads) <= 4):
            return False
        if self.comm_passes != 1 and not self._multi_pass_ok():   # comm_passes > 1: the one-launch kernels, once per pass
            return False
        if sum(int(o) for o in a.naction_heads) + 1 > 16:
            return False
        return isinstance(x, (list, tuple)) and x[0].is_cuda and self.encoder.weight.dtype == torch.float32

    def _commnet_ok(self, x):
        """The non-recurrent module on the fused path: every communication pass in one ic3_commnet_forward launch."""
        a = self.args
        if torch.is_grad_enabled() or not getattr(a, 'fused_policy', True) or not getattr(a, 'mega_policy', True):
            return False
        if a.recurrent or getattr(self, 'continuous', False) or len(self.heads) > 4:
            return False
        if sum(int(o) for o in a.naction_heads) + 1 > 16 or not torch.is_tensor(x) or not x.is_cuda:
            return False
        return self.encoder.weight.dtype == torch.float32 and ops.commnet_forward_supported(self.hid_size, self.nagents)

    def _commnet_cache(self):
        ps = [self.encoder.weight, self.encoder.bias, self.value_head.weight, self.value_head.bias] + \
            [q for m in list(self.C_modules) + list(self.f_modules) for q in (m.weight, m.bias)] + \
            [q for hd in self.heads for q in (hd.weight, hd.bias)]
        key = tuple((q._version, q.data_ptr()) for q in ps) + (bool(getattr(self.args, 'gate_split', True)),)
        if getattr(self, '_cn_key', None) != key:
            with torch.no_grad():
                wt = self.encoder.weight.t().contiguous()
                wp, bias = ops.commnet_pack([m.weight for m in self.C_modules], [m.weight for m in self.f_modules],
                                            [m.bias for m in self.C_modules], [m.bias for m in self.f_modules])
                # args.gate_split (the default): the [comm | h] . [C_i | F_i]^T product as exact bf16 split products too
                wp3 = ops.commnet_pack_split([m.weight for m in self.C_modules], [m.weight for m in self.f_modules]) \
                    if getattr(self.args, 'gate_split', True) else None
                self._cn = dict(wt=wt, enc_bias=self.encoder.bias.detach().contiguous(), wp=wp, wp3=wp3, bias=bias,
                                loc_table=self.obs_table(wt) if self.obs_table is not None else None,
                                w_heads=torch.cat([hd.weight for hd in self.heads] + [self.value_head.weight], 0).contiguous(),
                                b_heads=torch.cat([hd.bias for hd in self.heads] + [self.value_head.bias], 0).contiguous())
            self._cn_key = key
        return self._cn

    def _forward_commnet(self, x, info):
        n, H = self.nagents, self.hid_size
        batch = x.size(0)
        R, dev = batch * n, x.device
        cn = self._commnet_cache()
        alive = self._mask(info, 'alive_mask', batch, dev)
        comm_action = self._mask(info, 'comm_action', batch, dev) if self.args.hard_attn else None
        mode_avg = hasattr(self.args, 'comm_mode') and self.args.comm_mode == 'avg'
        buf = getattr(self, '_cnb', None)
        if buf is None or buf.shape[0] != R or buf.device != dev:
            buf = self._cnb = torch.empty((R, H), dtype=torch.float32, device=dev)
        if self._x_is_env_obs(x):
            self.obs_encoder(cn['wt'], cn['enc_bias'], out=buf, loc_table=cn['loc_table'])
        else:
            torch.addmm(cn['enc_bias'], x.reshape(R, -1), cn['wt'], out=buf)             # dense encoder GEMM
        out = ops.commnet_forward(buf, batch, n, cn['wp'], cn['bias'], cn['w_heads'], cn['b_heads'],
                                  self.args.naction_heads, mode_avg, bool(self.args.comm_mask_zero), alive, comm_action,
                                  wp3=cn.get('wp3'))
        self.commnet_forwards = getattr(self, 'commnet_forwards', 0) + 1
        self.sampled = False
        action, value = self._split_out(out, batch, n)
        return action, value.reshape(batch, n, 1)                 # comm.py:228 on a (B, N, H) hidden state (shape quirk Q25)

    def commnet_step_ok(self, env, x):
        """True when step_env_commnet() may replace forward + select_action + env.step for this input (the non-recurrent
        module, the env's own observation)."""
        tw = self._twin_for(x)
        if tw is not None:
            return tw.commnet_step_ok(env, x)
        if not self._commnet_ok(x) or not hasattr(env, '_h'):
            return False
        if getattr(self.obs_encoder, '__self__', None) is not env or not self._x_is_env_obs(x):
            return False
        if x.shape[0] != env.nenvs or self.nagents != env.nagents_env:
            return False
        return ops.commnet_step_supported(env, self.hid_size)

    def step_env_commnet(self, env, x, info, action, reward, done, alive=None, is_completed=None, obs=None, out=None, h_in=None,
                         h_out=None):
        """trainer.py:61-67 for the non-recurrent module in ONE launch (ic3_commnet_step): action_out, value =
        forward(x, info); `action` (heads, E, N) int32 <- the draws; env.step(action[0]) -> reward / done / alive /
        is_completed; `obs`, when given, receives the dense observation of the state this call acts on.  h_in / h_out (E*N, H):
        the tanh recurrence of models.RNN on this module as its stand-in (one pass, communication off: ic3_commnet_step's h_in)."""
        tw = self._twin_for(x)
        if tw is not None:
            assert h_in is None, "the tanh recurrence runs at the kernels' own hidden sizes"
            r = tw.step_env_commnet(env, x, info, action, reward, done, alive, is_completed, obs, out)
            self._twin_done(tw)
            return r
        n, H = self.nagents, self.hid_size
        batch = x.size(0)
        R, dev = batch * n, x.device
        cn = self._commnet_cache()
        alive_in = self._mask(info, 'alive_mask', batch, dev)
        comm_in = self._mask(info, 'comm_action', batch, dev) if self.args.hard_attn else None
        mode_avg = hasattr(self.args, 'comm_mode') and self.args.comm_mode == 'avg'
        heads = [int(a) for a in self.args.naction_heads]
        if out is None:
            out = torch.empty((R, sum(heads) + 1), dtype=torch.float32, device=dev)
        ops.commnet_step(env, cn, H, heads, mode_avg, bool(self.args.comm_mask_zero), alive_in, comm_in, out, action, reward,
                         done, alive, is_completed, obs, h_in=h_in, h_out=h_out)
        self.commnet_steps = getattr(self, 'commnet_steps', 0) + 1
        action_out, value = self._split_out(out, batch, n)
        return action_out, value.reshape(batch, n, 1)

    def _x_is_env_obs(self, x):
        """The sparse encoder evaluates encoder(obs(env's CURRENT integer state)) without reading x; that is only
        the answer when x IS that observation, i.e. the env's own obs buffer (which env.reset/step keep in sync with
        the state).  Any other tensor (a stored / foreign state) takes the dense GEMM."""
        if self.obs_encoder is None:
            return False
        env = getattr(self.obs_encoder, '__self__', None)
        own = getattr(env, '_obs', None)
        if own is None:
            return True                         # a custom encoder hook: the caller vouches for it
        return x.data_ptr() == own.data_ptr() and x.shape == own.shape

    def _gate_split(self):
        """args.gate_split (default on): the LSTM gate product of the one-launch kernels with every fp32 operand split
        EXACTLY into three bf16 terms and all nine cross products on the bf16 matrix cores (each product exact in fp32,
        fp32 accumulation: fp32-class arithmetic, measured error vs fp64 = the fp32 matrix instruction's).  False: the
        plain fp32 matrix instruction."""
        return bool(getattr(self.args, 'gate_split', DEFAULT_GATE_SPLIT))

    def _mega_wanted(self):
        return bool(getattr(self.args, 'mega_policy', True)) and self.hid_size in ops.POLICY_STEP_SIZES

    def _multi_pass_ok(self):
        """comm_passes > 1 on the fused path: ic3_policy_forward / ic3_policy_step once per communication pass
        (ic3_policy.pass_index / inner_pass) — recurrent LSTM policies the one-launch kernels cover."""
        a = self.args
        return self._mega_wanted() and getattr(a, 'rnn_type', '') == 'LSTM' and self.nagents <= 64

    def _fused_cache(self):
        tw = self._twin() if not torch.is_grad_enabled() else None
        if tw is not None:                                       # (the Trainer refreshes derived weights before replays)
            fc = tw._fused_cache()
            self._twin_done(tw)
            return fc
        ps = [self.encoder.weight, self.encoder.bias] + [q for m in self.C_modules for q in (m.weight, m.bias)] + [
              self.f_module.weight_ih, self.f_module.weight_hh, self.f_module.bias_ih, self.f_module.bias_hh,
              self.value_head.weight, self.value_head.bias] + [p for hd in self.heads for p in (hd.weight, hd.bias)]
        split = self._gate_split()
        key = tuple((p._version, p.data_ptr()) for p in ps) + (split,)
        if getattr(self, '_fc_key', None) != key:
            with torch.no_grad():
                new = dict(
                    wt=self.encoder.weight.t().contiguous(),
                    enc_bias=(self.encoder.bias + self.C_modules[0].bias).contiguous(),    # comm.py:206 bias, Q24
                    c_wt=self.C_modules[0].weight.t().contiguous(),
                    w_cat_t=torch.cat([self.f_module.weight_ih, self.f_module.weight_hh], 1).t().contiguous(),
                    b_cat=(self.f_module.bias_ih + self.f_module.bias_hh).contiguous(),
                    w_heads=torch.cat([hd.weight for hd in self.heads] + [self.value_head.weight], 0).contiguous(),
                    b_heads=torch.cat([hd.bias for hd in self.heads] + [self.value_head.bias], 0).contiguous())
                new['loc_table'] = self.obs_table(new['wt']) if self.obs_table is not None else None
                if self._mega_wanted():
                    new.update(ops.policy_step_pack(self.C_modules[0].weight, self.f_module.weight_ih,
                                                    self.f_module.weight_hh))
                    if split:   # the gate product as exact bf16 split products (ruling of round 3's verdict)
                        new['ps_l_wp3'] = ops.policy_pack_split(self.f_module.weight_ih, self.f_module.weight_hh)
                        if self.hid_size in (64, 128):             # the update half's fused input gradient (bptt)
                            new['ps_l_wp3_bwd'] = ops.policy_pack_split_bwd(self.f_module.weight_ih, self.f_module.weight_hh)
                    for i in range(1, self.comm_passes):         # comm_passes > 1: what pass i swaps in (C_modules[i])
                        ci = self.C_modules[i]
                        new['enc_bias_p%d' % i] = (self.encoder.bias + ci.bias).contiguous()
                        new['enc_dbias_p%d' % i] = (ci.bias - self.C_modules[0].bias).contiguous()
                        new['ps_c_wp_p%d' % i] = ops.policy_step_pack(ci.weight, self.f_module.weight_ih,
                                                                     self.f_module.weight_hh)['ps_c_wp']
                old = getattr(self, '_fc', None)
                same = old is not None and old.keys() == new.keys() and all(
                    (old[k] is None) == (new[k] is None) and (new[k] is None or (old[k].shape == new[k].shape
                                                                                 and old[k].device == new[k].device))
                    for k in new)
                if same:
                    # a new weight version: refresh the derived tensors IN PLACE — captured step graphs hold their
                    # addresses (a re-allocation would leave the graphs reading freed / stale memory)
                    for k, v in new.items():
                        if v is not None:
                            old[k].copy_(v)
                else:
                    self._fc = new
                    self.cache_generation = getattr(self, 'cache_generation', 0) + 1   # addresses changed: graphs stale
            self._fc_key = key
        return self._fc

    def _forward_fused(self, x, info):
        n, H = self.nagents, self.hid_size
        x, (hidden_state, cell_state) = x
        batch = x.size(0)
        R = batch * n
        dev = x.device
        fc = self._fused_cache()
        alive = self._mask(info, 'alive_mask', batch, dev)
        comm_action = self._mask(info, 'comm_action', batch, dev) if self.args.hard_attn else None
        mode_avg = hasattr(self.args, 'comm_mode') and self.args.comm_mode == 'avg'
        self.sampled = False
        if 'ps_l_wp' in fc and n <= 64:
            # everything after the encoder in ONE launch (ic3_policy_forward: communication block, C, LSTMCell, heads,
            # log_softmax; comm / inp / gates stay in LDS and registers)
            mb = getattr(self, '_mb', None)
            if mb is None or mb['h'].shape[0] != R or mb['h'].device != dev:
                mb = self._mb = dict(h=torch.empty((R, H), dtype=torch.float32, device=dev),
                                     c=torch.empty((R, H), dtype=torch.float32, device=dev))
            h, c = mb['h'], mb['c']
            if hidden_state.data_ptr() != h.data_ptr():            # fresh hidden state from the caller (t = 0)
                h.copy_(hidden_state.detach().reshape(R, H))
            if cell_state.data_ptr() != c.data_ptr():
                c.copy_(cell_state.detach().reshape(R, H))
            enc = mb.get('enc')
            if enc is None:
                enc = mb['enc'] = torch.empty((R, H), dtype=torch.float32, device=dev)
            if self._x_is_env_obs(x):
                self.obs_encoder(fc['wt'], fc['enc_bias'], out=enc, loc_table=fc['loc_table'])
            else:
                torch.addmm(fc['enc_bias'], x.reshape(R, -1), fc['wt'], out=enc)       # dense encoder GEMM
            mz = bool(self.args.comm_mask_zero)
            self.mega_forwards = getattr(self, 'mega_forwards', 0) + 1
            for i in range(self.comm_passes - 1):                 # comm.py:179: every pass but the last updates h, c only
                ops.policy_forward(fc, H, self.args.naction_heads, mode_avg, mz, self._enc_of_pass(fc, mb, enc, i), batch, n,
                                   h, c, alive, comm_action, pass_index=i, inner=True)
            last = self.comm_passes - 1
            out = ops.policy_forward(fc, H, self.args.naction_heads, mode_avg, mz, self._enc_of_pass(fc, mb, enc, last),
                                     batch, n, h, c, alive, comm_action, pass_index=last)
            return self._split_out(out, batch, n) + ((h, c),)
        buf = getattr(self, '_fb', None)
        if buf is None or buf['xh'].shape[0] != R or buf['xh'].device != dev:
            buf = self._fb = dict(xh=torch.empty((R, 2 * H), dtype=torch.float32, device=dev),
                                  c=torch.empty((R, H), dtype=torch.float32, device=dev),
                                  comm=torch.empty((batch, n, H), dtype=torch.float32, device=dev),
                                  gates=torch.empty((R, 4 * H), dtype=torch.float32, device=dev))
        xh, c = buf['xh'], buf['c']
        h_view = xh[:, H:]
        if hidden_state.data_ptr() != h_view.data_ptr():       # fresh hidden state from the caller (t = 0)
            h_view.copy_(hidden_state.detach().reshape(R, H))
        if cell_state.data_ptr() != c.data_ptr():
            c.copy_(cell_state.detach().reshape(R, H))
        # encoder(x) + C.bias -> XH[:, :H]
        if self._x_is_env_obs(x):
            self.obs_encoder(fc['wt'], fc['enc_bias'], out=xh[:, :H], loc_table=fc['loc_table'])
        else:
            enc = buf.get('enc')
            if enc is None:
                enc = buf['enc'] = torch.empty((R, H), dtype=torch.float32, device=dev)
            torch.addmm(fc['enc_bias'], x.reshape(R, -1), fc['wt'], out=enc)           # dense encoder GEMM
            xh[:, :H].copy_(enc)
        if self.args.comm_mask_zero:
            pass                                                                       # comm.py:40-41: C(0) = bias only
        else:
            ops.comm_masked_mean_raw(xh.view(batch, n, 2 * H)[:, :, H:], alive, comm_action, mode_avg, True,
                                     out=buf['comm'])
            xh[:, :H].addmm_(buf['comm'].view(R, H), fc['c_wt'])                      # inp = enc + C(comm_sum)
        out = None
        torch.addmm(fc['b_cat'], xh, fc['w_cat_t'], out=buf['gates'])                  # all four gates (hipBLASLt)
        if getattr(self.args, 'fused_heads', True) and ops.lstm_cell_heads_ok(H):
            # cell + heads + value + log_softmax (+ the action draws when the Trainer asked for them) in one launch
            sink = self.sample_into
            out = ops.lstm_cell_heads_(buf['gates'], c, h_view, fc['w_heads'], fc['b_heads'], self.args.naction_heads,
                                       env=sink[0] if sink else None, action=sink[1] if sink else None)
            self.sampled = sink is not None
        else:
            ops.lstm_cell_(buf['gates'], c, h_view)
        if out is None:
            out = ops.policy_heads(h_view, fc['w_heads'], fc['b_heads'], self.args.naction_heads)
        return self._split_out(out, batch, n) + ((h_view, c),)

    @staticmethod
    def _enc_of_pass(fc, mb, enc, i):
        """encoder(x) + encoder.bias + C_modules[i].bias: `enc` carries pass 0's bias; later passes add the difference."""
        if i == 0 or ('enc_dbias_p%d' % i) not in fc:
            return enc
        buf = mb.get('enc_p')
        if buf is None or buf.shape != enc.shape:
            buf = mb['enc_p'] = torch.empty_like(enc)
        torch.add(enc, fc['enc_dbias_p%d' % i], out=buf)
        return buf

    def _split_out(self, out, batch, n):
        """(R, OT) [log-probs of every head | value] -> ([ (E,N,A_k) ], value (R,1))"""
        OT = out.shape[1]
        action, off = [], 0
        for A in self.args.naction_heads:
            action.append(out.view(batch, n, OT)[:, :, off:off + A])
            off += A
        return action, out[:, off:off + 1]

    # ------------------------------------------------------------------------------------------
    # One-launch rollout iteration (ic3_policy_step, csrc/policy_step.hip): forward + select_action + env.step for a
    # tile of whole envs per workgroup; encoder output, communication vectors, gates and logits never reach HBM.
    # Same contract as _forward_fused (no autograd, recurrent LSTM policy, one communication pass, state read through
    # the env's integer state), plus: the caller hands over the env and the buffers env.step would fill.
    # ------------------------------------------------------------------------------------------
    def mega_ok(self, env, x):
        """True when step_env() may replace forward + select_action + env.step for this input."""
        tw = self._twin_for(x)
        if tw is not None:
            hp, cp, _ = self._twin_hidden(tw, x[1][0], x[1][1], x[0].size(0) * self.nagents)
            return tw.mega_ok(env, [x[0], (hp, cp)])
        if not (self._mega_wanted() and self._fused_ok(x) and hasattr(env, '_h')):
            return False
        if getattr(self.obs_encoder, '__self__', None) is not env or not self._x_is_env_obs(x[0]):
            return False
        if x[0].shape[0] != env.nenvs or self.nagents != env.nagents_env:
            return False
        ok = getattr(self, '_mega_sup', None)
        if ok is None or ok[0] is not env:
            ok = self._mega_sup = (env, ops.policy_step_supported(env, self.hid_size))
        return ok[1]

    def mega_supported(self, env):
        """mega_ok() without an input: the conditions that do not depend on the tensors of a step (the Trainer asks
        before an episode starts whether that episode will run on the one-launch path)."""
        a = self.args
        tw = self._twin()
        if tw is not None:
            return tw.mega_supported(env)
        if not (self._mega_wanted() and getattr(a, 'fused_policy', True) and hasattr(env, '_h')):
            return False
        if not (a.recurrent and getattr(a, 'rnn_type', '') == 'LSTM' and len(self.heads) <= 4):
            return False
        if sum(int(o) for o in a.naction_heads) + 1 > 16 or self.encoder.weight.dtype != torch.float32:
            return False
        if getattr(self.obs_encoder, '__self__', None) is not env or self.nagents != env.nagents_env:
            return False
        ok = getattr(self, '_mega_sup', None)
        if ok is None or ok[0] is not env:
            ok = self._mega_sup = (env, ops.policy_step_supported(env, self.hid_size))
        return ok[1]

    def zero_hidden(self, batch_size, device):
        """init_hidden() for the one-launch rollout path: the persistent (h, c) buffers step_env() updates in place,
        zeroed — two fills instead of two allocations + fills + two copies at every episode start."""
        R, H = batch_size * self.nagents, self.hid_size
        tw = self._twin()
        if tw is not None:
            h, c = tw.zero_hidden(batch_size, device)
            return (h[:, :H], c[:, :H])
        mb = getattr(self, '_mb', None)
        if mb is None or mb['h'].shape[0] != R or mb['h'].device != device:
            mb = self._mb = dict(h=torch.empty((R, H), dtype=torch.float32, device=device),
                                 c=torch.empty((R, H), dtype=torch.float32, device=device))
        mb['h'].zero_()
        mb['c'].zero_()
        return (mb['h'], mb['c'])

    def step_env(self, env, x, info, action, reward, done, alive=None, is_completed=None, obs=None, hidden_out=None,
                 out=None, record_out=None):
        """action_out, value, (h, c) = forward(x, info); `action` (heads, E, N) int32 <- select_action (Philox draws
        positioned by the env's own counters); env.step(action[0]) -> reward (E,N) f32, done (E,) i32, alive /
        is_completed (E,N) i32.  trainer.py:49-67 in one launch.  `obs` (E,N,obs_dim), when given, receives the dense
        observation of the state this call ACTS ON (the reference's `state` argument of policy_net at this step),
        assembled by the same launch; the observation of the new state is what the next call writes (or env.observe())."""
        n, H = self.nagents, self.hid_size
        tw = self._twin_for(x)
        if tw is not None:
            hp, cp, wide = self._twin_hidden(tw, x[1][0], x[1][1], x[0].size(0) * n)
            assert hidden_out is None or wide, "an in-place episode record has the twin's width (Trainer._kernel_net)"
            action_out, value, hc = tw.step_env(env, [x[0], (hp, cp)], info, action, reward, done, alive,
                                                is_completed, obs, hidden_out, out, record_out)
            self._twin_done(tw)
            return action_out, value, self._twin_state_out(hc, wide)
        x, (hidden_state, cell_state) = x
        batch = x.size(0)
        R = batch * n
        dev = x.device
        fc = self._fused_cache()
        mb = getattr(self, '_mb', None)
        if mb is None or mb['h'].shape[0] != R or mb['h'].device != dev:
            mb = self._mb = dict(h=torch.empty((R, H), dtype=torch.float32, device=dev),
                                 c=torch.empty((R, H), dtype=torch.float32, device=dev))
        h, c = mb['h'], mb['c']
        if hidden_out is not None and self.comm_passes == 1:
            # the caller keeps the state ENTERING every step (the update half's episode record): read it where it is, write
            # h', c' into the next slot (ic3_env_set_hidden_out) — no copies
            h, c = hidden_state, cell_state
            assert h.is_contiguous() and c.is_contiguous() and tuple(h.shape) == (R, H) and h.dtype == torch.float32
            env.set_hidden_out(hidden_out[0], hidden_out[1])
            if record_out is not None:     # + the cell's activated gates and the inp rows into the record (the backward
                g_out, x_out = record_out  #   skips the gate product and what leads up to it)
                assert g_out.is_contiguous() and tuple(g_out.shape) == (R, 4 * H) and g_out.dtype == torch.float32
                assert x_out is None or (x_out.is_contiguous() and tuple(x_out.shape) == (R, 2 * H) and x_out.dtype == torch.float32)
                env.set_record_out(g_out, x_out)
        else:
            assert record_out is None, "the gate record comes with the in-place episode record (one pass)"
            hidden_out = None
            if hidden_state.data_ptr() != h.data_ptr():        # fresh hidden state from the caller (t = 0)
                h.copy_(hidden_state.detach().reshape(R, H))
            if cell_state.data_ptr() != c.data_ptr():
                c.copy_(cell_state.detach().reshape(R, H))
        alive_in = self._mask(info, 'alive_mask', batch, dev)
        comm_in = self._mask(info, 'comm_action', batch, dev) if self.args.hard_attn else None
        mode_avg = hasattr(self.args, 'comm_mode') and self.args.comm_mode == 'avg'
        heads = [int(a) for a in self.args.naction_heads]
        OT = sum(heads) + 1
        if out is None:
            out = torch.empty((R, OT), dtype=torch.float32, device=dev)   # per call: a Transition keeps its action_out
        else:                        # the caller's buffer (hipGraph mode: nothing is allocated inside a captured step)
            assert out.is_contiguous() and tuple(out.shape) == (R, OT) and out.dtype == torch.float32
        mz = bool(self.args.comm_mask_zero)
        if ops.policy_step_passes_supported(fc, H, self.comm_passes) and getattr(self.args, 'passes_in_launch', True):
            # comm.py:179-218 as a loop inside ONE launch (round 5): the hidden state stays on chip between the passes
            ops.policy_step(env, fc, H, heads, mode_avg, mz, h, c, alive_in, comm_in, out, action, reward, done, alive,
                            is_completed, obs, passes=self.comm_passes)
        else:
            for i in range(self.comm_passes - 1):                 # comm.py:179: every pass but the last updates h, c only
                ops.policy_step_pass(env, fc, H, heads, mode_avg, mz, h, c, alive_in, comm_in, i)
            ops.policy_step(env, fc, H, heads, mode_avg, mz, h, c, alive_in, comm_in, out, action, reward, done, alive,
                            is_completed, obs, pass_index=self.comm_passes - 1)
        self.mega_steps = getattr(self, 'mega_steps', 0) + 1
        return self._split_out(out, batch, n) + ((h, c) if hidden_out is None else tuple(hidden_out),)

    def _encode(self, x):
        """self.encoder(x) (comm.py:51,119); during no-grad rollouts optionally via the env's sparse gather."""
        if self.hid_size % 4 == 0 and self._x_is_env_obs(x):
            w = self.encoder.weight
            key = (w._version, w.data_ptr())
            if self._wt_cache[0] != key:
                wt = w.detach().t().contiguous()
                self._wt_cache = (key, wt, self.obs_table(wt) if self.obs_table is not None else None)
            if not torch.is_grad_enabled():
                return self.obs_encoder(self._wt_cache[1], self.encoder.bias.detach(), loc_table=self._wt_cache[2])
            if self.obs_env is not None:       # update half: same gather, backward = ic3_env_encode_backward
                return ops.env_encode(self.obs_env, w, self.encoder.bias, self._wt_cache[1], self._wt_cache[2])
        return self.encoder(x)

    def init_hidden(self, batch_size):                            # comm.py:250-253
        p = self.encoder.weight
        return tuple((torch.zeros(batch_size * self.nagents, self.hid_size, requires_grad=True, device=p.device,
                                  dtype=p.dtype),
                      torch.zeros(batch_size * self.nagents, self.hid_size, requires_grad=True, device=p.device,
                                  dtype=p.dtype)))
