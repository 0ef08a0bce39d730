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


class _StandInArgs(object):
    """The args of a baseline, read live, as the communicating module's constructor and kernels read them: one pass, the
    communication block switched off (comm_mask_zero: C sees zeros, comm.py:40-41), no gate head."""
    _FIXED = dict(comm_passes=1, comm_mask_zero=True, hard_attn=False, share_weights=False, comm_init='uniform', commnet=True,
                  comm_mode='avg', comm_action_one=False)

    def __init__(self, base, recurrent):
        object.__setattr__(self, '_base', base)
        object.__setattr__(self, '_rec', bool(recurrent))

    def __getattr__(self, k):
        if k in _StandInArgs._FIXED:
            return _StandInArgs._FIXED[k]
        if k == 'recurrent':
            return object.__getattribute__(self, '_rec')
        if k == 'rnn_type':
            return 'LSTM' if object.__getattribute__(self, '_rec') else 'MLP'
        return getattr(object.__getattribute__(self, '_base'), k)

    def __setattr__(self, k, v):
        setattr(object.__getattribute__(self, '_base'), k, v)


class _KernelStandIn(object):
    """The ONE-LAUNCH rollout of a baseline (round 5).  models.MLP is the non-recurrent CommNet module with one pass and the
    communication block off — h = tanh(x + f(x) + C(0)), x = tanh(encoder(obs)) (comm.py:119-129,220-224 against
    models.py:23-34: encoder = affine1, f = affine2, C.bias = 0) — and models.RNN with the LSTM cell is the recurrent CommNet
    policy with the block off — LSTMCell(encoder(obs) + C.bias, (h, c)) (comm.py:209-215 against models.py:75-84).  So a
    baseline keeps a CommNetMLP of that shape whose parameters are COPIES of its own (C: zeros), refreshed when a parameter's
    version changes (like the zero-padded twin of comm.CommNetMLP), and its rollout steps run ic3_commnet_step /
    ic3_policy_step on that stand-in: sparse encoder, layer(s), heads, Philox draws, env.step and the obs rows in one launch.
    The tanh recurrence of models.RNN (rnn_type 'MLP') has no such kernel and stays a launch chain."""

    def __init__(self, owner, recurrent):
        self.owner, self.recurrent, self.net, self.key = owner, recurrent, None, None

    def get(self):
        from .comm import CommNetMLP
        o = self.owner
        w = o.affine1.weight
        if not w.is_cuda or w.dtype != torch.float32 or not getattr(o.args, 'mega_policy', True):
            return None
        if self.net is None or self.net.encoder.weight.device != w.device:
            self.net = CommNetMLP(_StandInArgs(o.args, self.recurrent), o.affine1.in_features).to(device=w.device, dtype=w.dtype)
            for q in self.net.parameters():
                q.requires_grad_(False)
                q.zero_()
            self.key = None
        key = tuple((q._version, q.data_ptr()) for q in o.parameters())
        if key != self.key:
            with torch.no_grad():
                n = self.net
                n.encoder.weight.copy_(o.affine1.weight)
                n.encoder.bias.copy_(o.affine1.bias)
                if self.recurrent:
                    for k in ('weight_ih', 'weight_hh', 'bias_ih', 'bias_hh'):
                        getattr(n.f_module, k).copy_(getattr(o.lstm_unit, k))
                else:
                    n.f_modules[0].weight.copy_(o.affine2.weight)
                    n.f_modules[0].bias.copy_(o.affine2.bias)
                for mine, theirs in zip(list(o.heads) + [o.value_head], list(n.heads) + [n.value_head]):
                    theirs.weight.copy_(mine.weight)
                    theirs.bias.copy_(mine.bias)
            self.key = key
        n = self.net
        n.obs_encoder, n.obs_table, n.sample_into = o.obs_encoder, getattr(o, 'obs_table', None), getattr(o, 'sample_into', None)
        return n

    def done(self):
        """mirror what the Trainer reads off the policy after a call"""
        for k in ('mega_steps', 'commnet_steps', 'cache_generation'):
            if k in self.net.__dict__:
                self.owner.__dict__[k] = self.net.__dict__[k]
        self.owner.sampled = self.net.sampled
        self.owner.__dict__['_fc'] = True          # (Trainer.begin_episode: derived weights exist -> refresh them before replays)

    def refresh(self):
        """Trainer.begin_episode in hipGraph mode: replays never run Python, so the copies and the packed weights derived from
        them are brought up to date here (in place: the graphs hold their addresses)."""
        n = self.get()
        if n is not None:
            n._fused_cache() if self.recurrent else n._commnet_cache()
            self.done()


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
        self.obs_table = None            # envs.encode_table (set by the Trainer like comm.CommNetMLP's)
        self.sampled = False
        self._wt_cache = (None, None)
        self.__dict__['_stand_in'] = _KernelStandIn(self, recurrent=False)     # (not a submodule: derived data)

    # ---- the whole iteration trainer.py:61-67 as ONE launch, on the stand-in (see _KernelStandIn) -------------------------
    def commnet_step_ok(self, env, x):
        if type(self) is not MLP or torch.is_grad_enabled() or not (torch.is_tensor(x) and x.is_cuda):
            return False
        n = self._stand_in.get()
        return n is not None and n.commnet_step_ok(env, x)

    def step_env_commnet(self, env, x, info, **kw):
        n = self._stand_in.get()
        out = n.step_env_commnet(env, x, info, **kw)
        self._stand_in.done()
        return out

    def _fused_cache(self):
        if self.__dict__.get('_stand_in') is not None:
            self._stand_in.refresh()

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
            self.__dict__['_stand_in'] = _KernelStandIn(self, recurrent=True)
            self.sample_into = None          # (the Trainer's fused-draw protocol: unused here, the step launch draws itself)
        else:
            self._recur = _TanhRecurrence(self)
            # round 6: h_t = tanh(affine1(obs) + affine2(h_{t-1})) is the non-recurrent stand-in's layer with x = enc (no tanh) and
            # h_0 = h_{t-1} (ic3_commnet_step's h_in): the whole iteration as one launch too
            self.__dict__['_stand_in'] = _KernelStandIn(self, recurrent=False)

    # ---- the tanh variant's iteration as ONE launch (ic3_commnet_step with h_in on the stand-in) -----------------------------
    def commnet_step_ok(self, env, x):                       # (MLP's: not for a recurrent policy)
        return False

    def rnn_step_supported(self, env):
        from . import ops
        if self.args.rnn_type == 'LSTM' or torch.is_grad_enabled() or ops.padded_hidden(self.hid_size) is not None:
            return False
        n = self._stand_in.get()
        return n is not None and hasattr(env, '_h') and getattr(self.obs_encoder, '__self__', None) is env \
            and self.nagents == env.nagents_env and ops.commnet_step_supported(env, self.hid_size)

    def rnn_step_ok(self, env, x):
        """True when step_env_rnn() may replace forward + select_action + env.step for this input: the env's own observation, a
        contiguous float32 (E, N, H) state on the device, hid_size 64 / 128 / 256."""
        if not self.rnn_step_supported(env):
            return False
        obs, h = x
        if not (torch.is_tensor(obs) and obs.is_cuda and torch.is_tensor(h) and h.is_cuda and h.dtype == torch.float32
                and h.is_contiguous() and h.numel() == obs.shape[0] * self.nagents * self.hid_size):
            return False
        return self._stand_in.get().commnet_step_ok(env, obs)

    def step_env_rnn(self, env, x, info, h_out, **kw):
        """trainer.py:61-67 for models.RNN with the tanh recurrence in ONE launch: (action_out, value, h_t), h_t in `h_out`."""
        n = self._stand_in.get()
        action_out, value = n.step_env_commnet(env, x[0], info, h_in=x[1], h_out=h_out, **kw)
        self._stand_in.done()
        return action_out, value, h_out

    # ---- the LSTM variant's iteration as ONE launch (ic3_policy_step on the stand-in) ------------------------------------------
    def _kernel(self, x=None):
        if self._stand_in is None or self.args.rnn_type != 'LSTM' or torch.is_grad_enabled():
            return None
        if x is not None and not (torch.is_tensor(x[0]) and x[0].is_cuda):
            return None
        return self._stand_in.get()

    def mega_supported(self, env):
        n = self._kernel()
        return n is not None and n.mega_supported(env)

    def mega_ok(self, env, x):
        n = self._kernel(x)
        return n is not None and n.mega_ok(env, x)

    def record_inplace_ok(self):
        """The native update's episode record can serve as the step launch's (h, c) buffers only when the stand-in runs at
        this hidden size itself (64 / 128 / 256), not as a zero-padded twin with wider rows."""
        from . import ops
        return ops.padded_hidden(self.hid_size) is None

    def zero_hidden(self, batch_size, device):
        n = self._kernel()
        return n.zero_hidden(batch_size, device) if n is not None else self.init_hidden(batch_size)

    def step_env(self, env, x, info, **kw):
        n = self._stand_in.get()
        out = n.step_env(env, x, info, **kw)
        self._stand_in.done()
        return out

    def forward(self, x, info={}):
        obs, state = x
        feat, state = self._recur.step(self._affine1(obs), state)
        return self._outputs(feat) + (state,)

    def init_hidden(self, batch_size):
        return self._recur.blank(batch_size)
