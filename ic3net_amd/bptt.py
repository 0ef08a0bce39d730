"""The update half without autograd — /root/reference/trainer.py:128-225 (`compute_grad`) for the recurrent CommNet /
IC3Net policy (/root/reference/comm.py:134-244), as an explicit backward-through-time over what a NO-GRAD rollout
recorded.

The reference (and round 1/2 here) keeps the autograd graph of the whole rollout: every intermediate of every step —
encoder output, communication vectors, gate pre-activations, log-softmax inputs — stays alive until `loss.backward()`
(59.5 GB for one PP-hard update at 8192 envs), and the rollout has to run through differentiable ops, i.e. not through
the one-launch kernel.  Here the rollout is the ordinary rollout (ic3_policy_step where it applies) and records per step
only what cannot be re-derived: the recurrent state (h, c) that ENTERED the step, a snapshot of the env's integer state
(a few hundred bytes per env) and the two masks.  The backward pass walks the steps in reverse and, per step,
  re-evaluates   enc (ic3_env_encode_at on the snapshot), comm (ic3_comm_masked_mean), inp, the gate pre-activations
  backpropagates heads -> LSTM cell (ic3_lstm_cell_backward: dgates, dc) -> [W_ih | W_hh] -> C -> comm (its mixing matrix
                 is symmetric: the same kernel on the gradient) -> encoder (ic3_env_encode_backward on the snapshot)
with the hidden state cut every `detach_gap` steps exactly where the reference detaches it (trainer.py:56-60).  The dense
products are plain library GEMMs (hipBLASLt fp32 MFMA through torch.mm / addmm_: [R x 4H] x [4H x 2H] and its
transposes — there is nothing to fuse into them that the HBM traffic of their operands would notice); everything
pointwise or sparse is a HIP kernel of libic3rollout.

Cost per step (PP-hard): forward 0.30 GFLOP-equivalents of one gate GEMM, backward = recompute (1x) + input gradient (1x)
+ weight gradient (1x): fp32 arithmetic bounds the update at ~4x the rollout's matrix work.
"""
import torch

from . import ops


def _is_baseline(net):
    from . import models
    return isinstance(net, models.MLP)


def supported(args, net, raw):
    """Recurrent LSTM CommNet / IC3Net (any number of communication passes), or the NON-recurrent CommNet module (round 4:
    `_backward_episode_commnet`), with the sparse encoder bound to this env."""
    if _is_baseline(net):
        # IC / IRIC baselines (models.MLP / models.RNN, no communication): round 4, _backward_episode_baseline
        if args.hid_size % 4 or not hasattr(raw, 'encode_at') or getattr(net, 'continuous', False):
            return False
        if getattr(net.obs_encoder, '__self__', None) is not raw or args.nagents != raw.nagents_env:
            return False
        if getattr(args, 'recurrent', False) and getattr(args, 'rnn_type', 'MLP') == 'LSTM':
            h4 = args.hid_size // 4
            if h4 > 64 or h4 & (h4 - 1):                          # ic3_lstm_cell_backward
                return False
        return net.affine1.weight.is_cuda and net.affine1.weight.dtype == torch.float32
    if not getattr(args, 'recurrent', False):
        if not (hasattr(net, 'f_modules') and hasattr(net, '_commnet_cache')) or getattr(net, 'continuous', False):
            return False
        if not hasattr(raw, 'encode_at') or not ops.commnet_forward_supported(args.hid_size, net.nagents):
            return False
        if getattr(net.obs_encoder, '__self__', None) is not raw or net.nagents != raw.nagents_env:
            return False
        return net.encoder.weight.is_cuda and net.encoder.weight.dtype == torch.float32
    if not (getattr(args, 'recurrent', False) and getattr(args, 'rnn_type', '') == 'LSTM' and hasattr(net, 'f_module')):
        return False
    if getattr(net, 'comm_passes', 1) < 1 or args.hid_size % 4 or not hasattr(raw, 'encode_at'):
        return False
    h4 = args.hid_size // 4
    if h4 > 64 or h4 & (h4 - 1):        # ic3_lstm_cell_backward (the non-fused step): H/4 a power of two <= 64;
        return False                    # other sizes keep the autograd update
    if getattr(net.obs_encoder, '__self__', None) is not raw or net.nagents != raw.nagents_env:
        return False
    return net.encoder.weight.is_cuda and net.encoder.weight.dtype == torch.float32


class EpisodeRecord(object):
    """What one batched episode leaves behind for the backward pass: (h, c) entering every step, the env state of every
    step, the masks of the communication block."""

    def __init__(self, T, R, H, state_words, device, recurrent=True):
        self.recurrent = recurrent
        if recurrent:
            self.hs = torch.empty((T + 1, R, H), dtype=torch.float32, device=device)     # slot t: h entering step t; slot T: h_T
            self.cs = torch.empty((T + 1, R, H), dtype=torch.float32, device=device)
        else:                                  # the non-recurrent module carries no state between steps: env snapshots + masks only
            self.hs = self.cs = None
            self.rows, self.device = R, device
        self.snaps = torch.empty((T, state_words), dtype=torch.int32, device=device)
        self.alive = [None] * T
        self.gate = [None] * T
        self.h_last = None
        self.n = 0
        # (T, R, 4H) the activated gates of every step's LSTM cell and (T, R, 2H) the inp half of its [inp | h] rows, stored by
        # the step launch itself (ic3_env_set_record_out: Trainer._record_gates) — the backward then skips the gate product and
        # what leads up to it; gates_n = the steps that stored theirs
        self.gates = None
        self.xh = None
        self.gates_n = 0
        self.stream = None     # collection mode (Trainer._run_batch_streams): per-slot cuts of the recurrence, see backward_episode
        # (T, R, OT) the [log-probs | value] rows of every step, written by the one-launch steps themselves (Trainer hands slice t
        # as their `out`); out_n = the steps that did
        self.out = None
        self.out_n = 0

    def release(self):
        """Drop the record's tensors now (tens of GB with recorded gates): the update is done with them, and whatever still
        refers to the record object — a reference cycle waiting for the garbage collector — must not keep them alive beside
        the next update's record."""
        self.hs = self.cs = self.gates = self.xh = self.snaps = self.h_last = self.out = None
        self.alive, self.gate, self.stream = [], [], None

    def start_from(self, h, c):
        """Collection mode: this window continues the streams of the previous one — the state it ended with is the state
        entering slot 0 (an env that starts an episode there is zeroed inside the step launch, and again in the backward)."""
        R, H = self.hs.shape[1:]
        if h.shape[-1] != H:
            self._put(self.hs[0], h)
            self._put(self.cs[0], c)
        else:
            self.hs[0].copy_(h.detach().reshape(R, H))
            self.cs[0].copy_(c.detach().reshape(R, H))
        return (self.hs[0], self.cs[0])

    def start(self):
        """Zero state of step 0, held in the record itself: a rollout that reads (h, c) from slot t and lets the step launch
        write slot t + 1 (ic3_env_set_hidden_out) records the recurrent state without a single copy."""
        self.hs[0].zero_()
        self.cs[0].zero_()
        return (self.hs[0], self.cs[0])

    def slot(self, t):
        return (self.hs[t], self.cs[t])

    def record(self, t, net, raw, prev_hid, info):
        if self.recurrent:
            h, c = prev_hid if isinstance(prev_hid, (tuple, list)) else (prev_hid, None)   # (models.RNN, rnn_type MLP: h only)
            R, H = self.hs.shape[1:]
            if h.shape[-1] != H:                               # a zero-padded twin's record, an (R, hid_size) state
                self._put(self.hs[t], h)
                if c is not None:
                    self._put(self.cs[t], c)
            elif h.data_ptr() != self.hs[t].data_ptr():        # (in-place rollouts hand slot t itself)
                self.hs[t].copy_(h.detach().reshape(R, H))
                if c is not None:
                    self.cs[t].copy_(c.detach().reshape(R, H))
            dev = self.hs.device
        else:
            R, dev = self.rows, self.device
        raw.snapshot(out=self.snaps[t])
        if hasattr(net, '_mask'):                              # the communication block's masks (CommNetMLP)
            E = R // net.nagents
            self.alive[t] = net._mask(info, 'alive_mask', E, dev)
            self.gate[t] = net._mask(info, 'comm_action', E, dev) if net.args.hard_attn else None
        self.n = t + 1

    @staticmethod
    def _put(slot, x):
        """slot (R, wide) <- x (R, narrower), zero beyond"""
        slot.zero_()
        slot[:, :x.shape[-1]].copy_(x.detach().reshape(slot.shape[0], x.shape[-1]))

    def finish(self, prev_hid):
        if not self.recurrent:
            return
        h = prev_hid[0] if isinstance(prev_hid, (tuple, list)) else prev_hid
        if h.shape[-1] != self.hs.shape[2]:
            self.h_last = torch.empty_like(self.hs[0])
            self._put(self.h_last, h)
        elif self.n < self.hs.shape[0] and h.data_ptr() == self.hs[self.n].data_ptr():
            self.h_last = self.hs[self.n]
        else:
            self.h_last = h.detach().reshape(self.hs.shape[1:]).clone()


def _returns(args, rewards, episode_masks, episode_mini_masks):
    """trainer.py:162-171: the reversed scan of the cooperative and the per-agent returns and their mix — one launch
    (ic3_returns_scan) on the GPU for up to 256 agents, the reference's loop otherwise."""
    T, E, n = rewards.shape
    if rewards.is_cuda and n <= 256 and rewards.dtype == torch.float32:
        return ops.returns_scan(rewards, episode_masks, episode_mini_masks, args.gamma, args.mean_ratio)
    coop_returns = torch.empty_like(rewards)
    ncoop_returns = torch.empty_like(rewards)
    prev_coop = torch.zeros_like(rewards[0])
    prev_ncoop = torch.zeros_like(rewards[0])
    for i in reversed(range(T)):
        coop_returns[i] = rewards[i] + args.gamma * prev_coop * episode_masks[i]
        ncoop_returns[i] = rewards[i] + args.gamma * prev_ncoop * episode_masks[i] * episode_mini_masks[i]
        prev_coop = coop_returns[i]
        prev_ncoop = ncoop_returns[i]
    return args.mean_ratio * coop_returns.mean(dim=2, keepdim=True) + (1 - args.mean_ratio) * ncoop_returns


def _recorded_out(records, T):
    """The (T, R, OT) rows the step launches of the batch's episodes wrote, when every step wrote into its record's buffer
    (Trainer: EpisodeRecord.out) — no stack of T x heads views then."""
    if not records or any(getattr(r, 'out', None) is None or r.out_n != r.n for r in records) or sum(r.n for r in records) != T:
        return None
    return records[0].out[:records[0].n] if len(records) == 1 else torch.cat([r.out[:r.n] for r in records])


def loss_gradients(args, batch, records=None):
    """trainer.py:128-218 up to the losses: returns (stat, d_out) with d_out (T, R, OT) = dL/d[logits of every head |
    value] of every transition (the log-softmax is folded in: gradients w.r.t. its INPUT).  On the device, with the step
    launches' rows at hand (`records`) and at most four heads: ONE launch (ic3_loss_gradients) behind the return scan; the tensor
    program below is the same arithmetic (tests/test_trainer_gpu.py compares them)."""
    n = args.nagents
    rewards = torch.stack(batch.reward)                                   # (T, E, N)
    T, E = rewards.shape[0], rewards.shape[1]
    episode_masks = torch.stack(batch.episode_mask)
    episode_mini_masks = torch.stack(batch.episode_mini_mask)
    out_rows = _recorded_out(records, T) if (rewards.is_cuda and bool(getattr(args, 'fused_loss', True))) else None
    if out_rows is not None and len(batch.action_out[0]) <= 4 and rewards.dtype == torch.float32:
        actions = torch.stack(batch.action).reshape(T, -1, E * n)             # (T, heads, R) int32
        alive_masks = torch.stack([m['alive_mask'] for m in batch.misc]).reshape(T, E * n)
        live = torch.stack([m['live'] for m in batch.misc])                   # (T, E)
        returns = _returns(args, rewards, episode_masks, episode_mini_masks).reshape(T, E * n)
        shift, scale = 0.0, 1.0
        if args.normalize_rewards:                                        # trainer.py:176-177 (live entries only)
            lv = live.unsqueeze(2).expand(T, E, n).reshape(T, E * n)
            adv = returns - out_rows[:, :, -1]
            cnt = lv.sum()
            mean = (adv * lv).sum() / cnt
            var = (((adv - mean) ** 2) * lv).sum() / (cnt - 1)
            shift, scale = float(mean), float(1.0 / var.sqrt())
        if actions.dtype != torch.int32:
            actions = actions.int()
        d_out, sums = ops.loss_gradients(out_rows, actions.contiguous(), returns.contiguous(), alive_masks.contiguous(),
                                         live.contiguous(), [int(a) for a in args.naction_heads], float(args.entr),
                                         float(args.value_coeff), shift, scale)
        sums = sums.tolist()
        return dict(action_loss=sums[0], value_loss=sums[1], entropy=sums[2]), d_out
    actions = torch.stack(batch.action).permute(0, 2, 3, 1).long()        # (T, E, N, heads)
    values = torch.stack([v.reshape(E, n) for v in batch.value])          # (T, E, N)
    nheads = len(batch.action_out[0])
    log_p_a = [torch.stack([ao[k] for ao in batch.action_out]) for k in range(nheads)]    # (T, E, N, A_k)
    alive_masks = torch.stack([m['alive_mask'] for m in batch.misc])      # (T, E, N), already x live
    live = torch.stack([m['live'] for m in batch.misc]).unsqueeze(2).expand(T, E, n)

    returns = _returns(args, rewards, episode_masks, episode_mini_masks)  # trainer.py:162-171
    advantages = returns - values                                         # trainer.py:173-174 (values carry no graph here)
    if args.normalize_rewards:                                            # trainer.py:176-177 (live entries only)
        cnt = live.sum()
        mean = (advantages * live).sum() / cnt
        var = (((advantages - mean) ** 2) * live).sum() / (cnt - 1)
        advantages = (advantages - mean) / var.sqrt()
    stat = dict()
    per_head = [lp.gather(3, actions[..., k:k + 1]).squeeze(3) for k, lp in enumerate(log_p_a)]
    if args.advantages_per_action:                                        # trainer.py:192-197 (the same sum either way)
        action_loss = sum((-advantages * lp * alive_masks).sum() for lp in per_head)
    else:
        action_loss = (-advantages * sum(per_head) * alive_masks).sum()
    value_loss = ((values - returns).pow(2) * alive_masks).sum()          # trainer.py:203-206
    entropy = 0
    for lp in log_p_a:                                                    # trainer.py:211-218 (no alive mask there)
        entropy = entropy - (lp * lp.exp() * live.unsqueeze(3)).sum()
    stat['action_loss'] = action_loss.item()
    stat['value_loss'] = value_loss.item()
    stat['entropy'] = entropy.item()

    cols = []
    w_act = (-advantages * alive_masks).unsqueeze(3)                      # d action_loss / d logp[a]
    for k, lp in enumerate(log_p_a):
        dlp = torch.zeros_like(lp)
        dlp.scatter_(3, actions[..., k:k + 1], w_act)
        if args.entr > 0:                                                 # loss -= entr * entropy
            dlp = dlp + args.entr * live.unsqueeze(3) * lp.exp() * (lp + 1.0)
        cols.append(dlp - lp.exp() * dlp.sum(3, keepdim=True))            # through log_softmax: gradient w.r.t. the logits
    cols.append((2.0 * args.value_coeff * (values - returns) * alive_masks).unsqueeze(3))
    d_out = torch.cat(cols, 3).reshape(T, E * n, -1).contiguous()
    return stat, d_out


def backward_episode(args, net, raw, rec, d_out, acc, carry=None):
    """Backward through one recorded episode; parameter gradients are ADDED into `acc` (fp32 tensors keyed like the
    fused weight cache).

    Collection mode (`rec.stream`, Trainer._run_batch_streams: the record is a WINDOW of consecutive slots of E streams of
    episodes; every policy family — the non-recurrent ones have no state to cut, only the masks of a starting env): `fresh[t]` marks the envs that start an episode at slot t — their
    rows of (h, c) entering the slot are zero, nobody is dead and the gate is 0 (trainer.py:38-51, quirks Q21 / Q22), as
    the step launch had it — and `keep[t]` the envs whose state leaving slot t reaches slot t + 1 with its gradient (no
    episode end, not a detach point of the env's own step count: trainer.py:56-60); `carry` = (dL/dh, dL/dc) arriving at
    the window's last slot from the next window, and the pair leaving the window's first slot is returned."""
    if _is_baseline(net):
        si = standin_for_backward(args, net, rec)
        if si is not None:
            return _backward_episode_standin(net, si, raw, rec, d_out, acc, carry)
        return _backward_episode_baseline(args, net, raw, rec, d_out, acc, carry)
    if not rec.recurrent:
        return _backward_episode_commnet(args, net, raw, rec, d_out, acc)
    if net.comm_passes > 1:
        return _backward_episode_multipass(args, net, raw, rec, d_out, acc, carry)
    fc = net._fused_cache()
    T, R, H = rec.n, rec.hs.shape[1], rec.hs.shape[2]
    N = net.nagents
    E = R // N
    dev = rec.hs.device
    mode_avg = getattr(args, 'comm_mode', 'avg') == 'avg'
    mask_zero = bool(args.comm_mask_zero)
    w_cat_t = fc['w_cat_t']                                               # (2H, 4H) = [W_ih | W_hh]^T
    z = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
    xh, comm, dgates = z(R, 2 * H), z(E, N, H), z(R, 4 * H)   # xh = [inp | h_{t-1}]
    dxh, dcomm = z(R, 2 * H), z(R, H)                                      # dxh = [d inp | d h_{t-1}]
    inp, dinp = xh[:, :H], dxh[:, :H]
    # gate recompute + cell backward in one launch when the packed gate weights of the rollout kernel exist (hid 64/128/256);
    # its bias partials accumulate over the episode's steps and are reduced once, behind the loop
    fused_gates = fc.get('ps_l_wp') is not None and ops.lstm_gates_backward_supported(H) and R * 4 * H * 4 < 2 ** 32
    # the input gradient of the gate product inside the same launch (ic3_lstm_gates_backward_dx: split mode, hid 64 / 128)
    fused_dx = fused_gates and fc.get('ps_l_wp3') is not None and fc.get('ps_l_wp3_bwd') is not None and \
        bool(getattr(args, 'fused_input_grad', True))
    # the gates of every step as the rollout recorded them: no gate product in the backward at all
    given = fused_dx and rec.gates is not None and rec.xh is not None and rec.gates_n == T and rec.gates.shape[2] == 4 * H
    # ... and then the whole window as ONE host call (round 6: ic3_bptt_backward — three launches per step, no library product,
    # nothing on the host between them; the weight gradient of the window in one launch behind it)
    if given and bool(getattr(args, 'bptt_native_loop', True)) and d_out.shape[-1] <= 16 and ops.bptt_backward_supported(raw, H):
        return _backward_window_native(args, net, raw, rec, d_out, acc, carry, fc)
    if fused_gates:
        bias_parts = torch.zeros(((R + 63) // 64, 4 * H), dtype=torch.float32, device=dev)
    else:
        gates, bias_parts, bsum = z(R, 4 * H), z(ops.LSTM_BWD_MAX_PARTIALS, 4 * H), z(4 * H)
    # the weight gradient dgates^T . [inp | h] has K = R: as NB products over row blocks (batched, then summed) the
    # library fills the chip (tools/exp/microbench_bptt_gemms.py: 118 instead of 83 TFLOP/s at R = 81920)
    NB = 8 if R % 8 == 0 and R >= 8192 else 1
    wpart = torch.zeros((NB, 2 * H, 4 * H), dtype=torch.float32, device=dev) if NB > 1 else None
    # C.weight's gradient d inp^T . comm has an H x H result over K = R: a single product is 24 workgroups of split-K; as
    # NBC products over row blocks, summed behind the loop, it fills the chip too
    NBC = 32 if R % 32 == 0 and R >= 8192 and not mask_zero else 1
    cpart = torch.zeros((NBC, H, H), dtype=torch.float32, device=dev) if NBC > 1 else None
    dh_rec = torch.zeros((R, H), dtype=torch.float32, device=dev)         # dL/dh_t, dL/dc_t arriving from step t + 1
    dc_rec = torch.zeros((R, H), dtype=torch.float32, device=dev)
    gap = int(getattr(args, 'detach_gap', 10000))
    enc_acc = None        # None: no state accumulated yet; True: partial sums hold the steps so far; False: per-step form
    stream = rec.stream
    if stream is not None:
        if carry is not None:
            dh_rec.copy_(carry[0])
            dc_rec.copy_(carry[1])
        _heads_grad_episode(rec, d_out, acc, T, R, H)                     # (reads h_t of every slot: before rows are zeroed)
        fresh_rows = stream['fresh'].to(torch.float32).repeat_interleave(N, dim=1)     # (T, R)
        keep_rows = stream['keep'].to(torch.float32).repeat_interleave(N, dim=1).unsqueeze(2)   # (T, R, 1)
        ones_mask = torch.ones((E, N), dtype=torch.int32, device=dev)
        zeros_mask = torch.zeros((E, N), dtype=torch.int32, device=dev)
    # collection mode on recorded gates: the per-row cuts ride inside the launches that touch the rows anyway
    # (ic3_lstm_gates_backward_given: row_live, row_keep; ic3_comm_masked_mean_add: out_row_scale) — no passes of their own
    cut_in_kernel = stream is not None and given
    if cut_in_kernel:
        keep_flat = keep_rows.squeeze(2).contiguous()                     # (T, R)
        live_flat = (1.0 - fresh_rows).contiguous()
        dh_rec.mul_(keep_rows[T - 1])                                     # what the next window handed over, cut at its border
    for t in reversed(range(T)):
        if cut_in_kernel:
            pass
        elif stream is not None:
            dh_rec.mul_(keep_rows[t])                                     # the cuts of the env's OWN episode / detach points
            dc_rec.mul_(keep_rows[t])
        elif (t + 1) % gap == 0:                                          # trainer.py:56-60: (h_t, c_t) handed on detached
            dh_rec.zero_()
            dc_rec.zero_()
        h_prev, c_prev = rec.hs[t], rec.cs[t]
        h_t = rec.hs[t + 1] if t + 1 < T else rec.h_last
        alive, gate = rec.alive[t], rec.gate[t]
        if stream is not None:
            fr = stream['fresh'][t].unsqueeze(1)                          # (E, 1) the env starts an episode at this slot
            alive = torch.where(fr, ones_mask, alive) if alive is not None else None
            if cut_in_kernel:
                # (the state that entered stays in the record; the launch multiplies it by row_live, and the communication
                #  block must not see it either: the env's agents are gated off — what zero hidden states amount to)
                gate = torch.where(fr, zeros_mask, gate if gate is not None else ones_mask)
            else:
                h_prev.mul_((1.0 - fresh_rows[t]).unsqueeze(1))           # in place: the record is not read again
                c_prev.mul_((1.0 - fresh_rows[t]).unsqueeze(1))
                gate = torch.where(fr, zeros_mask, gate) if gate is not None else None
        # ---- the forward of step t again: enc + C.bias -> inp, comm, gate pre-activations (comm.py:119,181-215)
        if given:                                                         # inp as the step launch stored it; comm for C's gradient
            xh = rec.xh[t]
            if not mask_zero:
                ops.comm_masked_mean_raw(h_prev.view(E, N, H), alive, gate, mode_avg, True, out=comm)
        else:
            raw.encode_at(rec.snaps[t], fc['wt'], fc['enc_bias'], out=inp, loc_table=fc['loc_table'])
            if not fused_gates:
                xh[:, H:].copy_(h_prev)                                   # (the fused gate launch fills it)
            if mask_zero:
                comm.zero_()                                              # comm.py:40-41: C sees zeros
            else:
                ops.comm_masked_mean_raw(h_prev.view(E, N, H), alive, gate, mode_avg, True, out=comm)
                inp.addmm_(comm.view(R, H), fc['c_wt'])
        if not fused_gates:
            torch.addmm(fc['b_cat'], xh, w_cat_t, out=gates)              # one K = 2H product, as in the rollout
        # ---- heads (comm.py:228,239) -> LSTM cell
        d = d_out[t]
        # dL/dh_t = what step t + 1 sent back + the heads' share — in place (addmm with another `out` first copies R x H floats);
        # dh_rec is rewritten with dL/dh_{t-1} at the end of the iteration, after the cell backward has consumed it
        dh = dh_rec.addmm_(d, fc['w_heads'])
        if given:                                                         # dc_rec <- dL/dc_{t-1}
            ops.lstm_gates_backward_given(rec.gates[t], c_prev, dh, dc_rec, dgates, dc_rec, bias_parts, True, xh=xh, h_prev=h_prev,
                                          lstm_wp3_bwd=fc['ps_l_wp3_bwd'], dxh=dxh,
                                          row_live=live_flat[t] if cut_in_kernel else None,
                                          row_keep=keep_flat[t] if cut_in_kernel else None)
        elif fused_gates:
            # (the heads' own weight gradient is one pass over the whole episode behind the loop: ic3_heads_grad)
            ops.lstm_gates_backward(xh, fc['ps_l_wp'], fc['b_cat'], c_prev, dh, dc_rec, dgates, dc_rec, bias_parts, True,
                                    h_prev=h_prev, lstm_wp3=fc.get('ps_l_wp3'), lstm_wp3_bwd=fc.get('ps_l_wp3_bwd') if fused_dx else None,
                                    dxh=dxh if fused_dx else None)
        else:
            if stream is None:
                acc['w_heads'].addmm_(d.t(), h_t)
                acc['b_heads'].add_(d.sum(0))
            parts = ops.lstm_cell_backward(gates, c_prev, dh, dc_rec, dgates, dc_rec, bias_parts)
            torch.sum(parts, 0, out=bsum)
            acc['b_cat'].add_(bsum)
        # ---- [W_ih | W_hh] (torch.nn.LSTMCell): weight gradient and input gradient, one product each
        if NB > 1:                            # wpart_b (2H, 4H) += xh_b^T . dgates_b; the NB partials are summed behind the loop
            wpart.baddbmm_(xh.view(NB, R // NB, 2 * H).transpose(1, 2), dgates.view(NB, R // NB, 4 * H))
        else:
            acc['w_cat_t'].addmm_(xh.t(), dgates)                         # (2H, R) x (R, 4H)
        if not fused_dx:
            torch.mm(dgates, w_cat_t.t(), out=dxh)                        # (R, 4H) x (4H, 2H) -> [d inp | d h_{t-1}]
        # ---- inp = encoder(obs) + C(comm) (+ both biases)
        if not mask_zero:
            if NBC > 1:
                cpart.baddbmm_(dinp.view(NBC, R // NBC, H).transpose(1, 2), comm.view(NBC, R // NBC, H))
            else:
                acc['c_w'].addmm_(dinp.t(), comm.view(R, H))
            torch.mm(dinp, fc['c_wt'].t(), out=dcomm)                     # d comm = d inp . C.weight
            # dL/dh_{t-1} (what step t - 1 receives) = d h of the gate product + the communication block's share, one pass
            ops.comm_masked_mean_raw(dcomm.view(E, N, H), alive, gate, mode_avg, True, out=dh_rec.view(E, N, H), addend=dxh[:, H:],
                                     row_scale=keep_flat[t - 1] if cut_in_kernel and t > 0 else None)
        elif cut_in_kernel and t > 0:
            torch.mul(dxh[:, H:], keep_rows[t - 1], out=dh_rec)
        else:
            dh_rec.copy_(dxh[:, H:])
        # encoder: the first stage per step adds to partial sums, the expansion into (obs_dim, H) runs once behind the loop
        if enc_acc is not False:
            enc_acc = raw.encode_backward_accumulate(dinp, rec.snaps[t], first=(enc_acc is None)) \
                if hasattr(raw, 'encode_backward_accumulate') else False
        if enc_acc is False:
            dwt, db = raw.encode_backward(dinp, rec.snaps[t], want_bias=True)    # db = sum of the d inp rows: both biases
            acc['wt'].add_(dwt)
            acc['enc_bias'].add_(db)
    if enc_acc:
        dwt, db = raw.encode_backward_finish(H, want_bias=True)
        acc['wt'].add_(dwt)
        acc['enc_bias'].add_(db)
    if NB > 1:
        acc['w_cat_t'].add_(wpart.sum(0))
    if NBC > 1:
        acc['c_w'].add_(cpart.sum(0))
    if fused_gates:
        acc['b_cat'].add_(bias_parts.sum(0))
        if stream is None:
            _heads_grad_episode(rec, d_out, acc, T, R, H)
    return (dh_rec, dc_rec)       # dL/d(h, c) entering the record's first slot (collection mode: the previous window's carry)


def _backward_window_native(args, net, raw, rec, d_out, acc, carry, fc):
    """backward_episode on recorded gates through ic3_bptt_backward (csrc/bptt_kernels.hip).  Per step, last to first:
      ic3_lstm_gates_backward_given   cell derivative from the recorded gates, IN PLACE (dgates over the gates), dL/dh_t taking the
                                      heads' share d_t . W_heads on the way in, [d inp | d h_direct] = dgates . [W_ih | W_hh]
      ic3_comm_backward               dL/dh_{t-1} = d h_direct + (M d inp) . C,  dC += (M d inp)^T h_{t-1}   (M: the mixing matrix of
                                      the communication block, symmetric — one mix feeds both products; comm itself is never formed)
      ic3_env_encode_backward_accumulate   the sparse encoder's stage 1 on the step's snapshot — or, with the steps' input gradients
                                      kept in a ring, ONE ic3_env_encode_backward_window launch over all of them behind the loop
    and behind the loop ONE weight-gradient launch over the window's T x R rows (the record's inp rows, the recorded h, the
    dgates now standing in the gate record), the heads' pass, the encoder's expansion, the partials' sums.  Same cuts as
    backward_episode (detach points; collection mode: row_live / row_keep / gated-off fresh envs)."""
    T, R, H = rec.n, rec.hs.shape[1], rec.hs.shape[2]
    N = net.nagents
    E = R // N
    dev = rec.hs.device
    mode_avg = getattr(args, 'comm_mode', 'avg') == 'avg'
    mask_zero = bool(args.comm_mask_zero)
    zeros = lambda *sh: torch.zeros(sh, dtype=torch.float32, device=dev)
    dh_rec, dc_rec = zeros(R, H), zeros(R, H)
    # the per-step input gradients: a ring of T of them when the encoder's backward has its window form (one launch over all the
    # window's states behind the loop instead of one per step) and the memory is there, else one buffer
    ring = bool(getattr(args, 'enc_window', True)) and raw.encode_window_work(H) is not None and \
        _ring_fits(dev, T * R * 2 * H * 4)
    dxh = torch.empty((T, R, 2 * H) if ring else (R, 2 * H), dtype=torch.float32, device=dev)
    bias_parts = zeros((R + 63) // 64, 4 * H)
    # two chains of launches (envs [0, E1) and [E1, E) on two streams): one fills the ragged last round of the other's launches
    two = ring and bool(getattr(args, 'bptt_two_chains', True)) and ops.first_chain_envs(E, N) < E
    dcw_parts = None if mask_zero else zeros(ops.bptt_dcw_partials(E, N, two), H, H)
    alive, gate = list(rec.alive[:T]), list(rec.gate[:T])
    live_flat = keep_flat = None
    stream = rec.stream
    gap = int(getattr(args, 'detach_gap', 10000))
    if stream is not None:
        if carry is not None:
            dh_rec.copy_(carry[0])
            dc_rec.copy_(carry[1])
        fresh, keep = stream['fresh'], stream['keep']                     # (T, E) bool
        live_flat = (~fresh).to(torch.float32).repeat_interleave(N, dim=1).contiguous()
        keep_flat = keep.to(torch.float32).repeat_interleave(N, dim=1).contiguous()
        ones = torch.ones((E, N), dtype=torch.int32, device=dev)
        fr = fresh.unsqueeze(2)
        if any(m is not None for m in alive):                             # an env that starts an episode: nobody is dead (Q21)
            al = torch.where(fr, ones, torch.stack([m if m is not None else ones for m in alive]))
            alive = list(al.unbind(0))
        # ... and its agents are gated off: what the zero state entering the slot amounts to for the communication block (Q22)
        gt = torch.where(fr, torch.zeros_like(ones), torch.stack([m if m is not None else ones for m in gate]))
        gate = list(gt.unbind(0))
        dh_rec.mul_(keep_flat[T - 1].unsqueeze(1))                        # what the next window handed over, cut at its border
        gap = 0                                                           # (the env's OWN detach points are in `keep`)
    elif gap > T:
        gap = 0
    dhead = d_out if d_out.is_contiguous() else d_out.contiguous()
    # the heads' weight gradient reads only d_out and the recorded h: it runs on a second stream BESIDE the chain (no LDS, a few
    # waves per CU: it fits next to the chain's workgroups and rides their idle issue slots instead of taking 0.7 ms of its own)
    side = None
    if dev.type == 'cuda' and bool(getattr(args, 'heads_grad_beside', True)) and not torch.cuda.is_current_stream_capturing():
        main = torch.cuda.current_stream(dev)
        side = _SIDE_STREAMS.get(dev.index)
        if side is None:
            side = _SIDE_STREAMS[dev.index] = torch.cuda.Stream(device=dev)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            _heads_grad_episode(rec, d_out, acc, T, R, H)
    ops.bptt_backward(raw, T, E, N, H, rec.gates, rec.hs, rec.cs, dhead, rec.snaps, alive, gate, fc['ps_l_wp3_bwd'], fc['w_heads'],
                      None if mask_zero else net.C_modules[0].weight.detach(), dh_rec, dc_rec, dxh, bias_parts, dcw_parts,
                      mode_avg=mode_avg, comm_zero=mask_zero, detach_gap=gap, row_live=live_flat, row_keep=keep_flat, enc_first=True,
                      gate_events=getattr(raw, 'gate_timer', None),     # (bench.py --mode train: HIP events around the gate launches)
                      two_chains=two)
    work = acc.setdefault('_work', {})
    ops.lstm_weight_grad(rec.xh[:T], rec.hs[:T], rec.gates[:T], acc['w_cat_t'], row_live=live_flat, accumulate=True, work=work,
                         split=bool(getattr(args, 'gate_split', True)))
    dwt, db = raw.encode_backward_window_finish(H, want_bias=True) if ring else raw.encode_backward_finish(H, want_bias=True)
    acc['wt'].add_(dwt)
    acc['enc_bias'].add_(db)
    acc['b_cat'].add_(bias_parts.sum(0))
    if not mask_zero:
        acc['c_w'].add_(dcw_parts.sum(0))
    if side is not None:
        torch.cuda.current_stream(dev).wait_stream(side)
    else:
        _heads_grad_episode(rec, d_out, acc, T, R, H)
    return (dh_rec, dc_rec)


_SIDE_STREAMS = {}     # per device: the stream the heads' gradient runs on beside the backward's chain


def _ring_fits(dev, nbytes):
    """Room for the ring of per-step input gradients: a quarter of what the device has free (+ what the caching allocator holds)."""
    if dev.type != 'cuda':
        return True
    free, _ = torch.cuda.mem_get_info(dev)
    cached = torch.cuda.memory_reserved(dev) - torch.cuda.memory_allocated(dev)
    return nbytes <= (free + cached) // 4


class _Cuts(object):
    """Collection mode (rec.stream) for the per-step Python loops: the masks of an env that STARTS an episode at slot t (nobody
    is dead, the gate is 0: trainer.py:38-51, quirks Q21 / Q22), the rows of its entering state (zero), and what of the
    gradient of the recurrent state crosses from slot t + 1 back to slot t (`keep`: no episode end, no detach point of the env's
    own step counter)."""

    def __init__(self, rec, E, N, dev):
        st = rec.stream
        self.on = st is not None
        if not self.on:
            return
        self.fresh = st['fresh']
        self.live_rows = (~st['fresh']).to(torch.float32).repeat_interleave(N, dim=1).unsqueeze(2)   # (T, R, 1)
        self.keep_rows = st['keep'].to(torch.float32).repeat_interleave(N, dim=1).unsqueeze(2)
        self.ones = torch.ones((E, N), dtype=torch.int32, device=dev)
        self.zeros = torch.zeros((E, N), dtype=torch.int32, device=dev)

    def masks(self, t, alive, gate):
        if not self.on:
            return alive, gate
        fr = self.fresh[t].unsqueeze(1)
        return (torch.where(fr, self.ones, alive) if alive is not None else None,
                torch.where(fr, self.zeros, gate) if gate is not None else None)

    def entering(self, t, h, c=None):
        """(h, c) entering slot t with the rows of starting envs zeroed (in place: the record is not read again)"""
        if self.on:
            h.mul_(self.live_rows[t])
            if c is not None:
                c.mul_(self.live_rows[t])
        return h, c

    def cut(self, t, gap, *grads):
        """the gradient of the state LEAVING slot t: lock-step — zero at the episode's detach points; collection — times keep"""
        if self.on:
            for g in grads:
                g.mul_(self.keep_rows[t])
        elif (t + 1) % gap == 0:
            for g in grads:
                g.zero_()


def _heads_grad_episode(rec, d_out, acc, T, R, H):
    """heads + value head over a whole record: dW += sum_t d_t^T h_t, db += sum_t sum_rows d_t — h_t of step t is the state
    ENTERING step t + 1 (slot t + 1 of the record).  ONE pass (ic3_heads_grad) for up to 16 output columns, a library product
    otherwise."""
    def grad(d, h):
        if d.shape[-1] <= ops.HEADS_GRAD_MAX_OT:
            ops.heads_grad(d, h, acc['w_heads'], acc['b_heads'], acc.setdefault('_work', {}))
        else:                                                             # (more than 15 actions in total)
            acc['w_heads'].addmm_(d.t(), h)
            acc['b_heads'].add_(d.sum(0))
    if rec.h_last is not None and rec.h_last.data_ptr() == rec.hs[T].data_ptr():
        grad(d_out[:T].reshape(T * R, -1), rec.hs[1:T + 1].reshape(T * R, H))
    else:
        if T > 1:
            grad(d_out[:T - 1].reshape((T - 1) * R, -1), rec.hs[1:T].reshape((T - 1) * R, H))
        grad(d_out[T - 1], rec.h_last)


def _backward_episode_multipass(args, net, raw, rec, d_out, acc, carry=None):
    """comm_passes > 1 (comm.py:179-218): the step's passes are re-evaluated forward from the (h, c) that entered the step
    — pass i: comm_i = mix(h_i), inp_i = enc + C_i(comm_i), (h_{i+1}, c_{i+1}) = LSTMCell(inp_i, (h_i, c_i)), every pass's
    [inp | h], comm and c kept for the duration of the step — and then differentiated last pass first; the encoder sees
    the sum of the passes' d inp.  Plain launch chain (library GEMMs + the pointwise kernels): this is f3 coverage."""
    fc = net._fused_cache()
    P = net.comm_passes
    T, R, H = rec.n, rec.hs.shape[1], rec.hs.shape[2]
    N = net.nagents
    E = R // N
    dev = rec.hs.device
    mode_avg = getattr(args, 'comm_mode', 'avg') == 'avg'
    mask_zero = bool(args.comm_mask_zero)
    w_cat_t = fc['w_cat_t']
    z = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
    c_wt = [m.weight.detach().t().contiguous() for m in net.C_modules]    # per pass C_i^T (H, H)
    dbias = [(m.bias - net.C_modules[0].bias).detach() for m in net.C_modules]
    xh = [z(R, 2 * H) for _ in range(P)]
    comm = [z(E, N, H) for _ in range(P)]
    cs = [z(R, H) for _ in range(P)]
    enc, gates, dgates, dxh = z(R, H), z(R, 4 * H), z(R, 4 * H), z(R, 2 * H)
    dcomm, dcomm_b, dh, denc = z(R, H), z(E, N, H), z(R, H), z(R, H)
    bias_parts, bsum = z(ops.LSTM_BWD_MAX_PARTIALS, 4 * H), z(4 * H)
    dinp = dxh[:, :H]
    dh_rec = torch.zeros((R, H), dtype=torch.float32, device=dev)
    dc_rec = torch.zeros((R, H), dtype=torch.float32, device=dev)
    gap = int(getattr(args, 'detach_gap', 10000))
    cuts = _Cuts(rec, E, N, dev)
    if cuts.on and carry is not None:
        dh_rec.copy_(carry[0])
        dc_rec.copy_(carry[1])
    if cuts.on:                                                   # (reads h_t of every slot: before the starting envs' rows are zeroed)
        heads_first = [(rec.hs[t + 1] if t + 1 < T else rec.h_last).clone() for t in range(T)]
    for t in reversed(range(T)):
        cuts.cut(t, gap, dh_rec, dc_rec)
        h_t = heads_first[t] if cuts.on else (rec.hs[t + 1] if t + 1 < T else rec.h_last)
        alive, gate = cuts.masks(t, rec.alive[t], rec.gate[t])
        # ---- forward: enc (+ encoder.bias + C_0.bias), then the passes
        raw.encode_at(rec.snaps[t], fc['wt'], fc['enc_bias'], out=enc, loc_table=fc['loc_table'])
        h_in, c_in = cuts.entering(t, rec.hs[t], rec.cs[t])
        xh[0][:, H:].copy_(h_in)
        cs[0].copy_(c_in)
        for i in range(P):
            inp_i = xh[i][:, :H]
            torch.add(enc, dbias[i], out=inp_i) if i else inp_i.copy_(enc)
            if mask_zero:
                comm[i].zero_()
            else:
                ops.comm_masked_mean_raw(xh[i].view(E, N, 2 * H)[:, :, H:], alive, gate, mode_avg, True, out=comm[i])
                inp_i.addmm_(comm[i].view(R, H), c_wt[i])
            if i + 1 < P:
                torch.addmm(fc['b_cat'], xh[i], w_cat_t, out=gates)
                cs[i + 1].copy_(cs[i])
                ops.lstm_cell_(gates, cs[i + 1], xh[i + 1][:, H:])
        # ---- backward: heads on the last pass's h, then the passes in reverse
        d = d_out[t]
        torch.addmm(dh_rec, d, fc['w_heads'], out=dh)
        acc['w_heads'].addmm_(d.t(), h_t)
        acc['b_heads'].add_(d.sum(0))
        denc.zero_()
        for i in reversed(range(P)):
            torch.addmm(fc['b_cat'], xh[i], w_cat_t, out=gates)
            parts = ops.lstm_cell_backward(gates, cs[i], dh, dc_rec, dgates, dc_rec, bias_parts)   # dc_rec <- dL/dc_i
            torch.sum(parts, 0, out=bsum)
            acc['b_cat'].add_(bsum)
            acc['w_cat_t'].addmm_(xh[i].t(), dgates)
            torch.mm(dgates, w_cat_t.t(), out=dxh)
            denc.add_(dinp)
            acc['c_b'][i].add_(dinp.sum(0))
            if not mask_zero:
                acc['c_w_p'][i].addmm_(dinp.t(), comm[i].view(R, H))
                torch.mm(dinp, c_wt[i].t(), out=dcomm)
                ops.comm_masked_mean_raw(dcomm.view(E, N, H), alive, gate, mode_avg, True, out=dcomm_b)
                torch.add(dxh[:, H:], dcomm_b.view(R, H), out=dh)         # dL/dh_i: what pass i - 1 (or step t - 1) receives
            else:
                dh.copy_(dxh[:, H:])
        dh_rec.copy_(dh)
        dwt, db = raw.encode_backward(denc, rec.snaps[t], want_bias=True)
        acc['wt'].add_(dwt)
        acc['enc_bias'].add_(db)
    return (dh_rec, dc_rec)


def _backward_episode_commnet(args, net, raw, rec, d_out, acc):
    """The NON-recurrent module (comm.py:127-129,179-205,220-224):  x = tanh(encoder(obs));  h_0 = x;
    h_{i+1} = tanh(x + f_i(h_i) + C_i(comm(h_i)));  [logits | value] = W_heads h_P + b.  No state crosses a step, so every
    recorded step is differentiated on its own: the forward is re-evaluated from the env snapshot (sparse encoder, the masked
    mean, library GEMMs for the H x H layers), then
        dz_i = dh_{i+1} (1 - h_{i+1}^2);   dx += dz_i;   dF_i += dz_i^T h_i;   dC_i += dz_i^T comm_i;
        dh_i = dz_i F_i + mix(dz_i C_i)            (the mixing matrix of the communication block is symmetric)
    and finally d enc = (dx + dh_0)(1 - x^2) through ic3_env_encode_backward on the snapshot."""
    cn = net._commnet_cache()
    P = net.comm_passes
    T, R, H = rec.n, rec.rows, net.hid_size
    N = net.nagents
    E = R // N
    dev = rec.device
    mode_avg = getattr(args, 'comm_mode', 'avg') == 'avg'
    mask_zero = bool(args.comm_mask_zero)
    z = lambda *sh: torch.empty(sh, dtype=torch.float32, device=dev)
    Cw = [m.weight.detach() for m in net.C_modules]               # (H, H): y = v C^T
    Fw = [m.weight.detach() for m in net.f_modules]
    bias = cn['bias']                                             # (P, H) = C_i.bias + f_i.bias
    enc = z(R, H)
    hs = [z(R, H) for _ in range(P + 1)]
    x = hs[0]                                                     # h_0 = x = tanh(encoder(obs))
    comm = [z(E, N, H) for _ in range(P)]
    dz, dh, dx, tmp, mixed = z(R, H), z(R, H), z(R, H), z(R, H), z(E, N, H)
    w_heads = cn['w_heads']
    tanh_bwd = torch.ops.aten.tanh_backward.grad_input            # grad (1 - out^2) in one launch
    # the H x H weight gradients have K = R: as NB products over row blocks, summed behind the loop, they fill the chip
    # (as in backward_episode)
    NB = 32 if R % 32 == 0 and R >= 8192 else 1
    fpart = [torch.zeros((NB, H, H), dtype=torch.float32, device=dev) for _ in range(P)] if NB > 1 else None
    cpart = [torch.zeros((NB, H, H), dtype=torch.float32, device=dev) for _ in range(P)] if NB > 1 and not mask_zero else None
    blk = lambda v: v.view(NB, R // NB, H)
    enc_acc = None        # None: nothing accumulated yet; True: partial sums hold the steps so far; False: per-step form
    cuts = _Cuts(rec, E, N, dev)      # collection mode: no state crosses a step — only the masks of an env that starts an episode
    for t in reversed(range(T)):
        alive, gate = cuts.masks(t, rec.alive[t], rec.gate[t])
        # ---- forward of step t again
        raw.encode_at(rec.snaps[t], cn['wt'], cn['enc_bias'], out=enc, loc_table=cn['loc_table'])
        torch.tanh(enc, out=x)
        for i in range(P):
            if mask_zero:
                comm[i].zero_()
            else:
                ops.comm_masked_mean_raw(hs[i].view(E, N, H), alive, gate, mode_avg, True, out=comm[i])
            torch.add(x, bias[i], out=tmp)                        # x + both biases
            tmp.addmm_(hs[i], Fw[i].t())                          # + f_i(h_i)
            if not mask_zero:
                tmp.addmm_(comm[i].view(R, H), Cw[i].t())         # + C_i(comm_i)
            torch.tanh(tmp, out=hs[i + 1])
        # ---- backward: heads, then the passes last to first
        d = d_out[t]
        acc['w_heads'].addmm_(d.t(), hs[P])
        acc['b_heads'].add_(d.sum(0))
        torch.mm(d, w_heads, out=dh)
        for i in reversed(range(P)):
            tanh_bwd(dh, hs[i + 1], grad_input=dz)                # dh (1 - h^2)
            if i == P - 1:
                dx.copy_(dz)
            else:
                dx.add_(dz)
            if NB > 1:
                fpart[i].baddbmm_(blk(dz).transpose(1, 2), blk(hs[i]))
            else:
                acc['f_w'][i].addmm_(dz.t(), hs[i])
            acc['cf_b'][i].add_(dz.sum(0))
            torch.mm(dz, Fw[i], out=dh)
            if not mask_zero:
                if NB > 1:
                    cpart[i].baddbmm_(blk(dz).transpose(1, 2), blk(comm[i].view(R, H)))
                else:
                    acc['c_w_p'][i].addmm_(dz.t(), comm[i].view(R, H))
                torch.mm(dz, Cw[i], out=tmp)
                # dh += mix(dz C_i): the mixing matrix is symmetric — the same kernel on the gradient, the addend along
                ops.comm_masked_mean_raw(tmp.view(E, N, H), alive, gate, mode_avg, True, out=mixed, addend=dh)
                dh, mixed = mixed.view(R, H), dh.view(E, N, H)
        dx.add_(dh)                                               # h_0 = x
        tanh_bwd(dx, x, grad_input=tmp)                           # through x = tanh(enc)
        # encoder: the first stage per step adds to partial sums, the expansion runs once behind the loop (as in backward_episode)
        if enc_acc is not False:
            enc_acc = raw.encode_backward_accumulate(tmp, rec.snaps[t], first=(enc_acc is None)) \
                if hasattr(raw, 'encode_backward_accumulate') else False
        if enc_acc is False:
            dwt, db = raw.encode_backward(tmp, rec.snaps[t], want_bias=True)
            acc['wt'].add_(dwt)
            acc['enc_bias'].add_(db)
    if enc_acc:
        dwt, db = raw.encode_backward_finish(H, want_bias=True)
        acc['wt'].add_(dwt)
        acc['enc_bias'].add_(db)
    if NB > 1:
        for i in range(P):
            acc['f_w'][i].add_(fpart[i].sum(0))
            if cpart is not None:
                acc['c_w_p'][i].add_(cpart[i].sum(0))


def standin_for_backward(args, net, rec=None):
    """models.RNN with the LSTM cell IS the recurrent CommNet policy with the communication block off (models._KernelStandIn): where
    that stand-in runs at the baseline's own hidden size with the fused gate launches (hid 64 / 128 / 256), its backward is
    backward_episode on the stand-in — gate launch with the input gradient, recorded gates when the rollout stored them —
    instead of the library chain of _backward_episode_baseline.  Returns the stand-in module, or None."""
    if not (getattr(args, 'recurrent', False) and getattr(args, 'rnn_type', 'MLP') == 'LSTM') or not hasattr(net, 'lstm_unit'):
        return None
    if not bool(getattr(args, 'baseline_fused_backward', True)):
        return None
    sk = getattr(net, '_stand_in', None)
    if sk is None or ops.padded_hidden(args.hid_size) is not None or not ops.lstm_gates_backward_supported(args.hid_size):
        return None
    if rec is not None and (not rec.recurrent or rec.hs.shape[2] != args.hid_size):
        return None
    with torch.no_grad():
        si = sk.get()
        if si is None or si._fused_cache().get('ps_l_wp') is None:
            return None
    return si


def _backward_episode_standin(net, si, raw, rec, d_out, acc, carry=None):
    """IRIC (models.RNN, LSTM) through the stand-in's backward; its accumulators are folded into the baseline's by name:
    encoder = affine1, [W_ih | W_hh] / b_ih + b_hh = lstm_unit's, heads; C (all zeros, never read: comm_mask_zero) has none."""
    acc2 = acc.get('_standin')
    if acc2 is None:
        acc2 = acc['_standin'] = new_accumulators(si)
    return backward_episode(si.args, si, raw, rec, d_out, acc2, carry=carry)


def fold_standin(acc):
    """(behind the last episode of the batch) the stand-in's accumulators -> the baseline's"""
    acc2 = acc.pop('_standin', None)
    if acc2 is None:
        return
    H = acc['l_b'].shape[0] // 4
    acc['wt'].add_(acc2['wt'])
    acc['a1_b'].add_(acc2['enc_bias'])
    acc['l_w_ih'].add_(acc2['w_cat_t'][:H].t())
    acc['l_w_hh'].add_(acc2['w_cat_t'][H:].t())
    acc['l_b'].add_(acc2['b_cat'])
    acc['w_heads'].add_(acc2['w_heads'])
    acc['b_heads'].add_(acc2['b_heads'])


def _backward_episode_baseline(args, net, raw, rec, d_out, acc, carry=None):
    """The IC / IRIC baselines of models.py:8-97 (no communication), differentiated by hand over the recorded rollout:
      MLP  (models.py:23-34)   x1 = tanh(affine1(obs));  h = tanh(affine2(x1) + x1)         every step on its own
      RNN  (models.py:68-92)   rnn_type 'MLP':  h_t = tanh(affine2(h_{t-1}) + affine1(obs))
                               rnn_type 'LSTM': (h_t, c_t) = LSTMCell(affine1(obs), (h_{t-1}, c_{t-1}))
    heads / value on h.  affine1 is the sparse encoder (ic3_env_encode_at on the step's snapshot, ic3_env_encode_backward);
    the recurrent gradient is cut where the Trainer detaches the hidden state (trainer.py:56-60)."""
    T, H = rec.n, args.hid_size
    R = rec.rows if not rec.recurrent else rec.hs.shape[1]
    dev = net.affine1.weight.device
    z = lambda *sh: torch.empty(sh, dtype=torch.float32, device=dev)
    wt = net.affine1.weight.detach().t().contiguous()
    b1 = net.affine1.bias.detach()
    w_heads = torch.cat([hd.weight for hd in net.heads] + [net.value_head.weight], 0).detach().contiguous()
    enc, dh, dz = z(R, H), z(R, H), z(R, H)
    recurrent = rec.recurrent
    lstm = recurrent and getattr(args, 'rnn_type', 'MLP') == 'LSTM'
    gap = int(getattr(args, 'detach_gap', 10000))
    if recurrent:
        dh_rec = torch.zeros((R, H), dtype=torch.float32, device=dev)
    if lstm:
        cell = net.lstm_unit
        w_ih, w_hh = cell.weight_ih.detach(), cell.weight_hh.detach()
        b_cat = (cell.bias_ih + cell.bias_hh).detach()
        gates, dgates = z(R, 4 * H), z(R, 4 * H)
        parts, bsum = z(ops.LSTM_BWD_MAX_PARTIALS, 4 * H), z(4 * H)
        dc_rec = torch.zeros((R, H), dtype=torch.float32, device=dev)
    elif recurrent:
        A2 = net.affine2.weight.detach()
    else:
        A2, b2 = net.affine2.weight.detach(), net.affine2.bias.detach()
        x1, hcur = z(R, H), z(R, H)
    tanh_bwd = torch.ops.aten.tanh_backward.grad_input            # grad (1 - out^2) in one launch
    NB = 32 if R % 32 == 0 and R >= 8192 else 1                   # affine2's weight gradient (K = R) as row-block products
    a2part = torch.zeros((NB, H, H), dtype=torch.float32, device=dev) if NB > 1 and not lstm else None
    blk = lambda v: v.view(NB, R // NB, H)
    enc_acc = None
    cuts = _Cuts(rec, R // net.args.nagents, net.args.nagents, dev)
    if cuts.on and recurrent:
        if carry is not None:
            dh_rec.copy_(carry[0])
            if lstm:
                dc_rec.copy_(carry[1])
        heads_first = [(rec.hs[t + 1] if t + 1 < T else rec.h_last).clone() for t in range(T)]   # (before rows are zeroed)
    for t in reversed(range(T)):
        d = d_out[t]
        raw.encode_at(rec.snaps[t], wt, b1, out=enc)              # affine1(obs_t)
        if not recurrent:                                         # ---- MLP
            torch.tanh(enc, out=x1)
            torch.addmm(b2, x1, A2.t(), out=hcur)
            hcur.add_(x1)
            torch.tanh(hcur, out=hcur)
            acc['w_heads'].addmm_(d.t(), hcur)
            acc['b_heads'].add_(d.sum(0))
            torch.mm(d, w_heads, out=dh)
            tanh_bwd(dh, hcur, grad_input=dz)                                     # through the outer tanh
            if a2part is not None:
                a2part.baddbmm_(blk(dz).transpose(1, 2), blk(x1))
            else:
                acc['a2_w'].addmm_(dz.t(), x1)
            acc['a2_b'].add_(dz.sum(0))
            torch.addmm(dz, dz, A2, out=dh)                                       # d x1 = dz A2 + dz (the skip)
            tanh_bwd(dh, x1, grad_input=dz)                                       # through x1 = tanh(enc)
        else:
            cuts.cut(t, gap, *((dh_rec, dc_rec) if lstm else (dh_rec,)))     # (h_t, c_t) were handed on detached / the episode ended
            h_prev, c_prev = cuts.entering(t, rec.hs[t], rec.cs[t] if lstm else None)
            h_t = heads_first[t] if cuts.on else (rec.hs[t + 1] if t + 1 < T else rec.h_last)
            acc['w_heads'].addmm_(d.t(), h_t)
            acc['b_heads'].add_(d.sum(0))
            torch.addmm(dh_rec, d, w_heads, out=dh)
            if lstm:                                              # ---- RNN, LSTM cell
                torch.addmm(b_cat, enc, w_ih.t(), out=gates)
                gates.addmm_(h_prev, w_hh.t())
                p_ = ops.lstm_cell_backward(gates, c_prev, dh, dc_rec, dgates, dc_rec, parts)   # dc_rec <- dL/dc_{t-1}
                torch.sum(p_, 0, out=bsum)
                acc['l_b'].add_(bsum)
                acc['l_w_ih'].addmm_(dgates.t(), enc)
                acc['l_w_hh'].addmm_(dgates.t(), h_prev)
                torch.mm(dgates, w_ih, out=dz)                    # d enc
                torch.mm(dgates, w_hh, out=dh_rec)                # dL/dh_{t-1}
            else:                                                 # ---- RNN, tanh recurrence
                tanh_bwd(dh, h_t, grad_input=dz)
                if a2part is not None:
                    a2part.baddbmm_(blk(dz).transpose(1, 2), blk(h_prev))
                else:
                    acc['a2_w'].addmm_(dz.t(), h_prev)
                acc['a2_b'].add_(dz.sum(0))
                torch.mm(dz, A2, out=dh_rec)
        if enc_acc is not False:                                  # first stage per step, the expansion once behind the loop
            enc_acc = raw.encode_backward_accumulate(dz, rec.snaps[t], first=(enc_acc is None)) \
                if hasattr(raw, 'encode_backward_accumulate') else False
        if enc_acc is False:
            dwt, db = raw.encode_backward(dz, rec.snaps[t], want_bias=True)
            acc['wt'].add_(dwt)
            acc['a1_b'].add_(db)
    if enc_acc:
        dwt, db = raw.encode_backward_finish(H, want_bias=True)
        acc['wt'].add_(dwt)
        acc['a1_b'].add_(db)
    if a2part is not None:
        acc['a2_w'].add_(a2part.sum(0))
    if recurrent:
        return (dh_rec, dc_rec if lstm else dh_rec)


def new_accumulators(net):
    if _is_baseline(net):
        H, dev = net.args.hid_size, net.affine1.weight.device
        z = lambda *sh: torch.zeros(sh, dtype=torch.float32, device=dev)
        OT = sum(hd.weight.shape[0] for hd in net.heads) + 1
        acc = dict(wt=z(net.affine1.weight.shape[1], H), a1_b=z(H), w_heads=z(OT, H), b_heads=z(OT), baseline=True)
        if hasattr(net, 'lstm_unit'):
            acc.update(l_w_ih=z(4 * H, H), l_w_hh=z(4 * H, H), l_b=z(4 * H))
        else:
            acc.update(a2_w=z(H, H), a2_b=z(H))
        return acc
    if not getattr(net.args, 'recurrent', False):                 # the non-recurrent module
        cn = net._commnet_cache()
        H, P, dev = net.hid_size, net.comm_passes, cn['wt'].device
        z = lambda *sh: torch.zeros(sh, dtype=torch.float32, device=dev)
        return dict(wt=z(*cn['wt'].shape), enc_bias=z(H), w_heads=z(*cn['w_heads'].shape), b_heads=z(cn['b_heads'].shape[0]),
                    c_w_p=[z(H, H) for _ in range(P)], f_w=[z(H, H) for _ in range(P)], cf_b=[z(H) for _ in range(P)])
    fc = net._fused_cache()
    H = net.hid_size
    dev = fc['wt'].device
    z = lambda *s: torch.zeros(s, dtype=torch.float32, device=dev)
    acc = dict(wt=z(*fc['wt'].shape), enc_bias=z(H), c_w=z(H, H), w_cat_t=z(2 * H, 4 * H), b_cat=z(4 * H),
               w_heads=z(*fc['w_heads'].shape), b_heads=z(fc['b_heads'].shape[0]))
    if net.comm_passes > 1:                                               # per pass: C_i.weight, C_i.bias
        acc['c_w_p'] = [z(H, H) for _ in range(net.comm_passes)]
        acc['c_b'] = [z(H) for _ in range(net.comm_passes)]
    return acc


def assign_grads(net, acc):
    """accumulators -> .grad of the reference's parameters (state_dict names of comm.py)."""
    def put(p, g):
        p.grad = g.reshape(p.shape).contiguous()
    if acc.get('baseline'):                                       # models.MLP / models.RNN
        fold_standin(acc)
        put(net.affine1.weight, acc['wt'].t())
        put(net.affine1.bias, acc['a1_b'])
        if 'l_b' in acc:
            put(net.lstm_unit.weight_ih, acc['l_w_ih'])
            put(net.lstm_unit.weight_hh, acc['l_w_hh'])
            put(net.lstm_unit.bias_ih, acc['l_b'].clone())
            put(net.lstm_unit.bias_hh, acc['l_b'].clone())
        else:
            put(net.affine2.weight, acc['a2_w'])
            put(net.affine2.bias, acc['a2_b'])
        off = 0
        for hd in net.heads:
            A = hd.weight.shape[0]
            put(hd.weight, acc['w_heads'][off:off + A])
            put(hd.bias, acc['b_heads'][off:off + A])
            off += A
        put(net.value_head.weight, acc['w_heads'][off:off + 1])
        put(net.value_head.bias, acc['b_heads'][off:off + 1])
        return
    put(net.encoder.weight, acc['wt'].t())
    put(net.encoder.bias, acc['enc_bias'].clone())
    if 'f_w' in acc:                                              # the non-recurrent module: C_i, f_i per pass (+ shared modules)
        done = {}
        for mods, wkey in ((net.C_modules, 'c_w_p'), (net.f_modules, 'f_w')):
            for i, m in enumerate(mods):
                if id(m) in done:                                 # share_weights: one module, the passes' sums
                    m.weight.grad.add_(acc[wkey][i])
                    m.bias.grad.add_(acc['cf_b'][i])
                else:
                    put(m.weight, acc[wkey][i].clone())
                    put(m.bias, acc['cf_b'][i].clone())
                    done[id(m)] = True
        off = 0
        for hd in net.heads:
            A = hd.weight.shape[0]
            put(hd.weight, acc['w_heads'][off:off + A])
            put(hd.bias, acc['b_heads'][off:off + A])
            off += A
        put(net.value_head.weight, acc['w_heads'][off:off + 1])
        put(net.value_head.bias, acc['b_heads'][off:off + 1])
        return
    if net.comm_passes > 1:
        done = {}
        for i, m in enumerate(net.C_modules):                             # share_weights: one module, the passes' sums
            if id(m) in done:
                m.weight.grad.add_(acc['c_w_p'][i])
                m.bias.grad.add_(acc['c_b'][i])
            else:
                put(m.weight, acc['c_w_p'][i].clone())
                put(m.bias, acc['c_b'][i].clone())
                done[id(m)] = True
    else:
        put(net.C_modules[0].weight, acc['c_w'])
        put(net.C_modules[0].bias, acc['enc_bias'].clone())               # inp = enc + C(comm): both biases see d inp
    H = net.hid_size
    put(net.f_module.weight_ih, acc['w_cat_t'][:H].t())
    put(net.f_module.weight_hh, acc['w_cat_t'][H:].t())
    put(net.f_module.bias_ih, acc['b_cat'].clone())
    put(net.f_module.bias_hh, acc['b_cat'].clone())
    off = 0
    for hd in net.heads:
        A = hd.weight.shape[0]
        put(hd.weight, acc['w_heads'][off:off + A])
        put(hd.bias, acc['b_heads'][off:off + A])
        off += A
    put(net.value_head.weight, acc['w_heads'][off:off + 1])
    put(net.value_head.bias, acc['b_heads'][off:off + 1])
