"""torch-facing wrappers of the two custom HIP policy ops (include/ic3_rollout.h).  No CPU fallback."""
import torch

from . import _lib
from ._lib import check, ptr, stream


def _need_cuda(t, name):
    if not t.is_cuda:
        raise _lib.IC3Error("%s: expected a CUDA (ROCm) tensor — the HIP op has no CPU fallback" % name)


class _CommMaskedMean(torch.autograd.Function):
    """comm.py:181-205 in closed form.  The map h -> out is linear with a symmetric (N x N) mixing matrix
    M[j,i] = m_j m_i s (i != j), so the backward pass is the same kernel applied to grad_out."""

    @staticmethod
    def forward(ctx, h, alive, comm_action, mode_avg, mask_self):
        ctx.alive, ctx.comm_action, ctx.mode_avg, ctx.mask_self = alive, comm_action, mode_avg, mask_self
        return comm_masked_mean_raw(h, alive, comm_action, mode_avg, mask_self)

    @staticmethod
    def backward(ctx, g):
        return comm_masked_mean_raw(g.contiguous(), ctx.alive, ctx.comm_action, ctx.mode_avg, ctx.mask_self), None, None, None, None


class _EnvEncode(torch.autograd.Function):
    """encoder(obs(env state)) with a backward: forward = the sparse gather ic3_env_encode, backward = the
    position-sum scatter ic3_env_encode_backward on a snapshot of the integer state (no dense obs is kept)."""

    @staticmethod
    def forward(ctx, weight, bias, env, weight_t, loc_table):
        out = env.encode(weight_t, bias.detach(), loc_table=loc_table)
        ctx.env = env
        ctx.snap = env.snapshot()
        ctx.need_bias = bias.requires_grad
        return out

    @staticmethod
    def backward(ctx, g):
        dwt, dbias = ctx.env.encode_backward(g.contiguous(), ctx.snap, want_bias=ctx.need_bias)
        return dwt.t().contiguous(), dbias, None, None, None      # contiguous: grads may be all-reduced (sharding.py)


def env_encode(env, weight, bias, weight_t=None, loc_table=None):
    """nn.Linear(obs_dim, H)(env's current observation) -> (E, N, H), differentiable w.r.t. weight (H, obs_dim) and
    bias; weight_t = weight.t().contiguous() if the caller caches it."""
    if weight_t is None:
        weight_t = weight.detach().t().contiguous()
    return _EnvEncode.apply(weight, bias, env, weight_t, loc_table)


def _rows(t, H):
    """(tensor usable by the kernels, row stride in floats) for a (..., H) fp32 tensor whose rows are unit-stride
    and evenly spaced (e.g. a column slice of a wider row-major buffer); anything else is made contiguous."""
    if t.dtype != torch.float32:
        t = t.float()
    flat = t.reshape(-1, H) if t.is_contiguous() else t
    if flat.dim() == 2 and flat.stride(1) == 1 and flat.stride(0) >= H:
        return flat, flat.stride(0)
    if t.dim() == 3 and t.stride(2) == 1 and t.stride(0) == t.shape[1] * t.stride(1) and t.stride(1) >= H:
        return t, t.stride(1)
    t = t.contiguous()
    return t, H


def comm_masked_mean_raw(h, alive, comm_action, mode_avg, mask_self, out=None, addend=None, row_scale=None):
    """The kernel launch without autograd: h (E,N,H) rows may be a strided column slice; `out` (E,N,H) contiguous.
    `addend` (E*N,H) / (E,N,H) rows (a strided column slice is fine): out = addend + comm(h) (ic3_comm_masked_mean_add);
    `row_scale` (E*N,) float32 with it: every output row times its factor."""
    _need_cuda(h, "comm_masked_mean")
    E, N, H = h.shape
    hk, ldh = _rows(h, H)
    if out is None:
        out = torch.empty((E, N, H), dtype=torch.float32, device=h.device)
    if addend is not None:
        ak, lda = _rows(addend, H)
        if row_scale is not None:
            assert row_scale.is_contiguous() and row_scale.dtype == torch.float32 and row_scale.numel() == E * N
        check(_lib.lib().ic3_comm_masked_mean_add(ptr(hk), ldh, ptr(alive), ptr(comm_action), ptr(ak), lda,
                                                  ptr(row_scale) if row_scale is not None else None, ptr(out), E, N, H,
                                                  int(mode_avg), int(mask_self), stream()))
        return out
    assert row_scale is None, "row_scale comes with an addend"
    check(_lib.lib().ic3_comm_masked_mean(ptr(hk), ldh, ptr(alive), ptr(comm_action), ptr(out), E, N, H,
                                          int(mode_avg), int(mask_self), stream()))
    return out


def lstm_cell_(gates, c, h_out):
    """In-place LSTM pointwise step: gates (R,4H) contiguous, c (R,H) contiguous (updated), h_out (R,H) rows
    with unit column stride (may be a column slice of a wider buffer)."""
    _need_cuda(gates, "lstm_cell")
    R, H = c.shape
    assert gates.is_contiguous() and c.is_contiguous() and h_out.stride(1) == 1
    check(_lib.lib().ic3_lstm_cell(ptr(gates), ptr(c), ptr(h_out), h_out.stride(0), R, H, stream()))
    return h_out, c


LSTM_BWD_MAX_PARTIALS = 2048


def lstm_cell_backward(gates, c_prev, dh, dc, dgates, dc_prev, dbias_partials=None):
    """Backward of lstm_cell_ from the re-computed gate pre-activations (ic3_lstm_cell_backward): gates (R,4H), c_prev,
    dh, dc (or None) (R,H) -> dgates (R,4H), dc_prev (R,H; may alias dc).  dbias_partials: (LSTM_BWD_MAX_PARTIALS, 4H)
    scratch — returns the view of the rows the call wrote (their sum over dim 0 = column sums of dgates), else None.
    All contiguous float32."""
    _need_cuda(gates, "lstm_cell_backward")
    R, H = c_prev.shape
    for t in (gates, c_prev, dh, dgates, dc_prev):
        assert t.is_contiguous() and t.dtype == torch.float32
    if dbias_partials is not None:
        assert dbias_partials.is_contiguous() and tuple(dbias_partials.shape) == (LSTM_BWD_MAX_PARTIALS, 4 * H)
    n = _lib.lib().ic3_lstm_cell_backward(ptr(gates), ptr(c_prev), ptr(dh), ptr(dc) if dc is not None else None,
                                          ptr(dgates), ptr(dc_prev),
                                          ptr(dbias_partials) if dbias_partials is not None else None, R, H, stream())
    if n < 0:
        check(n)
    return dbias_partials[:n] if dbias_partials is not None else None


def commnet_forward_supported(H, N):
    return bool(_lib.lib().ic3_commnet_forward_supported(int(H), int(N)))


def commnet_pack(c_weights, f_weights, c_biases, f_biases):
    """Per-pass [C_i | F_i] in the layout ic3_commnet_forward streams + the summed biases: lists of (H,H) / (H,) tensors."""
    _need_cuda(c_weights[0], "commnet_pack")
    H, P, dev = c_weights[0].shape[0], len(c_weights), c_weights[0].device
    wp = torch.empty((P, 2 * H * H), dtype=torch.float32, device=dev)
    for i in range(P):
        check(_lib.lib().ic3_commnet_pack(ptr(c_weights[i].detach().contiguous().float()),
                                          ptr(f_weights[i].detach().contiguous().float()), ptr(wp[i]), H, stream()))
    bias = torch.stack([(c_biases[i] + f_biases[i]).detach().float() for i in range(P)]).contiguous()
    return wp, bias


def commnet_pack_split(c_weights, f_weights):
    """Per-pass [C_i | F_i] as three exact bf16 planes in MFMA fragment order (ic3_commnet_pack_split): (P, 3 * H * H) float32
    words, or None when hid_size is not a multiple of 32."""
    H, P, dev = c_weights[0].shape[0], len(c_weights), c_weights[0].device
    if H % 32:
        return None
    wp3 = torch.empty((P, 3 * H * H), dtype=torch.float32, device=dev)
    for i in range(P):
        check(_lib.lib().ic3_commnet_pack_split(ptr(c_weights[i].detach().contiguous().float()),
                                                ptr(f_weights[i].detach().contiguous().float()), ptr(wp3[i]), H, stream()))
    return wp3


def commnet_forward(enc, E, N, wp, bias, head_w, head_b, head_sizes, mode_avg, comm_zero, alive_in, comm_in, out=None,
                    h_out=None, wp3=None):
    """The non-recurrent CommNet module after the encoder, every communication pass in one launch (ic3_commnet_forward):
    enc (E*N, H) = encoder(obs) incl. bias -> out (E*N, OT) = [log-probs of every head | value]."""
    import ctypes as C
    _need_cuda(enc, "commnet_forward")
    R, H = enc.shape
    assert R == E * N and enc.is_contiguous() and enc.dtype == torch.float32
    for m in (alive_in, comm_in):
        assert m is None or (m.dtype == torch.int32 and m.is_contiguous() and m.numel() == R)
    OT = sum(int(a) for a in head_sizes) + 1
    if out is None:
        out = torch.empty((R, OT), dtype=torch.float32, device=enc.device)
    sizes = (C.c_int32 * len(head_sizes))(*[int(a) for a in head_sizes])
    assert wp3 is None or (wp3.is_contiguous() and tuple(wp3.shape) == (wp.shape[0], 3 * H * H))
    check(_lib.lib().ic3_commnet_forward(ptr(enc), E, N, H, wp.shape[0], ptr(wp), ptr(wp3), ptr(bias), ptr(head_w), ptr(head_b), sizes,
                                         len(head_sizes), int(bool(mode_avg)), int(bool(comm_zero)), ptr(alive_in),
                                         ptr(comm_in), ptr(out), ptr(h_out), stream()))
    return out


def commnet_step_supported(env, H):
    return _lib.lib().ic3_commnet_step_supported(env._h, int(H)) > 0


def commnet_step(env, cn, H, head_sizes, mode_avg, comm_zero, alive_in, comm_in, out, action, reward, done, alive=None,
                 is_completed=None, obs=None, h_in=None, h_out=None):
    """One whole rollout iteration of the NON-recurrent module (sparse encoder -> communication passes -> heads -> draws ->
    env.step, + the dense obs rows of the state acted on) in one launch — ic3_commnet_step.  `cn`: the module's derived
    weights (wt, enc_bias, loc_table, wp, bias, w_heads, b_heads; wp3 = the split planes or None: the fp32 instruction)."""
    import ctypes as C
    _need_cuda(out, "commnet_step")
    R = out.shape[0]
    assert out.is_contiguous() and action.is_contiguous() and action.dtype == torch.int32
    assert action.numel() == len(head_sizes) * R and out.shape[1] == sum(int(a) for a in head_sizes) + 1
    for m in (alive_in, comm_in):
        assert m is None or (m.dtype == torch.int32 and m.is_contiguous() and m.numel() == R)
    for v in (h_in, h_out):
        assert v is None or (v.dtype == torch.float32 and v.is_contiguous() and v.numel() == R * int(H))
    sizes = (C.c_int32 * len(head_sizes))(*[int(a) for a in head_sizes])
    check(_lib.lib().ic3_commnet_step(env._h, ptr(cn['wt']), ptr(cn['enc_bias']), ptr(cn['loc_table']), int(H),
                                      cn['wp'].shape[0], ptr(cn['wp']), ptr(cn.get('wp3')), ptr(cn['bias']), ptr(cn['w_heads']),
                                      ptr(cn['b_heads']),
                                      sizes, len(head_sizes), int(bool(mode_avg)), int(bool(comm_zero)), ptr(alive_in),
                                      ptr(comm_in), ptr(h_in), ptr(h_out), ptr(out), ptr(action), ptr(obs), ptr(reward), ptr(done),
                                      ptr(alive), ptr(is_completed), stream()))
    return out


def lstm_gates_backward_supported(H):
    return bool(_lib.lib().ic3_lstm_gates_backward_supported(int(H)))


def lstm_gates_backward(xh, lstm_wp, bias, c_prev, dh, dc, dgates, dc_prev, dbias_partials=None, accumulate=False,
                        h_prev=None, lstm_wp3=None, lstm_wp3_bwd=None, dxh=None):
    """Gate recompute + LSTM cell backward in one launch (ic3_lstm_gates_backward): xh (R,2H) = [inp | h_prev] rows (unit
    column stride; with h_prev (R,H) given the launch fills the h half of xh from it), lstm_wp = policy_step_pack's 'ps_l_wp', bias (4H,) = b_ih + b_hh; c_prev, dh, dc (or None) (R,H) ->
    dgates (R,4H), dc_prev (R,H; may alias dc).  dbias_partials: (ceil(R/64), 4H), written (or added to: `accumulate`)."""
    _need_cuda(xh, "lstm_gates_backward")
    R, H = c_prev.shape
    assert xh.dtype == torch.float32 and xh.stride(1) == 1 and xh.shape == (R, 2 * H)
    for t in (c_prev, dh, dgates, dc_prev, bias, lstm_wp):
        assert t.is_contiguous() and t.dtype == torch.float32
    tiles = (R + 63) // 64
    if dbias_partials is not None:
        assert dbias_partials.is_contiguous() and tuple(dbias_partials.shape) == (tiles, 4 * H)
    if h_prev is not None:
        assert h_prev.is_contiguous() and h_prev.dtype == torch.float32 and tuple(h_prev.shape) == (R, H)
    if dxh is not None:       # + the input gradient [d inp | d h_prev] = dgates . [W_ih | W_hh] in the same launch
        assert lstm_wp3 is not None and lstm_wp3_bwd is not None and dxh.is_contiguous() and tuple(dxh.shape) == (R, 2 * H)
        n = _lib.lib().ic3_lstm_gates_backward_dx(ptr(xh), xh.stride(0), ptr(h_prev) if h_prev is not None else None, ptr(lstm_wp),
                                                  ptr(lstm_wp3), ptr(lstm_wp3_bwd), ptr(bias), ptr(c_prev), ptr(dh),
                                                  ptr(dc) if dc is not None else None, ptr(dgates), ptr(dc_prev),
                                                  ptr(dbias_partials) if dbias_partials is not None else None,
                                                  int(bool(accumulate)), ptr(dxh), R, H, stream())
        if n < 0:
            check(n)
        return n
    n = _lib.lib().ic3_lstm_gates_backward(ptr(xh), xh.stride(0), ptr(h_prev) if h_prev is not None else None, ptr(lstm_wp),
                                           ptr(lstm_wp3) if lstm_wp3 is not None else None,   # split gate product
                                           ptr(bias), ptr(c_prev), ptr(dh),
                                           ptr(dc) if dc is not None else None, ptr(dgates), ptr(dc_prev),
                                           ptr(dbias_partials) if dbias_partials is not None else None,
                                           int(bool(accumulate)), R, H, stream())
    if n < 0:
        check(n)
    return n


def _rowvec(v, R):
    if v is None:
        return None
    assert v.is_contiguous() and v.dtype == torch.float32 and v.numel() == R
    return ptr(v)


def lstm_gates_backward_given(gates, c_prev, dh, dc, dgates, dc_prev, dbias_partials=None, accumulate=False, xh=None,
                              h_prev=None, lstm_wp3_bwd=None, dxh=None, row_live=None, row_keep=None, dhead=None, w_heads=None):
    """The cell's derivative from the RECORDED activated gates (R, 4H) of the step (ic3_lstm_gates_backward_given; the
    rollout's launch stored them: envs.set_record_out) — no gate product.  xh + h_prev: h_prev is copied into the h half of xh;
    lstm_wp3_bwd + dxh: [d inp | d h_prev] in the same launch.  row_live / row_keep (R,) float32 (collection mode): c_prev and
    the copied h_prev times row_live, dc times row_keep, per row.  dgates may be `gates` itself (in place).  dhead (R, OT) +
    w_heads (OT, H): dh + dhead @ w_heads is what the cell sees (the heads' share of dL/dh_t, folded in)."""
    _need_cuda(gates, "lstm_gates_backward_given")
    R, H = c_prev.shape
    for t in (gates, c_prev, dh, dgates, dc_prev):
        assert t.is_contiguous() and t.dtype == torch.float32
    assert tuple(gates.shape) == (R, 4 * H) and tuple(dgates.shape) == (R, 4 * H)
    tiles = (R + 63) // 64
    if dbias_partials is not None:
        assert dbias_partials.is_contiguous() and tuple(dbias_partials.shape) == (tiles, 4 * H)
    if xh is not None:
        assert h_prev is not None and xh.dtype == torch.float32 and xh.stride(1) == 1 and xh.shape == (R, 2 * H)
        assert h_prev.is_contiguous() and tuple(h_prev.shape) == (R, H)
    if dxh is not None:
        assert lstm_wp3_bwd is not None and dxh.is_contiguous() and tuple(dxh.shape) == (R, 2 * H)
    OT = 0
    if dhead is not None:
        OT = dhead.shape[1]
        assert w_heads is not None and dhead.is_contiguous() and dhead.dtype == torch.float32 and dhead.shape[0] == R
        assert w_heads.is_contiguous() and w_heads.dtype == torch.float32 and tuple(w_heads.shape) == (OT, H)
    n = _lib.lib().ic3_lstm_gates_backward_given(ptr(gates), ptr(xh) if xh is not None else None, xh.stride(0) if xh is not None else 0,
                                                 ptr(h_prev) if xh is not None else None,
                                                 ptr(lstm_wp3_bwd) if dxh is not None else None, ptr(c_prev), ptr(dh),
                                                 ptr(dc) if dc is not None else None, ptr(dgates), ptr(dc_prev),
                                                 ptr(dbias_partials) if dbias_partials is not None else None,
                                                 int(bool(accumulate)), ptr(dxh) if dxh is not None else None,
                                                 _rowvec(row_live, R), _rowvec(row_keep, R),
                                                 ptr(dhead) if dhead is not None else None,
                                                 ptr(w_heads) if dhead is not None else None, OT, R, H, stream())
    if n < 0:
        check(n)
    return n


def comm_backward_partials(E, N):
    return int(_lib.lib().ic3_comm_backward_partials(int(E), int(N)))


def comm_backward(dxh, h_prev, alive, gate, c_weight, dh_out, dcw_partials, E, N, mode_avg=True, comm_zero=False,
                  out_scale=None, accumulate=True):
    """ic3_comm_backward: dh_out (R, H) = (dxh[:, H:] + mix(dxh[:, :H]) @ c_weight) * out_scale and dcw_partials (P, H, H)
    (+)= per-workgroup partial sums of mix(d inp)^T @ h_prev — the communication block's and C's share of one recorded step's
    backward in one launch.  comm_zero: dh_out = dxh[:, H:] * out_scale."""
    _need_cuda(dxh, "comm_backward")
    R, H2 = dxh.shape
    H = H2 // 2
    assert R == E * N and dxh.is_contiguous() and dxh.dtype == torch.float32 and dh_out.is_contiguous() and tuple(dh_out.shape) == (R, H)
    if not comm_zero:
        assert h_prev.is_contiguous() and tuple(h_prev.shape) == (R, H) and c_weight.is_contiguous() and tuple(c_weight.shape) == (H, H)
        assert dcw_partials.is_contiguous() and tuple(dcw_partials.shape) == (comm_backward_partials(E, N), H, H)
        for m in (alive, gate):
            assert m is None or (m.dtype == torch.int32 and m.is_contiguous() and m.numel() == R)
    n = _lib.lib().ic3_comm_backward(ptr(dxh), 2 * H, ptr(h_prev), ptr(alive), ptr(gate), ptr(c_weight), _rowvec(out_scale, R),
                                     ptr(dh_out), ptr(dcw_partials), int(bool(accumulate)), E, N, H, int(bool(mode_avg)),
                                     int(bool(comm_zero)), stream())
    if n < 0:
        check(n)
    return n


def lstm_weight_grad(inp, h_prev, dgates, dW, row_live=None, accumulate=True, work=None, split=True):
    """ic3_lstm_weight_grad: dW (2H, 4H) (+)= [inp | h_prev]^T @ dgates over all Q rows of a window of recorded steps in one
    launch.  inp (Q, >= H) rows with unit column stride (the first H floats count: the record's [inp | h] rows), h_prev (Q, H),
    dgates (Q, 4H) contiguous; leading dims may be (T, R).  split (default): exact bf16 split products on the bf16 matrix cores
    (the rollout's gate_split arithmetic); False: the fp32 matrix instruction."""
    _need_cuda(dgates, "lstm_weight_grad")
    H = h_prev.shape[-1]
    Q = dgates.numel() // (4 * H)
    inp2 = inp.reshape(Q, inp.shape[-1])
    assert inp2.stride(1) == 1 and inp2.shape[1] >= H and inp2.dtype == torch.float32
    assert h_prev.is_contiguous() and h_prev.numel() == Q * H and dgates.is_contiguous() and dgates.dtype == torch.float32
    assert dW.is_contiguous() and tuple(dW.shape) == (2 * H, 4 * H)
    if row_live is not None:
        assert row_live.is_contiguous() and row_live.dtype == torch.float32 and row_live.numel() == Q
    work = work if work is not None else dict()
    n = int(_lib.lib().ic3_lstm_weight_grad_scratch_floats(Q, H))
    if n == 0:
        raise NotImplementedError("lstm_weight_grad: hid_size 64 / 128")
    key = ('wgrad', str(dgates.device))
    if key not in work or work[key].numel() < n:
        work[key] = torch.empty((n,), dtype=torch.float32, device=dgates.device)
    check(_lib.lib().ic3_lstm_weight_grad(ptr(inp2), inp2.stride(0), ptr(h_prev), ptr(dgates), ptr(row_live), Q, H, ptr(dW),
                                          int(bool(accumulate)), int(bool(split)), ptr(work[key]), stream()))


def first_chain_envs(E, N):
    """ic3_bptt_first_chain_envs: with ic3_bptt.two_chains, envs [0, E1) run on the caller's stream and [E1, E) on a second one."""
    return int(_lib.lib().ic3_bptt_first_chain_envs(int(E), int(N)))


def bptt_dcw_partials(E, N, two_chains):
    """Slots of ic3_bptt.dcw_partials."""
    E1 = first_chain_envs(E, N) if two_chains else E
    return comm_backward_partials(E1, N) + (comm_backward_partials(E - E1, N) if E1 < E else 0)


def bptt_backward_supported(env, H):
    return bool(_lib.lib().ic3_bptt_backward_supported(env._h, int(H)))


def bptt_backward(env, T, E, N, H, gates, hs, cs, dhead, snaps, alive, gate, lstm_wp3_bwd, w_heads, c_weight, dh, dc, dxh,
                  dbias_partials, dcw_partials, mode_avg=True, comm_zero=False, detach_gap=0, row_live=None, row_keep=None,
                  enc_first=True, gate_events=None, two_chains=False):
    """ic3_bptt_backward: the backward through a window of T recorded steps as one host call (gate launch in place on the
    record, communication backward, encoder backward stage 1 — three launches per step).  alive / gate: lists of T tensors
    (E, N) int32 or None entries, or None.  gate_events: a list that receives (start, stop, t) DispatchEvent triples stamped
    around every step's gate launch (measurement: bench.py --mode train)."""
    import ctypes as C
    _need_cuda(gates, "bptt_backward")
    R = E * N
    OT = dhead.shape[-1]
    assert gates.is_contiguous() and gates.numel() == T * R * 4 * H and gates.dtype == torch.float32
    assert hs.is_contiguous() and cs.is_contiguous() and hs.shape[0] >= T and tuple(hs.shape[1:]) == (R, H) == tuple(cs.shape[1:])
    assert dhead.is_contiguous() and dhead.numel() == T * R * OT and dhead.dtype == torch.float32
    assert snaps.is_contiguous() and snaps.dtype == torch.int32 and snaps.shape[0] >= T
    for v in (dh, dc):
        assert v.is_contiguous() and tuple(v.shape) == (R, H) and v.dtype == torch.float32
    assert dxh.is_contiguous() and tuple(dxh.shape) in ((R, 2 * H), (T, R, 2 * H))      # one buffer, or a ring of T (dxh_step)
    assert dbias_partials.is_contiguous() and tuple(dbias_partials.shape) == ((R + 63) // 64, 4 * H)
    two_chains = bool(two_chains) and dxh.dim() == 3 and first_chain_envs(E, N) < E
    if not comm_zero:
        assert dcw_partials.is_contiguous() and tuple(dcw_partials.shape) == (bptt_dcw_partials(E, N, two_chains), H, H)
        assert c_weight.is_contiguous() and tuple(c_weight.shape) == (H, H)
    for v in (row_live, row_keep):
        assert v is None or (v.is_contiguous() and v.dtype == torch.float32 and tuple(v.shape) == (T, R))

    def ptrs(ms):
        if ms is None or all(m is None for m in ms):
            return None
        assert len(ms) >= T
        for m in ms[:T]:
            assert m is None or (m.dtype == torch.int32 and m.is_contiguous() and m.numel() == R)
        return (C.c_void_p * T)(*[None if m is None else m.data_ptr() for m in ms[:T]])
    pa, pg = ptrs(alive), ptrs(gate)
    b = _lib.Bptt()
    b.struct_size = C.sizeof(b)
    b.T, b.E, b.N, b.H, b.OT = T, E, N, H, OT
    b.mode_avg, b.comm_zero, b.detach_gap, b.enc_first = int(bool(mode_avg)), int(bool(comm_zero)), int(detach_gap), int(bool(enc_first))
    b.gates, b.hs, b.cs, b.dhead, b.snaps = gates.data_ptr(), hs.data_ptr(), cs.data_ptr(), dhead.data_ptr(), snaps.data_ptr()
    b.snap_words = snaps.stride(0)
    b.alive = C.cast(pa, C.POINTER(C.c_void_p)) if pa is not None else None
    b.gate = C.cast(pg, C.POINTER(C.c_void_p)) if pg is not None else None
    b.row_live = row_live.data_ptr() if row_live is not None else None
    b.row_keep = row_keep.data_ptr() if row_keep is not None else None
    b.lstm_wp3_bwd, b.w_heads = lstm_wp3_bwd.data_ptr(), w_heads.data_ptr()
    b.c_weight = c_weight.data_ptr() if c_weight is not None else None
    b.dh, b.dc, b.dxh = dh.data_ptr(), dc.data_ptr(), dxh.data_ptr()
    b.dbias_partials = dbias_partials.data_ptr()
    b.dcw_partials = dcw_partials.data_ptr() if dcw_partials is not None else None
    b.two_chains = int(two_chains)
    if dxh.dim() == 3:                     # a ring of per-step input gradients: the encoder's first stage once, over the window
        b.dxh_step = dxh.stride(0)
        b.enc_work = env.encode_window_work(H).data_ptr()
    else:
        b.enc_work = env._encb_work(H).data_ptr()
    evs = None
    if gate_events is not None:
        from .envs import DispatchEvent
        evs = [DispatchEvent() for _ in range(2 * T)]
        arr = (C.c_void_p * (2 * T))(*[e.handle.value for e in evs])
        b.gate_events = C.cast(arr, C.POINTER(C.c_void_p))
    check(_lib.lib().ic3_bptt_backward(env._h, C.byref(b), stream()))
    if evs is not None:
        gate_events.extend((evs[2 * t], evs[2 * t + 1], t) for t in range(T))


HEADS_GRAD_MAX_OT = 16      # ic3_heads_grad: at most 16 output columns (the heads' actions in total + the value)


def heads_grad(d, h, dW, db, work=None):
    """dW (OT,H) += d^T . h, db (OT,) += column sums of d over all M rows of d (M,OT) / h (M,H) — ic3_heads_grad: the heads'
    weight gradient of a whole episode in one pass."""
    _need_cuda(d, "heads_grad")
    M, OT = d.shape
    H = h.shape[1]
    assert d.is_contiguous() and h.is_contiguous() and h.shape[0] == M and d.dtype == h.dtype == torch.float32
    assert dW.is_contiguous() and tuple(dW.shape) == (OT, H) and db.is_contiguous() and db.numel() == OT
    work = work if work is not None else dict()
    key = ('heads_grad', H, str(d.device))
    if key not in work:
        work[key] = torch.empty((int(_lib.lib().ic3_heads_grad_scratch_floats(H)),), dtype=torch.float32, device=d.device)
    check(_lib.lib().ic3_heads_grad(ptr(d), ptr(h), M, H, OT, ptr(dW), ptr(db), ptr(work[key]), stream()))


def policy_heads(h, W, b, head_sizes, out=None):
    """h (R,H) rows (unit column stride), W (OT,H), b (OT,) -> out (R,OT) = [log_softmax heads | value]."""
    import ctypes as C
    _need_cuda(h, "policy_heads")
    R, H = h.shape
    OT = W.shape[0]
    assert h.stride(1) == 1 and W.is_contiguous() and OT == sum(head_sizes) + 1
    if out is None:
        out = torch.empty((R, OT), dtype=torch.float32, device=h.device)
    sizes = (C.c_int32 * len(head_sizes))(*[int(a) for a in head_sizes])
    check(_lib.lib().ic3_policy_heads(ptr(h), h.stride(0), ptr(W), ptr(b), sizes, len(head_sizes), ptr(out), R, H,
                                      stream()))
    return out


POLICY_STEP_SIZES = (64, 128, 256)


def padded_hidden(H):
    """The next hidden size the one-launch kernels are built for, or None (H is one of them, or larger than all):
    comm.CommNetMLP runs other sizes zero-padded to it (exact: a padded unit's pre-activations, state and output are 0)."""
    H = int(H)
    if H in POLICY_STEP_SIZES or H < 1:
        return None
    for s in POLICY_STEP_SIZES:
        if H < s:
            return s
    return None


def policy_step_pack(c_weight, w_ih, w_hh):
    """C.weight (H,H) and [W_ih | W_hh] (4H, 2H) in the layout ic3_policy_step streams (ic3_policy_pack)."""
    _need_cuda(c_weight, "policy_step_pack")
    H = c_weight.shape[0]
    dev = c_weight.device
    c_wp = torch.empty((H * H,), dtype=torch.float32, device=dev)
    l_wp = torch.empty((8 * H * H,), dtype=torch.float32, device=dev)
    check(_lib.lib().ic3_policy_pack(ptr(c_weight.detach().contiguous().float()), ptr(w_ih.detach().contiguous().float()),
                                     ptr(w_hh.detach().contiguous().float()), ptr(c_wp), ptr(l_wp), H, stream()))
    return dict(ps_c_wp=c_wp, ps_l_wp=l_wp)


def policy_step_supported(env, H):
    """True when ic3_policy_step can run this env / hid_size (env: the raw batched env object)."""
    return H in POLICY_STEP_SIZES and _lib.lib().ic3_policy_step_supported(env._h, int(H)) > 0


def policy_pack_split(w_ih, w_hh):
    """ic3_policy.gate_split (the default): [W_ih | W_hh] as three exact bf16 planes in MFMA fragment order."""
    _need_cuda(w_ih, "policy_pack_split")
    H = w_ih.shape[1]
    wp3 = torch.empty((3 * 2 * H * 4 * H,), dtype=torch.bfloat16, device=w_ih.device)
    check(_lib.lib().ic3_policy_pack_split(ptr(w_ih.detach().contiguous().float()), ptr(w_hh.detach().contiguous().float()),
                                           ptr(wp3), H, stream()))
    return wp3


def policy_pack_split_bwd(w_ih, w_hh):
    """The split planes of [W_ih | W_hh] in the fragment order of the gate product's BACKWARD (ic3_lstm_gates_backward_dx)."""
    _need_cuda(w_ih, "policy_pack_split_bwd")
    H = w_ih.shape[1]
    wp3 = torch.empty((3 * 4 * H * 2 * H,), dtype=torch.bfloat16, device=w_ih.device)
    check(_lib.lib().ic3_policy_pack_split_bwd(ptr(w_ih.detach().contiguous().float()), ptr(w_hh.detach().contiguous().float()),
                                               ptr(wp3), H, stream()))
    return wp3


def policy_step_passes_supported(fc, H, passes):
    """comm_passes > 1 as a loop INSIDE one ic3_policy_step launch (ic3_policy.npasses): the split gate product, hid 64 / 128,
    at most 4 passes; otherwise one launch per pass."""
    return 2 <= passes <= 4 and H in (64, 128) and fc.get('ps_l_wp3') is not None


def _policy_struct(fc, H, head_sizes, mode_avg, comm_zero, encoder=True, pass_index=0, inner=False, passes=1):
    """pass_index / inner: comm_passes > 1 — pass i uses C_modules[i] (cache keys '<name>_p<i>' for i > 0), every pass but
    the last is an `inner` one (h, c only).  passes >= 2: all of them in one launch (npasses)."""
    sfx = '' if pass_index == 0 else '_p%d' % pass_index
    pol = _lib.Policy()
    pol.pass_index, pol.inner_pass = int(pass_index), int(bool(inner))
    if passes >= 2:
        assert pass_index == 0 and not inner and encoder
        pol.npasses = int(passes)
        for i in range(passes):
            sf = '' if i == 0 else '_p%d' % i
            pol.c_wp_pass[i] = fc['ps_c_wp' + sf].data_ptr()
            pol.enc_bias_pass[i] = fc['enc_bias' + sf].data_ptr()
    pol.H, pol.nheads = int(H), len(head_sizes)
    for i, a in enumerate(head_sizes):
        pol.head_sizes[i] = int(a)
    pol.mode_avg, pol.comm_zero = int(bool(mode_avg)), int(bool(comm_zero))
    if encoder:
        pol.enc_wt, pol.enc_bias = fc['wt'].data_ptr(), fc['enc_bias' + sfx].data_ptr()
        pol.loc_table = fc['loc_table'].data_ptr() if fc.get('loc_table') is not None else None
    pol.c_wp, pol.lstm_wp, pol.lstm_bias = fc['ps_c_wp' + sfx].data_ptr(), fc['ps_l_wp'].data_ptr(), fc['b_cat'].data_ptr()
    pol.head_w, pol.head_b = fc['w_heads'].data_ptr(), fc['b_heads'].data_ptr()
    if fc.get('ps_l_wp3') is not None:                           # gate_split: exact split products on the bf16 matrix cores
        pol.gate_split, pol.lstm_wp3 = 1, fc['ps_l_wp3'].data_ptr()
    return pol


def policy_forward(fc, H, head_sizes, mode_avg, comm_zero, enc, E, N, h, c, alive_in, comm_in, out=None, pass_index=0,
                   inner=False):
    """The policy half of policy_step for a caller-supplied encoder output enc (E*N, H) = encoder(x) + C.bias
    (ic3_policy_forward): communication block, C, LSTMCell, heads, log_softmax in one launch; h, c in place.
    inner=True: a non-final communication pass (comm_passes > 1) — h, c only, returns None."""
    import ctypes as C
    _need_cuda(enc, "policy_forward")
    R = E * N
    assert enc.is_contiguous() and enc.shape == (R, H) and h.is_contiguous() and c.is_contiguous()
    for m in (alive_in, comm_in):
        assert m is None or (m.dtype == torch.int32 and m.is_contiguous() and m.numel() == R)
    if out is None and not inner:
        out = torch.empty((R, sum(head_sizes) + 1), dtype=torch.float32, device=enc.device)
    pol = _policy_struct(fc, H, head_sizes, mode_avg, comm_zero, encoder=False, pass_index=pass_index, inner=inner)
    check(_lib.lib().ic3_policy_forward(C.byref(pol), ptr(enc), E, N, ptr(h), ptr(c), ptr(alive_in), ptr(comm_in),
                                        None if inner else ptr(out), stream()))
    return None if inner else out


def policy_step_pass(env, fc, H, head_sizes, mode_avg, comm_zero, h, c, alive_in, comm_in, pass_index):
    """A non-final communication pass of a comm_passes > 1 policy on the env's own state (ic3_policy_step with
    ic3_policy.inner_pass): sparse encoder, communication block, C_modules[pass_index], LSTMCell — h, c in place."""
    _need_cuda(h, "policy_step_pass")
    import ctypes as C
    pol = _policy_struct(fc, H, head_sizes, mode_avg, comm_zero, pass_index=pass_index, inner=True)
    check(_lib.lib().ic3_policy_step(env._h, C.byref(pol), ptr(h), ptr(c), ptr(alive_in), ptr(comm_in), None, None, None,
                                     None, None, None, None, stream()))


def policy_step(env, fc, H, head_sizes, mode_avg, comm_zero, h, c, alive_in, comm_in, out, action, reward, done,
                alive=None, is_completed=None, obs=None, pass_index=0, passes=1):
    """One whole rollout iteration (policy forward -> action draws -> env.step) in one launch — ic3_policy_step.
    `fc`: the policy's derived-weight cache (wt, enc_bias, loc_table, ps_c_wp, ps_l_wp, b_cat, w_heads, b_heads);
    h, c (E*N, H) contiguous, updated in place; out (E*N, OT); action (heads, E, N) int32."""
    _need_cuda(h, "policy_step")
    R = h.shape[0]
    assert h.is_contiguous() and c.is_contiguous() and h.shape == (R, H) and c.shape == (R, H)
    assert out.is_contiguous() and action.is_contiguous() and action.dtype == torch.int32
    assert action.numel() == len(head_sizes) * R and out.shape == (R, sum(head_sizes) + 1)
    for m in (alive_in, comm_in):
        assert m is None or (m.dtype == torch.int32 and m.is_contiguous() and m.numel() == R)
    pol = _policy_struct(fc, H, head_sizes, mode_avg, comm_zero, pass_index=pass_index, passes=passes)
    import ctypes as C
    check(_lib.lib().ic3_policy_step(env._h, C.byref(pol), ptr(h), ptr(c), ptr(alive_in), ptr(comm_in), ptr(out),
                                     ptr(action), ptr(obs), ptr(reward), ptr(done), ptr(alive), ptr(is_completed), stream()))
    return out


def episode_finalize(done, reward, alive=None, is_completed=None, gate=None, gate_ones=False, auto_reset=False,
                     forced_last=False, work=None):
    """The per-step derivations of get_episode (trainer.py:70-105,109-110) for n slots of E envs in one launch —
    ic3_episode_finalize.  done (n, E) int32; reward (n, E, N) f32; alive / is_completed (n, E, N) int32 or None;
    gate (n, E, N) int32 view with contiguous (E, N) slots (the talk head's actions) or None.
    Returns dict(live (n,E), alive_mask, episode_mask (n,E), episode_mini_mask, live_after (E,), stats) where `stats`
    is a device fp64 tensor [num_steps, zero_len_envs, reward_sum[N], gate_sum[N]] (read it after a sync).
    `work`: dict reused between calls for the scratch buffers."""
    import ctypes as C
    _need_cuda(reward, "episode_finalize")
    n, E, N = reward.shape
    dev = reward.device
    assert done.dtype == torch.int32 and done.shape == (n, E) and done.is_contiguous()
    assert reward.dtype == torch.float32 and reward.is_contiguous()
    for m in (alive, is_completed):
        assert m is None or (m.dtype == torch.int32 and m.shape == (n, E, N) and m.is_contiguous())
    gate_stride = 0
    if gate is not None:
        assert gate.dtype == torch.int32 and gate.shape == (n, E, N) and gate[0].is_contiguous()
        gate_stride = gate.stride(0) if n > 1 else E * N
    work = work if work is not None else dict()
    key = (E, N, str(dev))
    if work.get('key') != key:
        nbytes = int(_lib.lib().ic3_episode_scratch_bytes(E, N))
        if nbytes == 0:
            raise ValueError("episode_finalize: unsupported sizes E=%d N=%d" % (E, N))
        work.update(key=key, scratch=torch.empty(nbytes // 8, dtype=torch.float64, device=dev),
                    counter=torch.zeros(1, dtype=torch.int32, device=dev))
    f = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
    res = dict(live=f(n, E), alive_mask=f(n, E, N), episode_mask=f(n, E), episode_mini_mask=f(n, E, N),
               live_after=f(E), stats=torch.empty(2 + 2 * N, dtype=torch.float64, device=dev))
    ep = _lib.Episode(n, E, N, int(bool(auto_reset)), int(bool(forced_last)), int(bool(gate_ones)), ptr(done),
                      ptr(alive), ptr(is_completed), ptr(reward), ptr(gate), gate_stride, ptr(res['live']),
                      ptr(res['alive_mask']), ptr(res['episode_mask']), ptr(res['episode_mini_mask']),
                      ptr(res['live_after']), ptr(res['stats']), ptr(work['scratch']), ptr(work['counter']))
    check(_lib.lib().ic3_episode_finalize(C.byref(ep), stream()))
    return res


def returns_scan(rewards, episode_masks, episode_mini_masks, gamma, mean_ratio):
    """trainer.py:162-171: returns (T, E, N) = mean_ratio * mean over agents of the cooperative returns + (1 - mean_ratio) *
    the per-agent returns, both scanned backwards over the T slots with episode_mask (and episode_mini_mask) cutting the
    recursion — ic3_returns_scan, one launch instead of ~9 tensor ops per slot.  episode_masks (T, E) or (T, E, N) expanded."""
    _need_cuda(rewards, "returns_scan")
    T, E, N = rewards.shape
    em = episode_masks[:, :, 0] if episode_masks.dim() == 3 else episode_masks
    rewards, em, mm = rewards.contiguous().float(), em.contiguous().float(), episode_mini_masks.contiguous().float()
    out = torch.empty_like(rewards)
    check(_lib.lib().ic3_returns_scan(ptr(rewards), ptr(em), ptr(mm), float(gamma), float(mean_ratio), ptr(out), T, E, N, stream()))
    return out


def loss_gradients(out, action, returns, alive_mask, live, head_sizes, entr, value_coeff, adv_shift=0.0, adv_scale=1.0):
    """compute_grad's losses and dL/d[logits | value] of every transition in one launch (ic3_loss_gradients): out (T, R, OT) the
    step launches' rows, action (T, heads, R) int32, returns / alive_mask (T, R), live (T, E).  Returns (d_out (T, R, OT),
    (action_loss, value_loss, entropy) as a float64 tensor of 3 on the device)."""
    import ctypes as C
    _need_cuda(out, "loss_gradients")
    T, R, OT = out.shape
    E = live.shape[1]
    N = R // E
    nh = len(head_sizes)
    assert OT == sum(int(a) for a in head_sizes) + 1 and tuple(action.shape) == (T, nh, R) and action.dtype == torch.int32
    for v in (out, action, returns, alive_mask, live):
        assert v.is_contiguous()
    assert returns.numel() == T * R == alive_mask.numel() and live.numel() == T * E and E * N == R
    d_out = torch.empty_like(out)
    sums = torch.empty((int(_lib.lib().ic3_loss_gradients_partials(T, R)), 3), dtype=torch.float64, device=out.device)
    sizes = (C.c_int32 * nh)(*[int(a) for a in head_sizes])
    check(_lib.lib().ic3_loss_gradients(ptr(out), ptr(action), ptr(returns), ptr(alive_mask), ptr(live), sizes, nh,
                                        float(adv_shift), float(adv_scale), float(entr), float(value_coeff), ptr(d_out), ptr(sums),
                                        T, E, N, stream()))
    return d_out, sums.sum(0)


def lstm_cell_heads_ok(H):
    """ic3_lstm_cell_heads needs H/4 to be a power of two <= 64."""
    return H % 4 == 0 and H // 4 <= 64 and (H // 4) & (H // 4 - 1) == 0


def lstm_cell_heads_(gates, c, h_out, W, b, head_sizes, out=None, env=None, action=None):
    """lstm_cell_ + policy_heads (+ the action draws of every head into `action` (heads, E, N) int32 when `env`, the
    raw batched env, is given) in one launch.  Returns out (R, OT)."""
    import ctypes as C
    _need_cuda(gates, "lstm_cell_heads_")
    R, H = c.shape
    OT = W.shape[0]
    assert gates.is_contiguous() and c.is_contiguous() and h_out.stride(1) == 1 and W.is_contiguous()
    assert OT == sum(head_sizes) + 1
    if out is None:
        out = torch.empty((R, OT), dtype=torch.float32, device=c.device)
    if action is not None:
        assert env is not None and action.is_contiguous() and action.dtype == torch.int32 \
            and action.numel() == len(head_sizes) * R
    sizes = (C.c_int32 * len(head_sizes))(*[int(a) for a in head_sizes])
    check(_lib.lib().ic3_lstm_cell_heads(ptr(gates), ptr(c), ptr(h_out), h_out.stride(0), R, H, ptr(W), ptr(b), sizes,
                                         len(head_sizes), ptr(out), env._h if action is not None else None, ptr(action),
                                         stream()))
    return out


def comm_masked_mean(h, alive=None, comm_action=None, mode_avg=True, mask_self=True):
    """h (E,N,H) f32; alive / comm_action (E,N) int32 CUDA tensors or None -> (E,N,H)."""
    if alive is not None:
        alive = alive.to(torch.int32).contiguous()
    if comm_action is not None:
        comm_action = comm_action.to(torch.int32).contiguous()
    return _CommMaskedMean.apply(h, alive, comm_action, mode_avg, mask_self)


def sample_actions(logp, head, seed, env_id_offset, episode, t, want_logp=False, out=None):
    """logp (E,N,A) f32 -> action (E,N) int32 [, chosen log-prob (E,N) f32]; action_utils.py:32-36."""
    _need_cuda(logp, "sample_actions")
    E, N, A = logp.shape
    logp, ld = _rows(logp.detach(), A)
    action = out if out is not None else torch.empty((E, N), dtype=torch.int32, device=logp.device)
    chosen = torch.empty((E, N), dtype=torch.float32, device=logp.device) if want_logp else None
    check(_lib.lib().ic3_sample_actions(ptr(logp), ld, A, int(head), int(seed) & 0xffffffff, int(env_id_offset),
                                        int(episode), int(t), ptr(action), ptr(chosen), E, N, stream()))
    return (action, chosen) if want_logp else action


def sample_actions_env(env, logp, head, out=None, want_logp=False):
    """sample_actions with (seed, env_id_offset, episode, t) taken from the env handle's own device-side counters
    (ic3_env_sample_actions): identical draws, but every launch argument is constant -> hipGraph-capturable."""
    _need_cuda(logp, "sample_actions_env")
    E, N, A = logp.shape
    logp, ld = _rows(logp.detach(), A)
    action = out if out is not None else torch.empty((E, N), dtype=torch.int32, device=logp.device)
    chosen = torch.empty((E, N), dtype=torch.float32, device=logp.device) if want_logp else None
    check(_lib.lib().ic3_env_sample_actions(env._h, ptr(logp), ld, A, int(head), ptr(action), ptr(chosen), stream()))
    return (action, chosen) if want_logp else action


def random_actions(E, N, naction, seed, env_id_offset, episode, t, device='cuda', out=None):
    action = out if out is not None else torch.empty((E, N), dtype=torch.int32, device=device)
    with torch.cuda.device(action.device):
        check(_lib.lib().ic3_random_actions(ptr(action), int(naction), int(seed) & 0xffffffff, int(env_id_offset),
                                            int(episode), int(t), E, N, stream()))
    return action
