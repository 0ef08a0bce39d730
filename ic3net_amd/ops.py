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
        return _launch(h, alive, comm_action, mode_avg, mask_self)

    @staticmethod
    def backward(ctx, g):
        return _launch(g.contiguous(), ctx.alive, ctx.comm_action, ctx.mode_avg, ctx.mask_self), None, None, None, None


def _launch(h, alive, comm_action, mode_avg, mask_self):
    _need_cuda(h, "comm_masked_mean")
    E, N, H = h.shape
    h = h.contiguous().float()
    out = torch.empty_like(h)
    with torch.cuda.device(h.device):
        check(_lib.lib().ic3_comm_masked_mean(ptr(h), ptr(alive), ptr(comm_action), ptr(out), E, N, H, int(mode_avg),
                                              int(mask_self), stream()))
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
    logp = logp.detach().contiguous().float()
    action = out if out is not None else torch.empty((E, N), dtype=torch.int32, device=logp.device)
    chosen = torch.empty((E, N), dtype=torch.float32, device=logp.device) if want_logp else None
    with torch.cuda.device(logp.device):
        check(_lib.lib().ic3_sample_actions(ptr(logp), A, int(head), int(seed) & 0xffffffff, int(env_id_offset),
                                            int(episode), int(t), ptr(action), ptr(chosen), E, N, stream()))
    return (action, chosen) if want_logp else action


def random_actions(E, N, naction, seed, env_id_offset, episode, t, device='cuda', out=None):
    action = out if out is not None else torch.empty((E, N), dtype=torch.int32, device=device)
    with torch.cuda.device(action.device):
        check(_lib.lib().ic3_random_actions(ptr(action), int(naction), int(seed) & 0xffffffff, int(env_id_offset),
                                            int(episode), int(t), E, N, stream()))
    return action
