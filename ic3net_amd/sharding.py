"""Multi-GPU: one process per GPU, environments sharded by contiguous global env-id ranges, NO collective
on the rollout path (envs are independent).  The only exchange is at update time: the
reference's MultiProcessTrainer sums worker gradients and stats and divides by the total num_steps
(/root/reference/multi_processing.py:74-98) — here one all-reduce(sum) over RCCL (gloo in CPU tests).
Whenever a process group is initialised the collectives run — also for a one-rank group, which is how the RCCL code
path is exercised on a single GPU (tests/test_rccl_world1_gpu.py).
"""
import numpy as np
import torch


def shard_range(total_envs, rank, world):
    """Global env ids [lo, hi) owned by `rank`: contiguous, sizes differ by at most one."""
    base, rem = divmod(int(total_envs), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def worker_seed(seed, rank):
    """multi_processing.py:16-17: worker `id` seeds torch/numpy with seed + id + 1 (master keeps `seed`).
    The env streams do NOT use this: they are keyed by (seed, global env id) and are shard-invariant."""
    return seed if rank == 0 else seed + (rank - 1) + 1


def _dist_device(group=None):
    import torch.distributed as dist
    return torch.device('cuda', torch.cuda.current_device()) if dist.get_backend(group) == 'nccl' else torch.device('cpu')


def broadcast_seed(seed, src=0, group=None):
    """Every rank adopts rank `src`'s seed (main.py:157-159 draws it once; the reference's workers inherit it)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return int(seed)
    t = torch.tensor([int(seed)], dtype=torch.int64, device=_dist_device(group))
    dist.broadcast(t, src=src, group=group)
    return int(t.item())


def broadcast_parameters(module, optimizer=None, src=0, group=None):
    """Replicas start from rank `src`'s parameters, buffers and optimizer state tensors (the reference keeps one
    shared-memory parameter set, main.py:177-178 / multi_processing.py:24-27)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return
    tensors = [p.data for p in module.parameters()] + [b.data for b in module.buffers()]
    if optimizer is not None:
        for pid in sorted(optimizer.state_dict()['state']):
            st = optimizer.state_dict()['state'][pid]
            tensors += [st[k] for k in sorted(st) if torch.is_tensor(st[k])]
    dev = _dist_device(group)
    for t in tensors:
        if t.device == dev:
            dist.broadcast(t, src=src, group=group)
        else:                                   # e.g. CPU tensors under an RCCL group (RMSprop's `step`)
            tmp = t.to(dev)
            dist.broadcast(tmp, src=src, group=group)
            t.copy_(tmp)


def allreduce_stats(stat, group=None):
    """merge_stat across ranks (multi_processing.py:86-88): numeric / ndarray entries are summed."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return stat
    stat = dict(stat)
    keys = sorted(k for k, v in stat.items() if isinstance(v, (int, float, np.ndarray, np.floating, np.integer)))
    flat = np.concatenate([np.atleast_1d(np.asarray(stat[k], np.float64)).ravel() for k in keys])
    dev = 'cuda' if dist.get_backend(group) == 'nccl' else 'cpu'
    t = torch.from_numpy(flat).to(dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    flat = t.cpu().numpy()
    out, i = dict(stat), 0
    for k in keys:
        n = np.atleast_1d(np.asarray(stat[k])).size
        v = flat[i:i + n]
        out[k] = v.reshape(np.shape(stat[k])) if isinstance(stat[k], np.ndarray) else float(v[0])
        i += n
    return out


def allreduce_grads(params, num_steps_total, group=None):
    """multi_processing.py:90-97: grads summed over workers, divided by the global num_steps; params without
    a grad (e.g. the unused hidd_encoder, quirk Q18) are skipped, exactly like `p._grad is not None`."""
    import torch.distributed as dist
    grads = [p.grad for p in params if p.grad is not None]
    if not grads:
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    if dist.is_available() and dist.is_initialized():     # (also a one-rank group: the same RCCL path)
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)   # one bucket: ~0.6 M fp32 for PP-hard
    flat /= float(num_steps_total)
    i = 0
    for g in grads:
        n = g.numel()
        g.copy_(flat[i:i + n].view_as(g))
        i += n
