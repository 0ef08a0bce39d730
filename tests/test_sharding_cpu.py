"""CPU, world_size 2 over gloo: the N>1 path.  Envs shard by contiguous global-id ranges with no
data-path collective; streams are keyed by the GLOBAL env id, so a sharded run reproduces the single-
process run env for env (checked with the oracle envs standing in for the HIP kernels, which share the
stream contract); stats / grads are all-reduced (sum) like multi_processing.py:74-98."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ic3net_amd.sharding import shard_range, allreduce_stats, allreduce_grads, worker_seed


def test_shard_range_partitions():
    for total, world in ((8192, 8), (65536, 8), (10, 3), (7, 8), (1, 1)):
        spans = [shard_range(total, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == total
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 1
    assert [worker_seed(5, r) for r in range(3)] == [5, 6, 7]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rollout(env_ids, T, seed):
    """Oracle PP envs with stream-driven pseudo-random actions: returns per-env trajectories + a stat dict."""
    import oracle
    from oracle import philox
    out, reward_sum, steps = {}, np.zeros(3), 0
    for gid in env_ids:
        env = oracle.PPOracle(3, 5, 1, 'mixed', seed=seed, env_gid=gid)
        env.reset()
        traj = []
        for t in range(T):
            act = [(philox.x24(seed, gid, philox.DOMAIN_BENCH, 0, t, n) * 5) >> 24 for n in range(3)]
            obs, rew, done = env.step(act)
            traj.append((env.loc.copy(), rew.copy()))
            reward_sum += rew
            steps += 1
            if done:
                break
        out[gid] = traj
    return out, {'reward': reward_sum, 'num_steps': float(steps), 'num_episodes': float(len(env_ids))}


def _worker(rank, world, port, total, T, seed, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_range(total, rank, world)
    traj, stat = _rollout(range(lo, hi), T, seed)
    stat = allreduce_stats(stat)
    # gradient all-reduce against a closed form: rank r contributes (r+1) * ones
    p = torch.nn.Parameter(torch.zeros(5))
    unused = torch.nn.Parameter(torch.zeros(2))            # no grad -> skipped (quirk Q18)
    p.grad = torch.full((5,), float(rank + 1))
    allreduce_grads([p, unused], stat['num_steps'])
    q.put((rank, {g: [(l.tolist(), r.tolist()) for l, r in tr] for g, tr in traj.items()},
           {k: np.asarray(v).tolist() for k, v in stat.items()}, p.grad.tolist(), unused.grad is None))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_rollout_matches_single_process():
    total, T, seed, world = 12, 15, 3, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, T, seed, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    single_traj, single_stat = _rollout(range(total), T, seed)
    merged = {}
    for rank, traj, stat, grad, unused_none in res:
        merged.update({int(k): v for k, v in traj.items()})
        # every rank holds the same all-reduced stats == the single-process stats
        np.testing.assert_allclose(stat['reward'], single_stat['reward'], rtol=0, atol=1e-12)
        assert stat['num_steps'] == single_stat['num_steps'] and stat['num_episodes'] == total
        np.testing.assert_allclose(grad, [3.0 / single_stat['num_steps']] * 5)
        assert unused_none
    assert sorted(merged) == list(range(total))
    for gid in range(total):
        assert len(merged[gid]) == len(single_traj[gid])
        for (l, r), (l2, r2) in zip(merged[gid], single_traj[gid]):
            assert l == l2.tolist() and r == r2.tolist()
