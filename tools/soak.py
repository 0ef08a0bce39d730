"""Soak: many graph-replayed episodes; throughput per block of episodes and allocated memory must stay flat."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

for workload, E in (("pp_hard", 8192), ("tj_hard", 4096)):
    tr, a = bench.build_trainer(workload, E, 0, 0, 0)
    a.hip_graph = True
    for ep in range(3):
        tr.get_episode(ep)
    torch.cuda.synchronize()
    base = torch.cuda.memory_allocated()
    for block in range(4):
        t0 = time.perf_counter()
        steps = 0
        for ep in range(12):
            _, st = tr.get_episode(3 + block * 12 + ep)
            steps += st['num_steps']
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("%s block %d: %.1f M agent-steps/s, allocated %+.1f MB vs start, reserved %.2f GB" %
              (workload, block, a.nagents * steps / dt / 1e6, (torch.cuda.memory_allocated() - base) / 2 ** 20,
               torch.cuda.memory_reserved() / 2 ** 30), flush=True)
    del tr
    torch.cuda.empty_cache()

# round 5: the update half on recorded gates — 60 updates per workload in blocks of 15; throughput and memory must stay flat
# (every update allocates and releases an episode record of 20 / 40 GB)
for workload, E in (("pp_hard", 8192), ("tj_hard", 8192)):
    tr, a = bench.build_trainer(workload, E, 0, 0, 0)
    a.__dict__.update(gamma=1.0, normalize_rewards=False, entr=0, value_coeff=0.01, advantages_per_action=False,
                      batch_size=E * a.max_steps)
    for u in range(2):
        tr.train_batch(u)
    torch.cuda.synchronize()
    base = torch.cuda.memory_allocated()
    for block in range(4):
        t0 = time.perf_counter()
        steps = 0
        for u in range(15):
            st = tr.train_batch(2 + block * 15 + u)
            steps += st['num_steps']
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("train_batch %s block %d: %.1f M agent-steps/s, allocated %+.1f MB vs start, reserved %.2f GB, peak %.1f GB" %
              (workload, block, a.nagents * steps / dt / 1e6, (torch.cuda.memory_allocated() - base) / 2 ** 20,
               torch.cuda.memory_reserved() / 2 ** 30, torch.cuda.max_memory_allocated() / 2 ** 30), flush=True)
    del tr
    torch.cuda.empty_cache()
