import sys, time
sys.path.insert(0, '/root/repo')
import torch, bench
tr, a = bench.build_trainer('pp_hard', 8192, 0, 0, 0)
a.__dict__.update(gamma=1.0, normalize_rewards=False, entr=0, value_coeff=0.01, advantages_per_action=False, batch_size=8192 * a.max_steps)
for u in range(34):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    st = tr.train_batch(u)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(u, "%.1f ms" % (dt * 1e3), "steps", st['num_steps'], "episodes", st['num_episodes'], "peak %.1f GB" % (torch.cuda.max_memory_allocated() / 2**30), flush=True)
