"""Allocated / peak memory around the first updates of a PP-hard train_batch (is a second episode record ever alive?)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, gc
import bench
tr, a = bench.build_trainer('pp_hard', 8192, 0, 0, 0)
a.__dict__.update(gamma=1.0, normalize_rewards=False, entr=0, value_coeff=0.01, advantages_per_action=False, batch_size=8192 * a.max_steps)
G = 2 ** 30
for u in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    torch.cuda.reset_peak_memory_stats()
    before = torch.cuda.memory_allocated()
    tr.train_batch(u)
    torch.cuda.synchronize()
    if u < 3 or torch.cuda.max_memory_allocated() / G > 30:
      print("update %d: allocated before %.2f GB, after %.2f GB, peak inside %.2f GB, gc counts %s" % (
        u, before / G, torch.cuda.memory_allocated() / G, torch.cuda.max_memory_allocated() / G, gc.get_count()), flush=True)
