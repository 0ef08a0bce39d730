"""Worker of tests/test_rccl_world1_gpu.py: a ONE-rank RCCL process group on the one GPU of the box, and the update-time
exchange of ic3net_amd.sharding on device tensors through it (/root/reference/multi_processing.py:74-98)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ic3net_amd import sharding  # noqa: E402

torch.cuda.set_device(0)
dist.init_process_group(backend='nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
assert dist.get_backend() == 'nccl'
dist.barrier(device_ids=[0])

# gradients: summed over ranks (one), divided by the global num_steps (multi_processing.py:90-97)
lin = torch.nn.Linear(7, 5).cuda()
unused = torch.nn.Parameter(torch.zeros(3, device='cuda'))          # no grad: skipped like `p._grad is not None`
lin(torch.ones(2, 7, device='cuda')).sum().backward()
want = [p.grad.clone() / 40.0 for p in lin.parameters()]
sharding.allreduce_grads(list(lin.parameters()) + [unused], 40.0)
for p, w in zip(lin.parameters(), want):
    assert p.grad.is_cuda and torch.allclose(p.grad, w)
assert unused.grad is None

# stats: numeric and ndarray entries summed (multi_processing.py:86-88), others passed through
st = sharding.allreduce_stats({'num_steps': 80.0, 'reward': np.arange(3.0), 'note': 'x', 'success': 2})
assert st['num_steps'] == 80.0 and st['success'] == 2.0 and st['note'] == 'x' and np.array_equal(st['reward'], np.arange(3.0))

# one seed / one parameter set for all replicas (main.py:157-159,177-178)
assert sharding.broadcast_seed(1234) == 1234
opt = torch.optim.RMSprop(lin.parameters(), lr=0.001, alpha=0.97, eps=1e-6)
opt.step()
before = [p.detach().clone() for p in lin.parameters()]
sharding.broadcast_parameters(lin, opt)
for p, b in zip(lin.parameters(), before):
    assert torch.equal(p.detach(), b)

# the MAX / SUM reductions bench.py uses for its timing, on device tensors
t = torch.tensor([3.5], dtype=torch.float64, device='cuda')
dist.all_reduce(t, op=dist.ReduceOp.MAX)
g = [torch.zeros_like(t)]
dist.all_gather(g, t)
assert float(t.item()) == 3.5 and float(g[0].item()) == 3.5
dist.barrier(device_ids=[0])
dist.destroy_process_group()
print("RCCL_WORLD1_OK")
