"""ic3_episode_finalize (masks + reduced statistics of get_episode, trainer.py:70-105,109-110) against the same
derivations written as tensor ops (Trainer._finalize_torch): bit-equal masks, equal fp64 sums."""
import numpy as np
import pytest
import torch

from ic3net_amd import ops
from ic3net_amd.trainer import Trainer

pytestmark = pytest.mark.gpu

CASES = [  # n, E, N, info (alive / is_completed present), gate: None | 'ones' | 'head', auto_reset, forced_last
    (20, 8192, 10, False, 'head', False, True),
    (20, 8192, 10, False, 'head', True, True),
    (40, 777, 20, True, 'head', False, True),
    (40, 777, 20, True, 'ones', True, True),
    (7, 37, 3, True, None, False, False),
    (1, 5, 1, False, 'head', False, False),
    (3, 1, 5, True, 'head', True, False),
    (80, 300, 20, True, 'head', False, True),
    (12, 513, 128, True, 'ones', False, True),
]


@pytest.mark.parametrize("n,E,N,info,gate,auto,forced", CASES)
def test_finalize_equals_tensor_ops(n, E, N, info, gate, auto, forced):
    dev = torch.device('cuda:0')
    g = torch.Generator(device='cpu').manual_seed(1000 * n + E + N)
    done = (torch.rand((n, E), generator=g) < 0.08).to(torch.int32).to(dev)
    done[0, ::7] = -1                                       # any non-zero value counts as done
    reward = (torch.randint(-40, 20, (n, E, N), generator=g).float() * 0.05).to(dev)
    alive = (torch.rand((n, E, N), generator=g) < 0.7).to(torch.int32).to(dev) if info else None
    comp = (torch.rand((n, E, N), generator=g) < 0.3).to(torch.int32).to(dev) if info else None
    gate_t = None
    if gate == 'head':                                      # the talk head inside a (n, heads, E, N) action buffer
        action = torch.randint(0, 2, (n, 2, E, N), generator=g, dtype=torch.int32).to(dev)
        gate_t = action[:, -1]
    work = dict()
    for rep in range(2):                                    # second call reuses the scratch / counter
        got = ops.episode_finalize(done, reward, alive, comp, gate_t, gate == 'ones', auto, forced, work=work)
        torch.cuda.synchronize()
        want = Trainer._finalize_torch(n, done, reward, alive, comp, gate_t, gate == 'ones', auto, forced)
        for k in ('live', 'alive_mask', 'episode_mask', 'episode_mini_mask', 'live_after'):
            assert got[k].shape == want[k].shape, k
            assert torch.equal(got[k], want[k]), k
        np.testing.assert_allclose(got['stats'].cpu().numpy(), want['stats'].cpu().numpy(), rtol=1e-13, atol=1e-9)
    assert int(work['counter'].item()) == 0


def test_finalize_is_reproducible():
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(5)
    n, E, N = 20, 4096, 10
    done = (torch.rand((n, E), generator=g) < 0.1).to(torch.int32).to(dev)
    reward = torch.randn((n, E, N), generator=g).to(dev)    # arbitrary floats: the summation order matters here
    a = ops.episode_finalize(done, reward, forced_last=True)['stats'].cpu()
    for _ in range(5):
        b = ops.episode_finalize(done, reward, forced_last=True)['stats'].cpu()
        assert torch.equal(a, b)


def test_finalize_rejects_cpu_tensors_and_bad_sizes():
    from ic3net_amd._lib import IC3Error
    with pytest.raises(IC3Error):
        ops.episode_finalize(torch.zeros((2, 3), dtype=torch.int32), torch.zeros((2, 3, 4)))
    dev = torch.device('cuda:0')
    with pytest.raises(ValueError):
        ops.episode_finalize(torch.zeros((1, 2), dtype=torch.int32, device=dev), torch.zeros((1, 2, 300), device=dev))
