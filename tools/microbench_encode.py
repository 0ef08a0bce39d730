"""Time ic3_env_encode with / without the per-position table, and the encode backward, in isolation (HIP events)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else 'pp_hard'
    E = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
    tr, a = bench.build_trainer(wl, E, 0, 0, 0)
    env = tr.env.env
    tr.begin_episode(0)
    for t in range(10):
        tr.step_episode(t)
    H = a.hid_size
    wt = tr.policy_net.encoder.weight.detach().t().contiguous()
    b = tr.policy_net.encoder.bias.detach()
    out = torch.empty((E * a.nagents, 2 * H), device='cuda')
    tab = env.encode_table(wt)
    g = torch.randn(E * a.nagents, H, device='cuda')
    print("%s E=%d  encode %.1f us | with table %.1f us | table build %.1f us | backward %.1f us" % (
        wl, E, timeit(lambda: env.encode(wt, b, out=out[:, :H])),
        timeit(lambda: env.encode(wt, b, out=out[:, :H], loc_table=tab)), timeit(lambda: env.encode_table(wt)),
        timeit(lambda: env.encode_backward(g))))


if __name__ == '__main__':
    main()
