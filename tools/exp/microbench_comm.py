"""comm block in isolation: comm_masked_mean + library GEMM (inp += comm @ C^T) vs ic3_comm_fused (HIP events)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ic3net_amd import ops


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
    E = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    H = int(sys.argv[3]) if len(sys.argv) > 3 else 128
    R = E * N
    xh = torch.randn(R, 2 * H, device='cuda') * 0.1
    Cw = torch.randn(H, H, device='cuda') * 0.05
    c_wt = Cw.t().contiguous()
    wp = ops.comm_pack_weights(Cw)
    gate = (torch.rand(E, N, device='cuda') < 0.5).int()
    comm = torch.empty(E, N, H, device='cuda')

    def old():
        ops.comm_masked_mean_raw(xh.view(E, N, 2 * H)[:, :, H:], None, gate, True, True, out=comm)
        xh[:, :H].addmm_(comm.view(R, H), c_wt)

    def only_comm():
        ops.comm_masked_mean_raw(xh.view(E, N, 2 * H)[:, :, H:], None, gate, True, True, out=comm)

    print("E=%d N=%d H=%d: comm %.1f us | comm + GEMM %.1f us | fused %.1f us" % (
        E, N, H, timeit(only_comm), timeit(old), timeit(lambda: ops.comm_fused_(xh, wp, None, gate, E, N, True))))


if __name__ == '__main__':
    main()
