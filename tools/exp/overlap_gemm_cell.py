"""Timing-only experiment: gates GEMM split into two column halves, cell(half 1) on a side stream while GEMM(half 2)
runs.  (Results of the overlapped variant are NOT numerically meaningful here: it ignores the h' write hazard.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ic3net_amd import ops

R, H = 81920, 128
torch.manual_seed(0)
xh = torch.randn(R, 2 * H, device='cuda') * 0.5
c = torch.randn(R, H, device='cuda')
w = torch.randn(2 * H, 4 * H, device='cuda') * 0.05
b = torch.randn(4 * H, device='cuda') * 0.1
gates = torch.empty(R, 4 * H, device='cuda')
wh = [w[:, :2 * H].contiguous(), w[:, 2 * H:].contiguous()]
bh = [b[:2 * H].contiguous(), b[2 * H:].contiguous()]
gh = [torch.empty(R, 2 * H, device='cuda') for _ in range(2)]
ch = [torch.randn(R, H // 2, device='cuda') for _ in range(2)]
hh = [torch.empty(R, H // 2, device='cuda') for _ in range(2)]
side = torch.cuda.Stream()

def seq():
    torch.addmm(b, xh, w, out=gates)
    ops.lstm_cell_(gates, c, xh[:, H:])

def split_seq():
    for p in range(2):
        torch.addmm(bh[p], xh, wh[p], out=gh[p])
        ops.lstm_cell_(gh[p], ch[p], hh[p])

def split_overlap():
    main = torch.cuda.current_stream()
    torch.addmm(bh[0], xh, wh[0], out=gh[0])
    side.wait_stream(main)
    with torch.cuda.stream(side):
        ops.lstm_cell_(gh[0], ch[0], hh[0])
    torch.addmm(bh[1], xh, wh[1], out=gh[1])
    ops.lstm_cell_(gh[1], ch[1], hh[1])
    main.wait_stream(side)

def timeit(fn, iters=40, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3

with torch.no_grad():
    for name, fn in (("gemm+cell", seq), ("2 halves sequential", split_seq), ("2 halves, cell1 || gemm2", split_overlap),
                     ("gemm+cell", seq), ("2 halves, cell1 || gemm2", split_overlap)):
        print("%-28s %.1f us" % (name, timeit(fn)))
