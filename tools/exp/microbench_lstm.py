"""Times the fused LSTM kernel vs hipBLASLt GEMM + lstm_cell on R = 81920 rows, H = 128."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ic3net_amd import ops

def timeit(fn, iters=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3

R, H = int(os.environ.get('R', 81920)), 128
torch.manual_seed(0)
cell = torch.nn.LSTMCell(H, H).cuda()
xh = torch.randn(R, 2 * H, device='cuda') * 0.5
c = torch.randn(R, H, device='cuda')
wp = ops.lstm_pack_weights(cell.weight_ih, cell.weight_hh)
b = (cell.bias_ih + cell.bias_hh).detach().contiguous()
wcat_t = torch.cat([cell.weight_ih, cell.weight_hh], 1).t().contiguous().detach()
gates = torch.empty(R, 4 * H, device='cuda')
flops = 2.0 * R * 2 * H * 4 * H
with torch.no_grad():
    t = timeit(lambda: ops.lstm_fused_(xh, wp, b, c))
    print("lstm_fused (IC3_LSTM_DBG=%s): %.1f us  %.1f TFLOP/s" % (os.environ.get('IC3_LSTM_DBG', '0'), t, flops / t / 1e6))
    t1 = timeit(lambda: torch.addmm(b, xh, wcat_t, out=gates))
    t2 = timeit(lambda: ops.lstm_cell_(gates, c, xh[:, H:]))
    print("hipBLASLt gemm %.1f us (%.1f TFLOP/s) + lstm_cell %.1f us = %.1f us" % (t1, flops / t1 / 1e6, t2, t1 + t2))
