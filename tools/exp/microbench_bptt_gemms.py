"""Timing of the dense products of the update half at PP-hard size (R = 81920, H = 128): which formulation of the
weight-gradient product dgates^T x [inp | h] (K = R) the library runs fastest.  python tools/exp/microbench_bptt_gemms.py"""
import os
import sys
import time

import torch

R, H = int(sys.argv[1]) if len(sys.argv) > 1 else 81920, 128
dev = 'cuda'
dg = torch.randn(R, 4 * H, device=dev)
xh = torch.randn(R, 2 * H, device=dev)
w = torch.randn(2 * H, 4 * H, device=dev)
if os.environ.get('TUNE', '1') == '1':
    import torch.cuda.tunable as tunable
    tunable.enable(True)
    tunable.tuning_enable(True)
    tunable.set_filename('/tmp/ic3_mb_tunable_%d.csv' % os.getpid())


def bench(name, fn, flops, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print("%-44s %7.1f us  %6.1f TFLOP/s" % (name, dt * 1e6, flops / dt / 1e12))


fl = 2.0 * R * 4 * H * 2 * H
acc = torch.zeros(4 * H, 2 * H, device=dev)
acc_t = torch.zeros(2 * H, 4 * H, device=dev)
bench("dW  acc.addmm_(dg.t(), xh)", lambda: acc.addmm_(dg.t(), xh), fl)
bench("dW  acc_t.addmm_(xh.t(), dg)", lambda: acc_t.addmm_(xh.t(), dg), fl)
for B in (2, 4, 8, 16, 32):
    dgv, xhv = dg.view(B, R // B, 4 * H), xh.view(B, R // B, 2 * H)
    out = torch.empty(B, 4 * H, 2 * H, device=dev)
    bench("dW  bmm batch %d (+ sum)" % B, lambda: acc.add_(torch.bmm(dgv.transpose(1, 2), xhv, out=out).sum(0)), fl)
    out2 = torch.empty(B, 2 * H, 4 * H, device=dev)
    bench("dW  bmm^T batch %d (+ sum)" % B, lambda: acc_t.add_(torch.bmm(xhv.transpose(1, 2), dgv, out=out2).sum(0)), fl)
dxh = torch.empty(R, 2 * H, device=dev)
bench("dX  mm(dg, w.t())", lambda: torch.mm(dg, w.t(), out=dxh), fl)
gates = torch.empty(R, 4 * H, device=dev)
b = torch.randn(4 * H, device=dev)
bench("fwd addmm(b, xh, w)", lambda: torch.addmm(b, xh, w, out=gates), fl)
bench("bias dg.sum(0)", lambda: dg.sum(0), 0.0)
