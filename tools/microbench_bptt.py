"""The update half's hand-written launches alone, at a BASELINE shape (default PP-hard: E = 8192, N = 10, H = 128):
ic3_comm_backward per step, ic3_lstm_weight_grad per window of T steps, ic3_lstm_gates_backward_given (in place, heads folded in).
python tools/microbench_bptt.py [E] [N] [H] [T]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ic3net_amd import ops


def timeit(name, fn, n=20, flop=None, nbytes=None):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    extra = ""
    if flop:
        extra += "  %.1f TFLOP/s (%.3f of the fp32 matrix peak 157.3)" % (flop / ms / 1e9, flop / ms / 1e9 / 157.3)
    if nbytes:
        extra += "  %.2f TB/s" % (nbytes / ms / 1e9)
    print("%-58s %8.3f ms%s" % (name, ms, extra))
    return ms


def main():
    E = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    H = int(sys.argv[3]) if len(sys.argv) > 3 else 128
    T = int(sys.argv[4]) if len(sys.argv) > 4 else 40
    dev = 'cuda'
    R = E * N
    g = lambda *s: torch.randn(s, device=dev, dtype=torch.float32)
    dxh, hp, cw = g(R, 2 * H), g(R, H), g(H, H) / H ** 0.5
    alive = (torch.rand((E, N), device=dev) < 0.9).to(torch.int32)
    gate = (torch.rand((E, N), device=dev) < 0.7).to(torch.int32)
    dh = torch.empty((R, H), device=dev)
    parts = torch.zeros((ops.comm_backward_partials(E, N), H, H), device=dev)
    timeit("ic3_comm_backward (E %d, N %d, H %d)" % (E, N, H),
           lambda: ops.comm_backward(dxh, hp, alive, gate, cw, dh, parts, E, N), flop=2 * 2 * R * H * H, nbytes=4 * R * H * 4)
    scale = torch.rand(R, device=dev)
    timeit("  ... with out_scale", lambda: ops.comm_backward(dxh, hp, alive, gate, cw, dh, parts, E, N, out_scale=scale))
    timeit("  ... comm_zero (copy)", lambda: ops.comm_backward(dxh, None, None, None, None, dh, None, E, N, comm_zero=True))
    # the two masked-mean launches + two library products it replaces
    comm, dcomm = torch.empty((E, N, H), device=dev), torch.empty((R, H), device=dev)
    c_acc = torch.zeros((H, H), device=dev)

    def old():
        ops.comm_masked_mean_raw(hp.view(E, N, H), alive, gate, True, True, out=comm)
        c_acc.addmm_(dxh[:, :H].t(), comm.view(R, H))
        torch.mm(dxh[:, :H], cw, out=dcomm)
        ops.comm_masked_mean_raw(dcomm.view(E, N, H), alive, gate, True, True, out=dh.view(E, N, H), addend=dxh[:, H:])
    timeit("  (rounds 3-5: 2 masked means + 2 library products)", old)
    # weight gradient of a window
    Q = T * R
    xh, hs, dg = g(T, R, 2 * H), g(T, R, H), g(T, R, 4 * H)
    dW = torch.zeros((2 * H, 4 * H), device=dev)
    work = {}
    timeit("ic3_lstm_weight_grad, bf16 x 9 (T %d: Q = %d rows)" % (T, Q), lambda: ops.lstm_weight_grad(xh, hs, dg, dW, work=work), n=5,
           flop=2 * Q * 2 * H * 4 * H, nbytes=Q * (4 * H * 4 + 2 * H * 4))
    timeit("ic3_lstm_weight_grad, fp32 instruction", lambda: ops.lstm_weight_grad(xh, hs, dg, dW, work=work, split=False), n=5,
           flop=2 * Q * 2 * H * 4 * H, nbytes=Q * (4 * H * 4 + 2 * H * 4))
    wpart = torch.zeros((8, 2 * H, 4 * H), device=dev)
    xcat = torch.cat([xh[0][:, :H], hs[0]], 1).contiguous()

    def lib():
        for t in range(T):
            wpart.baddbmm_(xcat.view(8, R // 8, 2 * H).transpose(1, 2), dg[t].view(8, R // 8, 4 * H))
    if R % 8 == 0:
        timeit("  (rounds 3-5: a library product per step, x %d)" % T, lib, n=3, flop=2 * Q * 2 * H * 4 * H)
    # the heads' weight gradient over the window
    dW_h, db_h, d_all = torch.zeros((8, H), device=dev), torch.zeros(8, device=dev), g(T * R, 8)
    timeit("ic3_heads_grad (T %d, OT 8)" % T, lambda: ops.heads_grad(d_all, hs.view(T * R, H), dW_h, db_h, work), n=5,
           nbytes=Q * (H + 8) * 4)
    # the gate launch: in place, heads folded in
    from ic3net_amd import _lib  # noqa: F401
    w_ih, w_hh = g(4 * H, H) / H ** 0.5, g(4 * H, H) / H ** 0.5
    wb3 = ops.policy_pack_split_bwd(w_ih, w_hh)
    gates = torch.rand((4, R, 4 * H), device=dev)
    cprev, dhh, dc = g(R, H), g(R, H), g(R, H)
    bias_parts = torch.zeros(((R + 63) // 64, 4 * H), device=dev)
    dxo = torch.empty((R, 2 * H), device=dev)
    dhead, wh = g(R, 8), g(8, H)
    k = [0]

    def given(fold):
        k[0] += 1
        gt = gates[k[0] % 4]
        ops.lstm_gates_backward_given(gt, cprev, dhh, dc, gt, dc, bias_parts, True, lstm_wp3_bwd=wb3, dxh=dxo,
                                      dhead=dhead if fold else None, w_heads=wh if fold else None)
    timeit("ic3_lstm_gates_backward_given in place", lambda: given(False), nbytes=R * H * 4 * (4 + 3 + 4 + 1 + 2))
    timeit("  ... + the heads' share folded in", lambda: given(True))


if __name__ == '__main__':
    main()
