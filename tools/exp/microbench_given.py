"""ic3_lstm_gates_backward_given at the PP-hard update's shape (R = 81920, H = 128): the whole launch, without the dx product,
without the h_prev copy — what each part costs.  python tools/exp/microbench_given.py [R] [H]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ic3net_amd import ops

R = int(sys.argv[1]) if len(sys.argv) > 1 else 81920
H = int(sys.argv[2]) if len(sys.argv) > 2 else 128
dev = 'cuda'
g = torch.Generator(device=dev).manual_seed(1)
rn = lambda *s: torch.randn(*s, device=dev, generator=g)
w_ih, w_hh = rn(4 * H, H) / H ** 0.5, rn(4 * H, H) / H ** 0.5
wb3 = ops.policy_pack_split_bwd(w_ih, w_hh)
NS = 6                                   # rotate over several step records like the update does (no L2 / MALL reuse)
gates = [torch.sigmoid(rn(R, 4 * H)) for _ in range(NS)]
xh = [rn(R, 2 * H) for _ in range(NS)]
hp = [rn(R, H) for _ in range(NS)]
cp = [rn(R, H) for _ in range(NS)]
dh, dc = rn(R, H), rn(R, H)
dgates, dxh = torch.empty(R, 4 * H, device=dev), torch.empty(R, 2 * H, device=dev)
parts = torch.zeros((R + 63) // 64, 4 * H, device=dev)


def run(name, fn, n=30):
    for i in range(5):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    print("%-46s %8.1f us" % (name, e0.elapsed_time(e1) / n * 1e3))


run("given: copy + cell + dx", lambda i: ops.lstm_gates_backward_given(gates[i % NS], cp[i % NS], dh, dc, dgates, dc, parts, True,
                                                                         xh=xh[i % NS], h_prev=hp[i % NS], lstm_wp3_bwd=wb3, dxh=dxh))
run("given: cell + dx (no h_prev copy)", lambda i: ops.lstm_gates_backward_given(gates[i % NS], cp[i % NS], dh, dc, dgates, dc, parts, True,
                                                                                   lstm_wp3_bwd=wb3, dxh=dxh))
run("given: copy + cell (no dx)", lambda i: ops.lstm_gates_backward_given(gates[i % NS], cp[i % NS], dh, dc, dgates, dc, parts, True,
                                                                            xh=xh[i % NS], h_prev=hp[i % NS]))
run("given: cell only", lambda i: ops.lstm_gates_backward_given(gates[i % NS], cp[i % NS], dh, dc, dgates, dc, parts, True))
run("copy of R x 4H floats (torch)", lambda i: dgates.copy_(gates[i % NS]))
