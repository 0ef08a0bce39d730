"""Time of ic3_lstm_gates_backward at PP-hard size (R = 81920, H = 128): python tools/exp/microbench_gates_bwd.py [R] [H]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from ic3net_amd import ops  # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 81920
H = int(sys.argv[2]) if len(sys.argv) > 2 else 128
rn = lambda *s: torch.randn(*s, device='cuda')
w_ih, w_hh, c_w, b = rn(4 * H, H) / H ** 0.5, rn(4 * H, H) / H ** 0.5, rn(H, H), rn(4 * H)
xh, h_prev, c_prev, dh, dc = rn(R, 2 * H), rn(R, H), rn(R, H), rn(R, H), rn(R, H)
wp = ops.policy_step_pack(c_w, w_ih, w_hh)['ps_l_wp']
dgates, parts = torch.empty(R, 4 * H, device='cuda'), torch.zeros((R + 63) // 64, 4 * H, device='cuda')
wp3 = ops.policy_pack_split(w_ih, w_hh) if os.environ.get('SPLIT', '0') == '1' else None    # EXPERIMENT gate_split
run = lambda: ops.lstm_gates_backward(xh, wp, b, c_prev, dh, dc, dgates, dc, parts, True, h_prev=h_prev, lstm_wp3=wp3)
for _ in range(5):
    run()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
torch.cuda.synchronize()
n = 100
ev[0].record()
for _ in range(n):
    run()
ev[1].record()
torch.cuda.synchronize()
us = ev[0].elapsed_time(ev[1]) * 1000 / n
print("lstm_gates_backward R=%d H=%d: %.1f us per call = %.1f TFLOP/s (%s)" % (
    R, H, us, 2.0 * R * 2 * H * 4 * H / us / 1e6, "gate_split" if wp3 is not None else os.path.basename(os.environ.get('IC3_ROLLOUT_LIB', 'default'))))
