"""Where ic3_lstm_gates_backward_dx differs from the float64 product (debug aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ic3net_amd import ops

for H, R in ((64, 150), (64, 64), (64, 128), (64, 640), (128, 150), (128, 6400)):
    gen = torch.Generator(device='cuda').manual_seed(3 * H + R)
    rn = lambda *s: torch.randn(*s, device='cuda', generator=gen)
    w_ih, w_hh, c_w = rn(4 * H, H) / H ** 0.5, rn(4 * H, H) / H ** 0.5, rn(H, H)
    b = rn(4 * H)
    xh, h_prev = rn(R, 2 * H), rn(R, H)
    c_prev, dh, dc = rn(R, H), rn(R, H), rn(R, H)
    wp = ops.policy_step_pack(c_w, w_ih, w_hh)['ps_l_wp']
    wp3, wb3 = ops.policy_pack_split(w_ih, w_hh), ops.policy_pack_split_bwd(w_ih, w_hh)
    tiles = (R + 63) // 64
    res = []
    dgs = []
    for rep in range(8):
        dgates = torch.empty(R, 4 * H, device='cuda'); dcp = torch.empty(R, H, device='cuda')
        parts = torch.zeros(tiles, 4 * H, device='cuda'); dxh = torch.full((R, 2 * H), float('nan'), device='cuda')
        ops.lstm_gates_backward(xh.clone(), wp, b, c_prev, dh, dc, dgates, dcp, parts, True, h_prev=h_prev, lstm_wp3=wp3,
                                lstm_wp3_bwd=wb3, dxh=dxh)
        res.append(dxh); dgs.append((dgates, dcp, parts))
    W = torch.cat([w_ih, w_hh], 1).double()
    ref = dgates.double() @ W
    e = (res[0].double() - ref).abs()
    for rep in range(1, 8):
        if not torch.equal(res[0], res[rep]):
            d = (res[0] != res[rep])
            print("  rep %d differs: %d elements, rows %s cols %s, max diff %.3e; err of this rep %.3e" % (rep, int(d.sum()),
                  d.any(1).nonzero().flatten().tolist()[:24], d.any(0).nonzero().flatten().tolist()[:24],
                  float((res[0] - res[rep]).abs().max()), float((res[rep].double() - ref).abs().max())))
        for k in range(3):
            if not torch.equal(dgs[0][k], dgs[rep][k]):
                print("  rep %d: output %d of the cell backward differs" % (rep, k))
    print("H %d R %d: max err %.3e, same twice %s, nan %d" % (H, R, float(e.max()), torch.equal(res[0], res[1]), int(torch.isnan(res[0]).sum())))
    bad = e > 1e-5
    print("  bad elements %d of %d; rows with bad %s; cols with bad %s" % (
        int(bad.sum()), bad.numel(), bad.any(1).nonzero().flatten().tolist()[:40], bad.any(0).nonzero().flatten().tolist()[:140]))
    # which K range explains the difference?  compare with partial products
    for k0 in range(0, 4 * H, H // 2):
        part = dgates[:, k0:k0 + H // 2].double() @ W[k0:k0 + H // 2]
        d = (res[0].double() - (ref - part)).abs().max()
        d16 = (res[0].double() - (ref - part + (dgates[:, k0:k0 + H // 2].bfloat16().double() @ W[k0:k0 + H // 2].float().bfloat16().double()))).abs().max()
        print("    without K [%d, %d): %.3e   with it in bf16: %.3e" % (k0, k0 + H // 2, float(d), float(d16)))
