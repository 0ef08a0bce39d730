import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import torch
import bench

def run():
    E, T, seed = 8192, 6, 11
    tr, a = bench.build_trainer("tj_medium", E, seed, 0, 0, add_rate_min=0.5, add_rate_max=0.5)
    a.max_steps = T
    tr.begin_episode(0)
    N, H = a.nagents, a.hid_size
    out = []
    for t in range(T):
        xh = torch.zeros((E * N, 2 * H), device='cuda'); g = torch.empty((E * N, 4 * H), device='cuda')
        tr.env.env.set_record_out(g, xh)
        tr.step_episode(t)
        h, c = tr._prev_hid
        out.append((h.clone(), xh[:, :H].clone(), xh[:, H:].clone()))
    return out

gold = run()
for it in range(int(os.environ.get('ITERS', '60'))):
    cur = run()
    for t, ((h0, x0, d0), (h1, x1, d1)) in enumerate(zip(gold, cur)):
        for name, u, v in (('inp', x0, x1), ('h', h0, h1)):
            if not torch.equal(u, v):
                d = (u != v)
                rows = d.any(1).nonzero().flatten().tolist()
                cols = d.any(0).nonzero().flatten().tolist()
                print(it, 't', t, name, 'differs: rows', rows[:20], '(n=%d)' % len(rows), 'cols', cols[:6], '...', cols[-3:], '(n=%d)' % len(cols),
                      'max', float((u - v).abs().max()))
                for r in sorted(set(rr // 10 * 10 for rr in rows))[:3]:
                    names = ['S', 'scl', 'sm'] + ['h row %d as read' % i for i in range(3, 10)]
                    for k, nm in enumerate(names):
                        a_, b_ = d0[r + k], d1[r + k]
                        if not torch.equal(a_, b_):
                            dc = (a_ != b_).nonzero().flatten().tolist()
                            print('         differing columns:', dc, ' gold', [round(float(a_[c]), 5) for c in dc[:8]], ' cur', [round(float(b_[c]), 5) for c in dc[:8]])
                        print('      env %d dump %-16s equal=%s  max diff %.3e   gold[:3] %s cur[:3] %s' % (r // 10, nm, bool(torch.equal(a_, b_)), float((a_ - b_).abs().max()), [round(x, 5) for x in a_[:3].tolist()], [round(x, 5) for x in b_[:3].tolist()]))
                break
        else:
            continue
        break
print('done')
