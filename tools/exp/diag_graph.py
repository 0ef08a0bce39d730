import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench

def run(mode, workload='pp_hard'):
    tr, a = bench.build_trainer(workload, 48, 7, 200, 0)
    a.hip_graph = mode == 'graph'
    a.mega_policy = mode != 'chain'
    a.fused_obs = os.environ.get('FUSED', '1') == '1'
    out = []
    for ep in range(3):
        e, s = tr.get_episode(ep)
        out.append(dict(value=[t.value.clone() for t in e], logp=[t.action_out[0].clone() for t in e],
                        act=[t.action.clone() for t in e], rew=[t.reward.clone() for t in e]))
    return out

runs = {m: run(m) for m in ('eager', 'graph', 'chain')}
for ep in range(3):
    for t in (0, 1, 2, 40):
        def d(a, b, k): return float((runs[a][ep][k][t].float() - runs[b][ep][k][t].float()).abs().max())
        print("ep %d t %2d | value e-g %.2e e-c %.2e g-c %.2e | logp e-g %.2e e-c %.2e g-c %.2e | act e==g %s e==c %s" % (
            ep, t, d('eager', 'graph', 'value'), d('eager', 'chain', 'value'), d('graph', 'chain', 'value'),
            d('eager', 'graph', 'logp'), d('eager', 'chain', 'logp'), d('graph', 'chain', 'logp'),
            bool(torch.equal(runs['eager'][ep]['act'][t], runs['graph'][ep]['act'][t])),
            bool(torch.equal(runs['eager'][ep]['act'][t], runs['chain'][ep]['act'][t]))))
