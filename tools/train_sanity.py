"""End-to-end sanity: IC3Net on Predator-Prey easy with the batched Trainer.train_batch — reward / success must
improve.  One update = nenvs x max_steps env-steps (400 x 20 = the reference's 16 processes x batch_size 500)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench


def main():
    updates = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    mode = sys.argv[2] if len(sys.argv) > 2 else 'native'        # native: ic3net_amd.bptt (default path) | native_recompute | autograd
    bench.WORKLOADS['pp_easy_train'] = ('predator_prey', dict(nagents=3, dim=5, vision=0, max_steps=20, hid_size=128,
                                                            ic3net=True, recurrent=True, detach_gap=10, mode='mixed'))
    # TJ-easy (README's Traffic-Junction easy command: 5 cars, 6 x 6, vision 0, add rate 0.3, no curriculum, 20 steps)
    bench.WORKLOADS['tj_easy_train'] = ('traffic_junction', dict(nagents=5, dim=6, vision=0, max_steps=20, hid_size=128,
                                                                 ic3net=True, recurrent=True, detach_gap=10, difficulty='easy',
                                                                 add_rate_min=0.3, add_rate_max=0.3))
    wl = sys.argv[3] if len(sys.argv) > 3 else 'pp_easy_train'
    tr, a = bench.build_trainer(wl, 400, 1, 0, 0)
    a.native_update = mode != 'autograd'
    a.record_gates = mode != 'native_recompute'   # (native: the rollout records its gates / inp rows for the backward, round 5)
    print("update path: %s (native supported: %s)" % (mode, tr._native_update()), flush=True)
    a.__dict__.update(gamma=1.0, normalize_rewards=False, entr=0, value_coeff=0.01, advantages_per_action=False,
                      batch_size=400 * 20, lrate=0.001)
    tr.optimizer = torch.optim.RMSprop(tr.policy_net.parameters(), lr=a.lrate, alpha=0.97, eps=1e-6)
    t0 = time.time()
    hist = []
    for u in range(updates):
        st = tr.train_batch(u // 10)
        hist.append((st['reward'].sum() / st['num_episodes'], st['success'] / st['num_episodes'],
                     st['steps_taken'] / st['num_episodes']))
        if (u + 1) % 25 == 0:
            r, s, k = np.mean(hist[-25:], axis=0)
            print("update %4d  reward/episode %7.3f  success %.3f  steps %.2f  (%.1f s)" % (u + 1, r, s, k, time.time() - t0),
                  flush=True)
    first, last = np.mean(hist[:25], axis=0), np.mean(hist[-25:], axis=0)
    print("first25", first, "last25", last)
    assert last[0] > first[0] and (last[1] > first[1] or wl.startswith('tj')), "no learning progress"   # (TJ: reward rises;
    # success — an episode without a collision — starts high at this add rate)
    print("LEARNING OK")


if __name__ == '__main__':
    main()
