"""What collection mode (args.auto_reset through Trainer.train_batch; trainer.py:227-242) buys and what it throws away, on
workloads whose episodes DO end early (round-5 verdict item 3).

Lock-step: one batched episode per update — an env that ends early idles (frozen) until the lock-step reset; its idle slots
are played by the launches but carry no transition.  Collection: E streams of consecutive whole episodes over W windows of
max_steps slots — a finished env restarts inside the step launch; what is still running when the batch closes is DISCARDED
(the reference never sees part of an episode).  Reported for both, same policy: live fraction (transitions that count / slots
played), for collection the discarded-tail fraction, and agent-steps/s of LIVE transitions through a whole train_batch.

python tools/bench_collection.py [workload] [nenvs] [pretrain updates] [windows per collection batch]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

bench.WORKLOADS['pp_easy_train'] = ('predator_prey', dict(nagents=3, dim=5, vision=0, max_steps=20, hid_size=128, ic3net=True,
                                                        recurrent=True, detach_gap=10, mode='mixed'))
bench.WORKLOADS['pp_small'] = ('predator_prey', dict(nagents=3, dim=3, vision=1, max_steps=20, hid_size=128, ic3net=True,
                                                   recurrent=True, detach_gap=10, mode='mixed'))
bench.WORKLOADS['pp_medium_train'] = ('predator_prey', dict(nagents=5, dim=10, vision=1, max_steps=40, hid_size=128, ic3net=True,
                                                          recurrent=True, detach_gap=10, mode='mixed'))


def measure(tr, a, collect, windows, updates=6):
    E, T = a.nenvs, a.max_steps
    a.auto_reset = bool(collect)
    a.batch_size = E * T * (windows if collect else 1)
    tr.train_batch(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    live = slots = eps = 0.0
    for u in range(updates):
        st = tr.train_batch(u)
        live += st['num_steps']
        eps += st['num_episodes']
        # (lock-step: run_batch plays batched episodes until the batch holds batch_size LIVE steps — two of them as soon as
        #  one env ends early; every batched episode is E x T slots of launches)
        slots += E * T * windows if collect else st['num_episodes'] * T
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return dict(live_frac=live / slots, agent_steps_per_s=a.nagents * live / dt, slots_per_s=a.nagents * slots / dt,
                steps_per_episode=live / max(eps, 1), ms_per_update=dt / updates * 1e3)


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else 'pp_easy_train'
    E = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
    pre = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    W = int(sys.argv[4]) if len(sys.argv) > 4 else 4
    tr, a = bench.build_trainer(wl, E, 1, 0, 0)
    a.__dict__.update(gamma=1.0, normalize_rewards=False, entr=0, value_coeff=0.01, advantages_per_action=False, lrate=0.001,
                      batch_size=E * a.max_steps)
    a.bptt_two_chains = os.environ.get('TWO_CHAINS', '1') == '1'
    a.enc_window = os.environ.get('ENC_WINDOW', '1') == '1'
    tr.optimizer = torch.optim.RMSprop(tr.policy_net.parameters(), lr=a.lrate, alpha=0.97, eps=1e-6)
    for u in range(pre):                                   # (lock-step updates: the policy learns to end its episodes early)
        st = tr.train_batch(u // 10)
    tag = "%s E=%d, %s policy" % (wl, E, ("after %d lock-step updates" % pre) if pre else "random-init")
    lock = measure(tr, a, False, 1)
    coll = measure(tr, a, True, W)
    print("%s | lock-step : live %.3f of the slots, %.1f steps per episode, %.1f M live agent-steps/s (%.1f M slots/s), %.1f ms per update"
          % (tag, lock['live_frac'], lock['steps_per_episode'], lock['agent_steps_per_s'] / 1e6, lock['slots_per_s'] / 1e6, lock['ms_per_update']))
    print("%s | collection, %d windows per batch: live %.3f of the slots (discarded tails %.3f), %.1f steps per episode, %.1f M live "
          "agent-steps/s (%.1f M slots/s), %.1f ms per update -> %.2f x lock-step"
          % (tag, W, coll['live_frac'], 1.0 - coll['live_frac'], coll['steps_per_episode'], coll['agent_steps_per_s'] / 1e6,
             coll['slots_per_s'] / 1e6, coll['ms_per_update'], coll['agent_steps_per_s'] / lock['agent_steps_per_s']))


if __name__ == '__main__':
    main()
