"""Throughput of the update path Trainer.train_batch (rollout + compute_grad + RMSprop), PP-hard by default:
python tools/bench_train.py [nenvs] [updates] [native|autograd] [workload] [dense_obs 0|1] [gate_split 0|1] [collection 0|1]   (default native: no-grad
one-launch rollout + the explicit backward through time of ic3net_amd.bptt; autograd: the rollout keeps the autograd
graph, rounds 1-2; dense_obs 0: the training rollout does not assemble observation rows nothing reads)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench


def main():
    E = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    updates = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    mode = sys.argv[3] if len(sys.argv) > 3 else 'native'
    wl = sys.argv[4] if len(sys.argv) > 4 else 'pp_hard'
    dense = int(sys.argv[5]) if len(sys.argv) > 5 else 1
    split = int(sys.argv[6]) if len(sys.argv) > 6 else 1      # gate product: 1 exact bf16 split products (default), 0 fp32 MFMA
    coll = int(sys.argv[7]) if len(sys.argv) > 7 else 0       # 1: collection mode (args.auto_reset): E streams of whole episodes
    tr, a = bench.build_trainer(wl, E, 0, 0, 0)
    a.train_dense_obs = bool(dense) and len(sys.argv) > 5     # (default since round 5: the training rollout assembles no obs rows)
    a.gate_split = bool(split)
    a.auto_reset = bool(coll)
    a.native_update = mode == 'native'
    a.bptt_two_chains = os.environ.get('TWO_CHAINS', '1') == '1'     # A/B: the backward's launches as two concurrent chains of envs
    a.fused_loss = os.environ.get('FUSED_LOSS', '1') == '1'           # A/B: compute_grad's losses + their gradients as one launch
    a.enc_window = os.environ.get('ENC_WINDOW', '1') == '1'           # A/B: the encoder backward's stage 1 once per window
    a.__dict__.update(gamma=1.0, normalize_rewards=False, entr=0, value_coeff=0.01, advantages_per_action=False,
                      batch_size=E * a.max_steps)
    tune = os.environ.get('TUNE', '1') == '1'
    if tune:                                  # TunableOp picks the GEMM solutions during the first (untimed) update
        import torch.cuda.tunable as tunable
        tunable.enable(True)
        tunable.tuning_enable(True)
        tunable.set_filename(os.path.join(os.environ.get('TMPDIR', '/tmp'), 'ic3_tunableop_%d.csv' % os.getpid()))
    tr.train_batch(0)
    if tune:
        torch.cuda.tunable.tuning_enable(False)
    import gc
    gc.collect()
    gc.freeze()
    torch.cuda.reset_peak_memory_stats()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    steps = 0
    for u in range(updates):
        st = tr.train_batch(u)
        steps += st['num_steps']
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    label = wl + (" (obs rows assembled in the rollout)" if a.train_dense_obs else "") + ("" if split else " (fp32 MFMA gate product)") + \
        (" (collection mode: %d whole episodes)" % int(st['num_episodes']) if coll else "")
    print("train_batch [%s] %s E=%d: %.0f env-steps/s = %.2f M agent-steps/s (%.2f s per update of %d env-steps), "
          "peak mem %.1f GB, gemm %s" % (mode, label, E, steps / dt, a.nagents * steps / dt / 1e6, dt / updates,
                                         steps // updates, torch.cuda.max_memory_allocated() / 2 ** 30,
                                         'TunableOp' if tune else 'default'))


if __name__ == '__main__':
    main()
