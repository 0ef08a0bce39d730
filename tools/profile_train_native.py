"""Where one native update (Trainer.train_batch, ic3net_amd.bptt) spends its GPU time: kernel table of one update after
warm-up.  python tools/profile_train_native.py [nenvs] [workload]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

E = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
WL = sys.argv[2] if len(sys.argv) > 2 else 'pp_hard'
tr, a = bench.build_trainer(WL, E, 0, 0, 0)
a.__dict__.update(gamma=1.0, normalize_rewards=False, entr=0, value_coeff=0.01, advantages_per_action=False,
                  batch_size=E * a.max_steps)
if os.environ.get('TUNE', '1') == '1':
    import torch.cuda.tunable as tunable
    tunable.enable(True)
    tunable.tuning_enable(True)
    tunable.set_filename(os.path.join(os.environ.get('TMPDIR', '/tmp'), 'ic3_tunableop_%d.csv' % os.getpid()))
tr.train_batch(0)
if os.environ.get('TUNE', '1') == '1':
    torch.cuda.tunable.tuning_enable(False)
tr.train_batch(1)
torch.cuda.synchronize()
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA]) as prof:
    tr.train_batch(2)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=28, max_name_column_width=70))
