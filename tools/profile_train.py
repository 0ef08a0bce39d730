"""Where does Trainer.train_batch spend its time?  (rollout with autograd | compute_grad incl. backward | optimizer)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

E = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
tr, a = bench.build_trainer('pp_hard', E, 0, 0, 0)
a.__dict__.update(gamma=1.0, normalize_rewards=False, entr=0, value_coeff=0.01, advantages_per_action=False,
                  batch_size=E * a.max_steps)
if os.environ.get('TUNE', '0') == '1':
    import torch.cuda.tunable as tunable
    tunable.enable(True); tunable.tuning_enable(True)
    tunable.set_filename(os.path.join(os.environ.get('TMPDIR', '/tmp'), 'ic3_tunableop_%d.csv' % os.getpid()))
    tunable.set_max_tuning_duration(30); tunable.set_max_tuning_iterations(20)
tr.train_batch(0)
if os.environ.get('TUNE', '0') == '1':
    torch.cuda.tunable.tuning_enable(False)
def sync(): torch.cuda.synchronize(); return time.perf_counter()
for it in range(2):
    a.rollout_grad = True
    t0 = sync(); batch, stat = tr.run_batch(0); t1 = sync()
    tr.optimizer.zero_grad(); s = tr.compute_grad(batch); t2 = sync()
    for p in tr.params:
        if p.grad is not None: p.grad /= stat['num_steps']
    tr.optimizer.step(); t3 = sync()
    a.rollout_grad = False
    print("E=%d  rollout(with grad) %.1f ms | compute_grad+backward %.1f ms | optimizer %.1f ms | steps %d" %
          (E, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, stat['num_steps']))
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA]) as prof:
    a.rollout_grad = True
    batch, stat = tr.run_batch(0); tr.optimizer.zero_grad(); tr.compute_grad(batch)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=14, max_name_column_width=60))
