export TMPDIR=/tmp
python -m pytest tests/test_policy_gpu.py tests/test_auto_reset_gpu.py tests/test_trainer_gpu.py tests/test_policy_step_gpu.py -m gpu -x -q -k "p2 or p3share or passes or auto_reset_stream or graph_replay_equals or multipass or comm_passes" 2>&1 | tail -4
python - <<'PY'
import time, torch, bench
for wl, over in (("pp_hard", dict(comm_passes=2)), ("tj_hard", dict(comm_passes=2)), ("pp_hard", dict(comm_passes=3))):
    for inl in (True, False):
        tr, a = bench.build_trainer(wl, 8192, 0, 0, 0, **over)
        a.passes_in_launch = inl
        T = a.max_steps
        for ep in range(2): tr.get_episode(ep)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for ep in range(3): tr.get_episode(ep)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / (3 * T)
        print("%s comm_passes=%d %s: %.4f ms/step  %.1f M agent-steps/s" % (wl, over['comm_passes'], "ONE launch" if inl else "one launch per pass", dt * 1e3, a.nagents * 8192 / dt / 1e6))
PY
