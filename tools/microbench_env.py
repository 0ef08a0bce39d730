"""Env-only micro-benchmark: times the obs-assembly / step kernels with HIP events on the launch stream."""
import argparse
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ic3net_amd.envs import PredatorPreyEnv, TrafficJunctionEnv
from ic3net_amd import _lib


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    p = argparse.ArgumentParser()
    p.add_argument('--E', type=int, default=8192)
    a = p.parse_args()
    E = a.E
    for name, mk in (("pp_hard", lambda: pp(10, 20, 1, E)), ("pp_scaled_E1024", lambda: pp(32, 40, 2, 1024)),
                     ("tj_medium_v1", lambda: tj(10, 14, 1, 'medium', E)), ("tj_hard_v1", lambda: tj(20, 18, 1, 'hard', E)),
                     ("tj_hard_v0", lambda: tj(20, 18, 0, 'hard', E))):
        env = mk()
        env.reset() if name.startswith('pp') else env.reset(0)
        En, N = env.nenvs, env.nagents_env
        act = torch.zeros((En, N), dtype=torch.int32, device='cuda')
        lib = _lib.lib()

        def rnd():
            lib.ic3_random_actions(_lib.ptr(act), env.dims.naction, 1, 0, 0, 0, En, N, _lib.stream())
        rnd()
        t_obs = timeit(lambda: env.observe())
        bytes_obs = En * N * env.obs_dim * 4
        t_step = timeit(lambda: lib.ic3_env_step(env._h, _lib.ptr(act), None, _lib.ptr(env._reward), _lib.ptr(env._done),
                                                 _lib.ptr(env._alive), _lib.ptr(env._completed), _lib.stream()))
        t_ms = timeit(lambda: env._obs.zero_())
        print("%-16s E=%d obs %.3f ms  %.1f GB/s (%.1f%% of 8TB/s) | memset %.3f ms %.1f GB/s | step %.4f ms" %
              (name, En, t_obs, bytes_obs / t_obs / 1e6, bytes_obs / t_obs / 1e6 / 80, t_ms, bytes_obs / t_ms / 1e6, t_step))
        del env


def pp(N, dim, v, E):
    env = PredatorPreyEnv()
    env.multi_agent_init(argparse.Namespace(nfriendly=N, nenemies=1, dim=dim, vision=v, moving_prey=False, mode='mixed',
                                            enemy_comm=False, no_stay=False, nenvs=E, seed=0, env_id_offset=0))
    return env


def tj(N, dim, v, diff, E):
    env = TrafficJunctionEnv()
    env.multi_agent_init(argparse.Namespace(nagents=N, dim=dim, vision=v, difficulty=diff, vocab_type='bool',
                                            add_rate_min=0.3, add_rate_max=0.3, curr_start=0, curr_end=0, nenvs=E,
                                            seed=0, env_id_offset=0))
    return env


if __name__ == '__main__':
    main()
