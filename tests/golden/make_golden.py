#!/usr/bin/env python
"""Generates tests/golden/*.npz by running the REFERENCE itself (/root/reference) in the build
container with the injected Philox stream (see ref_harness.py).  Run:

    PYTHONDONTWRITEBYTECODE=1 OMP_NUM_THREADS=1 python tests/golden/make_golden.py

The fixtures are data only (inputs + the reference's outputs); nothing of the reference's source is
stored.  /root/reference does not exist on the GPU box — tests read only the .npz files.
"""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.environ.get('IC3_GOLDEN_OUT', HERE)   # where the fixtures are written (tests/test_golden_recipes_cpu.py: a tmp dir)
sys.path.insert(0, HERE)

import zlib

import numpy as np  # noqa: E402
import torch  # noqa: E402

import ref_harness as rh  # noqa: E402
from oracle import philox  # noqa: E402

SEED = 1234


def coo(rows, e, ep, t, obs):
    """append sparse entries of a flattened (N, obs_dim) observation"""
    a, i = np.nonzero(obs)
    for aa, ii in zip(a, i):
        rows.append((e, ep, t, aa, ii, obs[aa, ii]))


# ------------------------------------------------------------------------------------------------
# F1: Predator-Prey trajectories
# ------------------------------------------------------------------------------------------------
def pp_fixture(name, N, dim, vision, mode, T, nenv=3, nep=2, greedy_env=None, no_stay=False, enemy_comm=False):
    """N = number of predators (args.nfriendly).  enemy_comm: the policy/env see N+1 agents (main.py:125-130)."""
    ref = rh.load_reference()
    a = rh.make_args('predator_prey', nagents=N, dim=dim, vision=vision, mode=mode, max_steps=T,
                     no_stay=no_stay, enemy_comm=enemy_comm)
    if enemy_comm:
        a.nagents += a.nenemies
    env = rh.make_env('predator_prey', a)
    R = N + (1 if enemy_comm else 0)
    raw = env.env
    rs = np.random.RandomState(zlib.crc32(name.encode()) & 0xffff)
    obs_rows = []
    init_loc = np.zeros((nenv, nep, N + 1, 2), np.int32)
    actions = np.zeros((nenv, nep, T, R), np.int32)
    loc = np.zeros((nenv, nep, T, N + 1, 2), np.int32)
    reached = np.zeros((nenv, nep, T, N), np.int32)
    reward = np.zeros((nenv, nep, T, R), np.float64)
    done = np.zeros((nenv, nep, T), np.int32)
    success = np.full((nenv, nep, T), -1, np.int32)
    nsteps = np.zeros((nenv, nep), np.int32)
    ndraws = np.zeros((nenv, nep), np.int32)
    for e in range(nenv):
        st = philox.Stream(SEED, 100 + e)
        for ep in range(nep):
            ref['rnd'].begin(st, philox.DOMAIN_PP_RESET, ep, 0)
            d0 = st.ndraws
            o = env.reset(0)
            ndraws[e, ep] = st.ndraws - d0
            init_loc[e, ep, :N] = raw.predator_loc
            init_loc[e, ep, N:] = raw.prey_loc
            coo(obs_rows, e, ep, 0, o[0].numpy())
            for t in range(T):
                if greedy_env is not None and e in greedy_env:
                    # steer towards the prey so that freezing / early termination is exercised
                    act = np.full(R, 4, np.int64)
                    for i in range(N):
                        dr = raw.prey_loc[0][0] - raw.predator_loc[i][0]
                        dc = raw.prey_loc[0][1] - raw.predator_loc[i][1]
                        if dr != 0 and (rs.rand() < 0.5 or dc == 0):
                            act[i] = 2 if dr > 0 else 0
                        elif dc != 0:
                            act[i] = 1 if dc > 0 else 3
                        if rs.rand() < 0.15:
                            act[i] = rs.randint(0, 6)
                else:
                    act = rs.randint(0, 6 if not no_stay else 5, size=R)   # includes the tolerated extra value (Q2)
                actions[e, ep, t] = act
                o, r, d, info = env.step([act])
                coo(obs_rows, e, ep, t + 1, o[0].numpy())
                loc[e, ep, t, :N] = raw.predator_loc
                loc[e, ep, t, N:] = raw.prey_loc
                reached[e, ep, t] = raw.reached_prey
                reward[e, ep, t] = r
                done[e, ep, t] = int(d)
                success[e, ep, t] = raw.stat.get('success', -1)
                nsteps[e, ep] = t + 1
                if d:
                    try:
                        env.step([act])
                        raise SystemExit("reference did not raise on step-after-done")
                    except RuntimeError:
                        pass
                    break
    np.savez_compressed(os.path.join(OUT, name + '.npz'),
                        cfg=np.array([N, dim, vision, {'mixed': 0, 'cooperative': 1, 'competitive': 2}[mode], T, int(no_stay)], np.int32),
                        enemy_comm=int(enemy_comm),
                        seed=SEED, env_gid0=100, obs_dim=env.observation_dim, init_loc=init_loc, actions=actions,
                        loc=loc, reached=reached, reward=reward, done=done, success=success, nsteps=nsteps,
                        ndraws=ndraws, obs_coo=np.array(obs_rows, np.float64))
    print(name, 'steps', nsteps.tolist(), 'done', done.sum(), 'nnz', len(obs_rows))


# ------------------------------------------------------------------------------------------------
# F2: Traffic-Junction tables
# ------------------------------------------------------------------------------------------------
def tj_tables_fixture():
    out = {}
    for diff, dim in [('easy', 6), ('easy', 8), ('medium', 14), ('medium', 8), ('medium', 6), ('hard', 18),
                      ('hard', 9), ('hard', 12)]:
        for v in (0, 1, 2):
            if diff != 'hard' and dim < 4 + v:
                continue
            a = rh.make_args('traffic_junction', nagents=5, dim=dim, vision=v, difficulty=diff)
            env = rh.make_env('traffic_junction', a).env
            key = '%s_%d_v%d' % (diff, dim, v)
            flat = [np.asarray(p) for r in env.routes for p in r]
            out[key + '_grid'] = env.grid.astype(np.int32)
            out[key + '_pad'] = env.pad_grid.astype(np.int32)
            out[key + '_off'] = np.concatenate([[0], np.cumsum([len(p) for p in flat])]).astype(np.int32)
            out[key + '_rc'] = np.concatenate(flat).astype(np.int32)
            out[key + '_meta'] = np.array([env.dims[0], env.dims[1], env.vocab_size, env.OUTSIDE_CLASS,
                                           env.CAR_CLASS, env.BASE, env.npath, len(env.routes),
                                           len(env.routes[0]), a_obs_dim(a, env)], np.int32)
    np.savez_compressed(os.path.join(OUT, 'tj_tables.npz'), **out)
    print('tj_tables', len(out) // 5, 'configs')


def a_obs_dim(a, env):
    from env_wrappers import GymWrapper
    return GymWrapper(env).observation_dim


# ------------------------------------------------------------------------------------------------
# F3: Traffic-Junction trajectories
# ------------------------------------------------------------------------------------------------
def tj_fixture(name, N, dim, vision, difficulty, add_rate, T, nenv=2, nep=2, brake_p=0.3, curriculum=None,
               vocab_type='bool'):
    ref = rh.load_reference()
    kw = dict(add_rate_min=add_rate, add_rate_max=add_rate, vocab_type=vocab_type)
    if curriculum:
        kw = dict(add_rate_min=curriculum[0], add_rate_max=curriculum[1], curr_start=curriculum[2],
                  curr_end=curriculum[3], vocab_type=vocab_type)
    a = rh.make_args('traffic_junction', nagents=N, dim=dim, vision=vision, difficulty=difficulty, max_steps=T, **kw)
    env = rh.make_env('traffic_junction', a)
    raw = env.env
    rs = np.random.RandomState(zlib.crc32(name.encode()) & 0xffff)
    obs_rows = []
    z = lambda *s, dt=np.int32: np.zeros((nenv, nep, T) + s, dt)
    actions, alive, wait, loc, last_act, route_loc, route_id, is_completed = \
        z(N), z(N), z(N), z(N, 2), z(N), z(N), z(N), z(N)
    cars_in_sys, has_failed, ndraws = z(), z(), z()
    reward = z(N, dt=np.float64)
    add_rate_seen = z(dt=np.float64)
    epochs = np.zeros((nenv, nep), np.int32)
    for e in range(nenv):
        st = philox.Stream(SEED, 200 + e)
        env = rh.make_env('traffic_junction', a)      # fresh curriculum state per env instance
        raw = env.env
        for ep in range(nep):
            epoch = ep if not curriculum else int(curriculum[2]) + 3 * ep + e   # exercise TJ:196-200 gating
            epochs[e, ep] = epoch
            o = env.reset(epoch)
            assert not o.numpy().any()
            for t in range(T):
                act = (rs.rand(N) < brake_p).astype(np.int64)
                if rs.rand() < 0.1:
                    act[rs.randint(N)] = 2                                      # tolerated extra value (Q2)
                actions[e, ep, t] = act
                ref['rnd'].begin(st, philox.DOMAIN_TJ_ADD, ep, t)
                d0 = st.ndraws
                o, r, d, info = env.step([act])
                ndraws[e, ep, t] = st.ndraws - d0
                coo(obs_rows, e, ep, t + 1, o[0].numpy())
                alive[e, ep, t] = raw.alive_mask
                wait[e, ep, t] = raw.wait
                loc[e, ep, t] = raw.car_loc
                last_act[e, ep, t] = raw.car_last_act
                route_loc[e, ep, t] = raw.car_route_loc
                route_id[e, ep, t] = raw.route_id
                is_completed[e, ep, t] = info['is_completed']
                cars_in_sys[e, ep, t] = raw.cars_in_sys
                has_failed[e, ep, t] = raw.has_failed
                reward[e, ep, t] = r
                add_rate_seen[e, ep, t] = raw.stat['add_rate']
                assert not d
                assert raw.stat['success'] == 1 - raw.has_failed
    np.savez_compressed(os.path.join(OUT, name + '.npz'),
                        cfg=np.array([N, dim, vision, {'easy': 0, 'medium': 1, 'hard': 2}[difficulty], T], np.int32),
                        add_rate=add_rate, curriculum=np.array(curriculum if curriculum else [0, 0, 0, 0], np.float64),
                        scalar=int(vocab_type == 'scalar'),
                        epochs=epochs, seed=SEED, env_gid0=200, obs_dim=env.observation_dim, actions=actions,
                        alive=alive, wait=wait, loc=loc, last_act=last_act, route_loc=route_loc, route_id=route_id,
                        is_completed=is_completed, cars_in_sys=cars_in_sys, has_failed=has_failed, reward=reward,
                        add_rate_seen=add_rate_seen, ndraws=ndraws, obs_coo=np.array(obs_rows, np.float64))
    print(name, 'alive-steps', int(alive.sum()), 'completed', int(is_completed.sum()), 'failed', int(has_failed.max()),
          'draws', int(ndraws.sum()), 'nnz', len(obs_rows))


def main():
    which = sys.argv[1:] or ['pp', 'tjt', 'tj', 'policy', 'trainer']
    if 'pp' in which:
        pp_fixture('pp_easy_mixed', 3, 5, 0, 'mixed', 20, greedy_env=[1])
        pp_fixture('pp_easy_coop', 3, 5, 0, 'cooperative', 20, greedy_env=[1, 2])
        pp_fixture('pp_easy_comp', 3, 5, 0, 'competitive', 20, greedy_env=[1, 2])
        pp_fixture('pp_medium_mixed', 5, 10, 1, 'mixed', 40, greedy_env=[2])
        pp_fixture('pp_hard_mixed', 10, 20, 1, 'mixed', 80, nenv=2, greedy_env=[1])
        pp_fixture('pp_edge_v2', 4, 6, 2, 'mixed', 30, greedy_env=[0], nenv=3)
        pp_fixture('pp_nostay_v1', 3, 4, 1, 'cooperative', 15, no_stay=True)
        pp_fixture('pp_enemycomm_mixed', 3, 5, 1, 'mixed', 20, greedy_env=[1], enemy_comm=True)
        pp_fixture('pp_enemycomm_coop', 4, 6, 0, 'cooperative', 20, greedy_env=[0, 2], enemy_comm=True)
    if 'tjt' in which:
        tj_tables_fixture()
    if 'tj' in which:
        tj_fixture('tj_easy_v0', 5, 6, 0, 'easy', 0.3, 20)
        tj_fixture('tj_easy_v1_full', 5, 6, 1, 'easy', 1.0, 20, brake_p=0.6)
        tj_fixture('tj_medium_v0', 10, 14, 0, 'medium', 0.05, 40)
        tj_fixture('tj_medium_v1', 10, 14, 1, 'medium', 0.3, 40)
        tj_fixture('tj_hard_v0', 20, 18, 0, 'hard', 0.05, 80, nenv=1)
        tj_fixture('tj_hard_v1', 20, 18, 1, 'hard', 0.3, 40, nenv=1, brake_p=0.4)
        tj_fixture('tj_hard9_v2', 8, 9, 2, 'hard', 1.0, 25, nenv=1, brake_p=0.5)
        tj_fixture('tj_easy_curr', 5, 6, 0, 'easy', 0.1, 12, nenv=2, nep=6, curriculum=(0.1, 0.3, 2, 12))
        tj_fixture('tj_scalar_medium_v1', 10, 14, 1, 'medium', 0.3, 30, vocab_type='scalar')
        tj_fixture('tj_scalar_easy_v0', 5, 6, 0, 'easy', 0.5, 20, vocab_type='scalar', brake_p=0.5)
        tj_fixture('tj_scalar_hard_v2', 12, 12, 2, 'hard', 0.4, 25, nenv=1, vocab_type='scalar')
    if 'policy' in which:
        import make_golden_policy
        make_golden_policy.main()
    if 'trainer' in which:
        import make_golden_policy
        make_golden_policy.trainer_main()


if __name__ == '__main__':
    main()
