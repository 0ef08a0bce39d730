"""CPU property tests (hypothesis) of the oracle — the size-independent invariants the GPU tests then check at
BASELINE sizes: PP move == clamp, exactly one location bit per window cell, TJ cars_in_sys == alive.sum(),
routes are unit-step paths, stream determinism / shard invariance."""
import numpy as np
from hypothesis import given, settings, strategies as st

import oracle
from oracle import philox


@settings(max_examples=40, deadline=None)
@given(N=st.integers(1, 8), dim=st.integers(3, 9), vision=st.integers(0, 2), seed=st.integers(0, 2 ** 31 - 1),
       mode=st.sampled_from(["mixed", "cooperative", "competitive"]))
def test_pp_invariants(N, dim, vision, seed, mode):
    if N + 1 > dim * dim:
        return
    env = oracle.PPOracle(N, dim, vision, mode, seed=seed, env_gid=seed % 97)
    obs = env.reset()
    # distinct initial cells (np.random.choice(replace=False), predator_prey_env.py:174)
    cells = {tuple(x) for x in env.loc}
    assert len(cells) == N + 1
    rs = np.random.RandomState(seed % 1000)
    W, vocab = 2 * vision + 1, dim * dim + 4
    for t in range(12):
        before, frozen = env.loc.copy(), env.reached.copy()
        act = rs.randint(0, 6, size=N)
        obs, rew, done = env.step(act)
        delta = {0: (-1, 0), 1: (0, 1), 2: (1, 0), 3: (0, -1)}
        for i in range(N):
            d = delta.get(int(act[i]), (0, 0))
            want = before[i] if frozen[i] else np.clip(before[i] + d, 0, dim - 1)
            assert (env.loc[i] == want).all()
        assert (env.loc[N] == before[N]).all()                                       # fixed prey
        o = obs.reshape(N, W * W, vocab)
        assert (o[:, :, :dim * dim + 2].sum(-1) == 1).all()                          # one location bit per cell
        assert (o[:, (W * W) // 2, vocab - 1] >= 1).all()                            # the agent itself is counted
        assert o[:, :, vocab - 1].sum() <= N * N and set(np.unique(rew)).issubset(
            {-0.05, 0.0} | {0.05 * k for k in range(1, N + 1)} | {0.05 / k for k in range(1, N + 1)})
        if done:
            assert mode == "mixed" and env.reached.all()
            break


@settings(max_examples=25, deadline=None)
@given(cfg=st.sampled_from([("easy", 6, 5), ("medium", 8, 6), ("medium", 14, 10), ("hard", 9, 8), ("hard", 18, 20)]),
       vision=st.integers(0, 1), rate=st.sampled_from([0.05, 0.3, 1.0]), seed=st.integers(0, 2 ** 31 - 1))
def test_tj_invariants(cfg, vision, rate, seed):
    diff, dim, N = cfg
    env = oracle.TJOracle(N, dim, vision, diff, add_rate_min=rate, add_rate_max=rate, seed=seed, env_gid=3)
    env.reset(0)
    rs = np.random.RandomState(seed % 1000)
    for t in range(25):
        prev_alive = env.alive.copy()
        obs, rew, _ = env.step((rs.rand(N) < 0.4).astype(int))
        assert env.cars_in_sys.value == env.alive.sum() <= N                         # the reference's bookkeeping
        assert ((env.loc == 0).all(1) | (env.alive == 1)).all()                      # dead cars park at (0,0), Q8
        assert (env.wait[env.alive == 0] == 0).all()
        assert (obs[env.alive == 0] == 0).all()                                      # dead rows are all-zero
        assert (rew[env.alive == 0] == 0).all()
        assert ((env.is_completed == 1) <= (prev_alive == 1)).all()
        alive_rows = obs[env.alive == 1]
        if len(alive_rows):
            win = alive_rows[:, 2:].reshape(len(alive_rows), (2 * vision + 1) ** 2, -1)
            assert (win[:, :, :env.tab['vocab'] - 1].sum(-1) == 1).all()     # one location bit per cell (any channel but CAR, Q9)
    for p in env.tab["routes"]:
        assert (np.abs(np.diff(p, axis=0)).sum(1) == 1).all()                        # _unittest_path, TJ:526-537


@settings(max_examples=50, deadline=None)
@given(seed=st.integers(0, 2 ** 32 - 1), gid=st.integers(0, 2 ** 20), ep=st.integers(0, 1000), t=st.integers(0, 100),
       d=st.integers(0, 200), n=st.integers(1, 1700))
def test_stream_scaling_is_exact(seed, gid, ep, t, d, n):
    x = philox.x24(seed, gid, 2, ep, t, d)
    assert 0 <= x < 2 ** 24
    assert int((x / 16777216.0) * n) == (x * n) >> 24          # what np.random.choice(n) becomes under injection
    assert oracle.lib().orc_x24(seed, gid, 2, ep, t, d) == x


def test_pp_reset_is_shard_invariant():
    a = [oracle.PPOracle(10, 20, 1, seed=9, env_gid=g) for g in range(16)]
    b = [oracle.PPOracle(10, 20, 1, seed=9, env_gid=g) for g in range(8, 16)]      # "second shard"
    for e in a + b:
        e.reset()
    for i in range(8):
        assert (a[8 + i].loc == b[i].loc).all()
