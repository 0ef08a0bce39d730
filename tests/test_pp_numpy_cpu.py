"""CPU: the reference-shaped numpy Predator-Prey env (oracle/pp_numpy.py, leg (ii) of bench.py's cpu_baseline) against
the trajectories recorded from the reference itself — positions, observations, rewards, done flags."""
import numpy as np
import pytest

from golden_util import load, SparseObs, MODES
from oracle.pp_numpy import PPNumpyEnv

FIXTURES = ["pp_easy_mixed", "pp_easy_coop", "pp_easy_comp", "pp_medium_mixed", "pp_hard_mixed", "pp_edge_v2",
            "pp_nostay_v1"]


@pytest.mark.parametrize("name", FIXTURES)
def test_reference_shaped_numpy_env_matches_reference(name):
    fx = load(name)
    N, dim, vision, mode, T, no_stay = [int(x) for x in fx["cfg"]]
    nenv, nep = fx["nsteps"].shape
    sp = SparseObs(fx["obs_coo"], N, int(fx["obs_dim"]))
    for e in range(nenv):
        env = PPNumpyEnv(N, dim, vision, MODES[mode], stay=not no_stay, seed=int(fx["seed"]),
                         env_gid=int(fx["env_gid0"]) + e)
        assert env.obs_dim == int(fx["obs_dim"])
        for ep in range(nep):
            obs = env.reset()
            np.testing.assert_array_equal(env.loc, fx["init_loc"][e, ep])
            np.testing.assert_array_equal(obs, sp.dense(e, ep, 0))
            for t in range(int(fx["nsteps"][e, ep])):
                obs, rew, done = env.step(fx["actions"][e, ep, t])
                np.testing.assert_array_equal(env.loc, fx["loc"][e, ep, t])
                np.testing.assert_array_equal(env.reached, fx["reached"][e, ep, t])
                np.testing.assert_array_equal(rew, fx["reward"][e, ep, t])
                assert int(done) == int(fx["done"][e, ep, t])
                np.testing.assert_array_equal(obs, sp.dense(e, ep, t + 1))
