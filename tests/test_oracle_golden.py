"""CPU: the oracle (oracle/) against golden vectors captured from the reference itself.
This is what pins the oracle (prompt §③); the GPU tests then compare the HIP path with the oracle."""
import numpy as np
import pytest

import oracle
from oracle import philox, tj_tables
from golden_util import load, SparseObs, PP_FIXTURES, TJ_FIXTURES, MODES, DIFFS


def test_philox_known_answers():
    # Random123 kat_vectors for philox4x32-10
    assert philox.philox4x32_10((0, 0, 0, 0), (0, 0)) == (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)
    assert philox.philox4x32_10((0xffffffff,) * 4, (0xffffffff,) * 2) == \
        (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)
    assert philox.philox4x32_10((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0)) == \
        (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)


def test_philox_c_matches_python():
    rs = np.random.RandomState(0)
    for _ in range(200):
        a = [int(x) for x in rs.randint(0, 2 ** 31, size=6)]
        assert oracle.lib().orc_x24(*a) == philox.x24(*a)
    v = philox.x24_vec(7, np.arange(50), 2, 3, 4, 5)
    assert [int(x) for x in v] == [philox.x24(7, e, 2, 3, 4, 5) for e in range(50)]


def test_rate_threshold_equivalence():
    for p in (0.0, 0.02, 0.05, 0.3, 0.07, 1.0, 0.1 + 0.02 * 3):
        thr = philox.rate_threshold(p)
        for x in (0, 1, thr - 1, thr, thr + 1, 2 ** 24 - 1):
            if 0 <= x < 2 ** 24:
                assert (x / 16777216.0 <= p) == (x <= thr)


@pytest.mark.parametrize("name", PP_FIXTURES)
def test_pp_oracle_matches_reference(name):
    fx = load(name)
    N, dim, vision, mode, T, no_stay = [int(x) for x in fx["cfg"]]
    nenv, nep = fx["nsteps"].shape
    ec = bool(int(fx["enemy_comm"])) if "enemy_comm" in fx.files else False
    sp = SparseObs(fx["obs_coo"], N + (1 if ec else 0), int(fx["obs_dim"]))
    for e in range(nenv):
        env = oracle.PPOracle(N, dim, vision, MODES[mode], stay=not no_stay, seed=int(fx["seed"]),
                              env_gid=int(fx["env_gid0"]) + e, enemy_comm=ec)
        assert env.obs_dim == int(fx["obs_dim"])
        for ep in range(nep):
            obs = env.reset()
            np.testing.assert_array_equal(env.loc, fx["init_loc"][e, ep])
            np.testing.assert_array_equal(obs, sp.dense(e, ep, 0))
            n = int(fx["nsteps"][e, ep])
            for t in range(n):
                obs, rew, done = env.step(fx["actions"][e, ep, t])
                np.testing.assert_array_equal(env.loc, fx["loc"][e, ep, t])
                np.testing.assert_array_equal(env.reached, fx["reached"][e, ep, t])
                np.testing.assert_array_equal(rew, fx["reward"][e, ep, t])          # float64, bit-exact
                assert int(done) == int(fx["done"][e, ep, t])
                if fx["success"][e, ep, t] >= 0:
                    assert env.success.value == fx["success"][e, ep, t]
                np.testing.assert_array_equal(obs, sp.dense(e, ep, t + 1))
            if fx["done"][e, ep, n - 1]:
                with pytest.raises(RuntimeError):
                    env.step(fx["actions"][e, ep, n - 1])


def test_tj_tables_match_reference():
    fx = load("tj_tables")
    keys = sorted(k[:-5] for k in fx.files if k.endswith("_meta"))
    assert len(keys) == 24
    for key in keys:
        diff, dim, v = key.split("_")
        t = tj_tables.build(int(dim), int(v[1:]), diff)
        np.testing.assert_array_equal(t["grid"], fx[key + "_grid"])
        np.testing.assert_array_equal(t["pad_grid"], fx[key + "_pad"])
        np.testing.assert_array_equal(t["route_off"], fx[key + "_off"])
        np.testing.assert_array_equal(t["route_rc"], fx[key + "_rc"])
        meta = [t["h"], t["w"], t["vocab"], t["outside"], t["car_class"], t["base"], t["npath"], t["narrival"],
                t["routes_per_arrival"], 2 + (2 * int(v[1:]) + 1) ** 2 * t["vocab"]]
        assert meta == [int(x) for x in fx[key + "_meta"]]


def test_tj_route_property():
    # the reference's own `_unittest_path` property (traffic_junction_env.py:526-537), on every route
    for diff, dim in (("medium", 14), ("hard", 18), ("easy", 6)):
        t = tj_tables.build(dim, 1, diff)
        for p in t["routes"]:
            assert (np.abs(np.diff(p, axis=0)).sum(1) == 1).all()


@pytest.mark.parametrize("name", TJ_FIXTURES)
def test_tj_oracle_matches_reference(name):
    fx = load(name)
    N, dim, vision, diff, T = [int(x) for x in fx["cfg"]]
    nenv, nep = fx["epochs"].shape
    sp = SparseObs(fx["obs_coo"], N, int(fx["obs_dim"]))
    cur = fx["curriculum"]
    has_curr = bool(cur[3] > cur[2])
    for e in range(nenv):
        kw = dict(add_rate_min=float(fx["add_rate"]), add_rate_max=float(fx["add_rate"]))
        if has_curr:
            kw = dict(add_rate_min=cur[0], add_rate_max=cur[1], curr_start=cur[2], curr_end=cur[3])
        env = oracle.TJOracle(N, dim, vision, DIFFS[diff], seed=int(fx["seed"]), env_gid=int(fx["env_gid0"]) + e,
                              vocab_type='scalar' if ("scalar" in fx.files and int(fx["scalar"])) else 'bool', **kw)
        assert env.obs_dim == int(fx["obs_dim"])
        for ep in range(nep):
            obs = env.reset(int(fx["epochs"][e, ep]))
            assert not obs.any()
            for t in range(T):
                obs, rew, done = env.step(fx["actions"][e, ep, t])
                for k in ("alive", "wait", "loc", "last_act", "route_loc", "route_id", "is_completed"):
                    np.testing.assert_array_equal(getattr(env, k), fx[k][e, ep, t], err_msg="%s t=%d" % (k, t))
                assert env.cars_in_sys.value == fx["cars_in_sys"][e, ep, t]
                assert env.has_failed.value == fx["has_failed"][e, ep, t]
                np.testing.assert_array_equal(rew, fx["reward"][e, ep, t])
                assert env.add_rate.value == fx["add_rate_seen"][e, ep, t]
                ref_obs = sp.dense(e, ep, t + 1)
                np.testing.assert_array_equal(obs, ref_obs)


def test_oracle_reproduces_reference_checksum_sweep():
    """120 random PP and 90 random TJ configurations run through the REFERENCE (tests/golden/make_golden_sweep.py):
    the oracle must reproduce the CRC32 of (state, reward, obs) at every step — pins it across the config space."""
    from golden_util import crc_of, SWEEP_RATES
    fx = load("sweep_checksums")
    seed = int(fx["seed"])
    for cfg, acts, crcs in zip(fx["pp_cfg"], fx["pp_act"], fx["pp_crc"]):
        N, dim, v, mode, ec, ns, gid = [int(x) for x in cfg]
        env = oracle.PPOracle(N, dim, v, MODES[mode], stay=not ns, seed=seed, env_gid=gid, enemy_comm=bool(ec))
        obs = env.reset()
        assert crc_of(env.loc[:N], env.loc[N:], obs) == crcs[0], cfg
        for t in range(acts.shape[0]):
            if env.over.value:
                assert crcs[t + 1] == 0
                continue
            obs, rew, done = env.step(acts[t, :N + ec])
            assert crc_of(env.loc[:N], env.reached, rew.astype(np.float32), obs, np.int32(int(done))) == crcs[t + 1], (cfg, t)
    for cfg, acts, crcs in zip(fx["tj_cfg"], fx["tj_act"], fx["tj_crc"]):
        N, dim, v, diff, rate_i, scalar, gid = [int(x) for x in cfg]
        r = SWEEP_RATES[rate_i]
        env = oracle.TJOracle(N, dim, v, DIFFS[diff], add_rate_min=r, add_rate_max=r, seed=seed, env_gid=gid,
                              vocab_type='scalar' if scalar else 'bool')
        env.reset(0)
        for t in range(acts.shape[0]):
            obs, rew, _ = env.step(acts[t, :N])
            got = crc_of(env.alive, env.wait, env.loc, env.last_act, env.route_loc, env.route_id, rew.astype(np.float32), obs)
            assert got == crcs[t], (cfg, t)
