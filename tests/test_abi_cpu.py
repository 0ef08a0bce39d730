"""CPU: the C-ABI shared library loads and exports every symbol include/ic3_rollout.h declares; the
host-only table generator of the product matches tables captured from the reference."""
import os
import re

import numpy as np
import pytest

from golden_util import load

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "ic3_rollout.h")).read()
    return sorted(set(re.findall(r"\b(ic3_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from ic3net_amd import _lib
    l = _lib.lib()
    syms = declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(l, s), s
    assert set(syms) == set(_lib.EXPORTS), set(syms) ^ set(_lib.EXPORTS)
    assert l.ic3_version() == 200


def test_product_tj_tables_match_reference():
    from ic3net_amd.envs import tj_build_tables
    fx = load("tj_tables")
    keys = sorted(k[:-5] for k in fx.files if k.endswith("_meta"))
    for key in keys:
        diff, dim, v = key.split("_")
        d, grid, off, rc = tj_build_tables(int(dim), int(v[1:]), diff)
        np.testing.assert_array_equal(grid, fx[key + "_grid"])
        np.testing.assert_array_equal(off, fx[key + "_off"])
        np.testing.assert_array_equal(rc, fx[key + "_rc"])
        m = [int(x) for x in fx[key + "_meta"]]
        assert [d.grid_h, d.grid_w, d.vocab, d.npath, d.narrival, d.obs_dim] == [m[0], m[1], m[2], m[6], m[7], m[9]]


def test_product_tj_tables_reject_bad_dims():
    from ic3net_amd.envs import tj_build_tables
    for args, msg in (((7, 0, "medium"), "even"), ((4, 1, "easy"), "Min dim"), ((10, 0, "hard"), "multiple of 3"),
                      ((6, 0, "hard"), "Min dim: 9")):
        with pytest.raises(AssertionError, match=msg):
            tj_build_tables(*args)


def test_no_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import argparse
    from ic3net_amd.envs import PredatorPreyEnv
    env = PredatorPreyEnv()
    with pytest.raises(Exception):
        env.multi_agent_init(argparse.Namespace(nfriendly=3, nenemies=1, dim=5, vision=0, moving_prey=False,
                                                mode='mixed', enemy_comm=False, no_stay=False, nenvs=2, seed=0))


def test_policy_state_dict_keys_match_reference():
    """Checkpoint compatibility surface: same parameter names and shapes as the reference's
    CommNetMLP.state_dict(), recorded in the policy fixtures."""
    import torch
    from policy_util import POLICY_FIXTURES, PolicyCase
    from ic3net_amd.comm import CommNetMLP
    for name in POLICY_FIXTURES:
        pc = PolicyCase(name)
        net = CommNetMLP(pc.args(), pc.obs_dim)
        sd = net.state_dict()
        names = [str(n) for n in pc.fx["param_names"]]
        shapes = [eval(str(s)) for s in pc.fx["param_shapes"]]
        assert sorted(sd.keys()) == names, name
        assert [tuple(sd[n].shape) for n in names] == shapes, name
