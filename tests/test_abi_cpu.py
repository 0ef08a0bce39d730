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
    hdr = int(re.search(r"#define\s+IC3_VERSION\s+(\d+)", open(os.path.join(ROOT, "include", "ic3_rollout.h")).read()).group(1))
    assert l.ic3_version() == hdr == _lib.ABI_VERSION


def test_library_exports_nothing_but_the_documented_c_abi():
    """Round-5 verdict item 7: `nm -D libic3rollout.so` lists the entry points of include/ic3_rollout.h and nothing else
    (csrc/exports.map keeps the ic3:: functions shared between the library's objects local), and INTEGRATION.md names every
    one of them."""
    import subprocess
    from ic3net_amd import _lib
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.SO_PATH], capture_output=True, text=True, check=True).stdout
    exported = sorted(l.split()[-1] for l in out.splitlines() if " T " in l)
    assert exported == declared_symbols(), sorted(set(exported) ^ set(declared_symbols()))
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    missing = [sym for sym in exported if not re.search(r"\b%s\b" % re.escape(sym), doc)]
    assert not missing, "INTEGRATION.md does not name: %s" % missing


def test_abi_handshake_refuses_other_versions_and_struct_sizes():
    """include/ic3_rollout.h: a binding built against another header version hands the library structs of another size
    (ic3_policy grew in rounds 2 and 3).  ic3_abi_check names the mismatch, and every entry point that takes a struct
    refuses one whose struct_size is not the library's with -EINVAL BEFORE it reads anything else (no GPU needed to
    see that: the check comes first)."""
    import ctypes as C
    from ic3net_amd import _lib
    l = _lib.lib()
    assert l.ic3_abi_check(_lib.ABI_VERSION, C.sizeof(_lib.Policy), C.sizeof(_lib.Episode)) == 0
    assert l.ic3_abi_check(200, C.sizeof(_lib.Policy), C.sizeof(_lib.Episode)) == -22
    assert b"version 200" in l.ic3_last_error()
    assert l.ic3_abi_check(_lib.ABI_VERSION, C.sizeof(_lib.Policy) - 16, C.sizeof(_lib.Episode)) == -22
    assert b"struct sizes differ" in l.ic3_last_error()
    pol = _lib.Policy()
    assert pol.struct_size == C.sizeof(_lib.Policy)
    pol.struct_size -= 16                        # round 2's ic3_policy: no gate_split / lstm_wp3
    fake = C.c_void_p(64)                        # never dereferenced
    assert l.ic3_policy_forward(C.byref(pol), fake, 1, 1, fake, fake, None, None, fake, None) == -22
    assert b"struct_size" in l.ic3_last_error()
    assert l.ic3_policy_step(fake, C.byref(pol), *([fake] * 12)) == -22
    assert b"struct_size" in l.ic3_last_error()
    ep = _lib.Episode(4, 2, 3, 0, 0, 0)
    assert (ep.struct_size, ep.n, ep.E, ep.N) == (C.sizeof(_lib.Episode), 4, 2, 3)
    ep.struct_size = 0
    assert l.ic3_episode_finalize(C.byref(ep), None) == -22
    assert b"struct_size" in l.ic3_last_error()
    bp = _lib.Bptt()                             # round 6: ic3_bptt — refused by its size before any pointer is read
    bp.struct_size = C.sizeof(bp) - 8
    assert l.ic3_bptt_backward(fake, C.byref(bp), None) == -22
    assert b"ic3_bptt has" in l.ic3_last_error()


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
