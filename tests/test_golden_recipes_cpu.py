"""The recipes of the golden vectors must keep RUNNING: every `tests/golden/make_golden*.py` entry point is re-run against
the reference (`/root/reference`, build container only — skipped where it is absent, e.g. on the GPU box) into a scratch
directory (`IC3_GOLDEN_OUT`) and what it writes is compared with the committed fixture of the same name: integers, strings
and float32 exactly, float64 to 1e-12 of the array's magnitude (the reference's fp64 BLAS products sum in a
thread-dependent order: last-ulp differences between two runs of the reference itself).  Round 4 broke `trainer_case`
(a NameError) without any test noticing; the fixtures of an oracle are only as pinned as their recipe is runnable.

The checkpoint recipe runs the reference's `main.py` on numpy's unseeded global stream (main.py:157-159 seeds torch only):
its VALUES differ from run to run by construction, so it is pinned in shape — the same checkpoint keys and tensor shapes,
the same stdout line structure (tests/test_checkpoint_cpu.py reads exactly those)."""
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, 'golden')
REF = '/root/reference'

pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason='the reference is only present in the build container')

RECIPES = {
    'pp': ['make_golden.py', 'pp'],
    'tj_tables': ['make_golden.py', 'tjt'],
    'tj': ['make_golden.py', 'tj'],
    'sweep': ['make_golden_sweep.py'],
    'policy': ['make_golden_policy.py'],
    'policy_fullsize': ['make_golden_policy.py', 'fullsize'],
    'policy_multipass': ['make_golden_policy.py', 'multipass'],
    'policy_nonrec': ['make_golden_policy.py', 'nonrec'],
    'trainer': ['make_golden_policy.py', 'trainer'],          # (runs trainer_fullsize too)
    'grad_nonrec': ['make_golden_policy.py', 'grad_nonrec'],
    'grad_fullsize': ['make_golden_policy.py', 'grad_fullsize'],
    'grad_scaled': ['make_golden_policy.py', 'grad_scaled'],
    'grad_stream': ['make_golden_policy.py', 'grad_stream'],
    'grad_stream_h128': ['make_golden_policy.py', 'grad_stream_h128'],
    'grad_stream_families': ['make_golden_policy.py', 'grad_stream_families'],
    'grad_baseline': ['make_golden_policy.py', 'grad_baseline'],
    'render': ['make_golden_render.py'],
    'ckpt': ['make_golden_ckpt.py'],
}
# every committed data fixture must come out of one of the recipes above
EXPECTED_MIN_NPZ = 74


@pytest.fixture(scope='module')
def regenerated(tmp_path_factory):
    """All recipes at once (they are independent processes; the slowest takes ~1.5 min on its own)."""
    root = tmp_path_factory.mktemp('golden_regen')
    procs = {}
    for name, argv in RECIPES.items():
        out = root / name
        out.mkdir()
        env = dict(os.environ, IC3_GOLDEN_OUT=str(out), OMP_NUM_THREADS='1', MKL_NUM_THREADS='1', PYTHONDONTWRITEBYTECODE='1')
        log = open(out / '_log.txt', 'w')
        procs[name] = (subprocess.Popen([sys.executable] + argv, cwd=GOLD, env=env, stdout=log, stderr=subprocess.STDOUT), out, log)
    done = {}
    for name, (p, out, log) in procs.items():
        try:
            rc = p.wait(timeout=900)
        except subprocess.TimeoutExpired:
            p.kill()
            rc = -9
        log.close()
        done[name] = (rc, out)
    return done


def _same(name, key, new, old):
    assert new.shape == old.shape and new.dtype == old.dtype, (name, key, new.shape, old.shape, new.dtype, old.dtype)
    if new.dtype == np.float64 and new.size:
        scale = max(1.0, float(np.nanmax(np.abs(old))) if np.isfinite(old).any() else 1.0)
        assert np.array_equal(np.isnan(new), np.isnan(old)), (name, key)
        err = np.nanmax(np.abs(new - old)) if np.isfinite(old).any() else 0.0
        assert err <= 1e-12 * scale, (name, key, err, scale)
    else:
        assert np.array_equal(new, old), (name, key)


@pytest.mark.parametrize('recipe', [r for r in RECIPES if r != 'ckpt'])
def test_recipe_reproduces_the_committed_fixtures(regenerated, recipe):
    rc, out = regenerated[recipe]
    log = open(out / '_log.txt').read()
    assert rc == 0, log[-3000:]
    made = sorted(f for f in os.listdir(out) if not f.startswith('_'))
    assert made, 'the recipe wrote nothing'
    for f in made:
        committed = os.path.join(GOLD, f)
        assert os.path.exists(committed), '%s is written by the recipe but not committed' % f
        if f.endswith('.npz'):
            new, old = np.load(out / f, allow_pickle=True), np.load(committed, allow_pickle=True)
            assert sorted(new.files) == sorted(old.files), (f, sorted(set(new.files) ^ set(old.files)))
            for k in new.files:
                _same(f, k, new[k], old[k])
        elif f.endswith('.json'):
            assert json.load(open(out / f)) == json.load(open(committed)), f
        else:
            assert open(out / f, 'rb').read() == open(committed, 'rb').read(), f


def test_every_committed_fixture_has_a_recipe(regenerated):
    made = set()
    for name, (rc, out) in regenerated.items():
        made |= {f for f in os.listdir(out) if not f.startswith('_')}
    committed = {f for f in os.listdir(GOLD) if f.endswith(('.npz', '.json', '.pt', '.txt'))}
    assert committed <= made, 'fixtures nobody regenerates: %s' % sorted(committed - made)
    assert len([f for f in made if f.endswith('.npz')]) >= EXPECTED_MIN_NPZ


def test_checkpoint_recipe_runs_and_keeps_its_shape(regenerated):
    sys.path.insert(0, os.path.dirname(HERE))
    from ic3net_amd import checkpoint
    rc, out = regenerated['ckpt']
    assert rc == 0, open(out / '_log.txt').read()[-3000:]
    new = checkpoint.read(str(out / 'ref_ckpt_pp_easy.pt'))
    old = checkpoint.read(os.path.join(GOLD, 'ref_ckpt_pp_easy.pt'))
    assert sorted(new) == sorted(old)
    assert {k: tuple(v.shape) for k, v in new['policy_net'].items()} == {k: tuple(v.shape) for k, v in old['policy_net'].items()}
    assert sorted(new['log']) == sorted(old['log'])
    for k in old['log']:
        assert len(new['log'][k].data) == len(old['log'][k].data), k
    # numbers -> '#', numpy's column padding inside [...] dropped: what stays is the line structure plot_script parses
    shape = lambda text: [re.sub(r'\s+', '', re.sub(r'-?\d+(\.\d*)?(e-?\d+)?', '# ', line)) for line in text.splitlines()]
    assert shape(open(out / 'ref_stdout_pp_easy.txt').read()) == shape(open(os.path.join(GOLD, 'ref_stdout_pp_easy.txt')).read())
    assert sorted(json.load(open(out / 'ref_plot_expect.json'))) == sorted(json.load(open(os.path.join(GOLD, 'ref_plot_expect.json'))))
