"""Checkpoint / log / stdout fixtures written BY THE REFERENCE, this container only.

Runs /root/reference/main.py itself (runpy, `__main__`) on a tiny Predator-Prey job with `--save`, with the same
outside accommodations as ref_harness.py (stub gym + visdom packages, numpy>=2 / torch>=2 patches; the env RNG is
NOT injected here — nothing is compared step by step).  Commits
    ref_ckpt_pp_easy.pt        the file main.py:260-265 wrote ({'policy_net','log','trainer'}, utils.LogField log)
    ref_stdout_pp_easy.txt     what main.py:229-244 printed
    ref_plot_expect.json       what /root/reference/plot_script.py:15-57 `read_file` extracts from that stdout
                               (the function body is taken from the reference file at generation time via ast, so
                               matplotlib is not needed)
"""
import ast
import contextlib
import io
import json
import os
import runpy
import sys

import numpy as np

import ref_harness as rh

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.environ.get('IC3_GOLDEN_OUT', HERE)   # where the fixtures are written (tests/test_golden_recipes_cpu.py: a tmp dir)


def main():
    ref = rh.load_reference()
    ref['pp'].np = np                       # real numpy RNG for this job
    ref['tj'].np = np
    import torch
    _load = torch.load
    torch.load = lambda f, *a, **kw: _load(f, *a, **dict(dict(weights_only=False), **kw))   # torch>=2.6 default
    ck = os.path.join(OUT, 'ref_ckpt_pp_easy.pt')
    argv = ['main.py', '--env_name', 'predator_prey', '--nagents', '3', '--nprocesses', '1', '--num_epochs', '3',
            '--epoch_size', '2', '--batch_size', '60', '--hid_size', '16', '--detach_gap', '10', '--lrate', '0.001',
            '--dim', '5', '--max_steps', '20', '--ic3net', '--vision', '0', '--recurrent', '--seed', '5', '--save', ck]
    old_argv, buf = sys.argv, io.StringIO()
    sys.argv = argv
    try:
        with contextlib.redirect_stdout(buf):
            runpy.run_path(os.path.join(rh.REF, 'main.py'), run_name='__main__')
    finally:
        sys.argv = old_argv
        torch.load = _load
    text = buf.getvalue()
    text = text[text.index('Epoch 1\t'):]          # drop the Namespace / model dump in front
    path = os.path.join(OUT, 'ref_stdout_pp_easy.txt')
    with open(path, 'w') as f:
        f.write(text)
    # plot_script.read_file on the reference's own output
    src = open(os.path.join(rh.REF, 'plot_script.py')).read()
    fn = [n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == 'read_file'][0]
    ns = {'np': np, 'print': lambda *a, **k: None}
    exec(compile(ast.Module([fn], []), 'plot_script.read_file', 'exec'), ns)
    expect = {}
    # (the non-scalar branch needs a tab in the line: only the 'Epoch ...\tReward [...]' line has one)
    for term, scalar in (('Epoch', False), ('Success', True), ('Steps-taken', True)):
        expect[term] = ns['read_file']([], path, scalar, term)
    json.dump(expect, open(os.path.join(OUT, 'ref_plot_expect.json'), 'w'), indent=1)
    d = _load(ck, weights_only=False)
    print('wrote', ck, os.path.getsize(ck), 'bytes; log epochs', d['log']['epoch'].data, 'plot', expect)


if __name__ == '__main__':
    main()
