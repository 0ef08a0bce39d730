#!/usr/bin/env python
"""Copies what tools/gpu_profile_r03.sh left under gpurun_out/r03prof into profiles/r03/ and refreshes the fused-kernel
PMC entries of profiles/obs_traffic.json:  python tools/collect_profile_r03.py"""
import collections
import csv
import glob
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(ROOT, 'gpurun_out', 'r03prof'), os.path.join(ROOT, 'profiles', 'r03')
os.makedirs(dst, exist_ok=True)


def newest(pat):
    fs = sorted(glob.glob(pat, recursive=True), key=os.path.getmtime)
    return fs[-1] if fs else None


def agg(path, counter):
    d = collections.defaultdict(lambda: [0, 0.0])
    if path is None:
        return d
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] == counter:
            d[r['Kernel_Name']][0] += 1
            d[r['Kernel_Name']][1] += float(r['Counter_Value'])
    return d


for f in sorted(os.listdir(src)):
    if f.endswith('.json') and f.startswith('bench_') or f in ('smoke.log', 'buf_probe.txt', 'phase_trace_pp_hard.txt') or \
            f.startswith('train_batch_'):
        shutil.copy(os.path.join(src, f), os.path.join(dst, f))
lines = open(src + '/tests_gpu.log').read().strip().split('\n')
open(dst + '/tests_gpu_summary.txt', 'w').write('\n'.join(lines[-3:]) + '\n')

tj = os.path.join(ROOT, 'profiles', 'obs_traffic.json')
t = json.load(open(tj))
ALG = {}
for w in ('pp_hard', 'tj_hard', 'tj_medium', 'pp_scaled'):
    ks = newest(src + '/kt_%s/**/*_kernel_stats.csv' % w)
    if ks:
        shutil.copy(ks, dst + '/bench_%s_kernel_stats.csv' % w)
    bj = os.path.join(src, 'bench_%s.json' % w)
    if not os.path.exists(bj):
        continue
    d = json.loads(open(bj).read().strip().splitlines()[-1])
    alg, alg_obs = d['roofline']['bytes_per_launch'], d['roofline']['obs_bytes_per_launch']
    wv = agg(newest(src + '/pmc_w_%s/**/*_counter_collection.csv' % w), 'WRITE_SIZE')
    fv = agg(newest(src + '/pmc_f_%s/**/*_counter_collection.csv' % w), 'FETCH_SIZE')
    ks_ = [k for k in wv if 'policy_step_kernel' in k]
    if not ks_:
        print(w, 'no PMC rows')
        continue
    k = max(ks_, key=lambda q: wv[q][1])       # (policy_step_kernel<H, 0> = the few tile-plan calibration launches)
    W, F = wv[k][1] / wv[k][0], (fv[k][1] / fv[k][0] if k in fv else 0.0)
    tot = int(round((W + 2 * F) * 1024))
    with open(dst + '/bench_%s_pmc_hbm.csv' % w, 'w') as o:
        o.write('"kernel","launches","WRITE_SIZE_KiB_avg","FETCH_SIZE_KiB_avg (x2 for bytes read on gfx950)"\n')
        for kk in sorted(wv, key=lambda q: -wv[q][1]):
            f2 = fv.get(kk, [1, 0.0])
            o.write('"%s",%d,%.1f,%.1f\n' % (kk[:110], wv[kk][0], wv[kk][1] / wv[kk][0], f2[1] / max(1, f2[0])))
    if d['config']['envs_per_gpu'] == t.get('_nenvs', 8192):
        t[w + '_fused'] = tot
    t['_raw'][w + '_fused'] = dict(kernel=k[:80], WRITE_SIZE_KiB_avg=round(W, 1), FETCH_SIZE_KiB_avg=round(F, 1),
                                   launches=wv[k][0], algorithmic_bytes=alg, algorithmic_obs_bytes=alg_obs,
                                   round="r03, profiles/r03/bench_%s_pmc_hbm.csv" % w)
    print("%-10s fused kernel PMC: WRITE %.0f KiB + 2 x FETCH %.0f KiB = %d B = %.4f x algorithmic (%d B)" %
          (w, W, F, tot, tot / alg, alg))
json.dump(t, open(tj, 'w'), indent=1)

# SQ counters of the PP-hard launch
rows = []
for i in range(1, 10):
    p = newest(src + '/sq_%d/**/*_counter_collection.csv' % i)
    if p is None:
        continue
    d = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(p)):
        if 'policy_step_kernel' in r['Kernel_Name']:
            d[r['Counter_Name']][0] += 1
            d[r['Counter_Name']][1] += float(r['Counter_Value'])
    for c in sorted(d):
        rows.append("%-34s %14.0f per launch (%d launches)" % (c, d[c][1] / d[c][0], d[c][0]))
if rows:
    open(dst + '/sq_counters_r03.txt', 'w').write(
        "rocprofv3 --pmc passes (3 counters each) over `bench.py --steps 20 --warmup 5`, policy_step_kernel<128, PP> of the "
        "round-3 code, per launch\n(cycle counters in units of 4 clocks except SQ_VALU_MFMA_BUSY_CYCLES):\n" + "\n".join(rows) + "\n")
for name in ('bench_pp_hard', 'bench_pp_hard_driver_args', 'bench_tj_hard', 'bench_tj_medium', 'bench_pp_scaled', 'bench_pp_easy'):
    p = os.path.join(dst, name + '.json')
    if not os.path.exists(p):
        continue
    d = json.loads(open(p).read().strip().splitlines()[-1])
    cb = d.get('cpu_baseline') or {}
    print(name, round(d['value'] / 1e6, 1), d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'],
          (d.get('roofline_mfma') or {}).get('achieved'), cb.get('value'))
print(lines[-1])
