#!/usr/bin/env python
"""Copies what tools/gpu_profile.sh left under gpurun_out/r02prof into profiles/<round>/ and refreshes the fused-kernel
PMC entry of profiles/obs_traffic.json:  python tools/collect_profile.py [r02]"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else 'r02'
src, dst = os.path.join(ROOT, 'gpurun_out', 'r02prof'), os.path.join(ROOT, 'profiles', rnd)


def newest(pat):
    return sorted(glob.glob(pat), key=os.path.getmtime)[-1]


def agg(path, counter):
    d = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] == counter:
            d[r['Kernel_Name']][0] += 1
            d[r['Kernel_Name']][1] += float(r['Counter_Value'])
    return d


for f in sorted(os.listdir(src)):
    if (f.startswith('bench_pp') or f.startswith('bench_tj')) and f.endswith('.json') or f in (
            'smoke.log', 'ws_probe.txt', 'two_stream.txt', 'policy_step_ablation_e8192.txt',
            'policy_step_ablation_lone_tile_e384.txt', 'policy_step_env_count_sweep.txt'):
        shutil.copy(os.path.join(src, f), os.path.join(dst, f))
ks = newest(src + '/kt/runc/*_kernel_stats.csv')
shutil.copy(ks, dst + '/bench_pp_hard_kernel_stats.csv')
shutil.copy(newest(src + '/kt_tj_hard/runc/*_kernel_stats.csv'), dst + '/bench_tj_hard_kernel_stats.csv')
m = open(newest(src + '/roctx/runc/*_marker_api_trace.csv')).read().split('\n')
open(dst + '/roctx_marker_trace_sample.csv', 'w').write('\n'.join(m[:60]) + '\n')
w = agg(newest(src + '/pmc_w/runc/*_counter_collection.csv'), 'WRITE_SIZE')
f = agg(newest(src + '/pmc_f/runc/*_counter_collection.csv'), 'FETCH_SIZE')
with open(dst + '/bench_pp_hard_pmc_hbm.csv', 'w') as o:
    o.write('"kernel","launches","WRITE_SIZE_KiB_avg","FETCH_SIZE_KiB_avg (x2 for bytes read on gfx950)"\n')
    for k in sorted(w, key=lambda k: -w[k][1]):
        fv = f.get(k, [1, 0.0])
        o.write('"%s",%d,%.1f,%.1f\n' % (k[:110], w[k][0], w[k][1] / w[k][0], fv[1] / max(1, fv[0])))
k = [k for k in w if 'policy_step_kernel' in k][0]
W, F = w[k][1] / w[k][0], f[k][1] / f[k][0]
tot = int(round((W + 2 * F) * 1024))
tj = os.path.join(ROOT, 'profiles', 'obs_traffic.json')
t = json.load(open(tj))
t['pp_hard_fused'] = tot
raw = t['_raw']['pp_hard_fused']
raw.update(WRITE_SIZE_KiB_avg=round(W, 1), FETCH_SIZE_KiB_avg=round(F, 1), launches=w[k][0])
json.dump(t, open(tj, 'w'), indent=1)
lines = open(src + '/tests_gpu.log').read().strip().split('\n')
open(dst + '/tests_gpu_summary.txt', 'w').write('\n'.join(lines[-3:]) + '\n')
print("fused kernel PMC: WRITE %.0f KiB + 2 x FETCH %.0f KiB = %d B = %.4f x algorithmic" % (W, F, tot, tot / raw['algorithmic_bytes']))
print(open(ks).read().split('\n')[1][:130])
print(lines[-1])
for name in ('bench_pp_hard', 'bench_pp_hard_driver_args', 'bench_tj_hard', 'bench_tj_medium'):
    d = json.loads(open(os.path.join(dst, name + '.json')).read().strip().splitlines()[-1])
    cb = d.get('cpu_baseline') or {}
    print(name, d['value'] / 1e6, d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'],
          (d.get('roofline_mfma') or {}).get('achieved'), cb.get('value'), (cb.get('reference_shaped') or {}).get('value'))
