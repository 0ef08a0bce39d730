#!/usr/bin/env python
"""Copies what `tools/gpu_call.sh evidence` (+ the train_batch lines) left under gpurun_out/ into profiles/r05/, writes
profiles/r05/README.md (one table: bench line, rocprofv3 kernel stats, PMC traffic, SQ counters of the same commands) and
refreshes the fused-kernel PMC entries of profiles/obs_traffic.json:  python tools/collect_profile_r05.py"""
import csv
import glob
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(ROOT, 'gpurun_out', 'evidence'), os.path.join(ROOT, 'profiles', 'r05')
os.makedirs(dst, exist_ok=True)


def line(path):
    try:
        return json.loads([l for l in open(path).read().splitlines() if l.startswith('{')][-1])
    except Exception:
        return None


def kstat(path, needle):
    """(avg_us, min_us, calls) of the kernel whose name contains `needle` with the most total time"""
    if not os.path.exists(path):
        return None
    best = None
    for r in csv.DictReader(open(path)):
        if needle in r['Name'] and (best is None or float(r['TotalDurationNs']) > float(best['TotalDurationNs'])):
            best = r
    if best is None:
        return None
    return float(best['AverageNs']) / 1e3, float(best['MinNs']) / 1e3, int(best['Calls'])


def pmc(path, needle):
    if not os.path.exists(path):
        return None
    best = None
    for r in csv.DictReader(open(path)):
        if needle in r['kernel'] and (best is None or float(r['avg_per_launch']) * int(r['launches']) > best[0] * best[1]):
            best = (float(r['avg_per_launch']), int(r['launches']))
    return best


for f in sorted(os.listdir(src)):
    if f.endswith(('.json', '.csv', '.txt')):
        shutil.copy(os.path.join(src, f), os.path.join(dst, f))
for extra in ('train/train_batch.txt', 'train/train_profile.txt', 'tests/pytest.txt'):
    p = os.path.join(ROOT, 'gpurun_out', extra)
    if os.path.exists(p):
        name = os.path.basename(p) if 'pytest' not in extra else 'tests_gpu.txt'
        if 'pytest' in extra:
            open(os.path.join(dst, 'tests_gpu_summary.txt'), 'w').write(''.join(open(p).readlines()[-6:]))
        else:
            shutil.copy(p, os.path.join(dst, name))

tj = os.path.join(ROOT, 'profiles', 'obs_traffic.json')
tab = json.load(open(tj))
rows = []
for w in ('pp_hard', 'tj_hard', 'tj_medium', 'pp_scaled', 'pp_easy', 'tj_medium_commnet_mlp', 'pp_hard_ic', 'pp_hard_iric'):
    d = line(os.path.join(src, 'bench_%s.json' % w))
    if d is None:
        continue
    kernel = 'commnet_forward_kernel' if ('commnet' in w or w == 'pp_hard_ic') else 'policy_step_kernel'
    ks = kstat(os.path.join(src, 'bench_%s_kernel_stats.csv' % w), kernel)
    wr = pmc(os.path.join(src, 'pmc_WRITE_SIZE_%s.csv' % w), kernel)
    fe = pmc(os.path.join(src, 'pmc_FETCH_SIZE_%s.csv' % w), kernel)
    r, m = d.get('roofline') or {}, d.get('roofline_mfma') or {}
    traffic = None
    if wr and fe and r.get('bytes_per_launch'):
        traffic = int(round((wr[0] + 2 * fe[0]) * 1024))          # gfx950: FETCH_SIZE counts 32-byte units (x2), KiB
        tab[w + '_fused'] = traffic
        tab.setdefault('_raw', {})[w + '_fused'] = {
            'kernel': kernel, 'WRITE_SIZE_KiB_avg': wr[0], 'FETCH_SIZE_KiB_avg': fe[0], 'launches': wr[1],
            'algorithmic_bytes': r['bytes_per_launch'], 'algorithmic_obs_bytes': r.get('obs_bytes_per_launch'),
            'round': 'r05, profiles/r05/pmc_WRITE_SIZE_%s.csv + pmc_FETCH_SIZE_%s.csv' % (w, w)}
    b16 = (m.get('bf16_issued') or {}).get('frac')
    rows.append("| %s | %.1f M | %.4f | %.4f (min %s / med %s / max %s) | %s | %s | %s | %s | %s |" % (
        w, d['value'] / 1e6, d['ms_per_step'], m.get('avg_launch_ms', 0), d['timing'].get('launch_ms_min'),
        d['timing'].get('launch_ms_median'), d['timing'].get('launch_ms_max'),
        ("%.1f us avg, %.1f min, %d calls" % ks) if ks else "-",
        ("%.0f GB/s = %.3f" % (r['achieved'], r['frac'])) if r.get('achieved') else "-",
        ("%.3f GB = %.3f x" % (traffic / 1e9, traffic / r['bytes_per_launch'])) if traffic else "-",
        ("%.1f TFLOP/s = %.3f" % (m['achieved'], m['frac'])) if m.get('achieved') else "-",
        ("%.3f" % b16) if b16 else "-"))
json.dump(tab, open(tj, 'w'), indent=1)

out = ["# profiles/r05 — round 5 evidence (one gpurun call: `bash tools/gpu_call.sh evidence`; MI355X, E = 8192 envs per GPU)", "",
       "Bench lines: `bench_<workload>.json` (default path: split gate product, obs rows rewritten every step, eager event-timed "
       "launches); `bench_*_fp32_instruction.json` = `--gate-split 0`; `bench_pp_hard_graph.json` = hipGraph replay "
       "(`--time-kernels 0`: no roofline by design); `bench_pp_hard_driver_args.json` = the driver's `--steps 20 --warmup 5`.  "
       "Kernel stats: `bench_<workload>_kernel_stats.csv` (rocprofv3 `--kernel-trace --stats` of `bench.py --steps 40 --warmup 8`, "
       "all launches of the run incl. its first episodes).  PMC: `pmc_WRITE_SIZE_<w>.csv`, `pmc_FETCH_SIZE_<w>.csv` (separate "
       "passes; KiB per launch; traffic = (WRITE + 2 x FETCH) x 1024 — gfx950's FETCH_SIZE counts 32-byte units).", "",
       "| workload | agent-steps/s | ms/step | launch by HIP events (ms) | rocprofv3 kernel stats | HBM: algorithmic / launch, frac of 8 TB/s | "
       "PMC traffic, ratio to algorithmic | MFMA fp32-equivalent, frac of 157.3 | bf16 issued, frac of 2.5 PFLOP/s |",
       "|---|---|---|---|---|---|---|---|---|"] + rows + [""]
out.append("Store stream alone, measured in the same runs (`roofline.store_stream_reference`: the stand-alone obs kernel on the same rows, HIP "
           "events behind the timed region) and the step launch's write rate against it:")
for w in ('pp_hard', 'tj_hard', 'tj_medium', 'pp_scaled'):
    d = line(os.path.join(src, 'bench_%s.json' % w))
    ref = ((d or {}).get('roofline') or {}).get('store_stream_reference')
    if ref:
        out.append("* %s: obs kernel alone %.4f ms = %.0f GB/s; the step launch writes %.0f GB/s = %.3f of it" % (
            w, ref['avg_launch_ms'], ref['GBps'], ref['step_launch_write_rate_GBps'], ref['step_launch_write_rate_over_reference']))
out.append("")
for name, label in (('bench_pp_hard_driver_args', 'PP-hard, the driver\'s --steps 20 --warmup 5'),
                    ('bench_pp_hard_fp32_instruction', 'PP-hard --gate-split 0 (fp32 matrix instruction)'),
                    ('bench_tj_hard_fp32_instruction', 'TJ-hard --gate-split 0'), ('bench_tj_medium_fp32_instruction', 'TJ-medium --gate-split 0'),
                    ('bench_pp_hard_graph', 'PP-hard, hipGraph replay'), ('bench_pp_hard_no_obs_diagnostic', 'PP-hard without obs rows (diagnostic)'),
                    ('bench_pp_hard_auto_reset', 'PP-hard --auto-reset 1'), ('bench_pp_hard_rccl_world1', 'PP-hard --rccl 1 (one-rank RCCL group)')):
    d = line(os.path.join(src, name + '.json'))
    if d:
        m = d.get('roofline_mfma') or {}
        out.append("* %s: %.1f M agent-steps/s, %.4f ms/step, launch %s ms%s" % (
            label, d['value'] / 1e6, d['ms_per_step'], m.get('avg_launch_ms', '-'),
            (", collectives %s" % d['collectives']) if d.get('collectives') else ""))
for f, title in (('sq_counters.csv', 'SQ counters of policy_step_kernel<128, PP, split> per launch (`--steps 20 --warmup 5`; cycle counters '
                                     'in units of 4 clocks except SQ_VALU_MFMA_BUSY_CYCLES)'),
                 ('train_batch.txt', '`tools/bench_train.py` (whole train_batch: rollout + backward through time + RMSprop)')):
    p = os.path.join(dst, f)
    if os.path.exists(p):
        out += ["", "**%s**" % title, "", "```"] + open(p).read().strip().splitlines() + ["```"]
out += ["", "**Other files of the round**", "",
        "* `ws_ab.txt`, `ws_phase_trace_pp_hard.txt`, `ws_phase_trace_tj_hard.txt` — the wave-specialised schedule "
        "(`csrc/policy_step_ws.hpp`, `IC3_PS_WS=1`): A/B against the default kernel, pacing sweep, per-tile phase traces "
        "(`tools/analyze_trace.py --ws`) and the reading (DESIGN.md section 10).",
        "* `train_batch.txt`, `train_profile.txt` — `tools/bench_train.py` lines (lock-step and collection mode) and the kernel table "
        "of one PP-hard update.",
        "* `mp_ab.txt` — comm_passes > 1 inside one launch against one launch per pass (`tools/exp/mp_ab.sh`).",
        "* `tests_gpu_summary.txt` — tail of `pytest -m gpu` on the final code."]
out.append("""
**Update half** (`tools/bench_train.py`, `tools/gpu_call.sh train`): `train_batch.txt` (agent-steps/s of whole updates: lock-step and
collection mode, obs rows on, E = 1024, the baselines), `train_profile*.txt` (kernel table of one update, torch profiler, GEMMs picked
by TunableOp), `train_batch_pp_hard_kernel_stats.csv` (rocprofv3 --kernel-trace --stats of 5 PP-hard updates with the library's
default GEMMs: `policy_step_kernel` 224.8 us x 400 — the training rollout incl. the gate / inp record —, `lstm_gates_bwd_kernel<128, 1, 1>`
213.4 us x 400), `train_sanity.txt` (IC3Net on PP-easy learns on the recorded-gates update), `soak.txt`, `mp_ab.txt` (comm_passes
inside one launch), and the A/B files of what was tried and not kept: `gates_given_stagger_experiment.txt`, `gates_given_nt_ab.txt`,
`bptt_side_stream_ab.txt`; `gates_bwd_dx_hazard_check.txt` (the stale-plane hazard of the asm MFMAs: error and run-to-run check after
the fix); `host_asan.txt` (the product's device code on the host under ASan + UBSan).""")
open(os.path.join(dst, 'README.md'), 'w').write("\n".join(out) + "\n")
print("\n".join(out[:20]))
