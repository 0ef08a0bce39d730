#!/usr/bin/env python
"""Copies what `tools/gpu_call.sh evidence` (+ the train_batch lines) left under gpurun_out/ into profiles/r04/, writes
profiles/r04/README.md (one table: bench line, rocprofv3 kernel stats, PMC traffic, SQ counters of the same commands) and
refreshes the fused-kernel PMC entries of profiles/obs_traffic.json:  python tools/collect_profile_r04.py"""
import csv
import glob
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(ROOT, 'gpurun_out', 'evidence'), os.path.join(ROOT, 'profiles', 'r04')
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
for w in ('pp_hard', 'tj_hard', 'tj_medium', 'pp_scaled', 'pp_easy', 'tj_medium_commnet_mlp'):
    d = line(os.path.join(src, 'bench_%s.json' % w))
    if d is None:
        continue
    kernel = 'commnet_forward_kernel' if 'commnet' in w else 'policy_step_kernel'
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
            'round': 'r04, profiles/r04/pmc_WRITE_SIZE_%s.csv + pmc_FETCH_SIZE_%s.csv' % (w, w)}
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

out = ["# profiles/r04 — round 4 evidence (one gpurun call: `bash tools/gpu_call.sh evidence`; MI355X, E = 8192 envs per GPU)", "",
       "Bench lines: `bench_<workload>.json` (default path: split gate product, obs rows rewritten every step, eager event-timed "
       "launches); `bench_*_fp32_instruction.json` = `--gate-split 0`; `bench_pp_hard_graph.json` = hipGraph replay "
       "(`--time-kernels 0`: no roofline by design); `bench_pp_hard_driver_args.json` = the driver's `--steps 20 --warmup 5`.  "
       "Kernel stats: `bench_<workload>_kernel_stats.csv` (rocprofv3 `--kernel-trace --stats` of `bench.py --steps 40 --warmup 8`, "
       "all launches of the run incl. its first episodes).  PMC: `pmc_WRITE_SIZE_<w>.csv`, `pmc_FETCH_SIZE_<w>.csv` (separate "
       "passes; KiB per launch; traffic = (WRITE + 2 x FETCH) x 1024 — gfx950's FETCH_SIZE counts 32-byte units).", "",
       "| workload | agent-steps/s | ms/step | launch by HIP events (ms) | rocprofv3 kernel stats | HBM: algorithmic / launch, frac of 8 TB/s | "
       "PMC traffic, ratio to algorithmic | MFMA fp32-equivalent, frac of 157.3 | bf16 issued, frac of 2.5 PFLOP/s |",
       "|---|---|---|---|---|---|---|---|---|"] + rows + [""]
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
        "* `prefill_experiment.txt` — the obs zero fill as a launch of its own beside the policy launch (`ic3_obs_prefill`): "
        "`tools/exp/ws_probe.hip` (a store stream without vector-ALU work beside an MFMA stream: free), every fill geometry tried, "
        "`tools/exp/prefill_probe.py` (what the policy launch costs in its obs modes), a rocprofv3 timeline: slower than the "
        "in-launch fill in every form (DESIGN.md section 10).",
        "* `split_pacing_sweep.txt` — the speed-only pacing knobs of the in-launch fill re-swept with the split gate product.",
        "* `shader_clock.txt` — the shader clock while the gate loops run (`-DIC3_PS_TRACE_CLK` build): 1.72 GHz on PP-hard, "
        "1.91 GHz on TJ-hard; `tcp_tcc_counters_tj_hard.csv` — L1 / L2 PMC counters of the TJ-hard launch (179 cycles per L1 -> L2 read).",
        "* `ab_runs.txt` — the second half of the round: variant libraries A/B on one box per call (cell epilogue, staged "
        "activation split, wave priority by phase, what had no effect, what stamping HIP events on every launch costs).",
        "* `phase_trace_pp_hard.txt`, `phase_trace_tj_hard.txt` — per-tile phase timeline of the final kernel "
        "(`tools/build_variant.sh trace -DIC3_PS_TRACE`, `tools/analyze_trace.py`); `phase_trace_epilogue_*.txt`: the "
        "`-DIC3_PS_TRACE_EPI` build's stamps inside the cell epilogue.",
        "* `train_batch.txt`, `train_profile.txt` — `tools/bench_train.py` lines and the kernel table of one PP-hard update.",
        "* `host_asan.txt` — the product's `.hip` sources on the host under ASan + UBSan (`tools/host_asan.sh`), incl. round 4's "
        "`obs_fill.hip`, `ic3_commnet_step`, `ic3_heads_grad`, `ic3_env_set_hidden_out`.",
        "* `tests_gpu_summary.txt` — tail of `pytest -m gpu` on the final code."]
open(os.path.join(dst, 'README.md'), 'w').write("\n".join(out) + "\n")
print("\n".join(out[:20]))
