#!/usr/bin/env python
"""Idle time between consecutive policy_step_kernel launches in a rocprofv3 --kernel-trace csv:  python tools/collect_gaps.py FILE
(eager launches vs hipGraph replays: where the 4 % of round 3's graph-mode line went)."""
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'policy_step_kernel' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
rows = rows[len(rows) // 2:]                   # the timed region (second half of the run)
gaps = [(int(b['Start_Timestamp']) - int(a['End_Timestamp'])) / 1e3 for a, b in zip(rows, rows[1:])]
gaps = sorted(g for g in gaps if g < 200.0)    # (episode boundaries have other launches in between)
durs = sorted((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows)
print("policy_step_kernel launches %d: duration median %.1f us; gap to the next launch median %.1f us, p10 %.1f, p90 %.1f, mean %.1f"
      % (len(rows), durs[len(durs) // 2], gaps[len(gaps) // 2], gaps[len(gaps) // 10], gaps[9 * len(gaps) // 10],
         sum(gaps) / len(gaps)))
