#!/usr/bin/env python
"""Per-kernel average of one PMC counter from a rocprofv3 --pmc run:  python tools/collect_pmc.py DIR COUNTER
(the *_counter_collection.csv under DIR; one row per dispatch and counter).  WRITE_SIZE / FETCH_SIZE are reported in KiB
per launch as rocprofv3 gives them (gfx950: FETCH_SIZE counts 32-byte units against the documented 64 — the bench's
traffic figure doubles it, /opt/skills/guides/MI355X_MICROARCH.md)."""
import collections
import csv
import glob
import os
import sys

d, counter = sys.argv[1], sys.argv[2]
files = sorted(glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True), key=os.path.getmtime)
if not files:
    sys.exit("no counter_collection.csv under " + d)
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(files[-1])):
    if r['Counter_Name'] == counter:
        k = r['Kernel_Name']
        agg[k][0] += 1
        agg[k][1] += float(r['Counter_Value'])
print("counter,kernel,launches,avg_per_launch")
for k, (n, tot) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%s,%s,%d,%.1f" % (counter, k[:90].replace(',', ';'), n, tot / n))
