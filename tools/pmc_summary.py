#!/usr/bin/env python3
"""Aggregate a rocprofv3 --pmc counter_collection CSV per kernel:
   python tools/pmc_summary.py <counter_collection.csv> <COUNTER> > profiles/<name>.csv
Writes kernel, launches, total, per-launch (the counter's own unit: FETCH_SIZE / WRITE_SIZE are KB)."""
import csv, sys, collections
rows = csv.DictReader(open(sys.argv[1]))
want = sys.argv[2]
tot, cnt = collections.defaultdict(float), collections.defaultdict(int)
for r in rows:
    if r["Counter_Name"] != want:
        continue
    k = r["Kernel_Name"].split("(")[0]
    tot[k] += float(r["Counter_Value"]); cnt[k] += 1
w = csv.writer(sys.stdout)
w.writerow(["kernel", "launches", want + "_total", want + "_per_launch"])
for k in sorted(tot, key=lambda k: -tot[k]):
    w.writerow([k, cnt[k], round(tot[k], 1), round(tot[k] / cnt[k], 1)])
