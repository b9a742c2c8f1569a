#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats CSV: python tools/prof_summary.py <kernel_stats.csv> [steps]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tot = sum(float(r['TotalDurationNs']) for r in rows)
print("total kernel time %.2f ms over %g steps = %.2f ms/step" % (tot / 1e6, steps, tot / 1e6 / steps))
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    n = r['Name'].replace('igemm::kernel', 'K').replace('convop::', '').replace('igemm::', '').replace('(anonymous namespace)::', '')
    n = n.split('(')[0][:70]
    print("%6.2f%% %7.2f ms/step calls/step=%6.1f avg=%8.1f us  %s" % (100 * float(r['TotalDurationNs']) / tot, float(r['TotalDurationNs']) / 1e6 / steps, float(r['Calls']) / steps, float(r['AverageNs']) / 1e3, n))
