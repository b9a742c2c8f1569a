"""Timeline of one overlapped step from a rocprofv3 kernel trace: python tools/trace_overlap.py <kernel_trace.csv>"""
import csv, collections, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r['Queue_Id']) for r in rows)
ad = [i for i, e in enumerate(ev) if 'adam_kernel' in e[2]]
a, b = ad[len(ad) // 2], ad[len(ad) // 2 + 1]
seg = ev[a + 1:b + 1]
t0 = seg[0][0]
qs = collections.Counter(e[3] for e in seg)
mainq = qs.most_common(1)[0][0]
side = [e for e in seg if e[3] != mainq]; main = [e for e in seg if e[3] == mainq]
print("step wall %.2f ms; main busy %.2f ms; side busy %.2f ms (%d kernels)" % ((seg[-1][1] - t0) / 1e6, sum(e[1] - e[0] for e in main) / 1e6, sum(e[1] - e[0] for e in side) / 1e6, len(side)))
tb = [e for e in main if 'WgradX' in e[2] or 'Im2colDgrad' in e[2]]
sb = [e for e in side if 'bwd' in e[2] or 'wgrad' in e[2] or 'narrow' in e[2]]
sf = [e for e in side if e not in sb]
print("trunk backward on main: %.2f -> %.2f ms" % ((tb[0][0] - t0) / 1e6, (tb[-1][1] - t0) / 1e6))
if sf: print("encoder forward on side: %.2f -> %.2f ms" % ((sf[0][0] - t0) / 1e6, (sf[-1][1] - t0) / 1e6))
if sb: print("encoder backward on side: %.2f -> %.2f ms, busy %.2f ms" % ((sb[0][0] - t0) / 1e6, (sb[-1][1] - t0) / 1e6, sum(e[1] - e[0] for e in sb) / 1e6))
prev = main[0]
for e in main[1:]:
    g = e[0] - prev[1]
    if g > 100e3: print("  main idle %.0f us at %.2f ms" % (g / 1e3, (prev[1] - t0) / 1e6))
    prev = e
