"""Busy time and idle gaps of the last learn() call in a rocprofv3 kernel trace (csv): usage trace_gaps.py trace.csv [ncalls]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ncalls = int(sys.argv[2]) if len(sys.argv) > 2 else 6
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('void ', '')) for r in rows), key=lambda e: e[0])
# split into calls at the largest gaps
gaps = sorted(((ev[i + 1][0] - ev[i][1], i) for i in range(len(ev) - 1)), reverse=True)[:ncalls]
cuts = sorted(i for _, i in gaps)
last = ev[cuts[-2] + 1:cuts[-1] + 1] if len(cuts) >= 2 else ev
t0, t1 = last[0][0], max(e[1] for e in last)
busy = sum(e[1] - e[0] for e in last)
print(f"last call: {len(last)} kernels, wall {(t1 - t0) / 1e6:.3f} ms, kernel time {busy / 1e6:.3f} ms, idle {(t1 - t0 - busy) / 1e6:.3f} ms")
per = collections.defaultdict(lambda: [0, 0])
for s, e, n in last:
    per[n][0] += e - s; per[n][1] += 1
for n, (t, c) in sorted(per.items(), key=lambda kv: -kv[1][0])[:14]:
    print(f"  {n[:60]:60s} {c:4d} x {t / c / 1e3:8.1f} us = {t / 1e6:7.3f} ms")
idle = sorted(((last[i + 1][0] - last[i][1], last[i][2], last[i + 1][2]) for i in range(len(last) - 1)), reverse=True)[:8]
for g, a, b in idle: print(f"  gap {g / 1e3:7.1f} us after {a[:40]} before {b[:40]}")
