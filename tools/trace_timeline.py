#!/usr/bin/env python3
"""Timeline view of a rocprofv3 --kernel-trace CSV of bench.py: for the steady-state window, how much wall time has an
accumulation kernel running, what runs when none does, and the per-kernel busy time.  Usage: trace_timeline.py kernel_trace.csv [skip_frac]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
def short(n):
    n = n.replace('zk::Curve<zk::Field<zk::FqParams> >', 'G1').replace('zk::Curve<zk::Fq2>', 'G2')
    m = re.match(r'(?:void )?(?:zk::)?([A-Za-z_0-9]+(?:<[^(]*>)?)', n)
    return m.group(1) if m else n[:40]
ev = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name']), int(r['Queue_Id'])) for r in rows]
ev.sort()
# steady-state window: G2 accumulations number FIRST .. LAST of the run (argv[2], argv[3]; default 8 .. 22: inside the timed
# steps of `bench.py --steps 20 --warmup 5`), start of the first to start of the last
g2 = [e for e in ev if e[2].startswith('k_msm_accumulate<G2')]
first = int(sys.argv[2]) if len(sys.argv) > 2 else 8
last = int(sys.argv[3]) if len(sys.argv) > 3 else 22
t0, t1 = g2[first][0], g2[last][0]
print("%d proofs in %.2f ms: %.3f ms per proof" % (last - first, (t1 - t0) / 1e6, (t1 - t0) / 1e6 / (last - first)))
win = [e for e in ev if e[0] >= t0 and e[0] < t1]
span = (t1 - t0) / 1e6
print("window %.2f ms, %d dispatches" % (span, len(win)))
def union(iv):
    iv = sorted(iv); tot = 0; cs, ce = None, None
    for s, e in iv:
        if cs is None: cs, ce = s, e
        elif s <= ce: ce = max(ce, e)
        else: tot += ce - cs; cs, ce = s, e
    if cs is not None: tot += ce - cs
    return tot
acc = [(s, e) for s, e, n, q in win if 'k_msm_accumulate' in n]
allk = [(s, e) for s, e, n, q in win]
print("some kernel running: %.1f%%   an accumulate kernel running: %.1f%%" % (100 * union(allk) / (t1 - t0), 100 * union(acc) / (t1 - t0)))
# time with no accumulate running: which kernels are running then
pts = sorted(set([t0, t1] + [x for s, e in acc for x in (s, e)]))
accu = sorted(acc)
def in_acc(t):
    return any(s <= t < e for s, e in accu)
gaps = []
for a, b in zip(pts, pts[1:]):
    if not in_acc((a + b) // 2): gaps.append((a, b))
from collections import defaultdict
busy = defaultdict(int)
for a, b in gaps:
    for s, e, n, q in win:
        o = min(e, b) - max(s, a)
        if o > 0: busy[n] += o
print("no accumulate running: %.2f ms in %d gaps; kernels running in those gaps (ms, overlapping counted each):" % (sum(b - a for a, b in gaps) / 1e6, len(gaps)))
for n, v in sorted(busy.items(), key=lambda x: -x[1])[:14]: print("   %-40s %8.3f" % (n, v / 1e6))
tot = defaultdict(lambda: [0, 0])
for s, e, n, q in win: tot[n][0] += e - s; tot[n][1] += 1
print("per-kernel total duration in window (ms), calls, avg us:")
for n, (v, c) in sorted(tot.items(), key=lambda x: -x[1][0])[:24]: print("   %-40s %8.3f %5d %9.1f" % (n, v / 1e6, c, v / c / 1e3))
