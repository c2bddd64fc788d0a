#!/usr/bin/env python3
"""Timeline of the LAST proof in a rocprofv3 --kernel-trace (--memory-copy-trace) run of tools/dev_sync_trace_target.py:
every dispatch / copy with start and end relative to the proof's first event, its queue, and the critical-path gaps.
usage: sync_timeline.py <dir with *_kernel_trace.csv> [launches_per_proof_guess]"""
import csv, glob, os, re, sys
d = sys.argv[1]
kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
mt = glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True)
def short(n):
    n = n.replace('zk::Curve<zk::Field<zk::FqParams> >', 'G1').replace('zk::Curve<zk::Fq2>', 'G2')
    m = re.match(r'(?:void )?(?:zk::)?(?:\(anonymous namespace\)::)?([A-Za-z_0-9]+(?:<[^(]*>)?)', n)
    return m.group(1) if m else n[:40]
ev = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name']), "q%s" % r['Queue_Id']) for r in csv.DictReader(open(kt))]
if mt:
    for r in csv.DictReader(open(mt[0])):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), "copy %s" % r.get('Direction', ''), "dma"))
ev.sort()
# the last proof starts at the last k_sort_count of the witness sort that follows a gap: find the last G2 accumulate, walk back to the preceding H2D copy / first kernel after an idle gap > 200 us
g2 = [i for i, e in enumerate(ev) if e[2].startswith('k_msm_accumulate<G2')]
i = g2[-1]
while i > 0 and ev[i][0] - max(e[1] for e in ev[:i]) < 150000: i -= 1
t0 = ev[i][0]
last = ev[i:]
print("last proof: %d events, %.3f ms from first start to last end" % (len(last), (max(e[1] for e in last) - t0) / 1e6))
for s, e, n, q in last:
    print("%9.3f %9.3f  %8.1f us  %-5s %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e3, q, n))
