#!/usr/bin/env python3
"""dev tool: VGPRs / scratch / occupancy per kernel from the *.resource.txt files the csrc Makefile writes; with two
directories (or files) prints old -> new.  usage: resource_diff.py NEW [OLD] [name filter]"""
import re, subprocess, sys
def parse(fn):
    out, cur = {}, None
    for line in open(fn, errors="replace"):
        m = re.search(r'Function Name: (\S+)', line)
        if m:
            cur = m.group(1); out[cur] = {}
            continue
        m = re.search(r'remark:\s+(VGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|VGPRs Spill):\s*(\d+)', line)
        if m and cur:
            out[cur]["Spill" if "Spill" in m.group(1) else m.group(1).split()[0]] = int(m.group(2))
    return out
def demangle(names):
    r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    fix = lambda d: re.sub(r'\(.*', '', d.replace('zk::Curve<zk::Field<zk::FqParams> >', 'G1').replace('zk::Curve<zk::Fq2>', 'G2').replace('void ', '').replace('zk::', ''))
    return dict(zip(names, map(fix, r)))
new = parse(sys.argv[1])
old = parse(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2] != "-" else {}
flt = sys.argv[3] if len(sys.argv) > 3 else ""
dm = demangle(list(new))
for f, v in new.items():
    if flt and flt not in dm[f]: continue
    o = old.get(f)
    s = "%-46s VGPR %3d spill %3d scratch %4d occ %d" % (dm[f][:46], v.get('VGPRs', -1), v.get('Spill', -1), v.get('ScratchSize', -1), v.get('Occupancy', -1))
    if o: s += "   (was VGPR %3d spill %3d scratch %4d occ %d)" % (o.get('VGPRs', -1), o.get('Spill', -1), o.get('ScratchSize', -1), o.get('Occupancy', -1))
    print(s)
