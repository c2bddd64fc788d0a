#!/usr/bin/env python3
"""dev tool: VGPR + AGPR, scratch and occupancy of every kernel in a *.resource.txt (hipcc -Rpass-analysis=kernel-resource-usage), filtered by
a substring.  The sum VGPR + AGPR, rounded up to 8, is what a wave holds of its SIMD's 512 registers -- what decides which waves fit beside it.
usage: resource_regs.py FILE [substring]"""
import re, sys
txt = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
def short(n):
    m = re.match(r"_ZN2zk\d+([a-z_0-9]+)I", n)
    cur = "G2" if "3Fq2" in n else ("G1" if "FqParams" in n else "")
    q = re.search(r"EELi(\d)E", n)
    return "%s<%s%s>" % (m.group(1) if m else n[:40], cur, (", " + q.group(1)) if q else "")
for m in re.finditer(r"Function Name: (\S+).*?VGPRs: (\d+).*?AGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+).*?Occupancy \[waves/SIMD\]: (\d+)", txt, re.S):
    n = short(m.group(1))
    if pat in n:
        v, a = int(m.group(2)), int(m.group(3))
        print("%-36s VGPR %3d AGPR %3d  held %3d  scratch %4s  occ %s" % (n, v, a, (v + a + 7) // 8 * 8, m.group(4), m.group(5)))
