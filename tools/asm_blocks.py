"""dev tool: instruction mix per basic block of one kernel in a device assembly file (ethsnarks_amd/csrc/*.strip.s).
Usage: python tools/asm_blocks.py msm_g1.strip.s k_msm_accumulate Li1E [min_instructions]"""
import re, sys
from collections import Counter
path, *keys = sys.argv[1:]
mn = 30
if keys and keys[-1].isdigit():
    mn = int(keys.pop())
lines = open(path).read().split("\n")
start = [i for i, l in enumerate(lines) if l.startswith("_Z") and re.match(r"^\S+:", l) and all(k in l for k in keys)][0]
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
body = lines[start:end + 1]
print(lines[start][:100], "...", len(body), "lines")
blocks, cur = [], None
for i, l in enumerate(body):
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        cur = [m.group(1), Counter(), i]; blocks.append(cur); continue
    t = l.strip()
    if not t or t.startswith((".", ";", "/")) or t.endswith(":"):
        continue
    if cur is None:
        cur = ["entry", Counter(), i]; blocks.append(cur)
    cur[1][t.split()[0]] += 1
for b in blocks:
    tot = sum(b[1].values())
    if tot >= mn:
        mads = sum(v for k, v in b[1].items() if k.startswith("v_mad_u64"))
        print("%-10s line %6d  %5d instr  (%d v_mad_u64_u32)  " % (b[0], b[2], tot, mads), b[1].most_common(10))
for i, l in enumerate(body):
    if "s_cbranch" in l or "s_branch" in l:
        print(i, l.strip())
