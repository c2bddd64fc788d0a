#!/usr/bin/env python3
"""Assembly post-pass of the build (ethsnarks_amd/csrc/Makefile): drops the `s_nop 0` hipcc pads behind an inline-asm
statement whose result the next instruction reads.

hipcc cannot see into an asm statement, assumes it may end in an SDWA / op_sel write (gfx940 dst_sel forwarding hazard,
one wait state), and counts the statement itself as zero wait states -- so a chain of dependent asm statements, which is
what the field arithmetic is (fips_asm.hpp), gets one s_nop per statement: 2-4 % of the accumulation kernels' time
(DESIGN section 4).  The statements of fips_asm.hpp hold only plain VOP2 / VOP3 integer instructions (v_mad_u64_u32,
v_addc_co_u32, v_add/sub/and/cndmask/mov): no SDWA, no op_sel, no transcendental, no DPP -- nothing that needs the wait
state -- and every statement is at least one VALU instruction long, so a one-wait-state hazard between an instruction
BEFORE the statement and one after it is satisfied by the statement itself.
Rules: only `s_nop 0` lines that directly follow `;;#ASMEND` go; not in functions that contain a DPP, SDWA, permlane or
swizzle instruction anywhere (those hazards need two wait states and may count this nop as one of them); and not when the
next instruction is a lane read / write (v_readfirstlane, v_readlane, v_writelane: one wait state behind a VALU write of
the register they read)."""
import re
import sys

src = open(sys.argv[1]).read().split("\n")
RISKY = re.compile(r"\b(dpp|row_shr|row_shl|quad_perm|row_bcast|sdwa|v_permlane|ds_swizzle|ds_bpermute|ds_permute)")
LANE = re.compile(r"^\s*(v_readlane|v_readfirstlane|v_writelane)")
n = len(src)
start = None
risky_line = [None] * n                      # None: outside any function
risky_funcs = 0
for k, line in enumerate(src):
    if re.match(r"^[A-Za-z_][\w$.]*:\s*(;.*)?$", line) and not line.startswith(".L"):
        start = k
    if line.startswith(".Lfunc_end") and start is not None:
        risky = any(RISKY.search(l) for l in src[start:k] if not l.lstrip().startswith(";"))
        for j in range(start, k + 1):
            risky_line[j] = risky
        risky_funcs += risky
        start = None
def next_instruction(k):
    for j in range(k + 1, min(k + 40, n)):
        t = src[j].strip()
        if t and not t.startswith(";") and not t.startswith(".") and not t.endswith(":"):
            return src[j]
    return ""
out, dropped = [], 0
for k, line in enumerate(src):
    if line.strip() == "s_nop 0" and k > 0 and src[k - 1].strip() == ";;#ASMEND" and risky_line[k] is False and not LANE.match(next_instruction(k)):
        dropped += 1
        continue
    out.append(line)
open(sys.argv[2], "w").write("\n".join(out))
print("strip_asm_nops: %d s_nop dropped, %d functions left untouched (DPP / lane ops)" % (dropped, risky_funcs), file=sys.stderr)
