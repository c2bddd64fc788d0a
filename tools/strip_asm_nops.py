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
import collections
import re
import sys

src = open(sys.argv[1]).read().split("\n")
RISKY = re.compile(r"\b(dpp|row_shr|row_shl|quad_perm|row_bcast|sdwa|v_permlane|ds_swizzle|ds_bpermute|ds_permute)")
LANE = re.compile(r"^\s*(v_readlane|v_readfirstlane|v_writelane)")
# The premise is CHECKED, not assumed: every ;;#ASMSTART .. ;;#ASMEND body must consist of these mnemonics only (plain
# VOP2 / VOP3 integer instructions, no modifiers); a function whose asm holds anything else keeps all its pads.
ALLOWED = {"v_mad_u64_u32", "v_addc_co_u32_e64", "v_addc_co_u32_e32", "v_add_co_u32_e32", "v_sub_co_u32_e32", "v_subb_co_u32_e32",
           "v_cndmask_b32_e32", "v_cndmask_b32_e64", "v_and_b32_e32", "v_mov_b32_e32"}
MODIFIER = re.compile(r"\b(op_sel|dst_sel|src0_sel|src1_sel|clamp|omod|mul:|div:|neg_lo|neg_hi|byte_sel)")
# A pad also stays when the instruction after it is a memory store (VMEM / FLAT / DS / scratch write of a just-written VGPR:
# the > 64-bit store-data hazard counts wait states too), a v_accvgpr_* move or an MFMA: none of them follows a pad in today's
# code objects (see the histogram this script prints), so the rule costs nothing and a compiler update cannot expose the hazard silently.
KEEP_BEFORE = re.compile(r"^\s*(global_store|flat_store|buffer_store|scratch_store|ds_write|ds_store|global_atomic|flat_atomic|buffer_atomic|ds_add|v_accvgpr|v_mfma|v_smfmac)")
n = len(src)
start = None
risky_line = [None] * n                      # None: outside any function
risky_funcs = 0
bad_asm = collections.Counter()
for k, line in enumerate(src):
    if re.match(r"^[A-Za-z_][\w$.]*:\s*(;.*)?$", line) and not line.startswith(".L"):
        start = k
    if line.startswith(".Lfunc_end") and start is not None:
        body = [l for l in src[start:k] if not l.lstrip().startswith(";") or l.strip().startswith(";;#ASM")]
        risky = any(RISKY.search(l) for l in body if not l.strip().startswith(";;#ASM"))
        inside = False
        for l in src[start:k]:                # the asm bodies of this function against the allow-list
            t = l.strip()
            if t == ";;#ASMSTART": inside = True; continue
            if t == ";;#ASMEND": inside = False; continue
            if inside and t and not t.startswith(";"):
                mn = t.split()[0]
                if mn not in ALLOWED or MODIFIER.search(t):
                    bad_asm[mn] += 1
                    risky = True
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
out, dropped, kept = [], 0, 0
followers = collections.Counter()
for k, line in enumerate(src):
    if line.strip() == "s_nop 0" and k > 0 and src[k - 1].strip() == ";;#ASMEND" and risky_line[k] is False:
        nxt = next_instruction(k)
        if LANE.match(nxt) or KEEP_BEFORE.match(nxt):
            kept += 1
        else:
            dropped += 1
            followers[nxt.split()[0] if nxt.split() else "(end)"] += 1
            continue
    out.append(line)
open(sys.argv[2], "w").write("\n".join(out))
print("strip_asm_nops: %d s_nop dropped, %d kept before a lane / store / accvgpr instruction, %d functions left untouched (DPP / lane ops / asm outside the allow-list%s)"
      % (dropped, kept, risky_funcs, (": " + ", ".join("%s x%d" % kv for kv in bad_asm.most_common(6))) if bad_asm else ""), file=sys.stderr)
print("strip_asm_nops: what follows the dropped pads: " + ", ".join("%s %d" % kv for kv in followers.most_common(12)), file=sys.stderr)
