"""Summarise a rocprofv3 --pmc SQ_* pass (counter_collection.csv) per kernel: what fraction of its wave-cycles a wave had a
VALU instruction in flight / waited for an instruction to issue / waited for anything (memory, barrier, ...).
Usage: sq_summary.py counter_collection.csv [substring ...]   (SQ_*_CYCLES style counters are in quad-cycles, ratios are unit-free)"""
import csv, collections, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
want = sys.argv[2:]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    n = re.sub(r'zk::', '', r['Kernel_Name']); n = re.sub(r'\(.*', '', n).replace('void ', '')
    n = n.replace('Curve<Field<FqParams> >', 'G1').replace('Curve<Fq2>', 'G2').replace(' >', '>')
    agg[n][r['Counter_Name']].append(float(r['Counter_Value']))
for n in sorted(agg):
    if want and not any(w in n for w in want):
        continue
    c = {k: sum(v) / len(v) for k, v in agg[n].items()}
    wc = c.get('SQ_WAVE_CYCLES', 0) or 1
    print("%-34s launches %3d INSTS_VALU=%.3e WAVE_CYCLES=%.3e  active_valu/wave=%.2f  wait_inst/wave=%.2f  wait_any/wave=%.2f" % (
        n[:34], len(next(iter(agg[n].values()))), c.get('SQ_INSTS_VALU', 0), wc, c.get('SQ_ACTIVE_INST_VALU', 0) / wc,
        c.get('SQ_WAIT_INST_ANY', 0) / wc, c.get("SQ_WAIT_ANY", 0) / wc))
