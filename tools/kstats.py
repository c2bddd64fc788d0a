"""dev tool: summarise a rocprofv3 kernel_trace.csv by kernel (+ grid) -> avg/min us, calls"""
import csv, re, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
div = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
agg = collections.defaultdict(list)
for r in rows:
    n = re.sub(r'zk::|\(anonymous namespace\)::', '', r['Kernel_Name']); n = re.sub(r'\(.*', '', n); n = n.replace('void ', '')
    n = n.replace('Curve<Field<FqParams> >', 'G1').replace('Curve<Fq2>', 'G2')
    agg[(n, r['Grid_Size_X'])].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
tot = sum(sum(v) for v in agg.values())
print("total kernel time %.2f ms (/%g = %.2f ms)" % (tot / 1e3, div, tot / 1e3 / div))
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print("%-44s grid=%-9s n=%4d avg=%9.1f us min=%9.1f  sum/div=%8.2f ms" % (k[0][:44], k[1], len(v), sum(v) / len(v), min(v), sum(v) / 1e3 / div))
