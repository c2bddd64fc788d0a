export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $R/gpurun_out/r03_w/sq -- python3 $R/tools/dev_wplan_time.py > $R/gpurun_out/r03_w/out.txt 2>&1
cd $R
F=$(find gpurun_out/r03_w/sq -name "*counter_collection.csv" | head -1)
python - <<PY
import csv, collections
rows=[r for r in csv.DictReader(open("$F")) if 'k_witness_tape' in r['Kernel_Name']]
by=collections.defaultdict(list)
for r in rows: by[(r['Dispatch_Id'], r['Counter_Name'])].append(float(r['Counter_Value']))
disp=sorted(set(d for d,_ in by), key=int)
for d in disp:
    c={n: sum(v) for (dd,n),v in by.items() if dd==d}
    print(d, {k: round(v/21345,1) for k,v in c.items()})
PY
rm -rf gpurun_out/r03_w/sq
