R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_h
mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
for rep in 1 2 3; do for v in nopairs default; do
  L=$R/variants/$v/libzkhip.so; [ $v = default ] && L=$R/ethsnarks_amd/libzkhip.so
  ZK_LIB=$L python bench.py --steps 30 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
done; done
for v in nopairs default; do
  L=$R/variants/$v/libzkhip.so; [ $v = default ] && L=$R/ethsnarks_amd/libzkhip.so
  ZK_LIB=$L python tools/dev_kernel_exclusive.py 20 > $O/excl_$v.txt 2>&1
  echo "== $v"; grep "phase timings" $O/excl_$v.txt | tr ' ' '\n' | grep "acc_\|compute_h" | tr '\n' ' '; echo
done
