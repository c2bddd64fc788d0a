R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_k
mkdir -p $O
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "msm or ntt or config3 or golden or batch" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
for rep in 1 2 3; do for v in ntt512 default; do
  L=$R/variants/$v/libzkhip.so; [ $v = default ] && L=$R/ethsnarks_amd/libzkhip.so
  ZK_LIB=$L python bench.py --steps 30 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'])"
done; done
for v in ntt512 default; do
  L=$R/variants/$v/libzkhip.so; [ $v = default ] && L=$R/ethsnarks_amd/libzkhip.so
  ZK_LIB=$L python tools/dev_kernel_exclusive.py 20 > $O/excl_$v.txt 2>&1
  echo "== $v"; grep "one proof\|ntt\|sort" $O/excl_$v.txt
done
