R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_i
mkdir -p $O
./tools/ntt_bench 20 20 > $O/ntt_bench.txt 2>&1; cat $O/ntt_bench.txt
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "ntt or config or option2 or chain or golden or batch" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
for rep in 1 2 3; do for v in head default; do
  L=$R/variants/$v/libzkhip.so; [ $v = default ] && L=$R/ethsnarks_amd/libzkhip.so
  ZK_LIB=$L python bench.py --steps 30 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'], d['phases_ms_last_step'].get('compute_h'))"
done; done
for v in head default; do
  L=$R/variants/$v/libzkhip.so; [ $v = default ] && L=$R/ethsnarks_amd/libzkhip.so
  ZK_LIB=$L python tools/dev_kernel_exclusive.py 20 > $O/excl_$v.txt 2>&1
  echo "== $v"; grep "one proof\|ntt\|pointwise" $O/excl_$v.txt;  grep "phase timings" $O/excl_$v.txt | tr ' ' '\n' | grep "compute_h" | tr '\n' ' '; echo
done
ZK_LIB=$R/variants/head/libzkhip.so python tools/dev_sync_latency.py 18 20 > $O/sync_head.txt 2>&1; python tools/dev_sync_latency.py 18 20 > $O/sync_new.txt 2>&1; cat $O/sync_head.txt $O/sync_new.txt
