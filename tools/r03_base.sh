set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r03_base
python -m pytest tests -m gpu -x -q > $R/gpurun_out/r03_base/pytest.txt 2>&1
echo "pytest done" 
python bench.py --steps 20 --warmup 5 > $R/gpurun_out/r03_base/bench.json 2> $R/gpurun_out/r03_base/bench.err
echo "bench done"
python bench.py --steps 40 --warmup 5 --witness host --no-cpu-baseline > $R/gpurun_out/r03_base/bench_host.json 2> $R/gpurun_out/r03_base/bench_host.err
echo "bench host done"
cd /tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/gpurun_out/r03_base/trace_host -- python $R/bench.py --steps 20 --warmup 5 --witness host --no-cpu-baseline --no-extras > $R/gpurun_out/r03_base/bench_host_traced.json 2> $R/gpurun_out/r03_base/bench_host_traced.err
cd $R
ls -la gpurun_out/r03_base/trace_host/*/ || true
python tools/dev_sync_latency.py 18 20 > gpurun_out/r03_base/sync_latency.txt 2>&1
