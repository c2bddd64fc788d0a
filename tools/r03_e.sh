export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r03_e
cd /tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/gpurun_out/r03_e/trace20 -- python $R/tools/dev_sync_trace_target.py 20 12 > $R/gpurun_out/r03_e/t20.txt 2>&1
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/gpurun_out/r03_e/trace18 -- python $R/tools/dev_sync_trace_target.py 18 12 > $R/gpurun_out/r03_e/t18.txt 2>&1
cd $R
python tools/sync_timeline.py gpurun_out/r03_e/trace20 > gpurun_out/r03_e/timeline20.txt 2>&1
python tools/sync_timeline.py gpurun_out/r03_e/trace18 > gpurun_out/r03_e/timeline18.txt 2>&1
rm -rf gpurun_out/r03_e/trace20 gpurun_out/r03_e/trace18
tail -2 gpurun_out/r03_e/t20.txt gpurun_out/r03_e/t18.txt
