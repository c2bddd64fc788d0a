# round-4 profile artifacts (one gpurun call, FINAL code): bench line, the same command under rocprofv3 --kernel-trace --stats, PMC traffic (separate
# passes, tied to the kernel sources by tools/pmc_summary.py), SQ counters, synchronous latency, one-at-a-time and 2^22 lines, small-circuit batch lines.
# Every run keeps its stderr under gpurun_out/r04_prof/*.err; a failed step ends the script (no GPU step after a failed one).
# usage (from the repo root, so that the commit is recorded):  gpurun --timeout 1200 -- "ZK_COMMIT=$(git rev-parse --short HEAD) bash tools/r04_profiles.sh"
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_prof
mkdir -p $O
cd $R
set -e
python bench.py --steps 20 --warmup 5 > $O/r04_bench.json 2> $O/bench.err
echo bench done
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 20 --warmup 5 > $O/r04_bench_under_rocprofv3.json 2> $O/stats.err
echo stats done
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --inflight 1 --no-extras --no-cpu-baseline > $O/pmc_fetch.json 2> $O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 2 --warmup 1 --inflight 1 --no-extras --no-cpu-baseline > $O/pmc_write.json 2> $O/pmc_write.err
echo pmc done
export ZK_SERIAL=1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/sq -- python3 $R/bench.py --steps 2 --warmup 1 --inflight 1 --no-extras --no-cpu-baseline > $O/sq.json 2> $O/sq.err
unset ZK_SERIAL
echo sq done
cd $R
F=$(find $O/pmc_fetch -name "*counter_collection.csv" | head -1); W=$(find $O/pmc_write -name "*counter_collection.csv" | head -1)
python tools/pmc_summary.py $F $W $O/r04_pmc_traffic.json > $O/pmc_summary.txt 2>&1
S=$(find $O/sq -name "*counter_collection.csv" | head -1)
python tools/sq_summary.py $S k_msm_accumulate k_msm_rowcol_sum k_msm_weighted_sum k_msm_bucket_finalize k_ntt_pass k_sort_partition k_sort_fine k_spmv_rows > $O/r04_sq_counters_body.txt 2>&1
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/r04_bench_kernel_stats.csv
python tools/dev_sync_latency.py merkle29 mimc11 14 16 18 20 > $O/r04_sync_latency.txt 2> $O/sync.err
python bench.py --steps 20 --warmup 3 --inflight 1 --no-extras --no-cpu-baseline > $O/r04_bench_inflight1.json 2> $O/inflight1.err
python bench.py --steps 10 --warmup 2 --logm 22 --no-extras --no-cpu-baseline > $O/r04_bench_2p22_1gpu.json 2> $O/2p22.err
for wl in merkle29 mimc11; do for b in 1 32 64; do
  python bench.py --workload $wl --batch $b --steps 30 --warmup 3 --witness resident --no-extras --no-cpu-baseline > $O/r04_bench_${wl}_batch$b.json 2> $O/${wl}_$b.err
done; done
rm -rf $O/stats $O/pmc_fetch $O/pmc_write $O/sq
ls -la $O
