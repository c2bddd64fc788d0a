#!/bin/bash
# round 4, call B: the FP64 52-bit-limb multiplier substrate, measured (tools/fp52.hpp, tools/mulbench.cpp) beside the shipped integer form on one box
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_b
mkdir -p $O
cd $R
timeout -k 10 300 ./tools/microbench > $O/microbench.txt 2> $O/microbench.err || { tail $O/microbench.err; exit 1; }
head -n 10 $O/microbench.txt
timeout -k 10 600 ./tools/mulbench r04 > $O/mulbench.txt 2> $O/mulbench.err || { tail $O/mulbench.err; exit 1; }
cat $O/mulbench.txt
