#!/bin/bash
# dev tool: build libzkhip.so with extra compiler flags into variants/<name>/libzkhip.so (for same-box A/B runs: ZK_LIB=variants/<name>/libzkhip.so)
# usage: tools/build_variant.sh NAME "-DZK_NO_X2 ..." ["POSTPASS=0 ..." (make variables)]
set -e -o pipefail
ARCH=${ARCH:-gfx950}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; EXTRA=$2; MAKEVARS=$3
W=$ROOT/build/variants/$NAME
mkdir -p $W/ethsnarks_amd/csrc $ROOT/variants/$NAME
cp $ROOT/ethsnarks_amd/csrc/*.cpp $ROOT/ethsnarks_amd/csrc/*.hpp $ROOT/ethsnarks_amd/csrc/Makefile $W/ethsnarks_amd/csrc/
ln -sfn $ROOT/include $W/include; ln -sfn $ROOT/tools $W/tools
rm -f $W/ethsnarks_amd/libzkhip.so $ROOT/variants/$NAME/libzkhip.so       # never ship (or compare against) a library an earlier build left behind
# the whole build log is kept; make's status decides, the grep only trims what is shown
make -C $W/ethsnarks_amd/csrc -j4 ARCH=$ARCH HIPFLAGS="-O3 -std=c++17 --offload-arch=$ARCH -fPIC -Wall -Wno-unused-function -Wno-unused-value -ffp-contract=off $EXTRA" $MAKEVARS > $W/build.log 2>&1 \
    || { echo "build_variant: make failed for $NAME (log: $W/build.log)"; grep -i -B2 -A8 "error" $W/build.log | tail -n 60; exit 1; }
grep -i "strip_asm" $W/build.log || true
cp $W/ethsnarks_amd/libzkhip.so $ROOT/variants/$NAME/libzkhip.so
cp $W/ethsnarks_amd/csrc/msm_g1.resource.txt $W/ethsnarks_amd/csrc/msm_g2.resource.txt $ROOT/variants/$NAME/
python3 $ROOT/tools/resource_diff.py $ROOT/variants/$NAME/msm_g1.resource.txt - k_msm_accumulate
python3 $ROOT/tools/resource_diff.py $ROOT/variants/$NAME/msm_g2.resource.txt - k_msm_accumulate
