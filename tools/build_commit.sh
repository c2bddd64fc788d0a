#!/bin/bash
# dev tool: build libzkhip.so of a given commit into variants/<name>/libzkhip.so (same-box A/B against history)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd $ROOT
REV=$1; NAME=$2
W=build/variants/$NAME; rm -rf $W; mkdir -p $W/ethsnarks_amd/csrc $W/tools variants/$NAME
for f in $(git ls-tree --name-only $REV ethsnarks_amd/csrc/); do git show $REV:$f > $W/$f; done
for f in $(git ls-tree -r --name-only $REV include/); do mkdir -p $W/$(dirname $f); git show $REV:$f > $W/$f; done
git show $REV:tools/strip_asm_nops.py > $W/tools/strip_asm_nops.py
make -C $W/ethsnarks_amd/csrc -j4 2>&1 | grep -i " error" || true
cp $W/ethsnarks_amd/libzkhip.so variants/$NAME/libzkhip.so
