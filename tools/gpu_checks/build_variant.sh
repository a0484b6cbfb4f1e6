#!/bin/bash
# Builds the library of another commit next to the working one, for same-box A/B timing (box-to-box clocks differ by ~15 %).
# usage: tools/gpu_checks/build_variant.sh <commit> <tag>   ->  romanimpreprocess_amd/libromanhip_<tag>.so
set -e
REPO=$(cd "$(dirname "$0")/../.." && pwd)
C=$1; T=$2
W=$REPO/gpurun_out/tmp/variant_$T
rm -rf $W && mkdir -p $W
git -C $REPO archive $C romanimpreprocess_amd/csrc include | tar -x -C $W
make -C $W/romanimpreprocess_amd/csrc -j8 LIB=$REPO/romanimpreprocess_amd/libromanhip_$T.so 2>&1 | grep -E "error" || true
ls -la $REPO/romanimpreprocess_amd/libromanhip_$T.so
