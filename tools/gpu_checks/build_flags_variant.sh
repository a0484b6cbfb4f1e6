#!/bin/bash
# Builds the WORKING tree's library with extra compiler flags next to the default build, for same-box A/B timing:
#   tools/gpu_checks/build_flags_variant.sh <tag> <flags...>   ->  romanimpreprocess_amd/libromanhip_<tag>.so
set -e
REPO=$(cd "$(dirname "$0")/../.." && pwd)
T=$1; shift
W=/tmp/rip_variant_$T
rm -rf $W && mkdir -p $W/romanimpreprocess_amd $W/include
cp -r $REPO/romanimpreprocess_amd/csrc $W/romanimpreprocess_amd/csrc
cp $REPO/include/*.h $W/include/
rm -f $W/romanimpreprocess_amd/csrc/*.o
make -C $W/romanimpreprocess_amd/csrc -j8 LIB=$REPO/romanimpreprocess_amd/libromanhip_$T.so EXTRA="$*" $MAKEVARS 2>&1 | grep -E "error|Error" || true
ls -la $REPO/romanimpreprocess_amd/libromanhip_$T.so
