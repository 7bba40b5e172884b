#!/bin/bash
# usage: tools/build_variant.sh <out .so> [SRC=<file under csrc/>] [extra hipcc flags]   -- A/B builds of ONE source (default conv_igemm.hip) with -D
# switches, linked against the objects of the last regular build.  Load with DMEL_LIB=<out .so>.
set -e
OUT=$1; shift
SRC=conv_igemm.hip
if [[ "$1" == SRC=* ]]; then SRC=${1#SRC=}; shift; fi
D=dmel_codec_amd/csrc
TMP=$(mktemp -d /tmp/abbuild.XXXX)
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -x hip "$@" -c $D/$SRC -o $TMP/variant.o
objs=$(ls dmel_codec_amd/build/*.o | grep -v "/${SRC%.*}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT $TMP/variant.o $objs
rm -rf $TMP
