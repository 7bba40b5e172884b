#!/bin/bash
# usage: tools/build_variant.sh <out .so> [extra hipcc flags]   -- A/B builds of the conv kernel (conv_igemm.hip with -D switches), linked
# against the objects of the last regular build.  Load with DMEL_LIB=<out .so>.
set -e
OUT=$1; shift
D=dmel_codec_amd/csrc
TMP=$(mktemp -d /tmp/abbuild.XXXX)
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -x hip "$@" -c $D/conv_igemm.hip -o $TMP/conv_variant.o
objs=$(ls dmel_codec_amd/build/*.o | grep -v conv_igemm.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT $TMP/conv_variant.o $objs
rm -rf $TMP
