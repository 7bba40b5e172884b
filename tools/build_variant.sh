#!/bin/bash
# usage: tools/build_variant.sh <conv_igemm source> <out .so> [extra hipcc flags]   -- A/B builds of the conv kernel
set -e
SRC=$1; OUT=$2; shift 2
D=dmel_codec_amd/csrc
mkdir -p /tmp/abbuild
cp $SRC /tmp/abbuild/conv_igemm_variant.hip
sed -i 's#"conv.h"#"'$PWD/$D'/conv.h"#' /tmp/abbuild/conv_igemm_variant.hip
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -x hip "$@" -c /tmp/abbuild/conv_igemm_variant.hip -o /tmp/abbuild/conv_variant.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT /tmp/abbuild/conv_variant.o dmel_codec_amd/build/common.o dmel_codec_amd/build/stft_logmel.o dmel_codec_amd/build/aa_snake.o dmel_codec_amd/build/conv_bwd.o dmel_codec_amd/build/train_ops.o dmel_codec_amd/build/small_ops.o dmel_codec_amd/build/modules.o
