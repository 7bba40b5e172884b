#!/bin/bash
# AddressSanitizer build of the HOST side of libdmel_hip.so on a fake HIP runtime (no GPU needed; GPU sanitizers are not available on
# this pool).  Usage: bash tools/asan/run.sh [build-dir]
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
OUT=${1:-/tmp/dmel_asan}
mkdir -p "$OUT"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="-O1 -g -fsanitize=address -fno-omit-frame-pointer -fPIC -std=c++17"
objs=""
for f in common.cpp stft_logmel.hip stft_bwd.hip conv_igemm.hip conv_pc.hip conv_snake.hip conv_bwd.hip train_ops.hip aa_snake.hip small_ops.hip wavenet_fused.hip modules.hip; do
  o="$OUT/${f%.*}.o"
  $HIPCC $FLAGS --offload-host-only -x hip -c "$ROOT/dmel_codec_amd/csrc/$f" -o "$o"
  objs="$objs $o"
done
# the host objects reference the (absent) device images by per-file hashed names: define them as empty blobs
nm -u $objs | awk '/__hip_fatbin_/ {print $2}' | sort -u | awk '{print "extern \"C\" { char " $1 "[16] = {0}; }"}' > "$OUT/fatbins.cpp"
g++ -O1 -g -fsanitize=address -fno-omit-frame-pointer -std=c++17 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include \
    "$ROOT/tools/asan/driver.cpp" "$ROOT/tools/asan/fake_hip.cpp" "$OUT/fatbins.cpp" $objs -include functional -o "$OUT/driver" -lpthread
ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 "$OUT/driver"
