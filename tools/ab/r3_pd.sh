#!/bin/bash
# deep weight prefetch (DMEL_PD2) with the corrected end-of-step wait, with / without the next chunk's x loads at the top of the chunk (DMEL_XTOP)
for lib in dmel_codec_amd/libdmel_hip.so tools/ab/conv_pd4.so tools/ab/conv_pd6.so tools/ab/conv_pd4x.so tools/ab/conv_pd6x.so; do
  echo "== $lib"
  DMEL_LIB=$PWD/$lib DMEL_CONV_PC=0 timeout -k 10 200 python tools/bench_conv.py --iters 10 --precision 3 --check 2>/dev/null | grep -v "max err" 
  DMEL_LIB=$PWD/$lib timeout -k 10 200 python tools/bench_conv.py --iters 10 --precision 3 --check --only wn_dec_k3 2>/dev/null
  DMEL_LIB=$PWD/$lib timeout -k 10 200 python bench.py --steps 20 --warmup 5 --cpu-budget 0 --median-steps 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'], d['one_batch_at_a_time'], d['kernel_ms_per_step'])"
  DMEL_LIB=$PWD/$lib timeout -k 10 300 python tools/bench_stream.py --pipeline-only 2>/dev/null | grep '"batch": 1' | cut -c1-200
done
