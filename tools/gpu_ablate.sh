#!/bin/bash
# timing-only ablations of k_var: libraries built with -DGPT_ABL=n (csrc/build/libgpt_abl<n>.so), selected through
# GPT_HIP_LIB; outputs of the ablated builds are wrong on purpose (bench sanity check relaxed by GPT_BENCH_ABLATE).
set -u
OUT=gpurun_out/${1:-ablate}; mkdir -p "$OUT"
B=$PWD/gaussian_process_transportation_amd/csrc/build
make -C gaussian_process_transportation_amd/csrc ablate > "$OUT/build.log" 2>&1 || { echo "ablation build failed"; exit 1; }
for v in 0 1 2 3 4 5 6; do
  if [ $v -eq 0 ]; then LIB=$PWD/gaussian_process_transportation_amd/libgpt_hip.so; else LIB=$B/libgpt_abl$v.so; fi
  GPT_BENCH_ABLATE=1 GPT_HIP_LIB=$LIB timeout -k 10 200 python bench.py --steps 2 --warmup 1 --cpu-sample 0 > "$OUT/abl$v.log" 2>&1
  rc=$?
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "variant $v timed out: stopping"; exit $rc; fi
  python3 - "$OUT/abl$v.log" $v <<'PY'
import json,sys
for line in open(sys.argv[1]):
    if line.startswith('{'):
        d=json.loads(line); r=d['roofline']
        print(f"ABL={sys.argv[2]} k_var {r['kernel_ms']:.1f} ms")
PY
done
