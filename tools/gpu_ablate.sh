#!/bin/bash
# timing-only ablations of k_var (GPT_VAR_ABLATE): what does each component cost?
set -u
OUT=gpurun_out/${1:-ablate}; mkdir -p "$OUT"
for v in ${VARIANTS:-0 1 2 3 4}; do
  GPT_VAR_ABLATE=$v timeout -k 10 200 python bench.py --steps 2 --warmup 1 --cpu-sample 0 > "$OUT/abl$v.log" 2>&1
  rc=$?
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "variant $v timed out: stopping"; exit $rc; fi
  python3 - "$OUT/abl$v.log" $v <<'PY'
import json,sys
for line in open(sys.argv[1]):
    if line.startswith('{'):
        d=json.loads(line); r=d['roofline']
        print(f"ABL={sys.argv[2]} k_var {r['kernel_ms']:.1f} ms  ({r['achieved']:.2f} TF-equivalent)")
PY
done
