#!/bin/bash
# A/B of k_var scheduling variants in one GPU session (same device): GPT_VAR_SCHED=0..3
set -u
OUT=gpurun_out/${1:-variants}; mkdir -p "$OUT"
for v in ${VARIANTS:-0 1 2 3}; do
  GPT_VAR_SCHED=$v timeout -k 10 200 python bench.py --steps ${STEPS:-3} --warmup 1 --cpu-sample 0 ${BENCH_ARGS:-} > "$OUT/sched$v.log" 2>&1
  rc=$?
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "variant $v timed out: stopping"; exit $rc; fi
  python3 - "$OUT/sched$v.log" $v <<'PY'
import json,sys
for line in open(sys.argv[1]):
    if line.startswith('{'):
        d=json.loads(line); r=d['roofline']
        print(f"SCHED={sys.argv[2]} value={d['value']:.0f} q/s  k_var {r['kernel_ms']:.1f} ms  {r['achieved']:.2f} TF ({r['frac']*100:.1f}%)  mean_jac {r['mean_jac_kernel_ms']:.2f} ms")
PY
done
