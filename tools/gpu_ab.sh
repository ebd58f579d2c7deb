#!/bin/bash
# same-session A/B of k_var variants.  usage: tools/gpu_ab.sh <tag> "ENV1=.. ENV2=.." "ENV.." ...
set -u
OUT=gpurun_out/$1; shift; mkdir -p "$OUT"
i=0
for cfg in "$@"; do
  i=$((i+1))
  env $cfg timeout -k 10 200 python bench.py --steps ${STEPS:-3} --warmup 1 --cpu-sample 0 ${BENCH_ARGS:-} > "$OUT/cfg$i.log" 2>&1
  rc=$?
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "config $cfg timed out: stopping"; exit $rc; fi
  python3 - "$OUT/cfg$i.log" "$cfg" <<'PY'
import json,sys
ok=False
for line in open(sys.argv[1]):
    if line.startswith('{'):
        d=json.loads(line); r=d['roofline']; ok=True
        print(f"[{sys.argv[2]}] value={d['value']:.0f} q/s  k_var {r['kernel_ms']:.1f} ms  {r['achieved']:.2f} TF ({r['frac']*100:.1f}%)  mean_jac {r['mean_jac_kernel_ms']:.2f} ms")
if not ok: print(f"[{sys.argv[2]}] FAILED"); print(open(sys.argv[1]).read()[-600:])
PY
done
