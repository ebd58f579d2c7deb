#!/bin/bash
# same-session comparison of Cholesky variants: tools/gpu_fit_ab.sh <tag> "ENV.." "ENV.." ...
set -u
OUT=gpurun_out/$1; shift; mkdir -p "$OUT"
for cfg in "$@"; do
  env $cfg timeout -k 10 240 python tools/fit_timing.py ${FIT_NS:-1024 2500 8192} 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/fit.log"
  rc=${PIPESTATUS[0]}
  if [ $rc -ne 0 ]; then echo "config [$cfg] failed rc=$rc: stopping"; exit $rc; fi
done
