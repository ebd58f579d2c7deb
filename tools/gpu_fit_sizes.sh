#!/bin/bash
# One rocprofv3 kernel trace per model size: one fit + one objective evaluation each.
# usage: tools/gpu_fit_sizes.sh <tag> [N ...]      -> gpurun_out/<tag>/N<N>/..., summary in gpurun_out/<tag>/summary.txt
set -u
TAG=$1; shift
Ns=${@:-1024 2500 8192}
OUT=gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp
: > "$OUT/summary.txt"
for N in $Ns; do
  timeout -k 10 240 rocprofv3 --kernel-trace --stats -d "$OUT/N$N" -o trace --output-format csv -- python3 tools/fit_once.py $N > "$OUT/N$N.log" 2>&1
  rc=$?
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "N=$N timed out: stopping"; exit $rc; fi
  grep "fit phases" "$OUT/N$N.log" >> "$OUT/summary.txt"
  python3 tools/fit_profile_summary.py "$OUT/N$N" $N >> "$OUT/summary.txt" 2>&1
  # keep the trace small in gpurun_out: the per-kernel stats stay, the raw trace goes
  find "$OUT/N$N" -name "*kernel_trace.csv" -size +40M -delete
done
cat "$OUT/summary.txt"
