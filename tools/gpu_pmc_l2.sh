#!/bin/bash
# FETCH_SIZE and L2 hit / miss of the dominant kernel for a choice of bench workloads (separate --pmc passes, kernel trace only).
# usage: tools/gpu_pmc_l2.sh <tag> "<bench args>" ["<bench args>" ...]
set -u
TAG=${1:-pmcl2}; shift
OUT=gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp
i=0
for args in "$@"; do
  i=$((i+1))
  for pass in "fetch FETCH_SIZE" "l2 TCC_HIT_sum TCC_MISS_sum"; do
    set -- $pass; name=$1; shift
    echo "=== [$args] pass $name"
    timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$OUT/w${i}_$name" -o pmc -- python3 bench.py --steps 1 --warmup 0 --cpu-sample 0 --no-secondary $args > "$OUT/w${i}_$name.log" 2>&1
    rc=$?
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timed out: stopping"; exit $rc; fi
    python3 tools/pmc_summary.py "$OUT/w${i}_$name" | head -4
  done
done
