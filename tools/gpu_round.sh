#!/bin/bash
# One GPU-box session: parity tests -> bench -> rocprofv3 kernel trace.  A step that times out or is
# killed stops the chain (no further GPU work after a hang); an ordinary test failure does not.
# usage: tools/gpu_round.sh <tag> [bench args...]
set -u
TAG=${1:-run}; shift || true
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
run_step() {  # name, timeout, cmd...
    local name=$1 to=$2; shift 2
    echo "=== $name ($(date +%T))"
    timeout -k 10 "$to" "$@" > "$OUT/$name.log" 2>&1
    local rc=$?
    echo "rc=$rc" >> "$OUT/$name.log"
    echo "=== $name rc=$rc"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step $name timed out: stopping"; exit $rc; fi
    return 0
}
if [ "${SKIP_TESTS:-0}" != "1" ]; then
    run_step pytest_gpu 600 python -m pytest tests -m gpu -x -q
    tail -n 15 "$OUT/pytest_gpu.log"
fi
run_step bench 400 python bench.py "$@"
tail -n 3 "$OUT/bench.log"
if [ "${SKIP_PROF:-0}" != "1" ]; then
    export TMPDIR=/tmp
    run_step rocprof 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -o trace -- python3 bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-secondary "$@"
    find "$OUT/prof" -name "*kernel_stats*.csv" | head -1 | xargs -r head -n 20
    # keep only the small summaries (the per-dispatch trace can be large)
    find "$OUT/prof" -name "*kernel_trace*.csv" -size +2M -delete
fi
echo "=== done"
