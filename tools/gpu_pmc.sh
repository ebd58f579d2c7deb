#!/bin/bash
# PMC passes for the dominant kernel (separate runs; --pmc never combined with tracing domains
# other than --kernel-trace).  usage: tools/gpu_pmc.sh <tag> [bench args...]
set -u
TAG=${1:-pmc}; shift || true
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
rocprofv3 -L > "$OUT/counters_list.txt" 2>&1
pass() {  # name, counters...
    local name=$1; shift
    echo "=== pmc pass $name: $*"
    timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$OUT/$name" -o pmc -- python3 bench.py --steps 1 --warmup 0 --cpu-sample 0 --no-secondary ${BENCH_ARGS:-} > "$OUT/$name.log" 2>&1
    local rc=$?
    echo "rc=$rc" >> "$OUT/$name.log"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pass $name timed out: stopping"; exit $rc; fi
    python3 tools/pmc_summary.py "$OUT/$name" || true
}
pass mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVES
pass waits SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass l2 TCC_HIT_sum TCC_MISS_sum
echo "=== done"
