#!/bin/bash
# PMC passes over a probe binary (separate runs, --kernel-trace only beside --pmc).  usage: tools/gpu_pmc_probe.sh <tag> <kernel substring> -- <probe> [args...]
set -u
TAG=$1; KSUB=$2; shift 3
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
pass() {
    local name=$1; shift
    local ctrs="$1"; shift
    timeout -k 10 200 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d "$OUT/$name" -o pmc -- "$@" > "$OUT/$name.log" 2>&1
    local rc=$?
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pass $name timed out: stopping"; exit $rc; fi
    python3 - "$OUT/$name" "$KSUB" <<'PY'
import csv, glob, sys
d, ksub = sys.argv[1], sys.argv[2]
acc, cnt = {}, {}
for f in glob.glob(d + "/**/*counter_collection*.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if ksub not in row.get("Kernel_Name", ""): continue
        c = row["Counter_Name"]; acc[c] = acc.get(c, 0.0) + float(row["Counter_Value"]); cnt[c] = cnt.get(c, 0) + 1
for c in sorted(acc): print(f"{c:34s} {acc[c]/cnt[c]:18.1f}  (avg over {cnt[c]} dispatches)")
PY
}
pass sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE" "$@"
pass sq2 "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAVES" "$@"
pass l2 "TCC_HIT_sum TCC_MISS_sum" "$@"
pass fetch "FETCH_SIZE" "$@"
pass write "WRITE_SIZE" "$@"
pass ta "TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum" "$@"
