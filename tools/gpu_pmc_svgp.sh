#!/bin/bash
# L2 hit rate and wait cycles of k_var<float,4> with and without its diagonal tiles (timing-only ablation
# build GPT_ABL=3), at Z = 2048 and Z = 4096.  Separate --pmc passes, --kernel-trace only beside them.
# usage: tools/gpu_pmc_svgp.sh <tag>     (needs csrc/build/libgpt_abl3.so: tools/build_variants.sh abl3:gpt_predict:"-DGPT_ABL=3")
set -u
TAG=$1
OUT=gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp
ABL=$PWD/gaussian_process_transportation_amd/csrc/build/libgpt_abl3.so
pass() {  # name lib Z counters...
    local name=$1 lib=$2 Z=$3; shift 3
    if [ -n "$lib" ]; then export GPT_HIP_LIB=$lib GPT_BENCH_ABLATE=1; else unset GPT_HIP_LIB GPT_BENCH_ABLATE; fi
    timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$OUT/$name" -o pmc -- python3 bench.py --config svgp --inducing $Z --steps 1 --warmup 0 --cpu-sample 0 > "$OUT/$name.log" 2>&1
    local rc=$?
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pass $name timed out: stopping"; exit $rc; fi
    echo "=== $name (Z=$Z, ${lib:+diagonal tiles ablated}${lib:-shipped})"
    python3 tools/pmc_summary.py "$OUT/$name" | head -12
}
for Z in 2048 4096; do
  pass l2_z$Z "" $Z TCC_HIT_sum TCC_MISS_sum
  pass l2_abl_z$Z "$ABL" $Z TCC_HIT_sum TCC_MISS_sum
  pass wait_z$Z "" $Z SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES
  pass wait_abl_z$Z "$ABL" $Z SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES
  pass mfma_z$Z "" $Z SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
  pass mfma_abl_z$Z "$ABL" $Z SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
done
# run times of the two builds (no profiler)
for Z in 2048 4096; do
  unset GPT_HIP_LIB GPT_BENCH_ABLATE
  python3 bench.py --config svgp --inducing $Z --steps 3 --warmup 1 --cpu-sample 0 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']; print('shipped Z=$Z: k_var', r['kernel_ms'], 'ms', r['achieved'], 'TFLOP/s', r['frac'])"
  GPT_HIP_LIB=$ABL GPT_BENCH_ABLATE=1 python3 bench.py --config svgp --inducing $Z --steps 3 --warmup 1 --cpu-sample 0 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']; print('no diagonal tiles Z=$Z: k_var', r['kernel_ms'], 'ms')"
done
