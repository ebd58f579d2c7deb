#!/bin/bash
# kernel stats of the reference's 3-D demo flow (N = 2500 transport fit with the hyper-parameter search): where the
# time of an optimizer-driven fit goes, per kernel.   usage: tools/gpu_example_prof.sh <tag>
set -u
OUT=gpurun_out/$1; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats -d "$OUT" -o trace --output-format csv -- python3 examples/surface_3d.py > "$OUT/run.log" 2>&1 || { tail -5 "$OUT/run.log"; exit 1; }
tail -3 "$OUT/run.log"
python3 - "$OUT" <<'PY'
import csv, glob, sys
out = sys.argv[1]
f = glob.glob(out + "/**/trace_kernel_stats.csv", recursive=True)[0]
rows = list(csv.reader(open(f)))
print(rows[0])
tot = sum(float(r[2]) for r in rows[1:])
print(f"total kernel time {tot/1e6:.1f} ms")
for r in rows[1:16]:
    print(f"{r[0][:70]:70s} calls {r[1]:>6s} total {float(r[2])/1e6:8.2f} ms avg {float(r[3])/1e3:8.1f} us  {float(r[2])/tot*100:5.1f} %")
PY
