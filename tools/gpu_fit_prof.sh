#!/bin/bash
# kernel trace of the fit at N=1024 and N=8192; prints per-kernel stats and the k_potrf_step durations by position in the panel
set -u
OUT=gpurun_out/$1; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats -d "$OUT" -o trace --output-format csv -- python3 tools/fit_timing.py 1024 8192 > "$OUT/run.log" 2>&1 || { tail -5 "$OUT/run.log"; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, sys
out = sys.argv[1]
f = glob.glob(out + "/**/trace_kernel_stats.csv", recursive=True)[0]
for r in list(csv.reader(open(f)))[:9]:
    print(r[0][:48], r[1], r[2], r[3], r[5], r[6])
rows = list(csv.DictReader(open(glob.glob(out + "/**/trace_kernel_trace.csv", recursive=True)[0])))
st = [r for r in rows if "k_potrf_step" in r["Kernel_Name"]]
fu = [r for r in rows if "k_potrf_step_update" in r["Kernel_Name"]]
dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
if st:
    print("N=1024 steps:", " ".join(f"{dur(r):.1f}" for r in st[:16]))
    last = st[-128:]
    print("N=8192 first 8:", " ".join(f"{dur(r):.1f}" for r in last[:8]), " mid:", " ".join(f"{dur(r):.1f}" for r in last[60:68]),
          " last 8:", " ".join(f"{dur(r):.1f}" for r in last[-8:]), f" sum {sum(dur(r) for r in last)/1e3:.2f} ms")
    if fu:
        lf = [r for r in last if "k_potrf_step_update" in r["Kernel_Name"]]
        print("fused launches (N=8192):", len(lf), " durations:", " ".join(f"{dur(r):.0f}" for r in lf))
    t0 = int(last[0]["Start_Timestamp"]); t1 = int(last[-1]["End_Timestamp"])
    print(f"span first..last step {(t1-t0)/1e6:.2f} ms")
    gm = [r for r in rows if "k_gemm<true" in r["Kernel_Name"] and int(r["Start_Timestamp"]) >= t0 and int(r["End_Timestamp"]) <= t1 + 2_000_000]
    byq = {}
    for r in gm: byq.setdefault(r["Queue_Id"], []).append(dur(r))
    for q, v in byq.items(): print(f"queue {q}: {len(v)} gemms, sum {sum(v)/1e3:.2f} ms, avg {sum(v)/len(v):.1f} us")
PY
