#!/bin/bash
# kernel timeline (start offset, duration, queue) of one N=8192 fit: tools/gpu_fit_timeline.sh <tag> [first-kernel-count]
set -u
OUT=gpurun_out/$1; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace -d "$OUT" -o trace --output-format csv -- python3 tools/fit_timing.py 8192 > "$OUT/run.log" 2>&1 || { tail -5 "$OUT/run.log"; exit 1; }
python3 - "$OUT" "${2:-60}" <<'PY'
import csv, glob, sys
out, n = sys.argv[1], int(sys.argv[2])
rows = list(csv.DictReader(open(glob.glob(out + "/**/trace_kernel_trace.csv", recursive=True)[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
grams = [i for i, r in enumerate(rows) if "k_gram" in r["Kernel_Name"]]
i0 = grams[-1]                                   # the last fit of the run
t0 = int(rows[i0]["Start_Timestamp"])
qs = {}
for r in rows[i0:i0 + n]:
    q = qs.setdefault(r["Queue_Id"], len(qs))
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    name = r["Kernel_Name"].split("(")[0].replace("gpt::", "")[:40]
    print(f"{s/1e3:9.1f} {e/1e3:9.1f} {(e-s)/1e3:7.1f} us  q{q} grid {r.get('Grid_Size_X', r.get('Grid_Size','?')):>7}  {name}")
PY
