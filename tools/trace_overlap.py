"""Summarise a rocprofv3 kernel trace of one fit: per-kernel totals, union (wall) time and overlap.
usage: trace_overlap.py <kernel_trace.csv> [name-substring ...]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
pats = sys.argv[2:] or ["potrf", "gemm"]
sel = [r for r in rows if any(p in r["Kernel_Name"] for p in pats)]
# last fit only: take the second half of potrf launches
pot = [r for r in sel if "potrf" in r["Kernel_Name"]]
if len(pot) >= 2:
    half = pot[len(pot) // 2]
    t0 = int(half["Start_Timestamp"])
    sel = [r for r in sel if int(r["Start_Timestamp"]) >= t0]
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40], r.get("Queue_Id", "?")) for r in sel)
tot = sum(e - s for s, e, _, _ in iv)
union = 0
cur_s, cur_e = iv[0][0], iv[0][1]
for s, e, _, _ in iv[1:]:
    if s > cur_e:
        union += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
union += cur_e - cur_s
span = iv[-1][1] - iv[0][0]
print(f"kernels {len(iv)}  sum of durations {tot/1e6:.2f} ms  union {union/1e6:.2f} ms  span {span/1e6:.2f} ms  gaps {(span-union)/1e6:.2f} ms")
byq = {}
for s, e, n, q in iv:
    byq.setdefault((q, n), [0, 0])
    byq[(q, n)][0] += 1
    byq[(q, n)][1] += e - s
for k, v in sorted(byq.items()):
    print(f"   queue {k[0]} {k[1]:<42s} n={v[0]:4d}  {v[1]/1e6:7.2f} ms  avg {v[1]/v[0]/1e3:7.1f} us")
