"""Sum rocprofv3 --pmc counter rows per kernel name.  usage: pmc_summary.py <dir>"""
import csv
import glob
import sys
from collections import defaultdict

d = sys.argv[1]
files = glob.glob(d + "/**/*counter_collection*.csv", recursive=True)
acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
for f in files:
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "?")
        short = k.split("(")[0][-40:]
        c = row.get("Counter_Name")
        acc[short][c] += float(row.get("Counter_Value", 0))
        cnt[short][c] += 1
for k in sorted(acc, key=lambda k: -sum(acc[k].values()))[:4]:
    print(k)
    for c, v in acc[k].items():
        print(f"   {c}: total {v:.6g} over {cnt[k][c]} dispatches -> {v / cnt[k][c]:.6g} per dispatch")
