"""Turn the PMC passes of tools/gpu_pmc.sh (run on the GPU box, merged back under gpurun_out/) into the record bench.py
quotes as `roofline.traffic`: HBM-side bytes of the dominant kernel per prediction call (all its launches of one step).

    python tools/pmc_traffic.py gpurun_out/<tag> j 8192 500000 [commit]   # key (j | jvar | svgp), n_source, queries per launch,
                                                                          # commit the GPU run was made from (default: HEAD)

FETCH_SIZE / WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE reports exactly half of the bytes of wide (16 B per lane)
coalesced streaming reads — /opt/skills/guides/MI355X_MICROARCH.md, section HBM: "double it before comparing with a byte
count"; every read of this kernel (A fragments, scratch image) is such a read, so FETCH is doubled; WRITE_SIZE is exact
for 16-B-per-lane stores.  The record carries the workload and the commit it was measured on; bench.py quotes it only
for that workload."""
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def counters(d, kernel_substr):
    acc, cnt = {}, {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection*.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if kernel_substr not in row.get("Kernel_Name", ""):
                continue
            c = row["Counter_Name"]
            acc[c] = acc.get(c, 0.0) + float(row["Counter_Value"])
            cnt[c] = cnt.get(c, 0) + 1
    # the passes run ONE step (bench.py --steps 1 --warmup 0): the sum over the dispatches of that step — k_var goes out as several
    # launches of 16 rounds since round 3 — is the figure per prediction call ("per launch" of the hot path)
    return acc


def main():
    d, key, n_source, queries = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    c = counters(d, "k_var<")
    need = ["FETCH_SIZE", "WRITE_SIZE"]
    if any(k not in c for k in need):
        raise SystemExit(f"missing counters in {d}: have {sorted(c)}")
    rec = {"n_source": n_source, "queries": queries,
           "hbm_bytes_per_launch": (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0,
           "fetch_size_kib": c["FETCH_SIZE"], "write_size_kib": c["WRITE_SIZE"],
           "commit": sys.argv[5] if len(sys.argv) > 5 else
           subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip(),
           "source": os.path.relpath(d, ROOT)}
    if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c:
        rec["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c:
        # busy cycles are summed over the SIMDs (1024), GRBM_GUI_ACTIVE over the 8 XCDs
        rec["mfma_busy"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / (c["GRBM_GUI_ACTIVE"] / 8.0)
        rec["gui_active_cycles_per_xcd"] = c["GRBM_GUI_ACTIVE"] / 8.0
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    allrec = json.load(open(path)) if os.path.exists(path) else {}
    allrec["note"] = ("HBM-side bytes per launch of the dominant kernel from rocprofv3 --pmc (separate passes, tools/gpu_pmc.sh): "
                      "2 x FETCH_SIZE (gfx950 wide-read correction, MI355X_MICROARCH.md section HBM) + WRITE_SIZE, KiB -> bytes; "
                      "each record names the workload and commit it was measured on (tools/pmc_traffic.py)")
    allrec[key] = rec
    json.dump(allrec, open(path, "w"), indent=1)
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main()
