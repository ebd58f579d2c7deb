"""Per-size summary of a rocprofv3 kernel trace of tools/fit_once.py: the LAST fit and the LAST objective evaluation of
the trace (the last two k_gram launches open them), kernels grouped by role.  Writes <out>/fit_N<N>_stats.csv (per kernel
name: calls, total us, average us) and prints the summary lines that go to profiles/r04_fit_summary.txt.
usage: python tools/fit_profile_summary.py <rocprof dir> <N>"""
import csv
import glob
import sys

out, N = sys.argv[1], int(sys.argv[2])
rows = list(csv.DictReader(open(glob.glob(out + "/**/*kernel_trace.csv", recursive=True)[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
grams = [i for i, r in enumerate(rows) if "k_gram" in r["Kernel_Name"]]
segs = {"fit": rows[grams[-2]:grams[-1]], "objective": rows[grams[-1]:]}
# k_scale_x precedes k_gram in a fit: it belongs to the segment it opens
NP = (N + 511) // 512 * 512
chol_flops, inv_flops, kinv_flops = NP ** 3 / 3, NP ** 3 / 3, NP ** 3 / 3 * 1.0     # K^-1 = W^T W on the lower triangle: N^3/3


def role(name):
    if "k_potrf_step" in name: return "chain (k_potrf_step)"
    if "k_potrf_finish" in name: return "finish"
    if "k_gemm<true" in name or "k_gemm<1" in name or "ILb1ELb0E" in name: return "cholesky GEMMs (BT)"
    if "ILb0ELb1E" in name or "k_gemm<false, true" in name: return "K^-1 = W^T W"
    if "k_gemm" in name: return "inverse GEMMs"
    if "k_gram" in name or "k_scale_x" in name: return "gram"
    if "k_alpha" in name: return "alpha"
    if "k_pack_w" in name or "k_store4" in name: return "pack"
    if "k_lml" in name or "k_sum_partials" in name or "k_logdet" in name or "k_dot" in name: return "lml terms"
    return "other"


dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for seg, rs in segs.items():
    rs = [r for r in rs if "k_" in r["Kernel_Name"]]
    if not rs:
        continue
    span = (int(rs[-1]["End_Timestamp"]) - int(rs[0]["Start_Timestamp"])) / 1e3
    by_role, by_name = {}, {}
    for r in rs:
        k = role(r["Kernel_Name"])
        by_role.setdefault(k, [0, 0.0]); by_role[k][0] += 1; by_role[k][1] += dur(r)
        nm = r["Kernel_Name"].split("(")[0][:70]
        by_name.setdefault(nm, [0, 0.0]); by_name[nm][0] += 1; by_name[nm][1] += dur(r)
    busy = sum(v[1] for v in by_role.values())
    print(f"N={N} {seg}: span {span / 1e3:.3f} ms, kernel time {busy / 1e3:.3f} ms (streams overlap when > span), {len(rs)} launches")
    for k, (n, us) in sorted(by_role.items(), key=lambda kv: -kv[1][1]):
        extra = ""
        if k.startswith("cholesky GEMMs"): extra = f"  = {chol_flops / (us * 1e-6) / 1e12:.1f} TFLOP/s on N^3/3"
        if k.startswith("inverse GEMMs"): extra = f"  = {inv_flops / (us * 1e-6) / 1e12:.1f} TFLOP/s on N^3/3"
        if k.startswith("K^-1"): extra = f"  = {kinv_flops / (us * 1e-6) / 1e12:.1f} TFLOP/s on N^3/3"
        print(f"    {k:28s} {n:4d} launches  {us / 1e3:8.3f} ms  avg {us / n:8.1f} us{extra}")
    with open(f"{out}/fit_N{N}_{seg}_stats.csv", "w") as f:
        f.write("kernel,calls,total_us,avg_us\n")
        for nm, (n, us) in sorted(by_name.items(), key=lambda kv: -kv[1][1]):
            f.write(f"\"{nm}\",{n},{us:.1f},{us / n:.2f}\n")
