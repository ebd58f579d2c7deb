"""k_var at the reference's own model sizes (N <= 2500) — VERDICT r3 item 1.  Device-resident queries, predict(mean, var)
(the configs[1] call) unless --jac; prints the variance kernel's time (the library's hipEvents around all its launches,
median of --reps calls) next to the MFMA floor  M (N^2 + 2N) / 78.6 TFLOP/s.
  python tools/small_n_probe.py [--shapes 1024x50000,2500x10000,...] [--reps 9] [--jac]
  GPT_HIP_LIB=.../build/libgpt_abl7.so python tools/small_n_probe.py       # timing-only ablation builds (make ablate)
  GPT_HIP_LIB=.../build/libgpt_vtrace.so python tools/small_n_probe.py --trace   # phase stamps of two workgroups (make trace)
"""
import argparse
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gaussian_process_transportation_amd import _lib  # noqa: E402

PEAK = 78.6e12
VT_STAMPS, VT_ITEMS, VT_WGS = 16, 24, 2
PHASES = ["item start->block barrier/fence", "->sweep set-up + first fill", "->A ring + barrier", "->lock-step chunks",
          "->diagonal tile (reload sweeps)", "->fold / partial store", "->GEN: scratch visible + barrier", "->slot epilogue"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="1024x49152,1024x50000,1024x4096,2500x460,2500x4096,2500x10000,2500x16384,2500x49152")
    ap.add_argument("--reps", type=int, default=9)
    ap.add_argument("--jac", action="store_true", help="mean + var + Jacobian (the DESIGN small-batch table) instead of mean + var")
    ap.add_argument("--f32", action="store_true", help="fp32 model (k_var<float>: the diagonal tile's image goes through LDS); the floor is then quoted on the fp32 peak")
    ap.add_argument("--trace", action="store_true")
    ap.add_argument("--trace-items", type=int, default=6)
    args = ap.parse_args()
    lib_tag = os.path.basename(os.environ.get("GPT_HIP_LIB", "libgpt_hip.so"))
    shapes = [tuple(int(v) for v in s.split("x")) for s in args.shapes.split(",")]
    handles = {}
    for N, M in shapes:
        rng = np.random.default_rng(0)
        if N not in handles:
            X = rng.uniform(0, 1, (N, 3)); Y = 0.05 * np.sin(4 * X) + 0.01 * rng.standard_normal((N, 3))
            h = _lib.Handle(0)
            if args.f32:
                h.set_dtype(_lib.GPT_F32)
            h.fit(X, Y, np.array([0.1] * 3), 0.1, 1e-4, 1e-10)
            handles[N] = h
        h = handles[N]
        tt = torch.float32 if args.f32 else torch.float64
        xq = torch.from_numpy(np.random.default_rng(1).uniform(-0.1, 1.1, (M, 3))).to(tt).cuda()
        mean = torch.empty((M, 3), dtype=tt, device="cuda"); var = torch.empty(M, dtype=tt, device="cuda")
        J = torch.empty((M, 3, 3), dtype=tt, device="cuda") if args.jac else None
        call = lambda: h.predict_all_dev(xq.data_ptr(), M, mean.data_ptr(), var.data_ptr(), J.data_ptr() if J is not None else 0, 0, 0)
        for _ in range(3):
            call()
        h.synchronize()
        h.set_profiling(True)
        vms, mms = [], []
        for _ in range(args.reps):
            call()
            t = h.predict_timings()
            vms.append(t["var_ms"]); mms.append(t["mean_jac_ms"])
        h.set_profiling(False)
        v = float(np.median(vms)); floor = M * (N * N + 2 * N) / (157.3e12 if args.f32 else PEAK) * 1e3
        print(f"[{lib_tag}] N={N} M={M}: k_var {v:.4f} ms (min {min(vms):.4f})  floor {floor:.4f} ms  = {floor / v * 100:.1f} % of the {'fp32' if args.f32 else 'fp64'} MFMA peak;"
              f"  mean{'+J' if args.jac else ''} kernel {np.median(mms):.4f} ms", flush=True)
        if args.trace:
            n_ph = VT_WGS * 8 * VT_ITEMS * VT_STAMPS
            buf = torch.zeros(n_ph + 2 * 1024, dtype=torch.int64, device="cuda")
            fn = h.lib.gpt_debug_set_var_trace
            fn.restype = ctypes.c_int; fn.argtypes = [ctypes.c_void_p]
            assert fn(ctypes.c_void_p(buf.data_ptr())) == 0
            call(); h.synchronize()
            assert fn(None) == 0
            raw = buf.cpu().numpy()
            tr = raw[:n_ph].reshape(VT_WGS, 8, VT_ITEMS, VT_STAMPS)
            wg = raw[n_ph:].reshape(1024, 2)
            wg = wg[wg[:, 0] != 0]
            if len(wg):          # (several launches per call: the last launch's stamps survive)
                d = (wg[:, 1] - wg[:, 0]).astype(np.float64)
                t0 = wg[:, 0].min()
                d *= 0.01; st = (wg[:, 0] - t0) * 0.01; en = (wg[:, 1] - t0) * 0.01       # 100 MHz chip-wide counter -> us
                print(f"  per-workgroup busy time of the last k_var launch, {len(wg)} workgroups (us): min {d.min():.1f}  median {np.median(d):.1f}  "
                      f"max {d.max():.1f}  mean {d.mean():.1f};  first start -> last end {en.max():.1f};  starts: median +{np.median(st):.1f} latest +{st.max():.1f};  "
                      f"ends: earliest {en.min():.1f} median {np.median(en):.1f}")
                print("    busy deciles (us): " + " ".join(f"{np.percentile(d, q):.1f}" for q in range(0, 101, 10)))
            for wi in range(VT_WGS):
                t0 = tr[wi, 0, 0, 0]
                print(f"  workgroup {'0' if wi == 0 else '37'}: (shader clocks; a full 512x512x64 tile = 128 k-steps x 2048 = 262144)")
                for it in range(min(VT_ITEMS, args.trace_items)):
                    if tr[wi, 0, it, 0] == 0:
                        break
                    for w in (0, 3, 4, 7):           # row groups 0, 3, 7, 4
                        s = tr[wi, w, it]
                        d = [int(s[i + 1] - s[i]) if s[i + 1] and s[i] else 0 for i in range(8)]
                        op = f"  [opening: to sweep entry +{int(s[11] - s[1])}, queries +{int(s[9] - s[11])}, diag image +{int(s[10] - s[9])}, first fill +{int(s[2] - s[10])}]" if s[9] and s[10] else ""
                        print(f"    item {it:2d} wave {w}: start +{int(s[0] - t0):8d} | " + " ".join(f"{x:7d}" for x in d) + f" | total {int(s[8] - s[0]):8d}" + op)
                    if tr[wi, 0, it, 12] and args.f32:   # fp32: image copied to LDS | the wave's k-steps | wait at the barrier
                        rows = []
                        for w in range(8):
                            s = tr[wi, w, it]
                            g = w if w < 4 else 11 - w
                            rows.append((g, int(s[12] - s[4]), int(s[14] - s[12]), int(s[5] - s[14])))
                        print("      diagonal tile, row group: image copied to LDS | its k-steps | wait at the barrier:  "
                              + "  ".join(f"g{g}: {a} | {b} | {c}" for g, a, b, c in sorted(rows)))
                    elif tr[wi, 0, it, 12]:              # HALF instantiation: the diagonal tile per wave (row groups 0 .. 7)
                        rows = []
                        for w in range(8):
                            s = tr[wi, w, it]
                            g = w if w < 4 else 11 - w
                            rows.append((g, int(s[12] - s[4]), int(s[13] - s[12]), int(s[14] - s[13]), int(s[5] - s[14])))
                        print("      diagonal tile, row group: image staged | k-steps < 64 (B from LDS) | k-steps >= 64 (B from scratch) | wait at the barrier:  "
                              + "  ".join(f"g{g}: {a} | {b} | {c} | {d}" for g, a, b, c, d in sorted(rows)))
                print("    columns: " + " | ".join(PHASES))


if __name__ == "__main__":
    main()
