"""Throughput of the wide layouts (input dimension 4 .. 8: rows of 8; 9 .. 15: rows of 16) next to the tuned D = 3 path, device-resident inputs,
N = 8192 sources: mean + variance + Jacobian ("J": one k* column per query) and the same with the Jacobian variance
("JVAR": 4 / 8 / 16 columns per query, of which 1 + D carry data).  k_var's rate is quoted on the columns it multiplies
(including the zero columns of the wide fused layout) and on the useful ones.
usage: python tools/wide_d_timing.py [N] [M]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gaussian_process_transportation_amd import _lib  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
M = int(sys.argv[2]) if len(sys.argv) > 2 else 200_000
for dtype, tname, peak in ((_lib.GPT_F64, "fp64", 78.6e12), (_lib.GPT_F32, "fp32", 157.3e12)):
    tt = torch.float64 if dtype == _lib.GPT_F64 else torch.float32
    for D in (3, 4, 6, 8, 9, 12, 15):
        rng = np.random.default_rng(0)
        X = rng.uniform(0, 1, (N, D)); Y = np.sin(4 * X[:, :3])
        h = _lib.Handle(0)
        h.set_dtype(dtype)
        h.fit(X, Y, np.full(D, 0.1 * (1.0 if D <= 3 else 2.0 * np.sqrt(D / 3.0))), 0.1, 1e-4, 1e-10)
        xq = torch.from_numpy(rng.uniform(0, 1, (M, D))).to(tt).cuda()
        mean = torch.empty((M, 3), dtype=tt, device="cuda"); var = torch.empty(M, dtype=tt, device="cuda")
        J = torch.empty((M, 3, D), dtype=tt, device="cuda"); Jv = torch.empty((M, D), dtype=tt, device="cuda")
        h.set_profiling(True)
        # J: mean + variance + Jacobian; JVAR: the same + Jacobian variance (fused columns); JV-ALONE: mean + Jacobian +
        # Jacobian variance without the variance (what the reference's transport_velocity asks for)
        for mode, vp, jv in (("J", var.data_ptr(), 0), ("JVAR", var.data_ptr(), Jv.data_ptr()), ("JV-ALONE", 0, Jv.data_ptr())):
            h.predict_all_dev(xq.data_ptr(), M, mean.data_ptr(), vp, J.data_ptr(), jv, 0)
            h.synchronize()
            reps = 3
            t0 = time.perf_counter()
            for _ in range(reps):
                h.predict_all_dev(xq.data_ptr(), M, mean.data_ptr(), vp, J.data_ptr(), jv, 0)
            h.synchronize()
            dt = (time.perf_counter() - t0) / reps
            tm = h.predict_timings()
            t_mj, t_var = tm["mean_jac_ms"], tm["var_ms"]
            fused = 4 if D <= 3 else (8 if D <= 7 else 16)
            cols = 1 if not jv else (fused if vp else (D if D <= 4 or D == 8 else fused))
            useful = 1 if not jv else (1 + D if vp else D)
            NP = (N + 511) // 512 * 512
            flops = M * cols * (NP * (NP + 512.0))          # block lower triangle incl. diagonal tiles, 2 flop per MAC
            print(f"{tname} D={D} {mode}: {M/dt:.0f} q/s ({dt*1e3:.1f} ms; mean+J {t_mj:.1f} ms, k_var {t_var:.1f} ms = "
                  f"{flops/t_var/1e9:.1f} TFLOP/s on {cols} columns/query = {flops/t_var/1e9*1e12/peak*100:.0f} % of peak, "
                  f"{useful}/{cols} useful)", flush=True)
        h.close()
