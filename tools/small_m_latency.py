"""Latency of the variance + Jacobian path for small query batches (device-resident inputs): the reference's own
batch sizes (M = 460 trajectories, 10^4-point grids) at N = 2500 and N = 8192.  Prints the MFMA floor next to each
number: M (N^2 + 2N) flop / 78.6 TFLOP/s — a call cannot be faster than that however the work is split.
usage: python tools/small_m_latency.py [order]   (order: GPT_VAR_TAIL_ORDER override, -1 = automatic)"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gaussian_process_transportation_amd import _lib  # noqa: E402

if len(sys.argv) > 1 and sys.argv[1] != "-1":
    os.environ["GPT_VAR_TAIL_ORDER"] = sys.argv[1]
print("tail order:", os.environ.get("GPT_VAR_TAIL_ORDER", "automatic"), flush=True)
for N, Ms in ((2500, (460, 1000, 4096, 10_000)), (8192, (64, 460, 1000, 4096, 10_000, 16384, 65536))):
    rng = np.random.default_rng(0)
    X = rng.uniform(0, 1, (N, 3)); Y = np.sin(4 * X)
    h = _lib.Handle(0)
    h.fit(X, Y, np.array([0.1] * 3), 0.1, 1e-4, 1e-10)
    for M in Ms:
        xq = torch.from_numpy(rng.uniform(0, 1, (M, 3))).cuda()
        mean = torch.empty((M, 3), dtype=torch.float64, device="cuda"); var = torch.empty(M, dtype=torch.float64, device="cuda")
        J = torch.empty((M, 3, 3), dtype=torch.float64, device="cuda")
        for _ in range(3):
            h.predict_all_dev(xq.data_ptr(), M, mean.data_ptr(), var.data_ptr(), J.data_ptr(), 0, 0)
        h.synchronize()
        reps = 20
        t0 = time.perf_counter()
        for _ in range(reps):
            h.predict_all_dev(xq.data_ptr(), M, mean.data_ptr(), var.data_ptr(), J.data_ptr(), 0, 0)
        h.synchronize()
        dt = (time.perf_counter() - t0) / reps
        floor = M * (N * N + 2 * N) / 78.6e12
        print(f"N={N} M={M}: {dt*1e3:.3f} ms per call = {M/dt:.0f} q/s   (MFMA floor {floor*1e3:.3f} ms, {floor/dt*100:.0f} % of it)", flush=True)
    h.close()
