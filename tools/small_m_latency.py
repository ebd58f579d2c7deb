"""Latency of the variance + Jacobian path for small query batches (device-resident inputs), N = 8192 and 1024."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gaussian_process_transportation_amd import _lib  # noqa: E402

for N in (1024, 8192):
    rng = np.random.default_rng(0)
    X = rng.uniform(0, 1, (N, 3)); Y = np.sin(4 * X)
    h = _lib.Handle(0)
    h.fit(X, Y, np.array([0.1] * 3), 0.1, 1e-4, 1e-10)
    for M in (16, 64, 400, 1000, 4096, 16384, 65536):
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
        print(f"N={N} M={M}: {dt*1e3:.3f} ms per call = {M/dt:.0f} q/s", flush=True)
    h.close()
