"""N = 8192, M = 10^4 (157 column blocks on 256 workgroups), device-resident mean + var + J: time per call and the k_var share,
for A/B runs of the tail plan (env GPT_VAR_SHORT_BIAS ...).  usage: python tools/m1e4_probe.py [M]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussian_process_transportation_amd import _lib  # noqa: E402

N = 8192
M = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000
rng = np.random.default_rng(0)
X = rng.uniform(0, 1, (N, 3)); Y = np.sin(4 * X)
h = _lib.Handle(0)
h.fit(X, Y, np.array([0.1] * 3), 0.1, 1e-4, 1e-10)
xq = torch.from_numpy(rng.uniform(0, 1, (M, 3))).cuda()
mean = torch.empty((M, 3), dtype=torch.float64, device="cuda"); var = torch.empty(M, dtype=torch.float64, device="cuda")
J = torch.empty((M, 3, 3), dtype=torch.float64, device="cuda")
for _ in range(5):
    h.predict_all_dev(xq.data_ptr(), M, mean.data_ptr(), var.data_ptr(), J.data_ptr(), 0, 0)
h.synchronize()
best = 1e9
for rep in range(5):
    t0 = time.perf_counter()
    for _ in range(20):
        h.predict_all_dev(xq.data_ptr(), M, mean.data_ptr(), var.data_ptr(), J.data_ptr(), 0, 0)
    h.synchronize()
    best = min(best, (time.perf_counter() - t0) / 20)
tag = " ".join(f"{k[4:]}={v}" for k, v in sorted(os.environ.items()) if k.startswith("GPT_")) or "defaults"
print(f"[{tag}] N={N} M={M}: {best*1e3:.3f} ms per call, var sum {float(var.sum()):.12g}", flush=True)
