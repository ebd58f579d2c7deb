"""Wall time of one evaluation of the hyper-parameter objective (gpt_lml_objective: Gram, Cholesky, inverse, alpha,
K^-1, gradient traces, one read-back) as the optimizer sees it, per problem size; and of the Python closure around it
(hyperopt.py) — the host's share matters at the reference's sizes (N = 400 .. 2500).
usage: python tools/objective_latency.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gaussian_process_transportation_amd import _lib  # noqa: E402

h = _lib.Handle(0)
for N, D in ((400, 2), (1000, 3), (2500, 3), (4096, 3), (8192, 3), (2500, 6)):
    rng = np.random.default_rng(0)
    X = rng.uniform(0, 1, (N, D)); Y = np.sin(4 * X[:, :3 if D >= 3 else 2])
    ls0 = np.full(D, 0.2 if D <= 3 else 0.5)
    for _ in range(3):
        h.lml_objective(X, Y, ls0, 0.1, 1e-3, 1e-10)
    reps = 40 if N <= 4096 else 10
    t0 = time.perf_counter()
    for i in range(reps):
        h.lml_objective(X, Y, ls0 * (1.0 + 0.01 * (i % 7)), 0.1, 1e-3, 1e-10)
    dt = (time.perf_counter() - t0) / reps
    print(f"N={N} D={D}: {dt*1e6:.0f} us per objective evaluation", flush=True)
h.close()
