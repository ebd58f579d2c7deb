"""What the FIRST predict on a fresh handle pays beyond the kernels (allocations of the variance kernel's scratch, plan
build + upload, kernel attribute opt-ins), against the steady state — the reference's demos call apply_transportation once."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gaussian_process_transportation_amd import _lib  # noqa: E402

rng = np.random.default_rng(0)
for N, M in ((2500, 460), (8192, 460)):
    X = rng.uniform(0, 1, (N, 3)); Y = np.sin(4 * X); Xq = rng.uniform(0, 1, (M, 3))
    h = _lib.Handle(0)
    h.fit(X, Y, np.array([0.1] * 3), 0.1, 1e-4, 1e-10)
    for label in ("first", "second", "third"):
        t0 = time.perf_counter()
        h.predict_all(Xq, mean=True, var=True, J=True, Jvar=True)
        print(f"N={N} M={M} {label} predict_all(mean,var,J,Jvar): {(time.perf_counter()-t0)*1e3:.2f} ms", flush=True)
    t0 = time.perf_counter()
    h.predict_all(Xq[:300], mean=True, var=True, J=True, Jvar=True)
    print(f"N={N} M=300 (new plan, buffers large enough): {(time.perf_counter()-t0)*1e3:.2f} ms", flush=True)
    h.close()
