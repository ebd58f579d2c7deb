"""One size per process for rocprofv3 (tools/gpu_fit_sizes.sh): two warm-up fits, then ONE fit and ONE objective evaluation
(gpt_lml_objective) — the trace's last two k_gram launches mark them (tools/fit_profile_summary.py).
usage: python tools/fit_once.py N"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussian_process_transportation_amd import _lib  # noqa: E402

N = int(sys.argv[1])
rng = np.random.default_rng(0)
X = rng.uniform(0, 1, (N, 3))
Y = 0.05 * np.sin(4 * X) + 0.01 * rng.standard_normal((N, 3))
ls = np.array([0.1, 0.1, 0.1])
h = _lib.Handle(0)
for _ in range(2):
    h.fit(X, Y, ls, 0.1, 1e-4, 1e-10)
    h.lml_objective(X, Y, ls, 0.1, 1e-4, 1e-10)
h.fit(X, Y, ls, 0.1, 1e-4, 1e-10)
t = h.fit_timings()
h.lml_objective(X, Y, ls, 0.1, 1e-4, 1e-10)
print(f"N={N} fit phases (device events, under the profiler): " + "  ".join(f"{k} {v:.3f}" for k, v in t.items()), flush=True)
h.close()
