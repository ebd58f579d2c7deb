"""One-off check beyond the bench size: fit at N = 16384 (and 12800, not a power of two), predictions against the CPU
oracle's fast path on a few hundred queries.  python tools/gpu_big_n.py [N ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gaussian_process_transportation_amd import _lib  # noqa: E402
from oracle import gp_oracle as orc  # noqa: E402

for N in [int(a) for a in sys.argv[1:]] or [12800, 16384]:
    X, Y, Xq = orc.synthetic_problem(N, 300)
    c, ls, noise, jit = 0.1, np.array([0.1, 0.1, 0.1]), 1e-4, 1e-10
    h = _lib.Handle(0)
    h.fit(X, Y, ls, c, noise, jit)
    t0 = time.perf_counter(); h.fit(X, Y, ls, c, noise, jit); t_fit = time.perf_counter() - t0
    out = h.predict_all(Xq, mean=True, var=True, J=True, Jvar=True)
    print(f"N={N}: GPU fit {t_fit*1e3:.1f} ms {h.fit_timings()}", flush=True)
    t0 = time.perf_counter()
    L, a = orc.gpr_fit(X, Y, c, ls, noise, jit)
    mean, var, J, Jvar = orc.posterior_all_fast(Xq, X, L, a, c, ls, noise, want_jvar=True)
    print(f"   oracle fit+predict {time.perf_counter()-t0:.1f} s", flush=True)
    rel = lambda x, y: float(np.max(np.abs(x - y)) / np.max(np.abs(y)))
    print(f"   rel err: mean {rel(out['mean'], mean):.2e}  var {rel(out['var'], var):.2e}  J {rel(out['J'], J):.2e}  Jvar {rel(out['Jvar'], Jvar):.2e}",
          flush=True)
    h.close()
