"""Wall time of a GaussianProcess fit with the hyper-parameter search (sklearn's protocol: 1 + n_restarts L-BFGS-B runs)
against the number of runs driven concurrently (GPT_OPT_WORKERS), at the reference's transport size.
usage: python tools/optimizer_concurrency.py [N] [workers ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sklearn.gaussian_process.kernels import RBF, WhiteKernel, ConstantKernel as C  # noqa: E402
from gaussian_process_transportation_amd import GaussianProcess  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2500
workers = [int(a) for a in sys.argv[2:]] or [1, 2, 3, 4, 6]
rng = np.random.default_rng(0)
X = rng.uniform(0, 1, (N, 3))
Y = np.column_stack([0.1 * np.sin(3 * X[:, 0] + X[:, 1]), 0.1 * np.cos(2 * X[:, 2]), 0.05 * X[:, 0] * X[:, 1]]) + 0.005 * rng.standard_normal((N, 3))
k0 = C(0.1) * RBF(length_scale=[0.1]) + WhiteKernel(1e-4)
ref = None
for w in workers:
    os.environ["GPT_OPT_WORKERS"] = str(w)
    ts = []
    for rep in range(3):
        np.random.seed(0)
        gp = GaussianProcess(kernel=k0, n_restarts_optimizer=5, verbose=False)
        t0 = time.perf_counter()
        gp.fit(X, Y)
        ts.append(time.perf_counter() - t0)
    th = np.asarray(gp.gp.kernel_.theta)
    ref = th if ref is None else ref
    print(f"N={N} workers={w}: fit {min(ts)*1e3:.0f} ms (runs: {' '.join(f'{t*1e3:.0f}' for t in ts)}), theta identical to workers={workers[0]}: {np.array_equal(th, ref)}",
          flush=True)
