"""Timing of the reference-shaped Python API at the bench size (numpy in, numpy out):
GaussianProcess.fit / predict(return_std) / derivative(return_var) / derivative_of_variance at N=8192, M=500k."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sklearn.gaussian_process.kernels import RBF, ConstantKernel as C, WhiteKernel  # noqa: E402

from gaussian_process_transportation_amd import GaussianProcess  # noqa: E402

N, M = 8192, 500_000
rng = np.random.default_rng(0)
X = rng.uniform(0, 1, (N, 3)); Y = 0.05 * np.sin(4 * X) + 0.01 * rng.standard_normal((N, 3))
Xq = np.random.default_rng(1).uniform(-0.1, 1.1, (M, 3))
gp = GaussianProcess(kernel=C(0.1) * RBF([0.1] * 3) + WhiteKernel(1e-4), optimizer=None, verbose=False)
gp.fit(X, Y); gp.predict(Xq[:1000], return_std=True); gp.derivative(Xq[:1000], return_var=True)


def t(fn, reps=2):
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); best = min(best, time.perf_counter() - t0)
    return best


tf = t(lambda: gp.fit(X, Y), 3)
tp = t(lambda: gp.predict(Xq, return_std=True))
td = t(lambda: gp.derivative(Xq))
tv = t(lambda: gp.derivative(Xq, return_var=True))
tg = t(lambda: gp.derivative_of_variance(Xq))
print(f"GaussianProcess (drop-in API, numpy in/out) N={N} M={M}: fit {tf*1e3:.1f} ms | predict(mean,std) {tp*1e3:.0f} ms = {M/tp/1e6:.2f} M q/s | "
      f"derivative {td*1e3:.0f} ms | derivative(return_var) {tv*1e3:.0f} ms = {M/tv/1e3:.0f} k q/s | derivative_of_variance {tg*1e3:.0f} ms")
