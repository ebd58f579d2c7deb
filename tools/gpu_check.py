"""Stage-by-stage diagnostic of the HIP path against the CPU oracle (run on the GPU box):
    python tools/gpu_check.py [--big]
Prints max-norm relative errors of L, W = L^-1, alpha, mean, var, J, Jvar, dvar for several sizes,
then rough timings.  Development aid; the judged tests live in tests/."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gaussian_process_transportation_amd import _lib  # noqa: E402
from oracle import gp_oracle as orc  # noqa: E402


def rel(a, b):
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


def check(N, M, D=3, O=3, ls=(0.1, 0.1, 0.1), seed=0):
    rng = np.random.default_rng(seed)
    X = rng.uniform(0, 1, (N, D))
    Y = 0.05 * np.sin(4 * X[:, :1] + np.arange(O)[None, :]) + 0.01 * rng.standard_normal((N, O))
    Xq = rng.uniform(-0.1, 1.1, (M, D))
    c, noise, jit = 0.1, 1e-4, 1e-10
    h = _lib.Handle(0)
    t0 = time.time()
    h.fit(X, Y, np.asarray(ls), c, noise, jit)
    t_fit = time.time() - t0
    g = orc.GaussianProcessOracle(c, np.asarray(ls), noise, jit).fit(X, Y)
    L, a = h.export()
    W = h.export_inverse_factor()
    Wref = np.linalg.inv(g.L_)
    print(f"N={N} D={D} O={O} M={M}  fit {t_fit*1e3:.1f} ms  timings {h.fit_timings()}")
    print(f"   L {rel(L, g.L_):.2e}  W {rel(np.tril(W), Wref):.2e}  upperW {np.abs(np.triu(W,1)).max():.1e}  alpha {rel(a, g.alpha_):.2e}")
    out = h.predict_all(Xq, mean=True, var=True, J=True, Jvar=True, dvar=True)
    out1 = h.predict_all(Xq, var=True)
    mean, std = g.predict(Xq, return_std=True)
    std = std if std.ndim == 1 else std[:, 0]
    var_ref = (std + np.sqrt(noise)) ** 2
    J, Jv = g.derivative(Xq, return_var=True)
    dv = g.derivative_of_variance(Xq)
    print(f"   mean {rel(out['mean'], mean.reshape(M, O)):.2e}  var4 {rel(out['var'], var_ref):.2e}  var1 {rel(out1['var'], var_ref):.2e}"
          f"  J {rel(out['J'], J):.2e}  Jvar {rel(out['Jvar'], Jv[:, 0, :]):.2e}  dvar {rel(out['dvar'], dv):.2e}")
    print(f"   lml {h.lml():.10g} vs {orc.log_marginal_likelihood(np.log(np.r_[c, ls, noise]), X, Y, len(ls), jit, False):.10g}")
    h.close()


def timing(N, M):
    X, Y, Xq = orc.synthetic_problem(N, M)
    h = _lib.Handle(0)
    h.fit(X, Y, np.array([0.1] * 3), 0.1, 1e-4, 1e-10)
    print(f"N={N}: fit timings (ms) {h.fit_timings()}")
    h.fit(X, Y, np.array([0.1] * 3), 0.1, 1e-4, 1e-10)
    print(f"N={N}: fit timings again (ms) {h.fit_timings()}")
    for kw in (dict(mean=True, J=True), dict(var=True), dict(mean=True, var=True, J=True),
               dict(mean=True, var=True, J=True, Jvar=True)):
        h.predict_all(Xq[:1024], **kw)
        t0 = time.time()
        h.predict_all(Xq, **kw)
        dt = time.time() - t0
        print(f"   N={N} M={M} {sorted(kw)}: {dt*1e3:.1f} ms  {M/dt:.0f} q/s (host buffers, end to end)")
    h.close()


if __name__ == "__main__":
    print("devices:", _lib.require_gpu(), _lib.load().gpt_version())
    check(64, 50)
    check(200, 77, D=2, O=2, ls=(0.2,))
    check(256, 100)
    check(300, 33, D=1, O=5, ls=(0.05,))
    check(1024, 500)
    check(1500, 300, D=3, O=1, ls=(0.1, 0.2, 0.15))
    timing(1024, 50000)
    if "--big" in sys.argv:
        timing(8192, 20000)
