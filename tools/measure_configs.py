"""Secondary measurements for DESIGN.md (run on the GPU box): BASELINE.json configs[1] (N=1024, M=50k, fit +
predict mean/std, no Jacobian) end to end through host buffers, and the PCIe-inclusive rate of configs[2]."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gaussian_process_transportation_amd import _lib  # noqa: E402


def synth(N, M, D=3):
    rng = np.random.default_rng(0)
    X = rng.uniform(0, 1, (N, D))
    Y = 0.05 * np.sin(4 * X) + 0.01 * rng.standard_normal((N, D))
    return X, Y, np.random.default_rng(1).uniform(-0.1, 1.1, (M, D))


def best(fn, reps=5):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return min(ts), float(np.median(ts))


if __name__ == "__main__":
    ls = np.array([0.1] * 3)
    h = _lib.Handle(0)
    X, Y, Xq = synth(1024, 50_000)
    h.fit(X, Y, ls, 0.1, 1e-4, 1e-10); h.predict_all(Xq, mean=True, var=True)
    tf = best(lambda: h.fit(X, Y, ls, 0.1, 1e-4, 1e-10))
    tp = best(lambda: h.predict_all(Xq, mean=True, var=True))
    print(f"cfg2 N=1024 M=50000: fit {tf[1]*1e3:.2f} ms (device phases {h.fit_timings()}), predict mean+std "
          f"{tp[1]*1e3:.2f} ms = {50_000/tp[1]:.0f} q/s host-to-host; fit+predict {(tf[1]+tp[1])*1e3:.2f} ms")
    X, Y, Xq = synth(8192, 500_000)
    h.fit(X, Y, ls, 0.1, 1e-4, 1e-10); h.predict_all(Xq[:1000], mean=True, var=True, J=True)
    tp = best(lambda: h.predict_all(Xq, mean=True, var=True, J=True), reps=3)
    print(f"cfg3 N=8192 M=500000 mean+var+J through host buffers (pageable numpy in/out, PCIe inclusive): "
          f"{tp[1]*1e3:.1f} ms = {500_000/tp[1]:.0f} q/s")
    tp = best(lambda: h.predict_all(Xq, mean=True, var=True, J=True, Jvar=True), reps=2)
    print(f"cfg3 + Jacobian variance through host buffers: {tp[1]*1e3:.1f} ms = {500_000/tp[1]:.0f} q/s")
    h.close()

    # configs[4]: SVGP exact-conversion predictor, Z = 2048 inducing points, T = D = 3 tasks, synthetic SPD pseudo-point
    # covariances, fp32 prediction as the reference computes it, through HOST buffers (numpy in / out);
    # the device-resident rate is bench.py --config svgp
    from gaussian_process_transportation_amd.svgp_exact import SVGPExactPredictor
    Z, T, M = 2048, 3, 1_000_000
    rng = np.random.default_rng(0)
    xz = rng.uniform(0, 1, (Z, 3))
    S = np.empty((T, Z, Z))
    for t in range(T):
        A = rng.standard_normal((Z, Z))
        S[t] = A @ A.T / Z + 1e-3 * np.eye(Z)
    yz = rng.standard_normal((T, Z))
    for dtype in ("float32", "float64"):
        t0 = time.perf_counter()
        pred = SVGPExactPredictor(xz, S, yz, np.ones(T), np.array([0.2, 0.2, 0.2]), dtype=dtype)
        t_conv = time.perf_counter() - t0
        xq = np.random.default_rng(1).uniform(0, 1, (M, 3))
        pred.posterior(xq[:1000])
        t_f = best(lambda: pred.posterior_f(xq, return_std=True), reps=2)[0]
        t_fp = best(lambda: pred.posterior_f_prime(xq, return_std=True), reps=2)[0]
        t_all = best(lambda: pred.posterior(xq), reps=2)[0]
        print(f"cfg5 SVGP exact conversion Z={Z} T={T} M={M} ({dtype}, host buffers): convert {t_conv*1e3:.0f} ms, posterior_f mean+std "
              f"{t_f*1e3:.0f} ms = {M/t_f:.0f} q/s, posterior_f_prime J+J_std {t_fp*1e3:.0f} ms = {M/t_fp:.0f} q/s, all four in one pass "
              f"{t_all*1e3:.0f} ms = {M/t_all:.0f} q/s")
        pred.close()
