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
