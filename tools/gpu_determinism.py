"""Same inputs twice -> bit-identical outputs (fit, all predict modes, covariance), on one handle and across handles.
Run on the GPU box: python tools/gpu_determinism.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gaussian_process_transportation_amd import _lib  # noqa: E402

rng = np.random.default_rng(3)
ok = True
for N, M in [(700, 5000), (2500, 20000), (8192, 100000)]:
    X = rng.uniform(0, 1, (N, 3)); Y = np.sin(4 * X); Xq = rng.uniform(-0.1, 1.1, (M, 3))
    ls = np.array([0.1, 0.12, 0.09])
    outs = []
    for rep in range(3):
        h = _lib.Handle(0) if rep != 1 else h            # rep 1 reuses the handle of rep 0
        h.fit(X, Y, ls, 0.1, 1e-4, 1e-10)
        L, a = h.export() if N <= 2500 else (None, h.export(want_L=False)[1])
        o = h.predict_all(Xq, mean=True, var=True, J=True, Jvar=True, dvar=True)
        _, cov = h.predict_cov(Xq[:50])
        outs.append((L, a, o, cov))
    for rep in (1, 2):
        same = all(np.array_equal(outs[0][2][k], outs[rep][2][k]) for k in outs[0][2]) and np.array_equal(outs[0][1], outs[rep][1]) \
            and np.array_equal(outs[0][3], outs[rep][3]) and (outs[0][0] is None or np.array_equal(outs[0][0], outs[rep][0]))
        print(f"N={N} M={M} run 0 vs run {rep} ({'same handle' if rep == 1 else 'new handle'}): {'bit-identical' if same else 'DIFFERENT'}")
        ok &= same
sys.exit(0 if ok else 1)
