"""Headless run of the reference's 3-D surface demo (example/3D/surface_generalization_3D.py:36-75) on the MI355X
path: dynamics GP (Matern 3/2, optimizer on) on the 460-point demo, transport with the default kernel and
optimizer from the 2500-point old surface to the new one (the reference needs 279 s of CPU for this fit), refit.

    python examples/surface_3d.py

Data: the arrays of the reference's example/3D/data/example.npz as stored in tests/golden/surface_3d.npz."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from sklearn.gaussian_process.kernels import Matern, WhiteKernel, ConstantKernel as C  # noqa: E402

from gaussian_process_transportation_amd import GaussianProcess as GPR  # noqa: E402
from gaussian_process_transportation_amd import GaussianProcessTransportation as Transport  # noqa: E402


def main(verbose=True):
    data = np.load(os.path.join(ROOT, "tests", "golden", "surface_3d.npz"))
    X, source_distribution, target_distribution = data["demo"], data["source"], data["target"]
    deltaX = np.zeros((len(X), 3))
    deltaX[:-1] = X[1:] - X[:-1]
    np.random.seed(0)
    t0 = time.perf_counter()
    gp_deltaX = GPR(kernel=C(constant_value=np.sqrt(0.1)) * Matern(1 * np.ones(3), nu=1.5) + WhiteKernel(0.01), verbose=verbose)
    gp_deltaX.fit(X, deltaX)
    t1 = time.perf_counter()
    transport = Transport(verbose=verbose)                       # default kernel C(0.1) * RBF([0.1]) + White(1e-4), optimizer on
    transport.source_distribution = source_distribution
    transport.target_distribution = target_distribution
    transport.training_traj = X
    transport.training_delta = deltaX
    transport.fit_transportation()
    t2 = time.perf_counter()
    transport.apply_transportation()
    t3 = time.perf_counter()
    X1, deltaX1 = transport.training_traj, transport.training_delta
    gp_deltaX1 = GPR(kernel=C(constant_value=np.sqrt(0.1)) * Matern(1 * np.ones(3), nu=1.5) + WhiteKernel(0.01), verbose=verbose)
    gp_deltaX1.fit(X1, deltaX1)
    if verbose:
        print(f"3-D demo: dynamics fit {t1-t0:.2f} s, transport fit (N=2500, 6 L-BFGS-B runs) {t2-t1:.2f} s, apply {1e3*(t3-t2):.1f} ms")
        print("transport kernel:", transport.method.delta_map.kernel)
    return dict(X1=X1, deltaX1=deltaX1, std=transport.std, var_vel=transport.var_vel_transported,
                fit_seconds=t2 - t1, apply_seconds=t3 - t2, theta=np.asarray(transport.method.delta_map.kernel.theta))


if __name__ == "__main__":
    main()
