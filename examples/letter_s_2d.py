"""Headless run of the reference's 2-D letter-S demo (example/2D/surface_generalization.py:30-80) on the
MI355X path: resample the drawings, fit the dynamics GP (Matern 5/2, optimizer on), transport the demo and
its velocities to the new surface (RBF, optimizer on), refit the dynamics on the transported data.

    python examples/letter_s_2d.py

Data: the arrays of the reference's example/2D/data/example.npz as stored in tests/golden/letterS_2d.npz."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from sklearn.gaussian_process.kernels import RBF, Matern, WhiteKernel, ConstantKernel as C  # noqa: E402

from gaussian_process_transportation_amd import GaussianProcess as GPR  # noqa: E402
from gaussian_process_transportation_amd import GaussianProcessTransportation as Transport  # noqa: E402
from gaussian_process_transportation_amd.utils import resample  # noqa: E402


def main(verbose=True):
    data = np.load(os.path.join(ROOT, "tests", "golden", "letterS_2d.npz"))
    X = resample(data["demo_raw"], num_points=400)
    source_distribution = resample(data["floor_raw"], num_points=20)
    target_distribution = resample(data["newfloor_raw"], num_points=20)
    deltaX = np.zeros((len(X), 2))
    deltaX[:-1] = X[1:] - X[:-1]

    np.random.seed(0)
    t0 = time.perf_counter()
    k_deltaX = C(constant_value=np.sqrt(0.1)) * Matern(1 * np.ones(2), nu=2.5) + WhiteKernel(0.01)
    gp_deltaX = GPR(kernel=k_deltaX, verbose=verbose)
    gp_deltaX.fit(X, deltaX)
    xg, yg = np.meshgrid(np.linspace(X[:, 0].min() - 10, X[:, 0].max() + 10, 100),
                         np.linspace(X[:, 1].min() - 10, X[:, 1].max() + 10, 100))
    grid = np.column_stack([xg.ravel(), yg.ravel()])
    field, field_std = gp_deltaX.predict(grid, return_std=True)          # the vector field the reference plots

    k_transport = C(constant_value=10) * RBF(4 * np.ones(2)) + WhiteKernel(0.01)
    transport = Transport(kernel_transport=k_transport, verbose=verbose)
    transport.source_distribution = source_distribution
    transport.target_distribution = target_distribution
    transport.training_traj = X
    transport.training_delta = deltaX
    transport.fit_transportation(do_scale=False, do_rotation=True)
    transport.apply_transportation()
    X1, deltaX1 = transport.training_traj, transport.training_delta

    k_deltaX1 = C(constant_value=np.sqrt(0.1)) * Matern(1 * np.ones(2), nu=2.5) + WhiteKernel(0.01)
    gp_deltaX1 = GPR(kernel=k_deltaX1, verbose=verbose)
    gp_deltaX1.fit(X1, deltaX1)
    field1 = gp_deltaX1.predict(grid)
    dt = time.perf_counter() - t0
    out = dict(X1=X1, deltaX1=deltaX1, std=transport.std, var_vel=transport.var_vel_transported,
               field=field, field_std=field_std, field1=field1, seconds=dt,
               transport_theta=np.asarray(transport.method.delta_map.kernel.theta))
    if verbose:
        print(f"letter-S demo: 3 GP fits with hyper-parameter search + transport + 2 x 10^4-point vector fields in {dt:.2f} s")
        print("transport kernel:", transport.method.delta_map.kernel)
    return out


if __name__ == "__main__":
    main()
