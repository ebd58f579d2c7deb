"""SVGPTransport — the attribute protocol of the reference's SVGP transport
(policy_transportation/transportation/torch/stocastic_variational_gaussian_process_transportation.py:11-102) over
the GPU exact-conversion predictor.

    inputs   source_distribution (N,D), target_distribution (N,D), training_traj (M,D),
             optional training_delta (M,D), training_ori (M,4) (w,x,y,z; D = 3)
    calls    fit_transportation(...), apply_transportation()
    outputs  training_traj, std, training_traj_old, training_delta + var_vel_transported, training_ori,
             affine_transform, gp_delta_map

Differences, all stated: the variational training (:59-60 -> gpytorch) is out of scope, so `fit_transportation`
takes the trained pseudo-point quantities through `pseudo_points=`; outputs are numpy arrays where the reference
holds torch tensors it converts at once (:68-69, :77-78); the quaternion algebra uses this package's own helpers
(the reference's `quaternion` package is absent: parity unpinned, as for transport_orientation)."""
from __future__ import annotations

import os
import pickle

import numpy as np

from .affine_transform import AffineTransform
from .svgp_exact import StocasticVariationalGaussianProcess


class SVGPTransport:
    def __init__(self, device=0, dtype="float32", verbose=True):
        self.device, self.dtype, self.verbose = device, dtype, verbose

    # ---- data plumbing of the reference (:17-43); files are this package's own pickles of numpy arrays
    def save_distributions(self, directory="distributions"):
        os.makedirs(directory, exist_ok=True)
        for name in ("source", "target"):
            with open(os.path.join(directory, name + ".pkl"), "wb") as f:
                pickle.dump(np.asarray(getattr(self, name + "_distribution")), f)

    def load_distributions(self, directory="distributions"):
        for name in ("source", "target"):
            try:
                with open(os.path.join(directory, name + ".pkl"), "rb") as f:
                    setattr(self, name + "_distribution", pickle.load(f))
            except OSError:
                print(f"No {name} distribution saved")

    def fit_transportation(self, num_epochs=20, num_inducing=100, pseudo_points=None):
        """(:46-60)  Affine pre-alignment, residual field, SVGP on (aligned source, residual).  `pseudo_points` =
        dict(x_inducing, var_inducing, y_inducing, outputscale, lengthscale) of the trained model; without it the
        call reaches the (unavailable) variational training and raises NotImplementedError."""
        if type(self.target_distribution) != type(self.source_distribution):
            raise TypeError("Both the distribution must be a numpy array.")
        if not isinstance(self.target_distribution, np.ndarray) and not isinstance(self.source_distribution, np.ndarray):
            self.convert_distribution_to_array()           # "a function of every sensor class" in the reference (:50)
        self.affine_transform = AffineTransform(verbose=self.verbose)
        self.affine_transform.fit(self.source_distribution, self.target_distribution)
        source_distribution = self.affine_transform.predict(self.source_distribution)
        delta_distribution = self.target_distribution - source_distribution
        self.gp_delta_map = StocasticVariationalGaussianProcess(source_distribution, delta_distribution,
                                                                num_inducing=num_inducing, device=self.device,
                                                                dtype=self.dtype)
        if pseudo_points is None:
            self.gp_delta_map.fit(num_epochs=num_epochs)
        else:
            self.gp_delta_map.set_pseudo_points(**pseudo_points)

    def apply_transportation(self):
        """(:62-102)"""
        from .quaternion import quaternion_from_nonorthogonal, quaternion_multiply
        self.training_traj_old = self.training_traj
        self.traj_rotated = self.affine_transform.predict(self.training_traj)
        mean, std = self.gp_delta_map.predict(self.traj_rotated, return_std=True)
        self.delta_map_mean, self.std = np.asarray(mean, dtype=np.float64), np.asarray(std, dtype=np.float64)
        self.training_traj = self.traj_rotated + self.delta_map_mean

        has_delta, has_ori = hasattr(self, "training_delta"), hasattr(self, "training_ori")
        if has_delta or has_ori:
            pos = np.array(self.traj_rotated)
            Jacobian, Jacobian_std = self.gp_delta_map.derivative(pos)
            Jacobian = np.asarray(Jacobian, dtype=np.float64)
            Jacobian_std = np.asarray(Jacobian_std, dtype=np.float64)
            rot_gp = np.eye(Jacobian[0].shape[0]) + Jacobian
            rot_affine = self.affine_transform.rotation_matrix
            derivative_affine = self.affine_transform.derivative(pos)
        if has_delta:
            delta = np.asarray(self.training_delta, dtype=np.float64)[:, :, np.newaxis]
            delta = derivative_affine @ delta
            self.var_vel_transported = (Jacobian_std ** 2 @ delta ** 2)[:, :, 0]
            self.training_delta = (rot_gp @ delta)[:, :, 0]
        if has_ori:
            quat_demo = np.asarray(self.training_ori, dtype=np.float64)
            quat_affine = quaternion_from_nonorthogonal(rot_affine)
            quat_gp = quaternion_from_nonorthogonal(rot_gp)
            self.training_ori = quaternion_multiply(quat_gp, quaternion_multiply(quat_affine, quat_demo))
