"""PolicyTransportation — Phi(x) = gamma(x) + Psi(gamma(x)) with gamma an AffineTransform and Psi
any `delta_map` exposing fit / predict / derivative / samples (duck-typed plugin slot of the
reference, policy_transportation/transportation/policy_transportation.py:11-84)."""
import numpy as np

from .affine_transform import AffineTransform


class PolicyTransportation:
    def __init__(self, method, verbose=True):
        self.delta_map = method
        self.verbose = verbose

    def fit(self, source_distribution, target_distribution, do_scale=False, do_rotation=True):
        self.affine_transform = AffineTransform(do_scale=do_scale, do_rotation=do_rotation, verbose=self.verbose)
        self.affine_transform.fit(source_distribution, target_distribution)
        source_aligned = self.affine_transform.predict(source_distribution)
        self.delta_distribution = np.asarray(target_distribution, dtype=np.float64) - source_aligned
        self.delta_map.fit(source_aligned, self.delta_distribution)

    def prefetch(self, pos):
        """Optional: lets a delta_map that can (GaussianProcess.prefetch_posterior) compute what transport(pos) and
        transport_velocity(pos, .) will ask for in one pass.  No effect on results."""
        hook = getattr(self.delta_map, "prefetch_posterior", None)
        if hook is None:
            return
        try:
            hook(self.affine_transform.predict(pos))
        except NotImplementedError:            # e.g. a Matern delta_map: the calls that follow decide what fails
            pass

    def transport(self, pos, return_std=True):
        """Returns (transported positions, std).  The reference raises NameError for
        return_std=False (its :35 returns an unbound name); here std is None in that case."""
        pos_rotated = self.affine_transform.predict(pos)
        if return_std:
            delta_mean, delta_std = self.delta_map.predict(pos_rotated, return_std=True)
        else:
            delta_mean, delta_std = self.delta_map.predict(pos_rotated, return_std=False), None
        return pos_rotated + delta_mean, delta_std

    def transport_velocity(self, pos, vel, return_var=True):
        """Push velocities through the Jacobian of Phi; variance from the Jacobian variance (:37-59)."""
        pos_rotated = self.affine_transform.predict(pos)
        J_gamma = self.affine_transform.derivative(pos)
        if return_var:
            J_psi, J_psi_var = self.delta_map.derivative(pos_rotated, return_var=True)
        else:
            J_psi, J_psi_var = self.delta_map.derivative(pos_rotated, return_var=False), None
        J_phi = J_gamma + J_psi @ J_gamma
        if self.verbose:
            print("Is the map locally diffeomorphic?", np.all(np.abs(np.linalg.det(J_phi)) > 0))
        vel = np.asarray(vel, dtype=np.float64)[:, :, None]
        vel_rotated = J_gamma @ vel
        vel_transported = (J_phi @ vel)[:, :, 0]
        if J_psi_var is None:
            return vel_transported, None
        var_vel_transported = (J_psi_var @ vel_rotated ** 2)[:, :, 0]
        return vel_transported, var_vel_transported

    def transport_orientation(self, pos, ori):
        """Rotate orientations (w,x,y,z quaternions) by the rotation closest to J_Phi, evaluated — as
        the reference does (:62) — at the UN-rotated positions."""
        from .quaternion import quaternion_from_nonorthogonal, quaternion_multiply
        J_phi = self.delta_map.derivative(pos)
        J_gamma = self.affine_transform.derivative(pos)
        J_phi = J_gamma + J_phi @ J_gamma
        if self.verbose:
            print("Is the map locally diffeomorphic?", np.all(np.linalg.det(J_phi) > 0))
        if J_phi[0].shape[0] != 3:
            print("The Jacobain of the map as shape ", J_phi[0].shape, " but it should be (3x3)")
            print("Robot orientation is not transported")
            return None
        return quaternion_multiply(quaternion_from_nonorthogonal(J_phi), np.asarray(ori, dtype=np.float64))

    def sample_transportation(self, pos):
        pos_rotated = self.affine_transform.predict(pos)
        return pos_rotated + self.delta_map.samples(pos_rotated)
