"""GaussianProcessTransportation — the user-facing attribute protocol of the reference
(policy_transportation/transportation/gaussian_process_transportation.py:11-30).

Protocol (names are the reference's, they are the API):
    inputs   source_distribution (N,D), target_distribution (N,D), training_traj (M,D)
             optional training_delta (M,D) velocities, training_ori (M,4) quaternions (D = 3)
    calls    fit_transportation(do_scale=False, do_rotation=True), apply_transportation(), sample_transportation()
    outputs  training_traj (moved), std, training_traj_old, and for the optional inputs
             training_delta + var_vel_transported, training_ori

The regressor behind it is the MI355X GaussianProcess; as in the reference the kernel is consumed by the
constructor, so assigning `.kernel_transport` afterwards changes nothing."""
from .gaussian_process import GaussianProcess
from .policy_transportation import PolicyTransportation

_MISSING = object()


def _reference_default_kernel():
    """C(0.1) * RBF([0.1]) + WhiteKernel(1e-4): the default argument of the reference's constructor (:12)."""
    from sklearn.gaussian_process.kernels import RBF, ConstantKernel, WhiteKernel
    return ConstantKernel(0.1) * RBF(length_scale=[0.1]) + WhiteKernel(0.0001)


class GaussianProcessTransportation:
    def __init__(self, kernel_transport=None, optimizer="fmin_l_bfgs_b", device=0, verbose=True, devices=None):
        """`devices=[0, 1, ...]`: apply_transportation() shards the demonstration's rows over these GPUs (the fit stays on
        devices[0]); default: the one `device`."""
        kernel = _reference_default_kernel() if kernel_transport is None else kernel_transport
        regressor = GaussianProcess(kernel=kernel, optimizer=optimizer, device=device, verbose=verbose, devices=devices)
        self.method = PolicyTransportation(regressor, verbose=verbose)

    def _input(self, name):
        value = getattr(self, name, _MISSING)
        if value is _MISSING:
            raise AttributeError(f"GaussianProcessTransportation: set .{name} before this call")
        return value

    def fit_transportation(self, do_scale=False, do_rotation=True):
        """Affine pre-alignment + GP fit of the residual displacement field (policy_transportation.py:16-24)."""
        source, target = self._input("source_distribution"), self._input("target_distribution")
        self.method.fit(source, target, do_scale=do_scale, do_rotation=do_rotation)

    def apply_transportation(self):
        """Moves the demonstration; velocities and orientations follow when they were provided (:19-27)."""
        before = self._input("training_traj")
        self.training_traj_old = before
        velocities = getattr(self, "training_delta", _MISSING)
        if velocities is not _MISSING:
            self.method.prefetch(before)              # std and Jacobian variance at the same positions: one pass
        self.training_traj, self.std = self.method.transport(before)
        if velocities is not _MISSING:
            self.training_delta, self.var_vel_transported = self.method.transport_velocity(before, velocities)
        orientations = getattr(self, "training_ori", _MISSING)
        if orientations is not _MISSING:
            self.training_ori = self.method.transport_orientation(before, orientations)

    def sample_transportation(self):
        """Posterior draws of the moved demonstration, at the positions of the last apply_transportation() (:29-30)."""
        return self.method.sample_transportation(self._input("training_traj_old"))
