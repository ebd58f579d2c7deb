"""GaussianProcessTransportation — user-facing attribute protocol of the reference
(policy_transportation/transportation/gaussian_process_transportation.py:11-30):
set source_distribution / target_distribution / training_traj [/ training_delta / training_ori],
call fit_transportation() then apply_transportation()."""
from .gaussian_process import GaussianProcess
from .policy_transportation import PolicyTransportation


def _default_kernel():
    from sklearn.gaussian_process.kernels import RBF, WhiteKernel, ConstantKernel as C
    return C(0.1) * RBF(length_scale=[0.1]) + WhiteKernel(0.0001)


class GaussianProcessTransportation:
    def __init__(self, kernel_transport=None, optimizer="fmin_l_bfgs_b", device=0, verbose=True):
        if kernel_transport is None:
            kernel_transport = _default_kernel()
        # as in the reference, the kernel is consumed here: assigning .kernel_transport later is a no-op
        self.method = PolicyTransportation(
            GaussianProcess(kernel=kernel_transport, optimizer=optimizer, device=device, verbose=verbose), verbose=verbose)

    def fit_transportation(self, do_scale=False, do_rotation=True):
        self.method.fit(self.source_distribution, self.target_distribution, do_scale=do_scale, do_rotation=do_rotation)

    def apply_transportation(self):
        self.training_traj_old = self.training_traj
        self.training_traj, self.std = self.method.transport(self.training_traj_old)
        if hasattr(self, "training_delta"):
            self.training_delta, self.var_vel_transported = self.method.transport_velocity(
                self.training_traj_old, self.training_delta)
        if hasattr(self, "training_ori"):
            self.training_ori = self.method.transport_orientation(self.training_traj_old, self.training_ori)

    def sample_transportation(self):
        return self.method.sample_transportation(self.training_traj_old)
