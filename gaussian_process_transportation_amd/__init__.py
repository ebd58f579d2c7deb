"""MI355X-native Gaussian-process transportation hot path (drop-in for the reference's
`policy_transportation` exports: AffineTransform, GaussianProcess, GaussianProcessTransportation)."""
from .affine_transform import AffineTransform
from .gaussian_process import GaussianProcess
from .policy_transportation import PolicyTransportation
from .gaussian_process_transportation import GaussianProcessTransportation

__all__ = ["AffineTransform", "GaussianProcessTransportation", "GaussianProcess", "PolicyTransportation"]
