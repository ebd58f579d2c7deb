"""MI355X-native Gaussian-process transportation hot path (drop-in for the reference's
`policy_transportation` exports: AffineTransform, GaussianProcess, GaussianProcessTransportation)."""
from .affine_transform import AffineTransform
from .gaussian_process import GaussianProcess
from .policy_transportation import PolicyTransportation
from .gaussian_process_transportation import GaussianProcessTransportation
from .svgp_exact import StocasticVariationalGaussianProcess, SVGPExactPredictor
from .svgp_transport import SVGPTransport

# the reference's three exports first; then the duck-typed caller and the SVGP exact-conversion path
__all__ = ["AffineTransform", "GaussianProcessTransportation", "GaussianProcess", "PolicyTransportation",
           "SVGPTransport", "StocasticVariationalGaussianProcess", "SVGPExactPredictor"]
