"""Exact-GP prediction on SVGP pseudo-points — the post-training half of the reference's
`SVGP` model (policy_transportation/models/torch/stocastic_variational_gaussian_process_derivatives.py):
`convert_to_exact_gp` (:72-78), `posterior_f` (:113-129) and `posterior_f_prime` (:132-153), and the wrapper class
`StocasticVariationalGaussianProcess` (:155-200) that the SVGP transport calls.

The variational training itself lives in gpytorch (not vendored by the reference, absent here) and is out of
scope; these classes take what training leaves behind — inducing points Z (Z,D), pseudo-point covariances
Sigma (T,Z,Z), pseudo-targets y (T,Z) or (T,Z,1), per-task outputscale (T,) and the ARD length-scale (D,) — and
run the prediction algebra on the GPU: ONE handle holds all T tasks (`gpt_fit_svgp`: the tasks' inverse factors are
stacked into one A operand and share one generated kernel-column operand), factorised in fp64, predicted in fp32
(the reference's arithmetic: it casts everything with `.float()`) or fp64.  The reference materialises M x M
matrices (:120-123, :142-144) and cannot reach M = 1e6; this path never does.

PARITY UNPINNED: the reference holds no fixture for this path; the CPU restatement is
oracle/gp_oracle.py:svgp_exact_oracle.  Where the reference's :142 broadcasts K_inv over the input-dimension axis
(it only type-checks for T == D), the intended per-task K_inv[t] is used."""
from __future__ import annotations

import numpy as np

from . import _lib

_DTYPES = {"float32": _lib.GPT_F32, "float64": _lib.GPT_F64, np.float32: _lib.GPT_F32, np.float64: _lib.GPT_F64,
           _lib.GPT_F32: _lib.GPT_F32, _lib.GPT_F64: _lib.GPT_F64}


def _dtype_code(dtype):
    try:
        key = dtype if dtype in _DTYPES else np.dtype(dtype).name
        return _DTYPES[key]
    except (KeyError, TypeError):
        raise ValueError(f"dtype must be float32 or float64, got {dtype!r}") from None


class SVGPExactPredictor:
    def __init__(self, x_inducing, var_inducing, y_inducing, outputscale, lengthscale, device=0, dtype="float32",
                 jitter=0.0):
        Z = np.asarray(x_inducing, dtype=np.float64)
        S = np.asarray(var_inducing, dtype=np.float64)
        y = np.asarray(y_inducing, dtype=np.float64)
        if y.ndim == 3:
            y = y[:, :, 0]
        os_ = np.atleast_1d(np.asarray(outputscale, dtype=np.float64))
        T = S.shape[0]
        if Z.ndim != 2 or S.shape != (T, len(Z), len(Z)) or y.shape != (T, len(Z)) or os_.shape != (T,):
            raise ValueError("expected x_inducing (Z,D), var_inducing (T,Z,Z), y_inducing (T,Z[,1]), outputscale (T,)")
        self.num_tasks, self.n_features = T, Z.shape[1]
        self.lengthscale = np.atleast_1d(np.asarray(lengthscale, dtype=np.float64)).reshape(-1)
        self.dtype = _dtype_code(dtype)
        self._handle = _lib.Handle(device)
        self._handle.fit_svgp(Z, y, S, self.lengthscale, os_, jitter=jitter, dtype=self.dtype)   # convert_to_exact_gp (:72-78)

    def posterior(self, x, return_std=True):
        """mean (M,T), std (M,T), Jacobian (M,T,D), Jacobian std (M,T,D) in ONE pass over the stacked factors
        (what test/svgp_derivatives_mimo.py:73-78 asks for in two calls)."""
        out = self._handle.predict_all(x, mean=True, var=bool(return_std), J=True, Jvar=bool(return_std))
        if not return_std:
            return out["mean"], out["J"]
        return out["mean"], np.sqrt(out["var"]), out["J"], np.sqrt(np.maximum(out["Jvar"], 0))

    def posterior_f(self, x, return_std=False):
        """mean (M,T) [, std (M,T)]  (:113-129).  (The reference's mean-only branch returns (T,M): it skips the permute
        of :125; here both branches use the (M,T) layout its callers rely on.)"""
        out = self._handle.predict_all(x, mean=True, var=bool(return_std))
        if not return_std:
            return out["mean"]
        return out["mean"], np.sqrt(out["var"])

    def posterior_f_prime(self, x, return_std=False):
        """Jacobian mean (M,T,D) [, its std (M,T,D)]  (:132-153)."""
        out = self._handle.predict_all(x, J=True, Jvar=bool(return_std))
        if not return_std:
            return out["J"]
        return out["J"], np.sqrt(np.maximum(out["Jvar"], 0))

    # the reference wrapper's names (StocasticVariationalGaussianProcess.predict / derivative, :189-200)
    def predict(self, x, return_std=False):
        return self.posterior_f(x, return_std=return_std)

    def derivative(self, x):
        return self.posterior_f_prime(x, return_std=True)

    def close(self):
        if self._handle is not None:
            self._handle.close()
            self._handle = None


class StocasticVariationalGaussianProcess:
    """Mirror of the reference wrapper (:155-200): `predict(x, return_std)` and `derivative(x)` over the exact-GP
    conversion of a trained SVGP.  `fit(num_epochs)` — Adam on the variational ELBO inside gpytorch (:168-187) — is NOT
    rebuilt; hand the trained quantities to `set_pseudo_points` (what `convert_to_exact_gp`, :72-78, reads from
    gpytorch: inducing points, pseudo-point covariances and targets, outputscales, length-scale)."""

    def __init__(self, X, Y, num_inducing=100, device=0, dtype="float32"):
        self.X = np.asarray(X, dtype=np.float64)
        self.Y = np.asarray(Y, dtype=np.float64)
        self.num_inducing = num_inducing
        self.device, self.dtype = device, dtype
        self.gp = None

    def fit(self, num_epochs=10):
        raise NotImplementedError(
            "variational training of the SVGP is gpytorch's (reference :168-187) and outside this GPU path; train "
            "it there and pass inducing points / pseudo-point covariances / pseudo-targets / outputscales / "
            "length-scale to set_pseudo_points()")

    def set_pseudo_points(self, x_inducing, var_inducing, y_inducing, outputscale, lengthscale, jitter=0.0):
        if self.gp is not None:
            self.gp.close()
        self.gp = SVGPExactPredictor(x_inducing, var_inducing, y_inducing, outputscale, lengthscale, device=self.device,
                                     dtype=self.dtype, jitter=jitter)
        return self

    def _require(self):
        if self.gp is None:
            raise RuntimeError("StocasticVariationalGaussianProcess has no model: call set_pseudo_points() first")
        return self.gp

    def predict(self, x, return_std=False):
        return self._require().posterior_f(x, return_std=return_std)

    def derivative(self, x):
        return self._require().posterior_f_prime(x, return_std=True)
