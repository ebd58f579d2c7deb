"""Exact-GP prediction on SVGP pseudo-points — the post-training half of the reference's
`SVGP` model (policy_transportation/models/torch/stocastic_variational_gaussian_process_derivatives.py):
`convert_to_exact_gp` (:72-78), `posterior_f` (:113-129) and `posterior_f_prime` (:132-153).

The variational training itself lives in gpytorch (not vendored by the reference, absent here) and is out of
scope; this class takes what training leaves behind — inducing points Z (Z,D), pseudo-point covariances
Sigma (T,Z,Z), pseudo-targets y (T,Z) or (T,Z,1), per-task outputscale (T,) and the ARD length-scale (D,) — and
runs the prediction algebra on the GPU, one factorisation per task, in fp64 (the reference computes it in fp32 on
M x M matrices and cannot reach M = 1e6).  PARITY UNPINNED: the reference holds no fixture for this path; the
CPU restatement is oracle/gp_oracle.py:svgp_exact_oracle.  Where the reference's :142 broadcasts K_inv over the
input-dimension axis (it only type-checks for T == D), the intended per-task K_inv[t] is used."""
from __future__ import annotations

import numpy as np

from . import _lib


class SVGPExactPredictor:
    def __init__(self, x_inducing, var_inducing, y_inducing, outputscale, lengthscale, device=0):
        Z = np.asarray(x_inducing, dtype=np.float64)
        S = np.asarray(var_inducing, dtype=np.float64)
        y = np.asarray(y_inducing, dtype=np.float64)
        if y.ndim == 3:
            y = y[:, :, 0]
        os_ = np.atleast_1d(np.asarray(outputscale, dtype=np.float64))
        T = S.shape[0]
        if S.shape != (T, len(Z), len(Z)) or y.shape != (T, len(Z)) or os_.shape != (T,):
            raise ValueError("expected var_inducing (T,Z,Z), y_inducing (T,Z[,1]), outputscale (T,)")
        self.num_tasks, self.n_features = T, Z.shape[1]
        self.lengthscale = np.atleast_1d(np.asarray(lengthscale, dtype=np.float64))
        self._handles = []
        for t in range(T):                                  # convert_to_exact_gp, per task (:72-78)
            h = _lib.Handle(device)
            h.fit_noise_matrix(Z, y[t][:, None], self.lengthscale, os_[t], S[t], alpha=0.0)
            self._handles.append(h)

    def posterior_f(self, x, return_std=False):
        """mean (M,T) [, std (M,T)]  (:113-129)."""
        outs = [h.predict_all(x, mean=True, var=bool(return_std)) for h in self._handles]
        mean = np.column_stack([o["mean"][:, 0] for o in outs])
        if not return_std:
            return mean
        return mean, np.sqrt(np.column_stack([o["var"] for o in outs]))

    def posterior_f_prime(self, x, return_std=False):
        """Jacobian mean (M,T,D) [, its std (M,T,D)]  (:132-153)."""
        outs = [h.predict_all(x, J=True, Jvar=bool(return_std)) for h in self._handles]
        J = np.stack([o["J"][:, 0, :] for o in outs], axis=1)
        if not return_std:
            return J
        Jvar = np.stack([o["Jvar"] for o in outs], axis=1)
        return J, np.sqrt(np.maximum(Jvar, 0.0))

    # the reference wrapper's names (StocasticVariationalGaussianProcess.predict / derivative, :189-200)
    def predict(self, x, return_std=False):
        return self.posterior_f(x, return_std=return_std)

    def derivative(self, x):
        return self.posterior_f_prime(x, return_std=True)

    def close(self):
        for h in self._handles:
            h.close()
        self._handles = []
