"""GaussianProcess — host-side mirror of the reference regressor
(policy_transportation/models/gaussian_process.py:16-126) running on an MI355X through
libgpt_hip.  Same constructor, methods, shapes and quirks; the numerics (Gram, Cholesky,
alpha, posterior mean / variance / Jacobian / Jacobian variance / d var/dx) are HIP kernels.
scikit-learn kernel objects are accepted as hyper-parameter containers only."""
from __future__ import annotations

import copy

import numpy as np

from . import _lib

_PARAM_C = "k1__k1__constant_value"
_PARAM_LS = "k1__k2__length_scale"
_PARAM_NOISE = "k2__noise_level"


def kernel_hyperparameters(kernel):
    """(constant_value, length_scale array, noise_level) of a `ConstantKernel * RBF + WhiteKernel`
    object — the parameter names the reference hard-codes (gaussian_process.py:39-41, 49)."""
    try:
        p = kernel.get_params()
        c, ls, noise = p[_PARAM_C], p[_PARAM_LS], p[_PARAM_NOISE]
    except (AttributeError, KeyError) as e:
        raise ValueError("kernel must be ConstantKernel * RBF + WhiteKernel (as the reference's parameter "
                         "names k1__k1__constant_value / k1__k2__length_scale / k2__noise_level require)") from e
    return float(c), np.atleast_1d(np.asarray(ls, dtype=np.float64)), float(noise)


def kernel_type(kernel):
    """GPT_KERNEL_* code of the stationary factor: RBF, or Matern with nu in {0.5, 1.5, 2.5, inf}."""
    k = kernel.get_params().get("k1__k2", None)
    name = type(k).__name__
    if name == "RBF":
        return 0
    if name == "Matern":
        nu = float(k.nu)
        codes = {0.5: 1, 1.5: 2, 2.5: 3, float("inf"): 0}
        if nu in codes:
            return codes[nu]
        raise NotImplementedError(f"Matern(nu={nu}) is not on the GPU path (nu must be 0.5, 1.5, 2.5 or inf)")
    raise NotImplementedError(f"only RBF and Matern kernels run on the GPU path, got {name}")


class _FittedView:
    """Stands in for the reference's `self.gp` (the sklearn estimator): the fitted attributes
    outside callers read, fetched from the device on first access."""

    def __init__(self, owner):
        self._o = owner
        self._L = None
        self._alpha_ = None
        self.alpha = owner.alpha
        self.kernel = owner._kernel_in
        self.kernel_ = None
        self.X_train_ = None
        self.y_train_ = None
        self._lml = None

    @property
    def alpha_(self):
        if self._alpha_ is None:
            self._alpha_ = self._o._handle.export(want_L=False)[1]
        return self._alpha_

    @property
    def log_marginal_likelihood_value_(self):
        """LML of the fitted theta (sklearn/_gpr.py:335-341), evaluated on the device on first access."""
        if self._lml is None:
            self._lml = self._o._handle.lml()
        return self._lml

    @property
    def L_(self):
        if self._L is None:
            self._L = self._o._handle.export(want_alpha=False)[0]
        return self._L


class GaussianProcess:
    def __init__(self, kernel, alpha=1e-10, optimizer="fmin_l_bfgs_b", n_restarts_optimizer=5, n_targets=None,
                 device=0, verbose=True, dtype="float64", devices=None):
        """Arguments as the reference's (:17-23); `device`, `devices`, `verbose` and `dtype` are additions.  dtype="float32"
        keeps the fp64 factorisation and runs the prediction kernels in fp32 (outputs float32; return_cov / samples need
        float64): twice the fp64 rate, ~1e-4 of the output scale.  devices=[0, 1, ...] (default: the one `device`): the fit
        and the hyper-parameter search run on devices[0], the fitted model is copied to the others and predict / derivative
        / derivative_of_variance shard their rows over all of them (device_group.py) — same results, same call."""
        self._kernel_in = kernel
        self.kernel = kernel
        self.alpha = alpha
        self.optimizer = optimizer
        self.n_restarts_optimizer = n_restarts_optimizer if optimizer is not None else 0
        self.n_targets = n_targets
        self.devices = [int(d) for d in devices] if devices is not None else None
        if self.devices is not None and not self.devices:
            raise ValueError("devices must name at least one GPU")
        self.device = self.devices[0] if self.devices else device
        self.verbose = verbose
        if np.dtype(dtype) not in (np.dtype(np.float64), np.dtype(np.float32)):
            raise ValueError("dtype must be float64 or float32")
        self._dtype = _lib.GPT_F32 if np.dtype(dtype) == np.dtype(np.float32) else _lib.GPT_F64
        self._handle = None
        self._K_inv = None
        self.gp = _FittedView(self)

    # ------------------------------------------------------------------ fit
    def fit(self, X, Y):
        X = np.asarray(X, dtype=np.float64)
        Y = np.asarray(Y, dtype=np.float64)
        if Y.ndim == 1:
            Y = Y[:, None]
        self.X = X
        self.Y = Y
        self._memo = None
        self.n_features = np.shape(X)[1]
        self.n_samples = np.shape(X)[0]          # pre-filter count, as the reference (:29)
        self.n_outputs = np.shape(Y)[1]
        mask = np.isnan(Y).any(axis=1)           # (:33-35)
        self.X = X[~mask]
        self.Y = Y[~mask]
        if self.n_targets is not None and self.n_outputs != self.n_targets:   # sklearn/_gpr.py:265-269
            raise ValueError(f"The number of targets seen in `y` is different from the parameter `n_targets`. "
                             f"Got {self.n_outputs} != {self.n_targets}.")
        c, ls, noise = kernel_hyperparameters(self._kernel_in)
        self._ktype = kernel_type(self._kernel_in)
        lml = None
        if self.optimizer is not None:
            from .hyperopt import optimize_hyperparameters
            c, ls, noise, lml = optimize_hyperparameters(self, c, ls, noise)
        if self._handle is None:
            self._handle = self._new_handle()
        self._handle.set_dtype(self._dtype)
        self._handle.fit(self.X, self.Y, ls, c, noise, self.alpha, self._ktype)
        self._K_inv = None
        # fitted kernel object with the reference's attribute protocol (:38-41)
        fitted = copy.deepcopy(self._kernel_in)
        keep_array = np.iterable(self._kernel_in.get_params()[_PARAM_LS])
        fitted.set_params(**{_PARAM_C: c, _PARAM_LS: (ls.copy() if keep_array else float(ls[0])), _PARAM_NOISE: noise})
        self.kernel = fitted
        self.gp = _FittedView(self)
        self.gp._lml = lml
        self.gp.kernel_ = fitted
        self.gp.X_train_ = self.X
        self.gp.y_train_ = self.Y
        self.kernel_params_ = [fitted.get_params()[_PARAM_LS], fitted.get_params()["k1"]]
        self.noise_var_ = self.alpha + noise
        self.prior_var = c
        self._c, self._ls, self._noise = c, ls, noise
        if self.verbose:
            print("lenghtscales", fitted.get_params()[_PARAM_LS])
        return self

    def _new_handle(self):
        """One handle on `device`, or — devices=[a, b, ...] — the group that fits on the first and predicts on all."""
        if self.devices is not None and len(self.devices) > 1:
            from .device_group import DeviceGroup
            return DeviceGroup(self.devices)
        return _lib.Handle(self.device)

    @property
    def K_inv(self):
        """(c*RBF + noise_var_*I)^-1 as the reference caches it (:42-43).  Only reference-internal
        code reads it; here it is assembled lazily from the device factor W = L^-1 (K^-1 = W^T W)."""
        if self._K_inv is None:
            W = self._handle.export_inverse_factor()
            self._K_inv = W.T @ W
        return self._K_inv

    def _require_rbf(self, what):
        if self._ktype != 0:
            raise NotImplementedError(f"{what}() implements the RBF formulas of the reference (gaussian_process.py:63-126); "
                                      "the reference silently applies them to any kernel, this path refuses")

    def _require_fit(self):
        if self._handle is None:
            raise RuntimeError("GaussianProcess is not fitted")

    # ------------------------------------------------------------------ predict
    def predict(self, x, return_std=False, return_cov=False):
        """(:46-55)  mean (M,O); with return_std also `std - sqrt(noise_level)` tiled over outputs."""
        self._require_fit()
        if return_std and return_cov:
            raise RuntimeError("At most one of return_std or return_cov can be requested.")
        if return_cov:                                  # sklearn/_gpr.py:458-470
            mean, cov = self._handle.predict_cov(x)
            if self.n_outputs == 1:
                return mean[:, 0], cov
            return mean, np.repeat(cov[:, :, None], self.n_outputs, axis=2)
        out = self._memo_lookup(x) or self._handle.predict_all(x, mean=True, var=bool(return_std))
        mean = out["mean"]
        if self.n_outputs == 1:
            mean = mean[:, 0]                       # sklearn squeezes single-target output (_gpr.py:449-451)
        if not return_std:
            return mean
        std = np.sqrt(out["var"])
        if self.n_outputs > 1:
            std = np.repeat(std[:, None], self.n_outputs, axis=1)   # _gpr.py:488
        return mean, std - np.sqrt(self._noise)     # reference quirk (:49)

    def samples(self, x):
        """(:57-60)  10 joint posterior draws per output, shape (10, M, O).  Mean and covariance come from the
        GPU; the draws follow sklearn's sample_y (sklearn/_gpr.py:498-535: RandomState(0),
        multivariate_normal per target) on the host."""
        self._require_fit()
        mean, cov = self._handle.predict_cov(x)
        rng = np.random.RandomState(0)
        per_target = [rng.multivariate_normal(mean[:, t], cov, 10).T[:, np.newaxis] for t in range(self.n_outputs)]
        y_samples = np.hstack(per_target)               # (M, O, 10)
        return np.transpose(y_samples, (2, 0, 1))

    # ------------------------------------------------------------------ derivatives
    def derivative(self, x, return_var=False):
        """(:63-102)  J (M,O,D) = d mean_o / d x_d; with return_var also its variance, tiled over outputs."""
        self._require_fit()
        self._require_rbf("derivative")
        out = self._memo_lookup(x) or self._handle.predict_all(x, J=True, Jvar=bool(return_var))
        if not return_var:
            return out["J"]
        Sigma = np.repeat(out["Jvar"][:, None, :], self.n_outputs, axis=1)
        return out["J"], Sigma

    def derivative_of_variance(self, x):
        """(:104-126)  (D, M) array of d var / d x_d."""
        self._require_fit()
        self._require_rbf("derivative_of_variance")
        return self._handle.predict_all(x, dvar=True)["dvar"]

    # ------------------------------------------------------------------ fused metric path
    def prefetch_posterior(self, x):
        """Everything predict(x, return_std=True) and derivative(x, return_var=True) return, computed in ONE pass over
        the factor (the variance and the three Jacobian-variance columns share the 4-column kernel: 4 column passes
        instead of 1 + 4) and kept for the next calls with the same x.  Used by
        GaussianProcessTransportation.apply_transportation, which asks for both at the same positions."""
        self._require_fit()
        self._require_rbf("prefetch_posterior")
        xk = np.array(x, dtype=np.float64, order="C", copy=True)
        self._memo = (xk, self._handle.predict_all(xk, mean=True, var=True, J=True, Jvar=True))

    def _memo_lookup(self, x):
        memo = getattr(self, "_memo", None)
        if memo is None:
            return None
        xk, out = memo
        x = np.asarray(x)
        if x.shape != xk.shape or not np.array_equal(x, xk):
            return None
        return out

    def posterior(self, x, jacobian_variance=False):
        """One call for mean (M,O), raw variance (M,), Jacobian (M,O,D) [and Jacobian variance (M,D)]."""
        self._require_fit()
        self._require_rbf("posterior")
        return self._handle.predict_all(x, mean=True, var=True, J=True, Jvar=bool(jacobian_variance))
