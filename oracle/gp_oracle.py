"""CPU oracle for the GP-transportation hot path.  TEST INFRASTRUCTURE ONLY.

This file is a numpy/scipy restatement of the reference algorithm
(`policy_transportation/models/gaussian_process.py`, the scikit-learn 1.7.2
`GaussianProcessRegressor` code it calls, `models/affine_trasformation.py` and
`transportation/policy_transportation.py`).  It is the checker for the HIP
path: only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py` may import it.  Nothing under `gaussian_process_transportation_amd/`
imports it and the product path never falls back to it.

Parity pin: `tests/golden/*.npz` hold outputs of the reference itself (imported
from /root/reference in the authoring container by `tests/golden/make_golden.py`,
scikit-learn 1.7.2); `tests/test_oracle_golden.py` checks every function here
against them.

Citations: `ref:` = path under /root/reference/policy_transportation,
`sklearn:` = sklearn/gaussian_process/{_gpr,kernels}.py (v1.7.2).
"""
from __future__ import annotations

import numpy as np
from scipy.linalg import cho_solve, cholesky, solve_triangular
from scipy.spatial.distance import cdist, pdist, squareform


# --------------------------------------------------------------------------- kernels
KINDS = ("rbf", "matern12", "matern32", "matern52")


def _shape(kind, r):
    """k(r) for r = |(x-y)/l|.  sklearn: kernels.py RBF 1553-1565, Matern 1717-1745 (nu = 0.5, 1.5, 2.5)."""
    if kind == "rbf":
        return np.exp(-0.5 * r * r)
    if kind == "matern12":
        return np.exp(-r)
    if kind == "matern32":
        t = r * np.sqrt(3.0)
        return (1.0 + t) * np.exp(-t)
    if kind == "matern52":
        t = r * np.sqrt(5.0)
        return (1.0 + t + t ** 2 / 3.0) * np.exp(-t)
    raise ValueError(kind)


def rbf_gram(X, Y=None, length_scale=1.0, kind="rbf"):
    """Stationary part of the kernel.  RBF: exp(-0.5 ||(x-y)/l||^2), sklearn kernels.py:1553-1565: the
    X-only form goes through pdist + squareform with the diagonal forced to 1, the two-argument form
    through cdist, both on inputs divided by length_scale (scalar or (D,)).  Matern (kernels.py:1717-1745)
    does the same with the euclidean metric."""
    ls = np.asarray(length_scale, dtype=np.float64)
    if kind == "rbf":
        if Y is None:
            d = pdist(X / ls, metric="sqeuclidean")
            K = squareform(np.exp(-0.5 * d))
            np.fill_diagonal(K, 1.0)
            return K
        return np.exp(-0.5 * cdist(X / ls, Y / ls, metric="sqeuclidean"))
    if Y is None:
        K = squareform(_shape(kind, pdist(X / ls, metric="euclidean")))
        np.fill_diagonal(K, 1.0)
        return K
    return _shape(kind, cdist(X / ls, Y / ls, metric="euclidean"))


def kernel_train(X, constant_value, length_scale, noise_level, kind="rbf"):
    """kernel_(X): c*k + noise*I.  sklearn: kernels.py Product 931-989,
    Sum 833-889, ConstantKernel 1239-1310, WhiteKernel 1369-1440 (Y is None)."""
    K = constant_value * rbf_gram(X, None, length_scale, kind)
    K[np.diag_indices_from(K)] += noise_level
    return K


def kernel_cross(Xq, X, constant_value, length_scale, kind="rbf"):
    """kernel_(Xq, X): WhiteKernel contributes zeros for an explicit second
    argument (sklearn: kernels.py:1413-1414)."""
    return constant_value * rbf_gram(Xq, X, length_scale, kind)


# --------------------------------------------------------------------------- sklearn GPR
def gpr_fit(X, Y, constant_value, length_scale, noise_level, alpha=1e-10, kind="rbf"):
    """L_, alpha_ as sklearn: _gpr.py:346-364 (K = kernel_(X); K[diag] += alpha;
    cholesky lower; cho_solve).  Raises numpy.linalg.LinAlgError on a non-PD K."""
    K = kernel_train(X, constant_value, length_scale, noise_level, kind)
    K[np.diag_indices_from(K)] += alpha
    L = cholesky(K, lower=True, check_finite=False)
    a = cho_solve((L, True), Y, check_finite=False)
    return L, a


def gpr_predict(Xq, X, L, a, constant_value, length_scale, noise_level,
                return_std=False, return_cov=False, kind="rbf"):
    """sklearn: _gpr.py:441-494.  y_mean = K* alpha_; V = L \\ K*^T;
    var = diag(k**) - sum(V*V) clipped at 0, tiled over the targets; std = sqrt."""
    Ks = kernel_cross(Xq, X, constant_value, length_scale, kind)
    mean = Ks @ a
    if not (return_std or return_cov):
        return mean
    V = solve_triangular(L, Ks.T, lower=True, check_finite=False)
    n_targets = a.shape[1] if a.ndim > 1 else 1
    if return_cov:
        cov = kernel_train(Xq, constant_value, length_scale, noise_level, kind) - V.T @ V
        if n_targets > 1:
            cov = np.repeat(cov[..., None], n_targets, axis=-1)
        return mean, cov
    var = np.full(Xq.shape[0], constant_value + noise_level, dtype=np.float64)
    var -= np.einsum("ij,ji->i", V.T, V)
    var[var < 0] = 0.0
    if n_targets > 1:
        var = np.repeat(var[:, None], n_targets, axis=1)
    return mean, np.sqrt(var)


def log_marginal_likelihood(theta, X, Y, n_ls, alpha=1e-10, eval_gradient=True, kind="rbf"):
    """LML and its gradient w.r.t. log-hyper-parameters.  sklearn: _gpr.py:537-652.

    theta = log([constant_value, length_scale (n_ls of them), noise_level]) — the
    parameter order of `(C * RBF) + WhiteKernel` (kernels.py theta concatenation
    of k1 then k2).  Gradient = 0.5 * sum_o trace((a_o a_o^T - K^-1) dK/dtheta_p)."""
    theta = np.asarray(theta, dtype=np.float64)
    c = np.exp(theta[0])
    ls = np.exp(theta[1:1 + n_ls])
    noise = np.exp(theta[1 + n_ls])
    N = X.shape[0]
    R = rbf_gram(X, None, ls if n_ls > 1 else ls[0], kind)
    K = c * R
    K[np.diag_indices_from(K)] += noise + alpha
    try:
        L = cholesky(K, lower=True, check_finite=False)
    except np.linalg.LinAlgError:
        return (-np.inf, np.zeros_like(theta)) if eval_gradient else -np.inf
    a = cho_solve((L, True), Y, check_finite=False)
    lml = -0.5 * np.einsum("ik,ik->k", Y, a)
    lml = lml - np.log(np.diag(L)).sum() - N / 2 * np.log(2 * np.pi)
    lml = lml.sum()
    if not eval_gradient:
        return lml
    K_inv = cho_solve((L, True), np.eye(N), check_finite=False)
    n_out = Y.shape[1]
    inner = a @ a.T - n_out * K_inv            # sum over outputs of (a_o a_o^T - K^-1)
    grad = np.empty_like(theta)
    cR = c * R
    grad[0] = 0.5 * np.sum(inner * cR)          # dK/dlog c = c*R       (kernels.py:1290-1300)
    # dK/dlog l_d = c * G * D_d with D_d = (x_d - x'_d)^2 / l_d^2 (summed over d when isotropic) and
    # G = R (RBF, kernels.py:1568-1580), R / r (nu=1/2), 3 exp(-sqrt3 r) (3/2), 5/3 (sqrt5 r + 1) exp(-sqrt5 r)
    # (5/2)  (Matern, kernels.py:1747-1778)
    lsv = np.broadcast_to(ls, (X.shape[1],)) if n_ls == 1 else ls
    Dd = (X[:, None, :] - X[None, :, :]) ** 2 / lsv ** 2          # (N, N, D)
    r = np.sqrt(Dd.sum(-1))
    if kind == "rbf":
        G = R
    elif kind == "matern12":
        G = np.divide(R, r, out=np.zeros_like(R), where=r != 0)
    elif kind == "matern32":
        G = 3.0 * np.exp(-np.sqrt(3.0) * r)
    else:
        t = np.sqrt(5.0) * r
        G = 5.0 / 3.0 * (t + 1.0) * np.exp(-t)
    if n_ls == 1:
        grad[1] = 0.5 * np.sum(inner * c * G * Dd.sum(-1))
    else:
        for d in range(n_ls):
            grad[1 + d] = 0.5 * np.sum(inner * c * G * Dd[:, :, d])
    grad[1 + n_ls] = 0.5 * noise * np.trace(inner)   # dK/dlog noise = noise*I (kernels.py:1403-1410)
    return lml, grad


# --------------------------------------------------------------------------- reference GaussianProcess
class GaussianProcessOracle:
    """Restates ref: models/gaussian_process.py:16-126 for optimizer=None
    (fixed hyper-parameters).  Same attribute names, shapes and quirks."""

    def __init__(self, constant_value, length_scale, noise_level, alpha=1e-10, kind="rbf"):
        self.kind = kind
        self.constant_value = float(constant_value)
        self.length_scale = np.atleast_1d(np.asarray(length_scale, dtype=np.float64))
        self.noise_level = float(noise_level)
        self.alpha = float(alpha)

    def _ls(self):
        return self.length_scale if self.length_scale.size > 1 else float(self.length_scale[0])

    def fit(self, X, Y):
        """ref: gaussian_process.py:25-44."""
        self.n_features = X.shape[1]
        self.n_samples = X.shape[0]            # pre-filter count (quirk, :29)
        self.n_outputs = Y.shape[1]
        mask = np.isnan(Y).any(axis=1)         # :33-35
        self.X = X[~mask]
        self.Y = Y[~mask]
        self.L_, self.alpha_ = gpr_fit(self.X, self.Y, self.constant_value, self._ls(),
                                       self.noise_level, self.alpha, self.kind)
        self.noise_var_ = self.alpha + self.noise_level   # :40
        self.prior_var = self.constant_value               # :41
        K_ = kernel_cross(self.X, self.X, self.constant_value, self._ls(), self.kind) \
            + self.noise_var_ * np.eye(len(self.X))        # :42
        self.K_inv = np.linalg.inv(K_)                     # :43
        return self

    def predict(self, x, return_std=False, return_cov=False):
        """ref: gaussian_process.py:46-55 (std - sqrt(noise_level) quirk at :49)."""
        args = (x, self.X, self.L_, self.alpha_, self.constant_value, self._ls(), self.noise_level)
        if return_std:
            y, std = gpr_predict(*args, return_std=True, kind=self.kind)
            return y, std - np.sqrt(self.noise_level)
        if return_cov:
            return gpr_predict(*args, return_cov=True, kind=self.kind)
        return gpr_predict(*args, kind=self.kind)

    def samples(self, x, n_samples=10):
        """ref: gaussian_process.py:57-60 -> sklearn/_gpr.py:498-535 (sample_y, random_state=0):
        per target multivariate_normal(mean, cov) draws from RandomState(0); (n_samples, M, O)."""
        mean, cov = self.predict(x, return_cov=True)
        rng = np.random.RandomState(0)
        if mean.ndim == 1:
            return rng.multivariate_normal(mean, cov, n_samples).T
        ys = [rng.multivariate_normal(mean[:, t], cov[..., t], n_samples).T[:, np.newaxis] for t in range(mean.shape[1])]
        return np.transpose(np.hstack(ys), (2, 0, 1))

    def _dk(self, x):
        """dk[d,m,n] = (X[n,d]-x[m,d]) / l_d^2 * k(x_m, X_n).  ref: :72-87."""
        lscale = self.length_scale.reshape(-1, 1)
        k_star = kernel_cross(x, self.X, self.constant_value, self._ls())
        diff = self.X.T[:, None, :] - x.T[:, :, None]
        dk = diff / (lscale[:, :, None] ** 2) * k_star
        return lscale, k_star, dk

    def derivative(self, x, return_var=False):
        """ref: gaussian_process.py:63-102.  J (M,O,D); Jvar (M,O,D)."""
        lscale, _, dk = self._dk(x)
        alfa = self.K_inv @ self.Y                                   # :73
        df_dx = (dk.transpose(1, 0, 2) @ alfa).transpose(0, 2, 1)    # :88-90
        if not return_var:
            return df_dx
        q = np.sum((dk @ self.K_inv) * dk, axis=2)                   # :95-97
        var = self.prior_var / (lscale ** 2) - q                     # :98  (D or 1, M)
        Sigma = np.repeat(var[None, :, :], self.n_outputs, axis=0).transpose(2, 0, 1)
        return df_dx, Sigma

    def derivative_of_variance(self, x):
        """ref: gaussian_process.py:104-126.  (D, M)."""
        _, k_star, dk = self._dk(x)
        return -2 * np.sum((dk @ self.K_inv) * k_star, axis=2)


# --------------------------------------------------------------------------- fast (BLAS-friendly) variant
def posterior_all_fast(Xq, X, L, a, constant_value, length_scale, noise_level,
                       want_jvar=False, chunk=2048):
    """Same numbers as GaussianProcessOracle.predict/derivative, but with
    C-contiguous operands and triangular solves instead of the explicit inverse
    (the reference's `dk @ K_inv` at :95 hits numpy's non-BLAS path).  Used as the
    timed CPU baseline; checked against the faithful forms in tests.

    Returns mean (M,O), var (M,) raw clipped variance, J (M,O,D), Jvar (M,D) or None.
    """
    ls = np.broadcast_to(np.atleast_1d(np.asarray(length_scale, np.float64)), (X.shape[1],))
    M, D = Xq.shape
    O = a.shape[1]
    mean = np.empty((M, O)); var = np.empty(M); J = np.empty((M, O, D))
    Jvar = np.empty((M, D)) if want_jvar else None
    Xs = X / ls
    for s in range(0, M, chunk):
        xq = Xq[s:s + chunk]
        Ks = constant_value * np.exp(-0.5 * cdist(xq / ls, Xs, metric="sqeuclidean"))  # (m,N)
        mean[s:s + chunk] = Ks @ a
        V = solve_triangular(L, Ks.T, lower=True, check_finite=False)
        v = constant_value + noise_level - np.einsum("ij,ij->j", V, V)
        var[s:s + chunk] = np.maximum(v, 0.0)
        for d in range(D):
            dk = np.ascontiguousarray(((X[None, :, d] - xq[:, None, d]) / ls[d] ** 2) * Ks)  # (m,N)
            J[s:s + chunk, :, d] = dk @ a
            if want_jvar:
                Vd = solve_triangular(L, dk.T, lower=True, check_finite=False)
                Jvar[s:s + chunk, d] = constant_value / ls[d] ** 2 - np.einsum("ij,ij->j", Vd, Vd)
    return mean, var, J, Jvar


# --------------------------------------------------------------------------- affine + transport algebra
class AffineTransformOracle:
    """ref: models/affine_trasformation.py:8-57 (Kabsch fit, predict, derivative)."""

    def __init__(self, do_scale=False, do_rotation=True):
        self.do_scale = do_scale
        self.do_rotation = do_rotation
        self.scale = 1

    def fit(self, S, T):
        assert len(S) == len(T)
        self.S_centroid = S.mean(axis=0)
        self.T_centroid = T.mean(axis=0)
        Sc = S - self.S_centroid
        Tc = T - self.T_centroid
        n, D = S.shape
        if (not self.do_rotation) or (D == 2 and n < 2) or (D == 3 and n < 3):   # :25
            R = np.eye(D)
        else:
            U, _, Vt = np.linalg.svd(Sc.T @ Tc)                                  # :29-31
            V = Vt.T
            R = V @ U.T
            if np.linalg.det(R) < 0:                                             # :35-37
                V[:, -1] *= -1
                R = V @ U.T
        self.rotation_matrix = R
        if self.do_scale:                                                        # :39-41
            Sr = (R @ Sc.T).T
            self.scale = np.sum(Sr * Tc) / np.sum(Sr ** 2)
        return self

    def predict(self, x):
        return self.scale * (self.rotation_matrix @ (x - self.S_centroid).T).T + self.T_centroid  # :51-53

    def derivative(self, x):
        return np.repeat(self.rotation_matrix[None], x.shape[0], axis=0)         # :55-57 (scale ignored)


def transport_oracle(gp: GaussianProcessOracle, source, target, traj, vel=None,
                     do_scale=False, do_rotation=True):
    """ref: transportation/policy_transportation.py:16-59 (fit, transport,
    transport_velocity) around an already-parameterised GaussianProcessOracle."""
    aff = AffineTransformOracle(do_scale, do_rotation).fit(source, target)
    src = aff.predict(source)
    gp.fit(src, target - src)                                                    # :20-24
    pos = aff.predict(traj)
    mean, std = gp.predict(pos, return_std=True)                                 # :30
    out = {"traj": pos + mean, "std": std, "rotation": aff.rotation_matrix, "scale": aff.scale}
    if vel is not None:
        Jg = aff.derivative(traj)
        Jp, Jpv = gp.derivative(pos, return_var=True)                            # :41
        Jphi = Jg + Jp @ Jg                                                      # :45
        v = vel[:, :, None]
        vr = Jg @ v
        out["var_vel"] = (Jpv @ vr ** 2)[:, :, 0]                                # :52
        out["vel"] = (Jphi @ v)[:, :, 0]                                         # :54
    return out


def resample_oracle(surface, num_points=20):
    """ref: utils.py:7-45 — arc-length resampling of a 2-D polyline."""
    def dist(p, q):
        return np.sqrt((q[0] - p[0]) ** 2 + (q[1] - p[1]) ** 2)
    total = np.sum([dist(surface[i], surface[i + 1]) for i in range(len(surface) - 1)])
    spacing = total / (num_points - 1)
    out = [surface[0]]
    cur = surface[0]
    remaining = spacing
    for p in surface[1:]:
        dn = dist(cur, p)
        if remaining <= dn:
            t = remaining / dn
            cur = [cur[0] + t * (p[0] - cur[0]), cur[1] + t * (p[1] - cur[1])]
            out.append(cur)
            remaining = spacing
        else:
            cur = p
            remaining -= dn
    while len(out) < num_points:
        out.append(surface[-1])
    return np.array(out)


def svgp_exact_oracle(x, Z, Sigma, y, outputscale, lengthscale):
    """ref: models/torch/stocastic_variational_gaussian_process_derivatives.py:72-78 (K_inv = inv(K_uu + Sigma),
    alpha = K_inv y), :113-129 (mean = k* alpha, std = sqrt(diag(k** - k* K_inv k*^T)), k** = outputscale),
    :132-153 (J = dk*/dx alpha, var' = outputscale/l_d^2 - diag(dk* K_inv dk*^T)); gpytorch ScaleKernel(RBF ARD):
    outputscale_t exp(-0.5 |(x-z)/l|^2).  K_inv[t] per task (see svgp_exact.py on the reference's :142).
    PARITY UNPINNED (gpytorch absent, no fixture in the reference).  Returns mean (M,T), std (M,T), J (M,T,D),
    J_std (M,T,D)."""
    ls = np.broadcast_to(np.atleast_1d(np.asarray(lengthscale, np.float64)), (Z.shape[1],))
    T = Sigma.shape[0]
    M, D = x.shape
    R = np.exp(-0.5 * cdist(x / ls, Z / ls, metric="sqeuclidean"))            # (M,Z)
    Ruu = np.exp(-0.5 * cdist(Z / ls, Z / ls, metric="sqeuclidean"))
    mean = np.empty((M, T)); std = np.empty((M, T)); J = np.empty((M, T, D)); Js = np.empty((M, T, D))
    for t in range(T):
        Kinv = np.linalg.inv(outputscale[t] * Ruu + Sigma[t])
        a = Kinv @ np.reshape(y[t], (-1,))
        ks = outputscale[t] * R
        mean[:, t] = ks @ a
        std[:, t] = np.sqrt(outputscale[t] - np.einsum("mz,zw,mw->m", ks, Kinv, ks))
        for d in range(D):
            dk = (Z[None, :, d] - x[:, None, d]) / ls[d] ** 2 * ks
            J[:, t, d] = dk @ a
            Js[:, t, d] = np.sqrt(outputscale[t] / ls[d] ** 2 - np.einsum("mz,zw,mw->m", dk, Kinv, dk))
    return mean, std, J, Js


def svgp_exact_oracle_fast(x, Z, Sigma, y, outputscale, lengthscale, dtype=np.float64):
    """Same quantities as svgp_exact_oracle with the quadratic forms as matrix products (BLAS) — for Z in the
    thousands — and in the arithmetic `dtype`: float32 restates what the reference computes (it casts inputs with
    `.float()` and inverts K_uu + Sigma in fp32, :72-78), float64 is the yardstick.  ref lines as above."""
    f = np.dtype(dtype).type
    x = np.asarray(x, dtype); Z = np.asarray(Z, dtype); Sigma = np.asarray(Sigma, dtype)
    ls = np.broadcast_to(np.atleast_1d(np.asarray(lengthscale, dtype)), (Z.shape[1],))
    T = Sigma.shape[0]
    M, D = x.shape

    def sq(a, b):
        return np.maximum((a * a).sum(1)[:, None] + (b * b).sum(1)[None, :] - f(2) * (a @ b.T), f(0))
    R = np.exp(f(-0.5) * sq(x / ls, Z / ls))
    Ruu = np.exp(f(-0.5) * sq(Z / ls, Z / ls))
    mean = np.empty((M, T), dtype); std = np.empty((M, T), dtype); J = np.empty((M, T, D), dtype); Js = np.empty((M, T, D), dtype)
    for t in range(T):
        os_t = f(outputscale[t])
        Kinv = np.linalg.inv(os_t * Ruu + Sigma[t])
        a = Kinv @ np.reshape(np.asarray(y[t], dtype), (-1,))
        ks = os_t * R
        mean[:, t] = ks @ a
        std[:, t] = np.sqrt(np.maximum(os_t - ((ks @ Kinv) * ks).sum(1), f(0)))
        for d in range(D):
            dk = (Z[None, :, d] - x[:, None, d]) / ls[d] ** 2 * ks
            J[:, t, d] = dk @ a
            Js[:, t, d] = np.sqrt(np.maximum(os_t / ls[d] ** 2 - ((dk @ Kinv) * dk).sum(1), f(0)))
    return mean, std, J, Js


def svgp_synthetic_problem(Zn, M, T=3, D=3, seed=0, qseed=1):
    """SURVEY §8d cfg5 inputs: inducing points U[0,1]^D, Sigma_pseudo = A A^T / Z + 1e-3 I (A ~ N(0,1)), y_pseudo ~
    N(0,1), outputscale 1, length-scale 0.2, queries U[-0.1,1.1]^D.  Returns Z, Sigma (T,Z,Z), y (T,Z),
    outputscale (T,), lengthscale (D,), Xq."""
    rng = np.random.default_rng(seed)
    Z = rng.uniform(0, 1, (Zn, D))
    Sigma = np.empty((T, Zn, Zn))
    for t in range(T):
        A = rng.standard_normal((Zn, Zn))
        Sigma[t] = A @ A.T / Zn + 1e-3 * np.eye(Zn)
    y = rng.standard_normal((T, Zn))
    Xq = np.random.default_rng(qseed).uniform(-0.1, 1.1, (M, D))
    return Z, Sigma, y, np.ones(T), np.full(D, 0.2), Xq


def synthetic_problem(N, M, D=3, seed=0, qseed=1):
    """SURVEY §8d synthetic inputs (identical for CPU and GPU)."""
    rng = np.random.default_rng(seed)
    X = rng.uniform(0, 1, (N, D))
    Y = 0.05 * np.sin(4 * X) + 0.01 * rng.standard_normal((N, D))
    Xq = np.random.default_rng(qseed).uniform(-0.1, 1.1, (M, D))
    return X, Y, Xq
