"""Hyper-parameter search for GaussianProcess(optimizer='fmin_l_bfgs_b').

The reference delegates this to scikit-learn (`GaussianProcessRegressor.fit`,
sklearn/gaussian_process/_gpr.py:296-338): L-BFGS-B on the negative log-marginal likelihood in
theta = log(hyper-parameters), first from the kernel's own theta, then from `n_restarts_optimizer`
starting points drawn log-uniformly inside the kernel's bounds with the GLOBAL numpy RNG
(`random_state=None` -> `np.random.mtrand._rand`), keeping the best optimum.

Here the driver is the same (scipy's L-BFGS-B on the host — or the caller's own `optimizer(obj_func,
initial_theta, bounds)` callable, sklearn's protocol — same bounds, same RNG draws in the same order); every objective evaluation — Gram, Cholesky, alpha, K^-1 and the traces of the gradient
(_gpr.py:537-652) — runs on the GPU (`gpt_lml_objective`).  The scikit-learn kernel object is
used only as the container of theta / bounds / fixed flags."""
from __future__ import annotations

import os
import warnings

import numpy as np
import scipy.optimize

from . import _lib

_PARAM_C = "k1__k1__constant_value"
_PARAM_LS = "k1__k2__length_scale"
_PARAM_NOISE = "k2__noise_level"


def _unpack(kernel, theta):
    """theta (log-space, non-fixed hyper-parameters only) -> (c, ls array, noise) via a kernel clone."""
    p = kernel.clone_with_theta(theta).get_params()
    return float(p[_PARAM_C]), np.atleast_1d(np.asarray(p[_PARAM_LS], dtype=np.float64)), float(p[_PARAM_NOISE])


def _make_unpack(kernel, free, n_ls):
    """The same map without cloning the kernel (sklearn's clone costs more than the GPU's share of a small objective
    evaluation): the full parameter vector [c, ls.., noise] with the fixed entries filled in once, the free ones
    exp(theta) as the kernel's theta setter does (sklearn/gaussian_process/kernels.py: `np.exp(theta[i:...])`)."""
    p = kernel.get_params()
    fixed = np.concatenate([[float(p[_PARAM_C])], np.atleast_1d(np.asarray(p[_PARAM_LS], dtype=np.float64)), [float(p[_PARAM_NOISE])]])

    def unpack(theta):
        v = fixed.copy()
        v[free] = np.exp(theta)
        return float(v[0]), v[1:1 + n_ls].copy(), float(v[1 + n_ls])
    return unpack


def _free_mask(kernel, n_ls):
    """Which entries of the full gradient [c, ls (n_ls), noise] belong to theta (hyper-parameters whose
    bounds are not "fixed"), in theta order."""
    mask = []
    for hp in kernel.hyperparameters:
        width = 1 if hp.n_elements == 1 else hp.n_elements
        mask.extend([not hp.fixed] * width)
    if len(mask) != 2 + n_ls:
        raise ValueError("unexpected hyper-parameter layout for ConstantKernel * RBF + WhiteKernel")
    return np.array(mask, dtype=bool)


def optimize_hyperparameters(gp, c0, ls0, noise0):
    """Returns (c, ls, noise, lml) maximising the log-marginal likelihood as sklearn would."""
    kernel = gp._kernel_in
    if kernel.n_dims == 0:
        return c0, ls0, noise0, None
    if gp.optimizer != "fmin_l_bfgs_b" and not callable(gp.optimizer):
        raise ValueError(f"Unknown optimizer {gp.optimizer}.")        # sklearn/_gpr.py:668-669
    if gp._handle is None:
        make = getattr(gp, "_new_handle", None)            # (devices=[...]: the group whose first handle runs the search)
        gp._handle = make() if make is not None else _lib.Handle(gp.device)
    h = gp._handle
    n_ls = int(np.size(ls0))
    free = _free_mask(kernel, n_ls)
    jitter = gp.alpha
    X = _lib.as_f64(gp.X, 2, "X")                          # validated once; every evaluation reuses these arrays
    Y = _lib.as_f64(gp.Y, 2, "y")

    unpack = _make_unpack(kernel, free, n_ls)
    ktype = gp._ktype

    def objective(theta, handle=h):
        c, ls, noise = unpack(theta)
        try:
            lml, grad = handle.lml_objective(X, Y, ls, c, noise, jitter, ktype)   # one C call, no prediction-side model
        except np.linalg.LinAlgError:                      # _gpr.py:587-590: -inf LML, zero gradient
            return np.inf, np.zeros_like(theta)
        return -lml, -grad[free]

    def obj_func(theta, eval_gradient=True):               # the callable-optimizer protocol of sklearn/_gpr.py:296-305
        value, grad = objective(np.asarray(theta, dtype=np.float64))
        return (value, grad) if eval_gradient else value

    def run(theta_init, bounds, handle=h, deferred=None):
        if callable(gp.optimizer):                          # sklearn/_gpr.py:664-667
            theta_opt, func_min = gp.optimizer(obj_func, theta_init, bounds=bounds)
            return np.asarray(theta_opt, dtype=np.float64), float(func_min)
        res = scipy.optimize.minimize(objective, theta_init, args=(handle,), method="L-BFGS-B", jac=True, bounds=bounds)
        if res.status != 0:                                # sklearn's _check_optimize_result("lbfgs", ...)
            msg = f"lbfgs failed to converge (status={res.status}): {res.message}"
            if deferred is None:
                warnings.warn(msg)
            else:
                deferred.append(msg)                        # a worker thread: the caller's thread warns (the warnings module's state is global)
        return res.x, res.fun

    bounds = kernel.bounds
    n_restarts = gp.n_restarts_optimizer
    if n_restarts > 0 and not np.isfinite(bounds).all():
        run(kernel.theta, bounds)                           # sklearn runs the first optimisation before it checks (_gpr.py:311-318)
        raise ValueError("Multiple optimizer restarts (n_restarts_optimizer>0) requires that all bounds are finite.")
    workers = _restart_workers(gp, n_restarts, X.shape[0])
    if workers <= 1:
        optima = [run(kernel.theta, bounds)]
        rng = np.random.mtrand._rand                       # check_random_state(None): the global RandomState
        for _ in range(n_restarts):
            theta_initial = rng.uniform(bounds[:, 0], bounds[:, 1])
            optima.append(run(theta_initial, bounds))
    else:
        optima = _run_concurrently(gp, h, run, kernel.theta, bounds, n_restarts, workers)
    values = [v for _, v in optima]
    best = int(np.argmin(values))
    c, ls, noise = _unpack(kernel, optima[best][0])
    return c, ls, noise, -values[best]


def _restart_workers(gp, n_restarts, n_points):
    """How many L-BFGS-B runs are driven at once.  The runs of sklearn's restart loop are independent: the start points come
    from the global RNG, which nothing inside a run touches, so drawing them all first gives the same points.  One
    evaluation of the objective at the reference's sizes (N = 400 .. 2500) is a chain of short launches that leaves most of
    the 256 CUs idle, so several runs on their own handles (streams, workspaces) overlap on the GPU; every run executes
    the same kernels on the same data as it would alone, so the optimum found is bit-identical.  Off for a caller's own
    optimizer (it may use the RNG or not be re-entrant), for large problems (one evaluation fills the GPU, and every handle
    holds 2 N^2 doubles) and with GPT_OPT_WORKERS=1."""
    if callable(gp.optimizer) or n_restarts < 1:
        return 1
    limit = int(os.environ.get("GPT_OPT_WORKERS", "6"))
    if n_points > int(os.environ.get("GPT_OPT_WORKERS_MAX_N", "4096")):
        return 1
    return max(1, min(limit, n_restarts + 1))


def _run_concurrently(gp, h, run, theta0, bounds, n_restarts, workers):
    import queue
    from concurrent.futures import ThreadPoolExecutor
    rng = np.random.mtrand._rand
    starts = [theta0] + [rng.uniform(bounds[:, 0], bounds[:, 1]) for _ in range(n_restarts)]   # same draws, same order
    extra = [_lib.Handle(gp.device) for _ in range(workers - 1)]
    pool = queue.SimpleQueue()
    for handle in [h] + extra:
        pool.put(handle)

    def job(theta_init):
        handle = pool.get()
        try:
            msgs = []
            return run(theta_init, bounds, handle, msgs), msgs
        finally:
            pool.put(handle)

    try:
        with ThreadPoolExecutor(max_workers=workers) as ex:
            results = list(ex.map(job, starts))
    finally:
        for handle in extra:
            handle.close()
    for _, msgs in results:
        for m in msgs:
            warnings.warn(m)
    return [out for out, _ in results]
