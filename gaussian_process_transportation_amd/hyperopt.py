"""Hyper-parameter search for GaussianProcess(optimizer='fmin_l_bfgs_b') — the reference delegates
it to sklearn (GaussianProcessRegressor.fit, sklearn/_gpr.py:296-338: L-BFGS-B on the negative
log-marginal likelihood from the kernel's theta plus n_restarts_optimizer log-uniform restarts
drawn from the global numpy RNG).  SURVEY §8f row 1: not on the GPU path yet."""


def optimize_hyperparameters(gp, c, ls, noise):
    raise NotImplementedError(
        "optimizer='fmin_l_bfgs_b' (log-marginal-likelihood search) is not implemented on the GPU path yet; "
        "construct GaussianProcess(..., optimizer=None) with fixed hyper-parameters.  There is no CPU fallback.")
