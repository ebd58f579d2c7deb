"""Pins the CPU oracle (oracle/gp_oracle.py) to outputs of the reference itself
(tests/golden/*.npz, written by tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest

from oracle import gp_oracle as orc
from tests.conftest import assert_parity, load_golden

TIGHT = 1e-9      # oracle vs reference: both fp64 CPU, same algorithm
SYN = ["synthetic_3d_N64", "synthetic_3d_N64_iso", "synthetic_3d_N64_nan", "synthetic_3d_N256",
       "synthetic_3d_N1024", "synthetic_5d_N200", "synthetic_8d_N128", "synthetic_12d_N160", "synthetic_15d_N96"]


def fitted(g):
    gp = orc.GaussianProcessOracle(g["constant_value"], g["length_scale"], g["noise_level"],
                                   alpha=float(g["alpha"]))
    gp.fit(g["X"], g["Y"])
    return gp


@pytest.mark.parametrize("name", SYN)
def test_fit_factor_and_alpha(name):
    g = load_golden(name)
    gp = fitted(g)
    assert gp.n_samples == int(g["n_samples"])            # pre-NaN-filter count quirk
    assert_parity(gp.alpha_, g["alpha_"], 1e-7, "alpha_")
    if "L_" in g:
        assert_parity(gp.L_, g["L_"], TIGHT, "L_")
    else:
        assert_parity(np.diag(gp.L_), g["L_diag"], TIGHT, "diag L")
        assert_parity(gp.L_[:, 0], g["L_col0"], TIGHT, "L[:,0]")
        assert_parity(gp.L_[-1], g["L_lastrow"], TIGHT, "L[-1]")
    assert float(gp.noise_var_) == pytest.approx(float(g["noise_var_"]), rel=1e-15)
    assert float(gp.prior_var) == pytest.approx(float(g["prior_var"]), rel=1e-15)


@pytest.mark.parametrize("name", SYN)
def test_predict_and_derivatives(name):
    g = load_golden(name)
    gp = fitted(g)
    Xq = g["Xq"]
    assert_parity(gp.predict(Xq), g["mean_only"], TIGHT, "mean only")
    m, s = gp.predict(Xq, return_std=True)
    assert_parity(m, g["mean"], TIGHT, "mean")
    assert_parity(s, g["std"], 1e-7, "std")
    J, Jv = gp.derivative(Xq, return_var=True)
    assert_parity(J, g["J"], 1e-7, "J")
    assert_parity(Jv, g["Jvar"], 1e-7, "Jvar")
    assert_parity(gp.derivative_of_variance(Xq), g["dvar"], 1e-7, "dvar")
    if "cov" in g:
        n = g["cov"].shape[0]
        _, cov = gp.predict(Xq[:n], return_cov=True)
        assert_parity(cov, g["cov"], 1e-7, "cov")
        assert_parity(gp.samples(Xq[:n]), g["samples"], 1e-6, "samples")


@pytest.mark.parametrize("name", ["synthetic_3d_N64", "synthetic_3d_N256", "synthetic_5d_N200", "synthetic_8d_N128", "synthetic_12d_N160", "synthetic_15d_N96"])
def test_fast_variant_matches_faithful(name):
    g = load_golden(name)
    gp = fitted(g)
    mean, var, J, Jvar = orc.posterior_all_fast(g["Xq"], gp.X, gp.L_, gp.alpha_, gp.constant_value,
                                                gp.length_scale, gp.noise_level, want_jvar=True, chunk=32)
    assert_parity(mean, g["mean"], TIGHT, "mean")
    assert_parity(np.sqrt(var) - np.sqrt(gp.noise_level), g["std"][:, 0], 1e-7, "std")
    assert_parity(J, g["J"], 1e-7, "J")
    assert_parity(Jvar, g["Jvar"][:, 0, :], 1e-7, "Jvar")


@pytest.mark.parametrize("name", ["synthetic_3d_N64", "synthetic_3d_N64_iso", "synthetic_3d_N256", "synthetic_5d_N200", "synthetic_8d_N128", "synthetic_12d_N160", "synthetic_15d_N96"])
def test_lml_and_gradient(name):
    g = load_golden(name)
    n_ls = g["length_scale"].size
    for th, v, gr in zip(g["lml_theta"], g["lml_value"], g["lml_grad"]):
        val, grad = orc.log_marginal_likelihood(th, g["X"], g["Y"], n_ls, alpha=float(g["alpha"]))
        assert val == pytest.approx(float(v), rel=1e-10)
        assert_parity(grad, gr, 1e-8, "lml grad")


def test_letterS_transport():
    g = load_golden("letterS_2d")
    # resample restatement against the reference's resampled arrays
    assert_parity(orc.resample_oracle(g["demo_raw"], 400), g["demo"], 1e-12, "resample demo")
    assert_parity(orc.resample_oracle(g["floor_raw"], 20), g["source"], 1e-12, "resample floor")
    assert_parity(orc.resample_oracle(g["newfloor_raw"], 20), g["target"], 1e-12, "resample newfloor")
    gp = orc.GaussianProcessOracle(g["constant_value"], g["length_scale"], g["noise_level"])
    out = orc.transport_oracle(gp, g["source"], g["target"], g["demo"], g["delta"])
    assert_parity(out["rotation"], g["rotation"], 1e-12, "R")
    assert_parity(gp.X, g["gp_X"], 1e-12, "rotated source")
    assert_parity(gp.alpha_, g["alpha_"], 1e-7, "alpha_")
    assert_parity(gp.L_, g["L_"], TIGHT, "L_")
    assert_parity(out["traj"], g["traj"], TIGHT, "traj")
    assert_parity(out["std"], g["std"], 1e-7, "std")
    assert_parity(out["vel"], g["vel"], 1e-7, "vel")
    assert_parity(out["var_vel"], g["var_vel"], 1e-6, "var_vel")
    # do_scale=True branch
    gp2 = orc.GaussianProcessOracle(g["constant_value"], g["length_scale"], g["noise_level"])
    out2 = orc.transport_oracle(gp2, g["source"], g["target"], g["demo"], g["delta"], do_scale=True)
    assert float(out2["scale"]) == pytest.approx(float(g["scale2"]), rel=1e-12)
    assert_parity(out2["traj"], g["traj2"], TIGHT, "traj2")
    assert_parity(out2["vel"], g["vel2"], 1e-7, "vel2")
    assert_parity(out2["var_vel"], g["var_vel2"], 1e-6, "var_vel2")
    # LML of the fitted theta as sklearn reported it
    val = orc.log_marginal_likelihood(g["theta_fit"], g["gp_X"], g["gp_Y"], 2, eval_gradient=False)
    assert val == pytest.approx(float(g["lml_fit"]), rel=1e-10)


def test_surface3d_transport():
    g = load_golden("surface_3d")
    gp = orc.GaussianProcessOracle(g["constant_value"], g["length_scale"], g["noise_level"])
    out = orc.transport_oracle(gp, g["source"], g["target"], g["demo"], g["delta"])
    assert_parity(out["rotation"], g["rotation"], 1e-12, "R")
    assert_parity(gp.alpha_, g["alpha_"], 1e-6, "alpha_")
    assert_parity(out["traj"], g["traj"], 1e-8, "traj")
    assert_parity(out["std"], g["std"], 1e-6, "std")
    assert_parity(out["vel"], g["vel"], 1e-6, "vel")
    assert_parity(out["var_vel"], g["var_vel"], 1e-5, "var_vel")


@pytest.mark.parametrize("tag,kind", [("12", "matern12"), ("32", "matern32"), ("52", "matern52")])
def test_matern_kernels(tag, kind):
    """The examples' dynamics-GP kernel family (C * Matern(nu) + White) against the reference."""
    g = load_golden("matern_2d")
    gp = orc.GaussianProcessOracle(0.3, np.array([1.5, 2.5]), 0.01, kind=kind).fit(g["X"], g["Y"])
    assert_parity(gp.alpha_, g[f"m{tag}_alpha_"], 1e-7, "alpha_")
    assert_parity(np.diag(gp.L_), g[f"m{tag}_Ldiag"], TIGHT, "diag L")
    m, s = gp.predict(g["grid"], return_std=True)
    assert_parity(m, g[f"m{tag}_mean"], 1e-8, "mean")
    assert_parity(s, g[f"m{tag}_std"], 1e-7, "std")
    _, cov = gp.predict(g["grid"][:12], return_cov=True)
    assert_parity(cov, g[f"m{tag}_cov"], 1e-7, "cov")
    for th, v, gr in zip(g[f"m{tag}_lml_theta"], g[f"m{tag}_lml_value"], g[f"m{tag}_lml_grad"]):
        val, grad = orc.log_marginal_likelihood(th, g["X"], g["Y"], 2, kind=kind)
        assert val == pytest.approx(float(v), rel=1e-10)
        assert_parity(grad, gr, 1e-7, "lml grad (ARD)")
    val, grad = orc.log_marginal_likelihood(g[f"m{tag}_iso_theta"], g["X"][::4], g["Y"][::4], 1, kind=kind)
    assert val == pytest.approx(float(g[f"m{tag}_iso_value"]), rel=1e-10)
    assert_parity(grad, g[f"m{tag}_iso_grad"], 1e-7, "lml grad (isotropic)")


@pytest.mark.parametrize("tag,kind", [("12", "matern12"), ("32", "matern32"), ("52", "matern52")])
def test_matern_kernels_in_five_dimensions(tag, kind):
    """C * Matern(nu) + White on a 5-D input, ARD and isotropic, against the reference class (matern_5d.npz)."""
    g = load_golden("matern_5d")
    for kk, ls, n_ls in (("ard", g["length_scale"], 5), ("iso", np.array([float(g["iso_length_scale"])]), 1)):
        pre = f"m{tag}_{kk}_"
        gp = orc.GaussianProcessOracle(0.3, ls, 0.01, kind=kind).fit(g["X"], g["Y"])
        assert_parity(gp.alpha_, g[pre + "alpha_"], 1e-7, "alpha_")
        assert_parity(np.diag(gp.L_), g[pre + "Ldiag"], TIGHT, "diag L")
        m, s = gp.predict(g["Xq"], return_std=True)
        assert_parity(m, g[pre + "mean"], 1e-8, "mean")
        assert_parity(s, g[pre + "std"], 1e-7, "std")
        _, cov = gp.predict(g["Xq"][:12], return_cov=True)
        assert_parity(cov, g[pre + "cov"], 1e-7, "cov")
        for th, v, gr in zip(g[pre + "lml_theta"], g[pre + "lml_value"], g[pre + "lml_grad"]):
            val, grad = orc.log_marginal_likelihood(th, g["X"], g["Y"], n_ls, kind=kind)
            assert val == pytest.approx(float(v), rel=1e-10)
            assert_parity(grad, gr, 1e-7, f"lml grad ({kk})")
