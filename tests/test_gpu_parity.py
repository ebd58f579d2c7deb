"""Parity of the HIP path (through the C ABI and the Python mirror of the reference API) against
golden vectors captured from the reference and against the CPU oracle.  Needs an MI355X.

Tolerance: north_star's 1e-5 relative (fp64), applied as max-norm relative error per output array
plus elementwise rtol/atol (tests/conftest.py:assert_parity)."""
import numpy as np
import pytest

from tests.conftest import assert_parity, load_golden, relmax

pytestmark = pytest.mark.gpu

RTOL = 1e-5
# Optimizer-on comparisons: the same L-BFGS-B driver fed objective values that agree to rounding takes the same steps, so the
# fitted hyper-parameters agree far better than an optimiser's tolerance suggests.  Bounds ~ 100 x the values measured in
# round 3 (printed by the tests; round 2 allowed 1e-3 everywhere): theta 1.4e-13 (surface-3D), 6.1e-13 (Matern dynamics GP),
# 2.7e-7 (6-D, 3 runs on noisy data), 1.4e-7 (surface-3D example: its dynamics GP is fitted on the transported demo);
# outputs 5e-14 .. 3e-12 (surface-3D, Matern), 4e-12 / 1.4e-11 (letter-S example), 8e-9 / 6.5e-8 (surface-3D example).
# (the absolute bounds assume the GPU objective drives L-BFGS-B along exactly the trajectory that produced the CPU golden;
# a benign change of summation order can flip a line-search step.  The gate of every optimizer-on test is
# `err <= what the theta difference explains + 1e-5`; these are the looser absolute backstops — measured values in DESIGN.md section 2:
# 1.4e-13 / 5e-14 (surface-3D), 6e-13 / 3e-12 (Matern).)
THETA_TOL = {"surface3d": 1e-6, "sixd": 1e-5, "matern": 1e-6, "surface3d_example": 1e-5}
OUT_TOL = {"surface3d": 1e-6, "matern": 1e-6, "letterS_example": 1e-6, "surface3d_example": 1e-5}


def sk_kernel(c, ls, noise):
    from sklearn.gaussian_process.kernels import RBF, WhiteKernel, ConstantKernel as C
    return C(float(c)) * RBF(length_scale=np.asarray(ls, dtype=float).tolist() if np.size(ls) > 1 else [float(np.ravel(ls)[0])]) \
        + WhiteKernel(float(noise))


def fitted_gp(g):
    from gaussian_process_transportation_amd import GaussianProcess
    gp = GaussianProcess(kernel=sk_kernel(g["constant_value"], g["length_scale"], g["noise_level"]),
                         alpha=float(g["alpha"]), optimizer=None, verbose=False)
    gp.fit(g["X"], g["Y"])
    return gp


SYN = ["synthetic_3d_N64", "synthetic_3d_N64_iso", "synthetic_3d_N64_nan", "synthetic_3d_N256", "synthetic_3d_N1024",
       "synthetic_5d_N200", "synthetic_8d_N128", "synthetic_12d_N160", "synthetic_15d_N96"]          # the last four: input dimension beyond 3 (the wide layouts: rows of 8 and of 16)


@pytest.mark.parametrize("name", SYN)
def test_golden_fit(name):
    g = load_golden(name)
    gp = fitted_gp(g)
    assert gp.n_samples == int(g["n_samples"])
    assert gp.X.shape[0] == g["alpha_"].shape[0]            # NaN rows dropped
    assert_parity(gp.gp.alpha_, g["alpha_"], RTOL, "alpha_")
    L = gp.gp.L_
    assert np.all(np.triu(L, 1) == 0.0)
    if "L_" in g:
        assert_parity(L, g["L_"], RTOL, "L_")
    else:
        assert_parity(np.diag(L), g["L_diag"], RTOL, "diag L")
        assert_parity(L[:, 0], g["L_col0"], RTOL, "L[:,0]")
        assert_parity(L[-1], g["L_lastrow"], RTOL, "L[-1]")
    assert gp.noise_var_ == pytest.approx(float(g["noise_var_"]), rel=1e-15)
    assert gp.prior_var == pytest.approx(float(g["prior_var"]), rel=1e-15)
    if "lml_value" in g:
        assert gp._handle.lml() == pytest.approx(float(g["lml_value"][0]), rel=1e-9)
    # K_inv as the reference caches it
    K = g["constant_value"] * np.exp(-0.5 * (((gp.X[:, None, :] - gp.X[None, :, :]) / g["length_scale"]) ** 2).sum(-1))
    K[np.diag_indices_from(K)] += gp.noise_var_
    assert_parity(gp.K_inv @ K, np.eye(len(K)), 1e-6, "K_inv K")


@pytest.mark.parametrize("name", SYN)
def test_golden_predict_and_derivatives(name):
    g = load_golden(name)
    gp = fitted_gp(g)
    Xq = g["Xq"]
    assert_parity(gp.predict(Xq), g["mean_only"], RTOL, "mean only")
    m, s = gp.predict(Xq, return_std=True)
    assert_parity(m, g["mean"], RTOL, "mean")
    assert_parity(s, g["std"], RTOL, "std")
    assert_parity(gp.derivative(Xq), g["J"], RTOL, "J (no var)")
    J, Jv = gp.derivative(Xq, return_var=True)
    assert_parity(J, g["J"], RTOL, "J")
    assert_parity(Jv, g["Jvar"], RTOL, "Jvar")
    assert_parity(gp.derivative_of_variance(Xq), g["dvar"], RTOL, "dvar")
    if "cov" in g:
        n = g["cov"].shape[0]
        m, cov = gp.predict(Xq[:n], return_cov=True)
        assert_parity(m, g["mean"][:n], RTOL, "mean (return_cov)")
        assert_parity(cov, g["cov"], RTOL, "posterior covariance")
        assert_parity(gp.samples(Xq[:n]), g["samples"], 1e-4, "samples (10 joint draws per output)")
    post = gp.posterior(Xq, jacobian_variance=True)
    assert_parity(post["mean"], g["mean"], RTOL, "fused mean")
    assert_parity(np.sqrt(post["var"]) - np.sqrt(float(g["noise_level"])), g["std"][:, 0], RTOL, "fused std")
    assert_parity(post["J"], g["J"], RTOL, "fused J")
    assert_parity(post["Jvar"], g["Jvar"][:, 0, :], RTOL, "fused Jvar")


def test_golden_n8192_spot_check():
    """Config-3 size against the reference itself: 256 queries at N=8192 (fixture holds no X/Y: regenerated
    from the documented seeds)."""
    g = load_golden("synthetic_3d_N8192")
    rng = np.random.default_rng(0)
    X = rng.uniform(0, 1, (8192, 3))
    Y = 0.05 * np.sin(4 * X) + 0.01 * rng.standard_normal((8192, 3))
    g = dict(g, X=X, Y=Y)
    gp = fitted_gp(g)
    assert_parity(gp.gp.alpha_, g["alpha_"], RTOL, "alpha_")
    L = gp.gp.L_
    assert_parity(np.diag(L), g["L_diag"], RTOL, "diag L")
    assert_parity(L[:, 0], g["L_col0"], RTOL, "L[:,0]")
    assert_parity(L[-1], g["L_lastrow"], RTOL, "L[-1]")
    del L
    m, s = gp.predict(g["Xq"], return_std=True)
    assert_parity(m, g["mean"], RTOL, "mean")
    assert_parity(s, g["std"], RTOL, "std")
    J, Jv = gp.derivative(g["Xq"], return_var=True)
    assert_parity(J, g["J"], RTOL, "J")
    assert_parity(Jv, g["Jvar"], RTOL, "Jvar")
    assert_parity(gp.derivative_of_variance(g["Xq"]), g["dvar"], RTOL, "dvar")


def _transport(g, do_scale=False):
    from gaussian_process_transportation_amd import GaussianProcessTransportation
    tr = GaussianProcessTransportation(kernel_transport=sk_kernel(g["constant_value"], g["length_scale"], g["noise_level"]),
                                       optimizer=None, verbose=False)
    tr.source_distribution = g["source"]
    tr.target_distribution = g["target"]
    tr.training_traj = g["demo"]
    tr.training_delta = g["delta"]
    tr.fit_transportation(do_scale=do_scale, do_rotation=True)
    tr.apply_transportation()
    return tr


def test_letterS_transport_fixed_theta():
    """Config 1 (2-D letter-S demo) with the hyper-parameters the reference's optimizer found."""
    g = load_golden("letterS_2d")
    tr = _transport(g)
    assert_parity(tr.method.affine_transform.rotation_matrix, g["rotation"], 1e-12, "R")
    assert_parity(tr.method.delta_map.gp.alpha_, g["alpha_"], RTOL, "alpha_")
    assert_parity(tr.method.delta_map.gp.L_, g["L_"], RTOL, "L_")
    assert_parity(tr.training_traj, g["traj"], RTOL, "traj")
    assert_parity(tr.std, g["std"], RTOL, "std")
    assert_parity(tr.training_delta, g["vel"], RTOL, "vel")
    assert_parity(tr.var_vel_transported, g["var_vel"], RTOL, "var_vel")
    assert tr.training_traj_old is g["demo"]
    smp = tr.sample_transportation()
    assert smp.shape == tuple(g["samples_full_shape"])
    # 400 closely spaced points under a smooth kernel: the posterior covariance has a large eigenspace degenerate at
    # ~noise_level, where the SVD basis used by multivariate_normal is arbitrary.  The golden file carries the evidence
    # (tests/golden/make_golden.py:case_letterS): recomputing the reference's OWN covariance in another order of
    # operations (it changes by 1.1e-13) moves the reference's own draws by up to 0.57 (`samples_recomputed_spread`).
    # So (1) the part of a draw that is basis-independent — its coordinates on the 29 leading, well-separated
    # eigenvectors of the reference covariance — is held to 1e-6, and (2) the rest stays inside the posterior band.
    pos_rot = tr.method.affine_transform.predict(g["demo"])
    mean_rot, cov = tr.method.delta_map.predict(pos_rot, return_cov=True)
    coords = np.einsum("nk,snt->skt", g["samples_eig_vectors"], smp - (pos_rot + mean_rot)[None])
    # a draw is z_i sqrt(s_i) v_i summed over the singular pairs, and LAPACK's SVD fixes the SIGN of v_i only up to
    # rounding-level details of its input: one sign per eigenvector (the same for all draws and targets) is free
    sign = np.sign(np.einsum("skt,skt->k", coords, g["samples_eig_coords"]))
    assert np.all(sign != 0)
    assert_parity(coords * sign[None, :, None], g["samples_eig_coords"], 1e-6, "draws on the leading eigenvectors")
    assert float(g["samples_recomputed_spread"]) > 0.1       # the reference is no closer to itself than this
    sd = np.sqrt(np.maximum(np.diag(cov[:, :, 0]), 0))[::8]
    assert np.all(np.abs(smp[:, ::8, :] - g["samples"]) <= 6 * np.sqrt(2) * sd[None, :, None] + 1e-9)
    tr2 = _transport(g, do_scale=True)
    assert float(tr2.method.affine_transform.scale) == pytest.approx(float(g["scale2"]), rel=1e-12)
    assert_parity(tr2.training_traj, g["traj2"], RTOL, "traj2")
    assert_parity(tr2.training_delta, g["vel2"], RTOL, "vel2")
    assert_parity(tr2.var_vel_transported, g["var_vel2"], RTOL, "var_vel2")


def test_surface3d_transport_fixed_theta():
    """The reference's 3-D surface demo (N=2500 sources, 460-point demo) at its fitted theta."""
    g = load_golden("surface_3d")
    tr = _transport(g)
    assert_parity(tr.method.delta_map.gp.alpha_, g["alpha_"], RTOL, "alpha_")
    assert_parity(tr.training_traj, g["traj"], RTOL, "traj")
    assert_parity(tr.std, g["std"], RTOL, "std")
    assert_parity(tr.training_delta, g["vel"], RTOL, "vel")
    assert_parity(tr.var_vel_transported, g["var_vel"], RTOL, "var_vel")


@pytest.mark.parametrize("N,M,D,O,ls", [(1, 5, 3, 3, (0.1,)), (3, 1, 2, 2, (0.3, 0.2)), (63, 257, 3, 3, (0.1, 0.2, 0.3)),
                                        (257, 300, 1, 5, (0.05,)), (700, 1031, 3, 1, (0.15,)), (1300, 513, 2, 6, (0.2, 0.1))])
def test_ragged_shapes_vs_oracle(N, M, D, O, ls):
    """Sizes off every tile boundary, D in 1..3, O from 1 to more than one alpha pass."""
    from gaussian_process_transportation_amd import _lib
    from oracle import gp_oracle as orc
    rng = np.random.default_rng(N + M)
    X = rng.uniform(0, 1, (N, D))
    Y = 0.05 * np.sin(4 * X[:, :1] + np.arange(O)[None, :]) + 0.01 * rng.standard_normal((N, O))
    Xq = rng.uniform(-0.1, 1.1, (M, D))
    c, noise, jit = 0.1, 1e-4, 1e-10
    h = _lib.Handle(0)
    h.fit(X, Y, np.asarray(ls), c, noise, jit)
    o = orc.GaussianProcessOracle(c, np.asarray(ls), noise, jit).fit(X, Y)
    L, a = h.export()
    assert_parity(L, o.L_, RTOL, "L")
    assert_parity(a, o.alpha_, RTOL, "alpha")
    out = h.predict_all(Xq, mean=True, var=True, J=True, Jvar=True, dvar=True)
    mean, std = o.predict(Xq, return_std=True)
    std = std if std.ndim == 1 else std[:, 0]
    J, Jv = o.derivative(Xq, return_var=True)
    assert_parity(out["mean"], np.reshape(mean, (M, O)), RTOL, "mean")
    assert_parity(out["var"], (std + np.sqrt(noise)) ** 2, RTOL, "var")
    assert_parity(out["J"], J, RTOL, "J")
    assert_parity(out["Jvar"], Jv[:, 0, :], RTOL, "Jvar")
    assert_parity(out["dvar"], o.derivative_of_variance(Xq), RTOL, "dvar")
    assert_parity(h.predict_all(Xq, var=True)["var"], out["var"], 1e-12, "var (1-column kernel) vs var (4-column kernel)")
    h.close()


@pytest.mark.parametrize("N,M,D,O,ls", [(5, 3, 4, 1, (0.5,)), (257, 300, 4, 5, (0.4, 0.6, 0.5, 0.3)), (700, 5000, 6, 2, (0.6,)),
                                        (1300, 460, 7, 3, (0.5, 0.7, 0.6, 0.8, 0.5, 0.9, 0.6)), (600, 1031, 8, 1, (0.8,)),
                                        (1100, 70, 5, 6, (0.3, 0.5, 0.4, 0.6, 0.35))])
def test_input_dimension_beyond_three_vs_oracle(N, M, D, O, ls):
    """The reference regressor takes any input dimension (models/gaussian_process.py:63-102 builds (D,M,N) operands for
    any D); here D = 4 .. 8 run on the wide layout (source rows of 8, 8 / 16 columns per query in the fused variance
    launch).  Sizes off the tile boundaries, few queries (cut sweeps) and many (whole rounds), isotropic and ARD."""
    from gaussian_process_transportation_amd import _lib
    from oracle import gp_oracle as orc
    rng = np.random.default_rng(N + M + D)
    X = rng.uniform(0, 1, (N, D))
    Y = 0.05 * np.sin(4 * X[:, :1] + np.arange(O)[None, :]) + 0.01 * rng.standard_normal((N, O))
    Xq = rng.uniform(-0.1, 1.1, (M, D))
    c, noise, jit = 0.1, 1e-4, 1e-10
    h = _lib.Handle(0)
    h.fit(X, Y, np.asarray(ls), c, noise, jit)
    o = orc.GaussianProcessOracle(c, np.asarray(ls), noise, jit).fit(X, Y)
    L, a = h.export()
    assert_parity(L, o.L_, RTOL, "L")
    assert_parity(a, o.alpha_, RTOL, "alpha")
    out = h.predict_all(Xq, mean=True, var=True, J=True, Jvar=True, dvar=True)
    sub = np.sort(rng.choice(M, min(M, 600), replace=False))          # the CPU oracle checks a subset of a large batch
    mean, std = o.predict(Xq[sub], return_std=True)
    std = std if std.ndim == 1 else std[:, 0]
    J, Jv = o.derivative(Xq[sub], return_var=True)
    assert_parity(out["mean"][sub], np.reshape(mean, (len(sub), O)), RTOL, "mean")
    assert_parity(out["var"][sub], (std + np.sqrt(noise)) ** 2, RTOL, "var")
    assert_parity(out["J"][sub], J, RTOL, "J")
    assert_parity(out["Jvar"][sub], Jv[:, 0, :], RTOL, "Jvar")
    assert_parity(out["dvar"][:, sub], o.derivative_of_variance(Xq[sub]), RTOL, "dvar")
    assert_parity(h.predict_all(Xq, var=True)["var"], out["var"], 1e-12, "var (1-column kernel) vs var (fused kernel)")
    assert_parity(h.predict_all(Xq, Jvar=True)["Jvar"], out["Jvar"], 1e-12, "Jvar alone vs fused")
    # the same model in fp32: against the fp64 result, at fp32's resolution of the prior scales
    h.set_dtype(_lib.GPT_F32)
    h.fit(X, Y, np.asarray(ls), c, noise, jit)
    o32 = h.predict_all(Xq, mean=True, var=True, J=True, Jvar=True)
    assert o32["var"].dtype == np.float32
    ils2 = float(np.max(1.0 / np.broadcast_to(np.asarray(ls), (D,)) ** 2))
    assert np.max(np.abs(o32["var"] - out["var"])) < 2e-4 * c
    assert np.max(np.abs(o32["Jvar"] - out["Jvar"])) < 2e-4 * c * ils2
    h.close()


def test_transportation_in_five_dimensions_vs_oracle():
    """The whole transport flow (affine pre-alignment, delta-map GP, transported positions / std / velocities / velocity
    variance: policy_transportation.py:16-59) on a 5-D state, against the CPU restatement of the same algebra."""
    from oracle import gp_oracle as orc
    rng = np.random.default_rng(8)
    D, N, M = 5, 300, 200
    src = rng.uniform(-1, 1, (N, D))
    Q, _ = np.linalg.qr(rng.standard_normal((D, D)))
    if np.linalg.det(Q) < 0:
        Q[:, 0] = -Q[:, 0]
    tgt = src @ Q.T + 0.3 + 0.1 * np.sin(2 * src) + 0.005 * rng.standard_normal((N, D))
    traj = rng.uniform(-0.9, 0.9, (M, D))
    vel = 0.05 * rng.standard_normal((M, D))
    c, ls, noise = 0.5, np.array([0.8, 1.0, 0.7, 0.9, 1.1]), 1e-3
    g = dict(constant_value=c, length_scale=ls, noise_level=noise, source=src, target=tgt, demo=traj, delta=vel)
    tr = _transport(g)
    want = orc.transport_oracle(orc.GaussianProcessOracle(c, ls, noise), src, tgt, traj, vel)
    assert_parity(tr.method.affine_transform.rotation_matrix, want["rotation"], 1e-10, "rotation")
    assert_parity(tr.training_traj, want["traj"], RTOL, "traj")
    assert_parity(tr.std, want["std"], RTOL, "std")
    assert_parity(tr.training_delta, want["vel"], RTOL, "vel")
    assert_parity(tr.var_vel_transported, want["var_vel"], RTOL, "var_vel")


def test_gaussian_process_class_in_six_dimensions_matches_sklearn_after_optimisation():
    """GaussianProcess (the reference's class surface) on a 6-D input with the optimizer on: the fitted hyper-parameters
    and the log-marginal likelihood against scikit-learn's own GaussianProcessRegressor run from the same start with the
    same global-RNG restarts (sklearn/_gpr.py:248-330), the predictions against sklearn's at the fitted kernel."""
    from sklearn.gaussian_process import GaussianProcessRegressor
    from gaussian_process_transportation_amd import GaussianProcess
    rng = np.random.default_rng(11)
    N, D = 300, 6
    X = rng.uniform(0, 1, (N, D))
    Y = np.column_stack([np.sin(3 * X[:, 0]) * X[:, 3], np.cos(2 * X[:, 1] + X[:, 5])]) + 0.02 * rng.standard_normal((N, 2))
    k0 = sk_kernel(1.0, 0.7 * np.ones(D), 0.01)
    np.random.seed(3)
    ref = GaussianProcessRegressor(kernel=k0, alpha=1e-10, n_restarts_optimizer=2).fit(X, Y)
    np.random.seed(3)
    gp = GaussianProcess(kernel=k0, alpha=1e-10, n_restarts_optimizer=2, verbose=False).fit(X, Y)
    assert gp.gp.log_marginal_likelihood_value_ == pytest.approx(ref.log_marginal_likelihood_value_, rel=1e-6)
    from tests.conftest import relmax
    print(f"6-D optimizer: theta vs sklearn {relmax(gp.gp.kernel_.theta, ref.kernel_.theta):.2e}")
    assert_parity(gp.gp.kernel_.theta, ref.kernel_.theta, THETA_TOL["sixd"], "fitted theta")
    Xq = rng.uniform(0, 1, (500, D))
    fixed = GaussianProcess(kernel=ref.kernel_, alpha=1e-10, optimizer=None, verbose=False).fit(X, Y)
    m_ref, s_ref = ref.predict(Xq, return_std=True)
    m, s = fixed.predict(Xq, return_std=True)
    assert_parity(m, m_ref, RTOL, "mean at sklearn's fitted kernel")
    noise = float(np.exp(ref.kernel_.theta[-1]))
    # the reference class reports sqrt(var without noise) (gaussian_process.py:57-58); sklearn's std includes the WhiteKernel
    assert_parity((s[:, 0] + np.sqrt(noise)) ** 2, s_ref[:, 0] ** 2, RTOL, "variance at sklearn's fitted kernel")


def test_svgp_exact_conversion_in_six_dimensions():
    """The multi-task SVGP exact-conversion model on a 6-D input (wide layout), fp64 against the CPU restatement and fp32
    within what the same algebra loses in numpy float32."""
    from gaussian_process_transportation_amd import _lib
    from oracle import gp_oracle as orc
    rng = np.random.default_rng(5)
    Z, T, D, M = 700, 3, 6, 2500
    Zp = rng.uniform(0, 1, (Z, D))
    A = rng.standard_normal((T, Z, Z))
    Sigma = A @ A.transpose(0, 2, 1) / Z * 0.05 + 1e-3 * np.eye(Z)
    y = rng.standard_normal((T, Z))
    osc = np.array([0.7, 1.3, 2.0])
    ls = np.array([0.5, 0.7, 0.6, 0.9, 0.55, 0.8])
    Xq = rng.uniform(-0.1, 1.1, (M, D))
    rm, rs, rJ, rJs = orc.svgp_exact_oracle_fast(Xq, Zp, Sigma, y, osc, ls)
    h = _lib.Handle(0)
    h.fit_svgp(Zp, y, Sigma, ls, osc, dtype=_lib.GPT_F64)
    out = h.predict_all(Xq, mean=True, var=True, J=True, Jvar=True)
    assert_parity(out["mean"], rm, 1e-7, "mean")
    assert_parity(out["var"], rs ** 2, 1e-7, "var")
    assert_parity(out["J"], rJ, 1e-7, "J")
    assert_parity(out["Jvar"], rJs ** 2, 1e-7, "Jvar")
    m32, s32, J32, Js32 = orc.svgp_exact_oracle_fast(Xq, Zp, Sigma, y, osc, ls, dtype=np.float32)
    h.fit_svgp(Zp, y, Sigma, ls, osc, dtype=_lib.GPT_F32)
    o32 = h.predict_all(Xq, mean=True, var=True, J=True, Jvar=True)
    vs, js = float(np.max(osc)), float(np.max(osc[:, None] / ls[None, :] ** 2))
    for name, got, want, ref32, scale in (("var", o32["var"], rs ** 2, s32.astype(float) ** 2, vs),
                                          ("Jvar", o32["Jvar"], rJs ** 2, Js32.astype(float) ** 2, js)):
        err = float(np.max(np.abs(got - want))) / scale
        lim = max(2e-4, 2 * float(np.max(np.abs(ref32 - want))) / scale)
        assert err <= lim, f"fp32 {name}: {err:.2e} > {lim:.2e}"
    h.close()


def test_empty_query_and_errors():
    from gaussian_process_transportation_amd import _lib, GaussianProcess
    h = _lib.Handle(0)
    with pytest.raises(_lib.GptError):
        h.predict_all(np.zeros((3, 3)), mean=True)                     # not fitted
    X = np.random.default_rng(0).uniform(0, 1, (40, 3))
    Y = np.sin(X)
    h.fit(X, Y, np.array([0.2]), 1.0, 1e-3, 1e-10)
    out = h.predict_all(np.zeros((0, 3)), mean=True, var=True, J=True)
    assert out["mean"].shape == (0, 3) and out["var"].shape == (0,) and out["J"].shape == (0, 3, 3)
    with pytest.raises(ValueError):
        h.predict_all(np.zeros((4, 2)), mean=True)                     # wrong feature count
    with pytest.raises(ValueError):
        h.fit(X, Y, np.array([0.2, 0.1]), 1.0, 1e-3, 1e-10)            # n_ls not in {1, D}
    with pytest.raises(ValueError):
        h.fit(X, Y, np.array([-0.2]), 1.0, 1e-3, 1e-10)
    # duplicated points with no noise and no jitter: not positive definite -> LinAlgError as sklearn
    Xd = np.vstack([X, X])
    with pytest.raises(np.linalg.LinAlgError):
        h.fit(Xd, np.vstack([Y, Y]), np.array([0.2]), 1.0, 0.0, 0.0)
    # a failed fit leaves the handle unfitted, a new fit recovers it
    with pytest.raises(_lib.GptError):
        h.predict_all(np.zeros((3, 3)), mean=True)
    h.fit(X, Y, np.array([0.2]), 1.0, 1e-3, 1e-10)
    assert np.all(np.isfinite(h.predict_all(X, mean=True)["mean"]))
    h.close()
    from sklearn.gaussian_process.kernels import RBF, Matern, WhiteKernel, ConstantKernel as C
    with pytest.raises(ValueError):
        GaussianProcess(kernel=C(1.0) * RBF(0.1) + WhiteKernel(1e-3), optimizer=None, n_targets=2, verbose=False).fit(X, Y)
    with pytest.raises(NotImplementedError):
        GaussianProcess(kernel=C(1.0) * Matern(0.1, nu=0.7) + WhiteKernel(1e-3), optimizer=None, verbose=False).fit(X, Y)
    mt = GaussianProcess(kernel=C(1.0) * Matern(0.3) + WhiteKernel(1e-3), optimizer=None, verbose=False).fit(X, Y)
    assert mt.predict(X[:5]).shape == (5, 3)
    with pytest.raises(NotImplementedError):
        mt.derivative(X[:5])                       # RBF-only formulas in the reference: refused for Matern
    with pytest.raises(NotImplementedError):
        mt.derivative_of_variance(X[:5])
    with pytest.raises(ValueError):
        GaussianProcess(kernel=RBF(0.1), optimizer=None, verbose=False).fit(X, Y)


@pytest.mark.parametrize("N", [2, 64, 65, 127, 129, 500, 513, 1100])
def test_cholesky_factor_across_panel_boundaries(N):
    """L_ against LAPACK for sizes on both sides of the 64-column step, the outer panel and the 512 padding
    (gpt_fit.hip: k_potrf_step / rank-OB update / k_potrf_finish), W = L^-1 and alpha with it."""
    import scipy.linalg
    from gaussian_process_transportation_amd import _lib
    from oracle import gp_oracle as orc
    rng = np.random.default_rng(N)
    X = rng.uniform(0, 1, (N, 3))
    Y = np.sin(3 * X[:, :2])
    ls, c, noise, jit = np.array([0.25, 0.3, 0.2]), 0.7, 1e-3, 1e-10
    h = _lib.Handle(0)
    h.fit(X, Y, ls, c, noise, jit)
    L, alpha = h.export()
    Kref = c * orc.rbf_gram(X / ls) + (noise + jit) * np.eye(N)
    Lref = np.linalg.cholesky(Kref)
    assert_parity(L, Lref, 1e-11, "L_")
    assert np.all(np.triu(L, 1) == 0.0)
    assert_parity(alpha, scipy.linalg.cho_solve((Lref, True), Y), 1e-7, "alpha_")
    W = h.export_inverse_factor()
    assert np.abs(W @ Lref - np.eye(N)).max() < 1e-9
    h.close()


def test_not_positive_definite_in_a_late_panel():
    """A matrix that stops being positive definite in a late panel: LinAlgError as with LAPACK's dpotrf (sklearn
    _gpr.py:348-358), the reported pivot inside the singular block."""
    import scipy.linalg.lapack
    from gaussian_process_transportation_amd import _lib
    from oracle import gp_oracle as orc
    rng = np.random.default_rng(5)
    X = rng.uniform(0, 1, (700, 3))
    X[650:] = X[100:150]                                  # exact duplicates: singular from row 651 on
    Y = np.sin(X)
    h = _lib.Handle(0)
    with pytest.raises(np.linalg.LinAlgError) as ei:
        h.fit(X, Y, np.array([0.3]), 1.0, 0.0, 0.0)
    K = orc.rbf_gram(X / 0.3)
    _, info = scipy.linalg.lapack.dpotrf(K, lower=1)
    assert info > 0
    got = int(str(ei.value).split("pivot")[1].split()[0])
    # rows 651.. are exact copies: in exact arithmetic every pivot from 651 on is 0, in floating point each comes out as
    # +-1e-16 and the first NEGATIVE one depends on the summation order (LAPACK's here: 651)
    assert info == 651 and 651 <= got <= 700, (got, info)
    # a clearly negative pivot (not a rounding matter) is reported exactly where LAPACK reports it
    Xr = rng.uniform(0, 1, (700, 3))
    Sigma = 1e-3 * np.eye(700)
    Sigma[300, 300] = -2.0
    with pytest.raises(np.linalg.LinAlgError) as ei:
        h.fit_noise_matrix(Xr, np.sin(Xr), np.array([0.3]), 1.0, Sigma)
    _, info = scipy.linalg.lapack.dpotrf(orc.rbf_gram(Xr / 0.3) + Sigma, lower=1)
    assert info == 301 and int(str(ei.value).split("pivot")[1].split()[0]) == 301
    h.close()


def test_single_target_shapes_follow_sklearn():
    from gaussian_process_transportation_amd import GaussianProcess
    from sklearn.gaussian_process.kernels import RBF, WhiteKernel, ConstantKernel as C
    rng = np.random.default_rng(3)
    X = rng.uniform(0, 1, (50, 2))
    y = np.sin(3 * X[:, 0])
    gp = GaussianProcess(kernel=C(1.0) * RBF([0.3, 0.3]) + WhiteKernel(1e-3), optimizer=None, verbose=False).fit(X, y)
    m, s = gp.predict(X[:7], return_std=True)
    assert m.shape == (7,) and s.shape == (7,)
    assert gp.derivative(X[:7]).shape == (7, 1, 2)


def test_full_size_properties():
    """Config 3 size (N=8192, M=500k) through size-independent identities:
    (a) K alpha = y  =>  mean(X_train) = y - noise_var * alpha;
    (b) the Jacobian is the central finite difference of the mean;
    (c) d var/dx is the central finite difference of the variance;
    (d) the 1-column and 4-column variance kernels agree; variance within [0, c + noise]."""
    from gaussian_process_transportation_amd import _lib
    N, M, D = 8192, 500_000, 3
    rng = np.random.default_rng(0)
    X = rng.uniform(0, 1, (N, D))
    Y = 0.05 * np.sin(4 * X) + 0.01 * rng.standard_normal((N, D))
    Xq = np.random.default_rng(1).uniform(-0.1, 1.1, (M, D))
    c, ls, noise, jit = 0.1, np.array([0.1] * 3), 1e-4, 1e-10
    h = _lib.Handle(0)
    h.fit(X, Y, ls, c, noise, jit)
    _, alpha = h.export(want_L=False)
    tr = h.predict_all(X, mean=True)["mean"]
    assert_parity(tr, Y - (noise + jit) * alpha, 1e-7, "mean at the training points")
    out = h.predict_all(Xq, mean=True, var=True, J=True)
    assert np.all(np.isfinite(out["mean"])) and np.all(np.isfinite(out["J"]))
    assert out["var"].min() >= 0.0 and out["var"].max() <= c + noise + 1e-12
    sub = slice(0, 4096)
    full = h.predict_all(Xq[sub], var=True, Jvar=True, dvar=True)
    assert_parity(full["var"], out["var"][sub], 1e-10, "var: 4-column kernel vs 1-column kernel")
    eps = 1e-5
    for d in range(D):
        e = np.zeros(D); e[d] = eps
        p = h.predict_all(Xq[sub] + e, mean=True, var=True)
        m = h.predict_all(Xq[sub] - e, mean=True, var=True)
        assert_parity(out["J"][sub][:, :, d], (p["mean"] - m["mean"]) / (2 * eps), 1e-5, f"J[..., {d}] vs finite difference")
        assert_parity(full["dvar"][d], (p["var"] - m["var"]) / (2 * eps), 1e-4, f"dvar[{d}] vs finite difference")
    assert np.all(full["Jvar"] <= c / ls[0] ** 2 + 1e-9) and np.all(full["Jvar"] > -1e-6)
    h.close()


@pytest.mark.parametrize("name", ["synthetic_3d_N64", "synthetic_3d_N64_iso", "synthetic_3d_N256", "synthetic_3d_N1024",
                                  "synthetic_5d_N200", "synthetic_8d_N128", "synthetic_12d_N160", "synthetic_15d_N96"])
def test_lml_value_and_gradient_vs_sklearn(name):
    """gpt_lml_gradient against sklearn's log_marginal_likelihood(theta, eval_gradient=True) at three thetas."""
    from gaussian_process_transportation_amd import _lib
    g = load_golden(name)
    n_ls = g["length_scale"].size
    h = _lib.Handle(0)
    for th, v, gr in zip(g["lml_theta"], g["lml_value"], g["lml_grad"]):
        c, ls, noise = np.exp(th[0]), np.exp(th[1:1 + n_ls]), np.exp(th[1 + n_ls])
        h.fit(g["X"], g["Y"], ls, c, noise, float(g["alpha"]))
        lml, grad = h.lml_gradient(n_ls)
        assert lml == pytest.approx(float(v), rel=1e-9)
        assert_parity(grad, gr, 1e-6, "d lml / d theta")
        with pytest.raises(_lib.GptError):
            h.export()                      # the factor was consumed by the gradient; a new fit restores it
        h.fit(g["X"], g["Y"], ls, c, noise, float(g["alpha"]))
        assert h.export()[0].shape == (len(g["X"]), len(g["X"]))
    h.close()


def test_letterS_with_optimizer_matches_reference_fit():
    """Config 1 end to end with the reference's default optimizer: same L-BFGS-B driver and RNG protocol as
    sklearn, objective on the GPU.  What is asserted: (1) the optimum is as good as sklearn's (LML not lower beyond
    1e-8 relative), (2) theta agrees to 1e-6 — the same driver fed objective values that agree to rounding takes the same
    steps (measured 1e-10) — and (3) the outputs differ from the reference's by no more than what
    that difference in theta explains: the same GPU path refitted at the REFERENCE's theta reproduces the golden
    outputs to 1e-5 (test_letterS_transport_fixed_theta), so |out(theta_gpu) - golden| must be within
    |out(theta_gpu) - out(theta_ref)| + 1e-5 of the array scale."""
    from gaussian_process_transportation_amd import GaussianProcessTransportation
    g = load_golden("letterS_2d")
    np.random.seed(0)
    tr = GaussianProcessTransportation(kernel_transport=sk_kernel(10.0, 4 * np.ones(2), 0.01), verbose=False)
    tr.source_distribution = g["source"]; tr.target_distribution = g["target"]
    tr.training_traj = g["demo"]; tr.training_delta = g["delta"]
    tr.fit_transportation(do_scale=False, do_rotation=True)
    tr.apply_transportation()
    gp = tr.method.delta_map
    lml_ref = float(g["lml_fit"])
    assert gp.gp.log_marginal_likelihood_value_ >= lml_ref - 1e-8 * abs(lml_ref)
    assert gp.gp.log_marginal_likelihood_value_ == pytest.approx(lml_ref, rel=1e-6)
    assert_parity(np.asarray(gp.kernel.theta), g["theta_fit"], 1e-6, "fitted theta")      # measured: ~1e-10
    at_ref = _transport(g)                                   # same path, hyper-parameters fixed at the reference's optimum
    for name, got, fixed, ref in (("traj", tr.training_traj, at_ref.training_traj, g["traj"]), ("std", tr.std, at_ref.std, g["std"]),
                                  ("vel", tr.training_delta, at_ref.training_delta, g["vel"]),
                                  ("var_vel", tr.var_vel_transported, at_ref.var_vel_transported, g["var_vel"])):
        scale = np.max(np.abs(ref))
        explained = np.max(np.abs(got - fixed)) / scale
        err = np.max(np.abs(got - ref)) / scale
        print(f"letter-S optimizer {name}: vs golden {err:.2e}, explained by theta {explained:.2e}")
        assert err <= explained + 1e-5, (name, err, explained)
        assert err <= 1e-6, (name, err)                     # measured 4e-12 .. 2e-11 (round 1 allowed 1e-3 here)


def test_surface3d_with_optimizer_reaches_reference_optimum():
    """The reference's 3-D demo with its default kernel and optimizer (279 s on 8 CPU cores): the GPU search
    must reach an optimum at least as good as sklearn's and the same hyper-parameters."""
    from gaussian_process_transportation_amd import GaussianProcess
    from gaussian_process_transportation_amd.affine_transform import AffineTransform
    g = load_golden("surface_3d")
    aff = AffineTransform(verbose=False).fit(g["source"], g["target"])
    src = aff.predict(g["source"])
    np.random.seed(0)
    gp = GaussianProcess(kernel=sk_kernel(0.1, [0.1], 1e-4), verbose=False)
    gp.fit(src, g["target"] - src)
    lml_ref = float(g["lml_fit"])
    print(f"surface-3D optimizer: LML {gp.gp.log_marginal_likelihood_value_!r} vs sklearn {lml_ref!r}")
    assert gp.gp.log_marginal_likelihood_value_ >= lml_ref - 1e-8 * abs(lml_ref)
    from tests.conftest import relmax
    print(f"surface-3D optimizer: theta vs sklearn {relmax(np.asarray(gp.kernel.theta), g['theta_fit']):.2e}")
    assert_parity(np.asarray(gp.kernel.theta), g["theta_fit"], THETA_TOL["surface3d"], "fitted theta")
    # and the predictions at the GPU's theta against the golden ones (the reference's, at ITS theta): within what the theta
    # difference explains (the same GPU path at the reference's theta) + 1e-5, as for letter-S
    m, s = gp.predict(aff.predict(g["demo"]), return_std=True)
    fixed = GaussianProcess(kernel=sk_kernel(g["constant_value"], g["length_scale"], g["noise_level"]), optimizer=None, verbose=False)
    fixed.fit(src, g["target"] - src)
    m0, s0 = fixed.predict(aff.predict(g["demo"]), return_std=True)
    for name, got, at_ref, ref in (("traj", aff.predict(g["demo"]) + m, aff.predict(g["demo"]) + m0, g["traj"]), ("std", s, s0, g["std"])):
        scale = np.max(np.abs(ref))
        explained = np.max(np.abs(got - at_ref)) / scale
        err = np.max(np.abs(got - ref)) / scale
        print(f"surface-3D optimizer {name}: vs golden {err:.2e}, explained by theta {explained:.2e}")
        assert err <= explained + 1e-5, (name, err, explained)
        assert err <= OUT_TOL["surface3d"], (name, err)


@pytest.mark.parametrize("N,M", [(1024, 150_000), (2500, 70_000)])
def test_many_column_blocks_per_workgroup(N, M):
    """Large M at moderate N: every persistent workgroup of the variance kernel walks many column blocks
    (scratch image and LDS reuse across pieces, ranges cut inside a block).  A strided subset is checked
    against the oracle, the rest through 1-column vs 4-column agreement."""
    from gaussian_process_transportation_amd import _lib
    from oracle import gp_oracle as orc
    X, Y, Xq = orc.synthetic_problem(N, M)
    c, ls, noise, jit = 0.1, np.array([0.1, 0.12, 0.09]), 1e-4, 1e-10
    h = _lib.Handle(0)
    h.fit(X, Y, ls, c, noise, jit)
    out1 = h.predict_all(Xq, mean=True, var=True, J=True)
    out4 = h.predict_all(Xq, var=True, Jvar=True, dvar=True)
    assert_parity(out4["var"], out1["var"], 1e-11, "var: 4-column vs 1-column kernel")
    idx = np.unique(np.r_[np.arange(0, M, 97), np.arange(M - 300, M), np.arange(0, 300)])
    L, a = orc.gpr_fit(X, Y, c, ls, noise, jit)
    mean, var, J, Jvar = orc.posterior_all_fast(Xq[idx], X, L, a, c, ls, noise, want_jvar=True)
    assert_parity(out1["mean"][idx], mean, RTOL, "mean")
    assert_parity(out1["var"][idx], var, RTOL, "var")
    assert_parity(out1["J"][idx], J, RTOL, "J")
    assert_parity(out4["Jvar"][idx], Jvar, RTOL, "Jvar")
    h.close()


def test_device_pointer_api_and_model_handoff():
    """gpt_predict_all_dev on torch tensors + the blob hand-off used for the multi-GPU broadcast
    (alloc on a second handle, copy the bytes, commit) reproduce the host-pointer results."""
    import torch
    from gaussian_process_transportation_amd import _lib
    from gaussian_process_transportation_amd.distributed import wrap_device_bytes
    rng = np.random.default_rng(5)
    N, M = 600, 3000
    X = rng.uniform(0, 1, (N, 3)); Y = np.cos(3 * X); Xq = rng.uniform(0, 1, (M, 3))
    h = _lib.Handle(0)
    h.fit(X, Y, np.array([0.2, 0.3, 0.25]), 0.5, 1e-3, 1e-10)
    ref = h.predict_all(Xq, mean=True, var=True, J=True, Jvar=True)
    dev = torch.device("cuda", 0)
    xq = torch.from_numpy(Xq).to(dev)
    mean = torch.empty((M, 3), dtype=torch.float64, device=dev); var = torch.empty(M, dtype=torch.float64, device=dev)
    J = torch.empty((M, 3, 3), dtype=torch.float64, device=dev); Jv = torch.empty((M, 3), dtype=torch.float64, device=dev)
    h.set_stream(torch.cuda.current_stream().cuda_stream)
    h.set_profiling(True)
    h.predict_all_dev(xq.data_ptr(), M, mean.data_ptr(), var.data_ptr(), J.data_ptr(), Jv.data_ptr())
    t = h.predict_timings()
    assert t["var_ms"] > 0 and t["mean_jac_ms"] > 0
    torch.cuda.synchronize()
    for k, v in (("mean", mean), ("var", var), ("J", J), ("Jvar", Jv)):
        assert np.array_equal(v.cpu().numpy(), ref[k]), k
    # hand-off: a second handle receives the model bytes
    src_ptr, nbytes = h.factor_blob()
    h2 = _lib.Handle(0)
    dst_ptr, nbytes2 = h2.factor_alloc(N, 3, 3)
    assert nbytes == nbytes2
    with pytest.raises(_lib.GptError):
        h2.factor_commit()                           # nothing broadcast yet: header check fails
    wrap_device_bytes(dst_ptr, nbytes, dev).copy_(wrap_device_bytes(src_ptr, nbytes, dev))
    torch.cuda.synchronize()
    h2.factor_commit()
    out2 = h2.predict_all(Xq, mean=True, var=True, J=True, Jvar=True)
    for k in ("mean", "var", "J", "Jvar"):
        assert np.array_equal(out2[k], ref[k]), k
    with pytest.raises(_lib.GptError):
        h2.export()                                   # L lives only on the fitting handle
    h.close(); h2.close()


def _two_rank_worker(rank, world, port, q, D=3):
    import os
    import torch
    import torch.distributed as dist
    from gaussian_process_transportation_amd import _lib
    from gaussian_process_transportation_amd.distributed import broadcast_model, shard_range
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)      # RCCL refuses two ranks on one GPU; gloo moves CUDA tensors too
    try:
        rng = np.random.default_rng(11)
        N, M = 700, 5000
        X = rng.uniform(0, 1, (N, D)); Y = np.sin(3 * X[:, :3]); Xq = rng.uniform(0, 1, (M, D))
        h = _lib.Handle(0)
        if rank == 0:
            h.fit(X, Y, np.linspace(0.2, 0.3, D) * (1.0 if D <= 3 else 2.5), 0.7, 1e-3, 1e-10)
        nbytes = broadcast_model(h, fitted=(rank == 0), src=0, device=torch.device("cuda", 0))
        a, b = shard_range(M, rank, world)
        out = h.predict_all(Xq[a:b], mean=True, var=True, J=True)
        ref = None
        if rank == 0:
            full = h.predict_all(Xq, mean=True, var=True, J=True)
            ref = {k: full[k] for k in ("mean", "var", "J")}
        box = [ref]
        dist.broadcast_object_list(box, src=0)
        # mean / J are bitwise independent of the batch; the variance kernel's work split (where sweeps are cut into partial
        # products, hence the order of those sums) depends on M, so shards agree to rounding, not to the bit.  Rounding here is
        # eps x the partial sums of W k*, which cancel heavily (|W| ~ noise^-1/2): 1e-11 of the prior variance, not 1e-16
        verr = float(np.max(np.abs(out["var"] - box[0]["var"][a:b])) / np.max(box[0]["var"]))
        ok = (np.array_equal(out["mean"], box[0]["mean"][a:b]) and np.array_equal(out["J"], box[0]["J"][a:b]) and verr <= 1e-11)
        q.put((rank, int(nbytes), bool(ok), verr))
        h.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("D", [3, 6])
def test_two_process_model_broadcast_and_sharded_predict(D):
    """Two ranks sharing the one GPU of the test box (gloo, since RCCL needs one GPU per rank): rank 0 fits,
    broadcast_model ships the blob, both predict their shard; shards must reproduce rank 0's full prediction.
    D = 6: the wide source layout travels in the same blob."""
    import os
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_two_rank_worker, args=(r, 2, port + 7 * D, q, D)) for r in range(2)]
    for p_ in procs:
        p_.start()
    res = sorted(q.get(timeout=300) for _ in range(2))
    for p_ in procs:
        p_.join(timeout=120)
        assert p_.exitcode == 0
    assert res[0][2] and res[1][2], res
    assert res[0][1] == res[1][1] > 0


def test_transport_orientation_runs_and_is_consistent():
    """transport_orientation (host quaternion algebra around the GPU Jacobian).  Parity unpinned: the
    reference's `Quaternion` module is absent; checked for unit norm and against rotating by the polar
    factor of J_Phi."""
    from gaussian_process_transportation_amd import GaussianProcessTransportation
    from gaussian_process_transportation_amd.quaternion import rotation_matrix_from_quaternion
    g = load_golden("surface_3d")
    rng = np.random.default_rng(2)
    ori = rng.standard_normal((len(g["demo"]), 4)); ori /= np.linalg.norm(ori, axis=1, keepdims=True)
    tr = GaussianProcessTransportation(kernel_transport=sk_kernel(g["constant_value"], g["length_scale"], g["noise_level"]),
                                       optimizer=None, verbose=False)
    tr.source_distribution = g["source"]; tr.target_distribution = g["target"]
    tr.training_traj = g["demo"]; tr.training_ori = ori
    tr.fit_transportation()
    tr.apply_transportation()
    out = tr.training_ori
    assert out.shape == ori.shape
    assert_parity(np.linalg.norm(out, axis=1), np.ones(len(ori)), 1e-9, "unit quaternions")
    # R(out) = polar(J_Phi) R(ori), J_Phi at the un-rotated positions (reference quirk)
    J = tr.method.delta_map.derivative(g["demo"])
    Jg = tr.method.affine_transform.derivative(g["demo"])
    Jphi = Jg + J @ Jg
    U, _, Vt = np.linalg.svd(Jphi)
    polar = U @ Vt
    assert np.max(np.abs(rotation_matrix_from_quaternion(out) - polar @ rotation_matrix_from_quaternion(ori))) < 5e-3


@pytest.mark.parametrize("tag,nu,code", [("12", 0.5, 1), ("32", 1.5, 2), ("52", 2.5, 3)])
def test_matern_kernels_vs_reference(tag, nu, code):
    """C * Matern(nu) + White (the examples' dynamics GP): fit, mean, std, covariance and the LML gradient."""
    from sklearn.gaussian_process.kernels import Matern, WhiteKernel, ConstantKernel as C
    from gaussian_process_transportation_amd import GaussianProcess, _lib
    g = load_golden("matern_2d")
    gp = GaussianProcess(kernel=C(0.3) * Matern([1.5, 2.5], nu=nu) + WhiteKernel(0.01), optimizer=None, verbose=False)
    gp.fit(g["X"], g["Y"])
    assert_parity(gp.gp.alpha_, g[f"m{tag}_alpha_"], RTOL, "alpha_")
    assert_parity(np.diag(gp.gp.L_), g[f"m{tag}_Ldiag"], RTOL, "diag L")
    m, s = gp.predict(g["grid"], return_std=True)
    assert_parity(m, g[f"m{tag}_mean"], RTOL, "mean")
    assert_parity(s, g[f"m{tag}_std"], RTOL, "std")
    _, cov = gp.predict(g["grid"][:12], return_cov=True)
    assert_parity(cov, g[f"m{tag}_cov"], RTOL, "cov")
    h = _lib.Handle(0)
    for th, v, gr in zip(g[f"m{tag}_lml_theta"], g[f"m{tag}_lml_value"], g[f"m{tag}_lml_grad"]):
        h.fit(g["X"], g["Y"], np.exp(th[1:3]), np.exp(th[0]), np.exp(th[3]), 1e-10, code)
        lml, grad = h.lml_gradient(2)
        assert lml == pytest.approx(float(v), rel=1e-9)
        assert_parity(grad, gr, 1e-6, "d lml / d theta (ARD)")
    th = g[f"m{tag}_iso_theta"]
    h.fit(g["X"][::4], g["Y"][::4], np.exp(th[1:2]), np.exp(th[0]), np.exp(th[2]), 1e-10, code)
    lml, grad = h.lml_gradient(1)
    assert lml == pytest.approx(float(g[f"m{tag}_iso_value"]), rel=1e-9)
    assert_parity(grad, g[f"m{tag}_iso_grad"], 1e-6, "d lml / d theta (isotropic)")
    h.close()


def test_matern_dynamics_gp_with_optimizer():
    """example/2D/surface_generalization.py:49-51: the dynamics GP with its default optimizer."""
    from sklearn.gaussian_process.kernels import Matern, WhiteKernel, ConstantKernel as C
    from gaussian_process_transportation_amd import GaussianProcess
    g = load_golden("matern_2d")
    np.random.seed(0)
    gp = GaussianProcess(kernel=C(constant_value=np.sqrt(0.1)) * Matern(1 * np.ones(2), nu=2.5) + WhiteKernel(0.01), verbose=False)
    gp.fit(g["X"], g["Y"])
    assert gp.gp.log_marginal_likelihood_value_ == pytest.approx(float(g["opt_lml"]), rel=1e-6)
    from tests.conftest import relmax
    print(f"Matern dynamics optimizer: theta vs sklearn {relmax(np.asarray(gp.kernel.theta), g['opt_theta']):.2e}")
    assert_parity(np.asarray(gp.kernel.theta), g["opt_theta"], THETA_TOL["matern"], "fitted theta")
    m, s = gp.predict(g["grid"], return_std=True)
    th = g["opt_theta"]
    fixed = GaussianProcess(kernel=C(np.exp(th[0])) * Matern(np.exp(th[1:3]), nu=2.5) + WhiteKernel(np.exp(th[3])), optimizer=None, verbose=False)
    fixed.fit(g["X"], g["Y"])
    m0, s0 = fixed.predict(g["grid"], return_std=True)
    for name, got, at_ref, ref in (("mean", m, m0, g["opt_mean"]), ("std", s, s0, g["opt_std"])):
        scale = np.max(np.abs(ref))
        explained = np.max(np.abs(got - at_ref)) / scale
        err = np.max(np.abs(got - ref)) / scale
        print(f"Matern dynamics optimizer {name}: vs golden {err:.2e}, explained by theta {explained:.2e}")
        assert err <= explained + 1e-5, (name, err, explained)
        assert err <= OUT_TOL["matern"], (name, err)


def test_letter_s_example_end_to_end():
    """Config 1: the reference's 2-D demo flow, headless (examples/letter_s_2d.py), against the golden
    transported trajectory / velocities."""
    import importlib.util
    import os
    from tests.conftest import ROOT
    spec = importlib.util.spec_from_file_location("letter_s_2d", os.path.join(ROOT, "examples", "letter_s_2d.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = mod.main(verbose=False)
    g = load_golden("letterS_2d")
    from tests.conftest import relmax
    print(f"letter-S example: traj {relmax(out['X1'], g['traj']):.2e}, vel {relmax(out['deltaX1'], g['vel']):.2e}")
    assert_parity(out["X1"], g["traj"], OUT_TOL["letterS_example"], "transported demo")
    assert_parity(out["deltaX1"], g["vel"], OUT_TOL["letterS_example"], "transported velocities")
    assert out["field"].shape == (10000, 2) and np.all(np.isfinite(out["field1"]))


@pytest.mark.parametrize("Z,M", [(200, 700), (1024, 3000)])
def test_svgp_exact_conversion_predictor(Z, M):
    """Config 5 algebra (synthetic SPD Sigma_pseudo = A A^T / Z + 1e-3 I, y ~ N(0,1), outputscale 1,
    l = 0.2, T = D = 3) against the CPU restatement.  Parity unpinned (no gpytorch, no reference fixture)."""
    from gaussian_process_transportation_amd.svgp_exact import SVGPExactPredictor
    from oracle import gp_oracle as orc
    rng = np.random.default_rng(0)
    T = D = 3
    Zp = rng.uniform(0, 1, (Z, D))
    A = rng.standard_normal((T, Z, Z))
    Sigma = A @ np.transpose(A, (0, 2, 1)) / Z + 1e-3 * np.eye(Z)
    y = rng.standard_normal((T, Z, 1))
    osc = np.array([1.0, 0.7, 1.3])
    ls = np.array([0.2, 0.25, 0.15])
    x = rng.uniform(0, 1, (M, D))
    sv = SVGPExactPredictor(Zp, Sigma, y, osc, ls, dtype="float64")
    mean, std = sv.posterior_f(x, return_std=True)
    J, Jstd = sv.posterior_f_prime(x, return_std=True)
    rm, rs, rJ, rJs = orc.svgp_exact_oracle(x, Zp, Sigma, y, osc, ls)
    assert mean.shape == (M, T) and J.shape == (M, T, D)
    assert_parity(mean, rm, RTOL, "mean")
    assert_parity(std, rs, RTOL, "std")
    assert_parity(J, rJ, RTOL, "J")
    assert_parity(Jstd, rJs, RTOL, "J std")
    assert_parity(sv.predict(x), rm, RTOL, "predict alias")
    m1, s1, J1, Js1 = sv.posterior(x)                      # all four in one pass over the stacked factors
    assert np.array_equal(m1, mean) and np.array_equal(J1, J)
    assert_parity(s1, rs, RTOL, "std (one pass)")
    assert_parity(Js1, rJs, RTOL, "J std (one pass)")
    sv.close()


def test_surface_3d_example_end_to_end():
    """The reference's 3-D demo flow, headless (examples/surface_3d.py), optimizer on, against the golden outputs."""
    import importlib.util
    import os
    from tests.conftest import ROOT
    spec = importlib.util.spec_from_file_location("surface_3d", os.path.join(ROOT, "examples", "surface_3d.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = mod.main(verbose=False)
    g = load_golden("surface_3d")
    from tests.conftest import relmax
    print(f"surface-3D example: theta {relmax(out['theta'], g['theta_fit']):.2e}, traj {relmax(out['X1'], g['traj']):.2e}, "
          f"vel {relmax(out['deltaX1'], g['vel']):.2e}")
    assert_parity(out["theta"], g["theta_fit"], THETA_TOL["surface3d_example"], "fitted theta")
    assert_parity(out["X1"], g["traj"], 1e-6, "transported demo")
    assert_parity(out["deltaX1"], g["vel"], OUT_TOL["surface3d_example"], "transported velocities")


def test_prefetch_changes_nothing_but_the_number_of_passes():
    """apply_transportation() asks for std and Jacobian variance at the same positions; the one-pass prefetch must give
    what the two separate calls give (variance from the 4-column instead of the 1-column kernel: last-bit level)."""
    from sklearn.gaussian_process.kernels import RBF, WhiteKernel, ConstantKernel as C
    from gaussian_process_transportation_amd import GaussianProcess
    rng = np.random.default_rng(4)
    X = rng.uniform(0, 1, (300, 3)); Y = 0.1 * np.sin(3 * X)
    Xq = rng.uniform(0, 1, (1000, 3))
    gp = GaussianProcess(kernel=C(0.5) * RBF([0.3, 0.2, 0.25]) + WhiteKernel(1e-3), optimizer=None, verbose=False).fit(X, Y)
    m0, s0 = gp.predict(Xq, return_std=True)
    J0, V0 = gp.derivative(Xq, return_var=True)
    gp.prefetch_posterior(Xq)
    m1, s1 = gp.predict(Xq, return_std=True)
    J1, V1 = gp.derivative(Xq, return_var=True)
    assert np.array_equal(m0, m1) and np.array_equal(J0, J1)
    assert_parity(s1, s0, 1e-10, "std from the 4-column pass")              # 1-column kernel vs 4-column kernel
    assert_parity(V1, V0, 1e-10, "Jacobian variance from the 4-column pass")  # 3-column kernel vs 4-column kernel
    other = gp.predict(Xq[:10], return_std=True)[0]            # different x: not served from the memo
    assert other.shape == (10, 3) and np.array_equal(other, m0[:10])
    gp.fit(X, 2 * Y)                                           # a new fit drops the memo
    assert not np.array_equal(gp.predict(Xq), m0)


@pytest.mark.parametrize("D", [1, 2, 3])
def test_jacobian_variance_alone_matches_the_four_column_path(D):
    """gpt_predict_all with Jvar but without var / dvar takes the 3-columns-per-query kernel (k_var<3>); with var it takes
    the 4-column one.  Same numbers to the last bits, for every input dimension (unused components are zero columns),
    and for a query count that leaves the last column block ragged."""
    from gaussian_process_transportation_amd import _lib
    from oracle import gp_oracle as orc
    rng = np.random.default_rng(10 + D)
    X = rng.uniform(0, 1, (700, D)); Y = np.sin(3 * X[:, :1])
    Xq = rng.uniform(-0.1, 1.1, (1237, D))
    ls = np.array([0.3, 0.2, 0.25][:D])
    h = _lib.Handle(0)
    h.fit(X, Y, ls, 0.8, 1e-3, 1e-10)
    alone = h.predict_all(Xq, J=True, Jvar=True)
    full = h.predict_all(Xq, var=True, J=True, Jvar=True)
    assert alone["Jvar"].shape == (1237, D) and np.array_equal(alone["J"], full["J"])
    assert_parity(alone["Jvar"], full["Jvar"], 1e-10, "Jvar 3-column vs 4-column")
    o = orc.GaussianProcessOracle(0.8, ls, 1e-3, 1e-10).fit(X, Y)
    _, Jv = o.derivative(Xq, return_var=True)
    assert_parity(alone["Jvar"], Jv[:, 0, :], RTOL, "Jvar vs oracle")
    h.close()


def test_svgp_multitask_handle_matches_per_task_handles():
    """gpt_fit_svgp stacks the tasks' inverse factors behind ONE generated kernel operand; per task it must give what a
    separate handle fitted by gpt_fit_noise_matrix gives (same factorisation, same kernels, other work split)."""
    from gaussian_process_transportation_amd import _lib
    from oracle import gp_oracle as orc
    Z, Sigma, y, osc, ls, Xq = orc.svgp_synthetic_problem(700, 2500)
    osc = np.array([1.0, 0.6, 1.7])
    hs = _lib.Handle(0)
    hs.fit_svgp(Z, y, Sigma, ls, osc, dtype=_lib.GPT_F64)
    assert hs.model_info() == (3, _lib.GPT_F64) and hs.info()[2] == 3
    out = hs.predict_all(Xq, mean=True, var=True, J=True, Jvar=True)
    assert out["var"].shape == (2500, 3) and out["Jvar"].shape == (2500, 3, 3)
    only_jvar = hs.predict_all(Xq, Jvar=True)["Jvar"]          # D columns per query, no k* column
    assert_parity(only_jvar, out["Jvar"], 1e-10, "Jvar alone vs with var")
    with pytest.raises(ValueError):
        hs.predict_all(Xq, dvar=True)                          # not defined for the multi-task model
    with pytest.raises(_lib.GptError):
        hs.predict_cov(Xq[:10])                                # needs a single-task fp64 fit on the handle
    for t in range(3):
        h1 = _lib.Handle(0)
        h1.fit_noise_matrix(Z, y[t][:, None], ls, osc[t], Sigma[t], alpha=0.0)
        o1 = h1.predict_all(Xq, mean=True, var=True, J=True, Jvar=True)
        assert_parity(out["mean"][:, t], o1["mean"][:, 0], 1e-10, f"mean task {t}")
        assert_parity(out["var"][:, t], o1["var"], 1e-10, f"var task {t}")
        assert_parity(out["J"][:, t], o1["J"][:, 0], 1e-10, f"J task {t}")
        assert_parity(out["Jvar"][:, t], o1["Jvar"], 1e-10, f"Jvar task {t}")
        h1.close()
    hs.close()


def test_svgp_fp32_config5_shape():
    """BASELINE configs[4] as written, at a query count the CPU oracle finishes in seconds: 2048 inducing points, T = D = 3,
    fp32 prediction (factorisation in fp64, results rounded once when the model is packed).
    PARITY UNPINNED (gpytorch absent, the reference holds no fixture): the yardstick is the fp64 CPU restatement.
    Tolerance: fp32 sums of 2048 products that cancel (alpha = K^-1 y has entries ~1/lambda_min) carry an error set by
    sum|r alpha|, not by the result.  Two bounds: (1) 1e-4 of the array scale (measured 3e-6 .. 2e-5), (2) no worse than
    the SAME algebra restated in the reference's arithmetic (numpy float32, explicit fp32 inverse, :72-78), which loses
    1.4e-4 .. 7.4e-4 against fp64 on these inputs — the GPU path factorises in fp64 and is 15-90x closer."""
    from gaussian_process_transportation_amd import SVGPExactPredictor
    from oracle import gp_oracle as orc
    Z, Sigma, y, osc, ls, Xq = orc.svgp_synthetic_problem(2048, 3000)
    ref64 = orc.svgp_exact_oracle_fast(Xq, Z, Sigma, y, osc, ls)
    ref32 = orc.svgp_exact_oracle_fast(Xq, Z, Sigma, y, osc, ls, dtype=np.float32)
    sv = SVGPExactPredictor(Z, Sigma, y, osc, ls, dtype="float32")
    got = sv.posterior(Xq)
    sv64 = SVGPExactPredictor(Z, Sigma, y, osc, ls, dtype="float64")
    got64 = sv64.posterior(Xq)
    for name, g32, g64, r64, r32 in zip(("mean", "std", "J", "J std"), got, got64, ref64, ref32):
        assert g32.dtype == np.float32 and g32.shape == r64.shape
        scale = np.max(np.abs(r64))
        err_gpu = np.max(np.abs(g32.astype(np.float64) - r64)) / scale
        err_ref_arith = np.max(np.abs(r32.astype(np.float64) - r64)) / scale
        print(f"svgp fp32 {name}: GPU {err_gpu:.2e}, reference arithmetic {err_ref_arith:.2e} (relative to {scale:.3g})")
        assert err_gpu <= err_ref_arith and err_gpu <= 1e-4, (name, err_gpu, err_ref_arith)
        assert_parity(g64, r64, 1e-6, name + " (fp64 model)")     # the algebra itself, cond(K) ~ 1e6
    sv.close(); sv64.close()


@pytest.mark.parametrize("N,order", [(1100, "0"), (1100, "1"), (2500, None)])
def test_small_batches_use_cut_sweeps_and_agree_with_large_batches(N, order, monkeypatch):
    """M = 1 .. 10^4 (the reference's own batch sizes) leaves fewer column blocks than workgroups: the plan then cuts
    sweeps at tile granularity and k_var_combine adds the partial products.  Results must agree with the same queries
    predicted inside a large batch (whole-block rounds) to rounding, for both tail orders, and with the oracle."""
    from gaussian_process_transportation_amd import _lib
    from oracle import gp_oracle as orc
    if order is not None:
        monkeypatch.setenv("GPT_VAR_TAIL_ORDER", order)
    X, Y, Xq = orc.synthetic_problem(N, 70_000)
    c, ls, noise, jit = 0.1, np.array([0.1, 0.12, 0.09]), 1e-4, 1e-10
    h = _lib.Handle(0)
    h.fit(X, Y, ls, c, noise, jit)
    big = h.predict_all(Xq, mean=True, var=True, J=True, Jvar=True, dvar=True)
    for m in (1, 63, 65, 460, 1000, 4096, 10_000):
        small = h.predict_all(Xq[:m], mean=True, var=True, J=True, Jvar=True, dvar=True)
        small1 = h.predict_all(Xq[:m], var=True)
        small3 = h.predict_all(Xq[:m], Jvar=True)
        assert np.array_equal(small["mean"], big["mean"][:m]) and np.array_equal(small["J"], big["J"][:m])
        assert_parity(small["var"], big["var"][:m], 1e-10, f"var M={m}")
        assert_parity(small1["var"], big["var"][:m], 1e-10, f"var (1 column) M={m}")
        assert_parity(small["Jvar"], big["Jvar"][:m], 1e-10, f"Jvar M={m}")
        assert_parity(small3["Jvar"], big["Jvar"][:m], 1e-10, f"Jvar (alone) M={m}")
        assert_parity(small["dvar"], big["dvar"][:, :m], 1e-9, f"dvar M={m}")
    L, a = orc.gpr_fit(X, Y, c, ls, noise, jit)
    mean, var, J, Jvar = orc.posterior_all_fast(Xq[:460], X, L, a, c, ls, noise, want_jvar=True)
    small = h.predict_all(Xq[:460], mean=True, var=True, J=True, Jvar=True)
    assert_parity(small["var"], var, RTOL, "var vs oracle")
    assert_parity(small["Jvar"], Jvar, RTOL, "Jvar vs oracle")
    h.close()


def test_rounds_dealt_to_several_launches_change_nothing(monkeypatch):
    """k_var's rounds of whole column blocks go out 16 per launch (the workgroups re-align at every kernel boundary,
    gpt_predict.hip launch_var_t).  A launch boundary must not change a bit: one launch for everything, one launch per
    round and an uneven split (3 + 1 rounds, the item list with the last) against the default, for the 1-column, the
    3-column and the fused 4-column kernels."""
    from gaussian_process_transportation_amd import _lib
    from oracle import gp_oracle as orc
    X, Y, Xq = orc.synthetic_problem(1100, 70_000)           # 1094 column blocks: 4 rounds of 256 + a tail of 70
    h = _lib.Handle(0)
    h.fit(X, Y, np.array([0.1, 0.12, 0.09]), 0.1, 1e-4, 1e-10)
    monkeypatch.delenv("GPT_VAR_ROUNDS_PER_LAUNCH", raising=False)
    monkeypatch.delenv("GPT_VAR_ROUNDS_EXACT", raising=False)
    ref = h.predict_all(Xq, var=True, Jvar=True, dvar=True)          # default: 16 rounds of the N = 8192 shape = all 4 here
    ref1 = h.predict_all(Xq, var=True)
    ref3 = h.predict_all(Xq, Jvar=True)
    for setting in ("1", "3", "2"):
        monkeypatch.setenv("GPT_VAR_ROUNDS_EXACT", setting)
        out = h.predict_all(Xq, var=True, Jvar=True, dvar=True)
        for k in ("var", "Jvar", "dvar"):
            assert np.array_equal(out[k], ref[k]), f"{k}, exactly {setting} rounds per launch"
        assert np.array_equal(h.predict_all(Xq, var=True)["var"], ref1["var"])
        assert np.array_equal(h.predict_all(Xq, Jvar=True)["Jvar"], ref3["Jvar"])
    h.close()


def test_non_finite_inputs_raise_like_sklearn():
    """sklearn's check_array refuses NaN / inf in X at fit and predict time (ValueError); Y rows with NaN are the
    reference's own filter (gaussian_process.py:33-35) and stay legal."""
    from gaussian_process_transportation_amd import GaussianProcess, _lib
    rng = np.random.default_rng(2)
    X = rng.uniform(0, 1, (50, 2)); Y = np.sin(3 * X)
    gp = GaussianProcess(kernel=sk_kernel(1.0, [0.3, 0.3], 1e-3), optimizer=None, verbose=False)
    Xbad = X.copy(); Xbad[7, 1] = np.nan
    with pytest.raises(ValueError, match="NaN or infinity"):
        gp.fit(Xbad, Y)
    Ynan = Y.copy(); Ynan[3, 0] = np.nan
    gp.fit(X, Ynan)                                            # filtered, not an error
    q = rng.uniform(0, 1, (5, 2)); q[2, 0] = np.inf
    for call in (lambda: gp.predict(q), lambda: gp.predict(q, return_std=True), lambda: gp.derivative(q),
                 lambda: gp.derivative_of_variance(q), lambda: gp.samples(q)):
        with pytest.raises(ValueError, match="NaN or infinity"):
            call()
    h = _lib.Handle(0)
    with pytest.raises(ValueError):
        h.fit(np.zeros((4, 16)), np.zeros((4, 1)), [1.0], 1.0, 1e-3, 0.0)       # D = 16: beyond the documented limit of 15


def test_covariance_needs_the_inverse_factor_of_the_committed_model():
    """A handle that fitted one model and then RECEIVED another of the same padded size (factor_alloc + commit) still
    has the old L^-1 in its workspace: predict_cov / samples must refuse instead of mixing two models."""
    import torch
    from gaussian_process_transportation_amd import _lib
    from gaussian_process_transportation_amd.distributed import wrap_device_bytes
    rng = np.random.default_rng(9)
    N = 300
    X = rng.uniform(0, 1, (N, 2)); Xq = rng.uniform(0, 1, (20, 2))
    ha, hb = _lib.Handle(0), _lib.Handle(0)
    ha.fit(X, np.sin(3 * X), np.array([0.2, 0.3]), 0.5, 1e-3, 1e-10)
    hb.fit(X + 0.05, np.cos(2 * X), np.array([0.4, 0.1]), 1.5, 1e-2, 1e-10)
    hb.predict_cov(Xq)                                         # fine: hb's own model
    src, nbytes = ha.factor_blob()
    dst, nbytes2 = hb.factor_alloc(N, 2, 2)
    assert nbytes == nbytes2
    dev = torch.device("cuda", 0)
    wrap_device_bytes(dst, nbytes, dev).copy_(wrap_device_bytes(src, nbytes, dev))
    torch.cuda.synchronize()
    hb.factor_commit()
    a = ha.predict_all(Xq, mean=True, var=True)
    b = hb.predict_all(Xq, mean=True, var=True)
    assert np.array_equal(a["mean"], b["mean"]) and np.array_equal(a["var"], b["var"])
    for call in (lambda: hb.predict_cov(Xq), hb.export_inverse_factor, hb.lml, lambda: hb.export(want_alpha=False)):
        with pytest.raises(_lib.GptError):
            call()
    ha.close(); hb.close()


def test_fp32_exact_gp_model():
    """gpt_set_dtype(GPT_F32) on the plain exact GP: fp64 factorisation, fp32 prediction kernels.  Tolerance 2e-4 of
    the array scale (fp32 sums of N = 900 cancelling products); the hand-off blob carries the element type."""
    from gaussian_process_transportation_amd import _lib
    from oracle import gp_oracle as orc
    X, Y, Xq = orc.synthetic_problem(900, 5000)
    c, ls, noise, jit = 0.1, np.array([0.1, 0.1, 0.1]), 1e-4, 1e-10
    h = _lib.Handle(0)
    h.set_dtype(_lib.GPT_F32)
    h.fit(X, Y, ls, c, noise, jit)
    assert h.model_info() == (1, _lib.GPT_F32)
    out = h.predict_all(Xq, mean=True, var=True, J=True, Jvar=True, dvar=True)
    assert all(v.dtype == np.float32 for v in out.values())
    L, a = orc.gpr_fit(X, Y, c, ls, noise, jit)
    mean, var, J, Jvar = orc.posterior_all_fast(Xq, X, L, a, c, ls, noise, want_jvar=True)
    for k, ref in (("mean", mean), ("var", var), ("J", J), ("Jvar", Jvar)):
        assert_parity(out[k], ref, 2e-4, k + " (fp32 model)")
    Lg, ag = h.export()                                         # the fp64 factor stays available on the fitting handle
    assert_parity(Lg, L, 1e-9, "L")
    assert_parity(ag, a, 1e-7, "alpha")
    h.close()
    # the same through the mirrored class
    from gaussian_process_transportation_amd import GaussianProcess
    gp = GaussianProcess(kernel=sk_kernel(c, ls, noise), optimizer=None, verbose=False, dtype="float32")
    gp.fit(X, Y)
    m32, s32 = gp.predict(Xq, return_std=True)
    assert m32.dtype == np.float32
    assert_parity(m32, mean, 2e-4, "mean (GaussianProcess, fp32)")
    assert_parity(s32, np.repeat((np.sqrt(var) - np.sqrt(noise))[:, None], 3, axis=1), 5e-3, "std (GaussianProcess, fp32)")
    with pytest.raises(_lib.GptError):
        gp.predict(Xq[:5], return_cov=True)                     # covariance / samples: fp64 models only
    # host-buffer path across the 131072-query chunk boundary (two staging sets, copy stream) in the 4-byte element type
    big = np.random.default_rng(8).uniform(-0.1, 1.1, (140_000, 3))
    full = gp._handle.predict_all(big, mean=True, var=True, J=True, Jvar=True)
    for a, b in ((0, 700), (130_900, 131_300), (139_500, 140_000)):
        part = gp._handle.predict_all(big[a:b], mean=True, var=True, J=True, Jvar=True)
        assert np.array_equal(part["mean"], full["mean"][a:b]) and np.array_equal(part["J"], full["J"][a:b])
        assert_parity(part["var"], full["var"][a:b], 1e-4, "var across chunks (fp32)")
        assert_parity(part["Jvar"], full["Jvar"][a:b], 1e-4, "Jvar across chunks (fp32)")


def test_callable_optimizer_follows_sklearn_protocol():
    """sklearn accepts `optimizer=callable(obj_func, initial_theta, bounds) -> (theta_opt, func_min)` (_gpr.py:296-305,
    664-667).  A callable that runs scipy's L-BFGS-B itself must land where the built-in string option lands, and
    obj_func must honour eval_gradient."""
    import scipy.optimize
    from gaussian_process_transportation_amd import GaussianProcess
    g = load_golden("synthetic_3d_N256")
    calls = {"n": 0, "value_only": None}

    def my_optimizer(obj_func, initial_theta, bounds):
        calls["n"] += 1
        calls["value_only"] = obj_func(initial_theta, eval_gradient=False)
        v, gr = obj_func(initial_theta)
        assert np.isscalar(calls["value_only"]) and calls["value_only"] == v and gr.shape == initial_theta.shape
        res = scipy.optimize.minimize(obj_func, initial_theta, method="L-BFGS-B", jac=True, bounds=bounds)
        return res.x, res.fun

    np.random.seed(3)
    a = GaussianProcess(kernel=sk_kernel(0.1, [0.1, 0.1, 0.1], 1e-4), optimizer=my_optimizer, n_restarts_optimizer=1, verbose=False)
    a.fit(g["X"], g["Y"])
    np.random.seed(3)
    b = GaussianProcess(kernel=sk_kernel(0.1, [0.1, 0.1, 0.1], 1e-4), optimizer="fmin_l_bfgs_b", n_restarts_optimizer=1, verbose=False)
    b.fit(g["X"], g["Y"])
    assert calls["n"] == 2                                    # the kernel's theta + one restart
    assert_parity(np.asarray(a.kernel.theta), np.asarray(b.kernel.theta), 1e-10, "theta")
    with pytest.raises(ValueError):
        GaussianProcess(kernel=sk_kernel(0.1, [0.1], 1e-4), optimizer="nelder", verbose=False).fit(g["X"], g["Y"])


@pytest.mark.parametrize("dtype,tol", [("float64", 1e-7), ("float32", 5e-4)])
def test_svgp_transport_attribute_protocol(dtype, tol):
    """SVGPTransport (reference: transportation/torch/stocastic_variational_gaussian_process_transportation.py:46-102) over
    the GPU exact-conversion predictor, against the same algebra written out with the CPU oracle: affine pre-alignment,
    mean / std of the residual field at the rotated demo, velocities through (I + J) after the affine derivative,
    their variance from the Jacobian std, orientations through the two rotations.  PARITY UNPINNED (gpytorch and the
    reference's `quaternion` package are absent): the oracle is this repo's restatement."""
    from gaussian_process_transportation_amd import AffineTransform, SVGPTransport
    from gaussian_process_transportation_amd.quaternion import quaternion_from_nonorthogonal, quaternion_multiply
    from oracle import gp_oracle as orc
    rng = np.random.default_rng(4)
    N, Z, M, T, D = 400, 96, 300, 3, 3
    src = rng.uniform(0, 1, (N, D))
    R = np.linalg.qr(rng.standard_normal((D, D)))[0]
    R *= np.sign(np.linalg.det(R))
    tgt = src @ R.T + 0.3 + 0.05 * np.sin(3 * src)
    demo = rng.uniform(0.1, 0.9, (M, D))
    vel = rng.standard_normal((M, D)) * 0.1
    ori = rng.standard_normal((M, 4)); ori /= np.linalg.norm(ori, axis=1, keepdims=True)
    aff = AffineTransform(verbose=False).fit(src, tgt)
    src_al = aff.predict(src)
    idx = rng.choice(N, Z, replace=False)
    A = rng.standard_normal((T, Z, Z))
    pp = dict(x_inducing=src_al[idx], var_inducing=A @ A.transpose(0, 2, 1) / Z * 1e-3 + 1e-4 * np.eye(Z),
              y_inducing=(tgt - src_al)[idx].T.copy(), outputscale=np.array([0.02, 0.03, 0.025]), lengthscale=np.array([0.3, 0.35, 0.25]))
    tr = SVGPTransport(dtype=dtype, verbose=False)
    tr.source_distribution, tr.target_distribution = src, tgt
    tr.training_traj, tr.training_delta, tr.training_ori = demo, vel, ori
    with pytest.raises(NotImplementedError):
        tr.fit_transportation()                                 # variational training is gpytorch's: not rebuilt
    tr.fit_transportation(pseudo_points=pp)
    tr.apply_transportation()
    # the same with the oracle
    rot = aff.predict(demo)
    mean, std, J, Jstd = orc.svgp_exact_oracle(rot, pp["x_inducing"], pp["var_inducing"], pp["y_inducing"], pp["outputscale"], pp["lengthscale"])
    dA = aff.derivative(rot)
    v1 = dA @ vel[:, :, None]
    assert tr.training_traj_old is demo
    assert_parity(tr.training_traj, rot + mean, tol, "training_traj")
    assert_parity(tr.std, std, 50 * tol, "std")                 # sqrt of a variance that nearly cancels at the pseudo-points
    assert_parity(tr.training_delta, ((np.eye(D) + J) @ v1)[:, :, 0], tol, "training_delta")
    assert_parity(tr.var_vel_transported, (Jstd ** 2 @ v1 ** 2)[:, :, 0], 50 * tol, "var_vel_transported")
    q = quaternion_multiply(quaternion_from_nonorthogonal(np.eye(D) + J),
                            quaternion_multiply(quaternion_from_nonorthogonal(aff.rotation_matrix), ori))
    assert_parity(tr.training_ori, q, 10 * tol, "training_ori")
    assert np.allclose(np.linalg.norm(tr.training_ori, axis=1), 1.0, atol=1e-6)


@pytest.mark.parametrize("N", [4700, 5300])
def test_overlapped_factor_and_inverse(N, monkeypatch):
    """For 4096 < NP <= 12288 the second half of the Cholesky runs on 3/8 of the CUs while W11 = L11^-1 and L21 W11 are
    computed on the others (CU-masked streams, gpt_fit.hip:launch_factor_inverse).  L against LAPACK, W L = I, alpha, the
    pivot of a matrix that stops being positive definite in the SECOND half, and the serial form (GPT_FIT_OVERLAP=0)."""
    import scipy.linalg
    import scipy.linalg.lapack
    from gaussian_process_transportation_amd import _lib
    from oracle import gp_oracle as orc
    rng = np.random.default_rng(N)
    X = rng.uniform(0, 1, (N, 3))
    Y = np.sin(3 * X[:, :2])
    ls, c, noise, jit = np.array([0.25, 0.3, 0.2]), 0.7, 1e-3, 1e-10
    Kref = c * orc.rbf_gram(X / ls) + (noise + jit) * np.eye(N)
    Lref = np.linalg.cholesky(Kref)
    aref = scipy.linalg.cho_solve((Lref, True), Y)
    for overlap in ("1", "0"):
        monkeypatch.setenv("GPT_FIT_OVERLAP", overlap)
        h = _lib.Handle(0)                                   # the environment is read when a handle first factorises
        h.fit(X, Y, ls, c, noise, jit)
        L, alpha = h.export()
        assert_parity(L, Lref, 1e-11, f"L_ (overlap {overlap})")
        assert_parity(alpha, aref, 1e-7, f"alpha_ (overlap {overlap})")
        W = h.export_inverse_factor()
        assert np.abs(W @ Lref - np.eye(N)).max() < 1e-9
        h.fit(X, Y, ls, c, noise, jit)                       # streams and events are reused by the next fit
        assert np.array_equal(h.export()[0], L)
        Sigma = 1e-3 * np.eye(N)
        Sigma[N - 700, N - 700] = -2.0
        with pytest.raises(np.linalg.LinAlgError) as ei:
            h.fit_noise_matrix(X, Y, ls, c, Sigma)
        assert int(str(ei.value).split("pivot")[1].split()[0]) == N - 700 + 1
        h.close()


def test_lml_objective_is_fit_plus_gradient_without_a_model():
    """gpt_lml_objective (the optimizer's inner loop: one C call, no packed model) returns exactly what gpt_fit_kernel +
    gpt_lml_gradient return, and leaves the handle without a model."""
    from gaussian_process_transportation_amd import _lib
    g = load_golden("synthetic_3d_N256")
    X, Y = g["X"], g["Y"]
    h = _lib.Handle(0)
    for ls, c, noise, kt in ((np.array([0.1, 0.1, 0.1]), 0.1, 1e-4, 0), (np.array([0.3]), 0.5, 1e-3, 0), (np.array([0.2, 0.3, 0.4]), 0.2, 1e-3, 2)):
        h.fit(X, Y, ls, c, noise, 1e-10, kt)
        v0, g0 = h.lml_gradient(ls.size)
        v1, g1 = h.lml_objective(X, Y, ls, c, noise, 1e-10, kt)
        assert v0 == v1 and np.array_equal(g0, g1)
        with pytest.raises(_lib.GptError):
            h.predict_all(g["Xq"], mean=True)
    h.fit(X, Y, np.array([0.1, 0.1, 0.1]), 0.1, 1e-4, 1e-10)
    assert_parity(h.predict_all(g["Xq"], mean=True)["mean"], g["mean"], RTOL, "mean after a real fit")
    with pytest.raises(np.linalg.LinAlgError):
        h.lml_objective(np.vstack([X, X[:5]]), np.vstack([Y, Y[:5]]), np.array([0.1]), 1.0, 0.0, 0.0)
    h.close()


def test_concurrent_restart_runs_are_bit_identical_to_sequential_ones(monkeypatch):
    """The L-BFGS-B runs of sklearn's restart loop are independent (start points from the global RNG, which a run does not
    touch): driven concurrently on their own handles they must return exactly what the sequential loop returns — same
    start points, same kernels on the same data — and leave the global RNG in the same state."""
    from gaussian_process_transportation_amd import GaussianProcess
    g = load_golden("letterS_2d")
    rng = np.random.default_rng(4)
    Xs = rng.uniform(0, 1, (600, 3))
    Ys = np.column_stack([np.sin(5 * Xs[:, 0]) * Xs[:, 1], np.cos(3 * Xs[:, 2])]) + 0.03 * rng.standard_normal((600, 2))
    for X, Y, k0, nres in ((g["gp_X"], g["gp_Y"], sk_kernel(10.0, 4 * np.ones(2), 0.01), 5),          # the letter-S transport GP
                           (Xs, Ys, sk_kernel(1.0, 0.5 * np.ones(3), 0.01), 3)):
        out = {}
        for workers in ("1", "6", "3"):
            monkeypatch.setenv("GPT_OPT_WORKERS", workers)
            np.random.seed(0)
            gp = GaussianProcess(kernel=k0, n_restarts_optimizer=nres, verbose=False).fit(X, Y)
            out[workers] = (np.asarray(gp.gp.kernel_.theta).copy(), gp.gp.log_marginal_likelihood_value_, np.random.uniform())
        for workers in ("6", "3"):
            assert np.array_equal(out[workers][0], out["1"][0]) and out[workers][1] == out["1"][1]
            assert out[workers][2] == out["1"][2]                      # the RNG stream continues from the same state
    assert_parity(out["1"][0], out["6"][0], 0.0, "theta")


# ---------------------------------------------------------------------------------------------------------------
# The launch shapes bench.py times, checked against reference vectors (not only against identities): the golden
# queries are scattered through batches of the benchmark's size, so they pass through the whole-round path of k_var
# (rounds of 256 column blocks, nbi = 16 i-blocks per block at N = 8192) in many rounds and workgroups.
# ---------------------------------------------------------------------------------------------------------------
def _predict_dev(h, Xq, mean=False, var=False, J=False, Jvar=False, dvar=False):
    """One gpt_predict_all_dev launch on device-resident queries — the call bench.py times — results as numpy."""
    import torch
    N, D, O, _ = h.info()
    nt, dt = h.model_info()
    from gaussian_process_transportation_amd import _lib
    ty = torch.float64 if dt == _lib.GPT_F64 else torch.float32
    dev = torch.device("cuda", 0)
    xq = torch.from_numpy(np.ascontiguousarray(Xq)).to(dev, dtype=ty)
    M = xq.shape[0]
    vshape, jvshape = ((M,), (M, D)) if nt == 1 else ((M, nt), (M, nt, D))
    bufs = {"mean": torch.empty((M, O), dtype=ty, device=dev) if mean else None,
            "var": torch.empty(vshape, dtype=ty, device=dev) if var else None,
            "J": torch.empty((M, O, D), dtype=ty, device=dev) if J else None,
            "Jvar": torch.empty(jvshape, dtype=ty, device=dev) if Jvar else None,
            "dvar": torch.empty((D, M), dtype=ty, device=dev) if dvar else None}
    torch.cuda.synchronize()
    h.predict_all_dev(xq.data_ptr(), M, *[(bufs[k].data_ptr() if bufs[k] is not None else 0) for k in ("mean", "var", "J", "Jvar", "dvar")])
    h.synchronize()
    return {k: (v.cpu().numpy() if v is not None else None) for k, v in bufs.items()}


def _golden_n8192_model():
    from gaussian_process_transportation_amd import _lib
    g = load_golden("synthetic_3d_N8192")
    rng = np.random.default_rng(0)
    X = rng.uniform(0, 1, (8192, 3))
    Y = 0.05 * np.sin(4 * X) + 0.01 * rng.standard_normal((8192, 3))
    h = _lib.Handle(0)
    h.fit(X, Y, g["length_scale"], float(g["constant_value"]), float(g["noise_level"]), float(g["alpha"]))
    return g, h


def _check_golden_rows(g, out1, out4, rows, what):
    noise = float(g["noise_level"])
    std1 = np.sqrt(out1["var"][rows]) - np.sqrt(noise)           # gaussian_process.py:49
    std4 = np.sqrt(out4["var"][rows]) - np.sqrt(noise)
    assert_parity(out1["mean"][rows], g["mean"], RTOL, f"mean ({what})")
    assert_parity(std1, g["std"][:, 0], RTOL, f"std, 1-column launch ({what})")
    assert_parity(out1["J"][rows], g["J"], RTOL, f"J ({what})")
    assert_parity(std4, g["std"][:, 0], RTOL, f"std, fused launch ({what})")
    assert_parity(out4["Jvar"][rows], g["Jvar"][:, 0, :], RTOL, f"Jvar ({what})")
    assert_parity(out4["dvar"][:, rows], g["dvar"], RTOL, f"dvar ({what})")
    print(f"{what}: mean {relmax(out1['mean'][rows], g['mean']):.1e}, std {relmax(std1, g['std'][:, 0]):.1e} / {relmax(std4, g['std'][:, 0]):.1e}, "
          f"J {relmax(out1['J'][rows], g['J']):.1e}, Jvar {relmax(out4['Jvar'][rows], g['Jvar'][:, 0, :]):.1e}, "
          f"dvar {relmax(out4['dvar'][:, rows], g['dvar']):.1e} (max-norm relative, vs the reference's vectors)")


def test_headline_launch_shape_against_reference_vectors():
    """BASELINE configs[2] exactly as bench.py launches it — N = 8192, ONE device-pointer call over M = 500 000 resident
    queries, mode J (mean + var + Jacobian, 1 column per query: 7813 column blocks = 30 whole rounds + a tail) and the
    fused 4-column launch (31 250 blocks = 122 rounds + a tail) — with the reference's own 256 golden queries
    (tests/golden/synthetic_3d_N8192.npz, captured from gaussian_process.py:46-55, 63-126 run in the authoring
    container) scattered through the batch at stride 1953, i.e. through different rounds and workgroups."""
    g, h = _golden_n8192_model()
    M, stride = 500_000, 1953
    Xq = np.random.default_rng(1).uniform(-0.1, 1.1, (M, 3))
    rows = np.arange(256) * stride
    assert rows[-1] < M
    Xq[rows] = g["Xq"]
    out1 = _predict_dev(h, Xq, mean=True, var=True, J=True)                 # bench.py's step
    out4 = _predict_dev(h, Xq, var=True, Jvar=True, dvar=True)              # bench.py --jvar's variance launch (+ dvar)
    _check_golden_rows(g, out1, out4, rows, "M=500k, whole rounds")
    # the golden rows sit in 256 different column blocks of the 1-column launch, in rounds 0 .. 30, and the two launches
    # agree on every row (two instantiations, two work splits)
    assert len(set((rows // 64).tolist())) == 256 and (rows[-1] // 64) // 256 >= 29
    assert_parity(out4["var"], out1["var"], 1e-10, "var: fused vs 1-column launch, all 500k rows")
    assert out1["var"].min() >= 0.0 and out1["var"].max() <= float(g["constant_value"]) + float(g["noise_level"]) + 1e-12
    h.close()


def test_reference_grid_size_at_n8192_against_reference_vectors():
    """N = 8192 with M = 10^4 queries — the size of the reference's plotting grids (plot_utils.py:13, 287-288): 157
    column blocks on 256 workgroups, i.e. the item-list / cut-sweep path with many blocks, against the golden queries
    scattered at stride 39."""
    g, h = _golden_n8192_model()
    M, stride = 10_000, 39
    Xq = np.random.default_rng(2).uniform(-0.1, 1.1, (M, 3))
    rows = np.arange(256) * stride
    Xq[rows] = g["Xq"]
    out1 = _predict_dev(h, Xq, mean=True, var=True, J=True)
    out4 = _predict_dev(h, Xq, var=True, Jvar=True, dvar=True)
    _check_golden_rows(g, out1, out4, rows, "M=10^4, cut sweeps")
    assert_parity(out4["var"], out1["var"], 1e-10, "var: fused vs 1-column launch")
    # and the host-buffer entry point on the same batch
    host = h.predict_all(Xq, mean=True, var=True, J=True)
    assert np.array_equal(host["mean"], out1["mean"]) and np.array_equal(host["J"], out1["J"]) and np.array_equal(host["var"], out1["var"])
    h.close()


def test_svgp_fp32_bench_launch_shape_against_oracle():
    """BASELINE configs[4] as `bench.py --config svgp` launches it: Z = 2048 inducing points, T = D = 3, fp32, ONE
    device-pointer call over M = 10^6 resident queries (the stacked-task whole-round path: 62 500 column blocks, nbi = 4),
    1024 strided rows against the fp64 CPU restatement.  PARITY UNPINNED (gpytorch absent, the reference holds no
    fixture); tolerance as test_svgp_fp32_config5_shape: no worse than the same algebra in numpy float32, and 1e-4 of
    the array scale."""
    from gaussian_process_transportation_amd import _lib
    from oracle import gp_oracle as orc
    M = 1_000_000
    Z, Sigma, y, osc, ls, Xq = orc.svgp_synthetic_problem(2048, M)
    rows = np.arange(1024) * 976 + 7
    assert rows[-1] < M
    ref64 = orc.svgp_exact_oracle_fast(Xq[rows], Z, Sigma, y, osc, ls)
    ref32 = orc.svgp_exact_oracle_fast(Xq[rows], Z, Sigma, y, osc, ls, dtype=np.float32)
    h = _lib.Handle(0)
    h.fit_svgp(Z, y, Sigma, ls, osc, dtype=_lib.GPT_F32)
    out = _predict_dev(h, Xq, mean=True, var=True, J=True, Jvar=True)       # bench.py's svgp step
    assert out["var"].dtype == np.float32 and out["var"].shape == (M, 3) and out["Jvar"].shape == (M, 3, 3)
    got = (out["mean"][rows], np.sqrt(np.maximum(out["var"][rows], 0)), out["J"][rows], np.sqrt(np.maximum(out["Jvar"][rows], 0)))
    for name, g32, r64, r32 in zip(("mean", "std", "J", "J std"), got, ref64, ref32):
        scale = np.max(np.abs(r64))
        err_gpu = np.max(np.abs(g32.astype(np.float64) - r64)) / scale
        err_ref_arith = np.max(np.abs(r32.astype(np.float64) - r64)) / scale
        print(f"svgp fp32 M=1e6 {name}: GPU {err_gpu:.2e}, reference arithmetic {err_ref_arith:.2e} (relative to {scale:.3g})")
        assert err_gpu <= err_ref_arith and err_gpu <= 1e-4, (name, err_gpu, err_ref_arith)
    assert np.all(np.isfinite(out["mean"])) and np.all(np.isfinite(out["J"]))
    assert out["var"].min() >= 0.0 and out["var"].max() <= 1.0 + 1e-5
    h.close()


@pytest.mark.parametrize("tag,nu,code", [("12", 0.5, 1), ("32", 1.5, 2), ("52", 2.5, 3)])
def test_matern_kernels_in_five_dimensions_vs_reference(tag, nu, code):
    """Matern x the wide layout (D = 5: k_gram<MAX_D>, k_mean_jac<.., KT_MATERN*, MAX_D>, k_var wide with a Matern KT,
    k_lml_terms wide): fit, mean, std, covariance, LML + gradient against the reference class (matern_5d.npz, ARD and
    isotropic), and the same model in fp32 against the fp64 result."""
    from sklearn.gaussian_process.kernels import Matern, WhiteKernel, ConstantKernel as C
    from gaussian_process_transportation_amd import GaussianProcess, _lib
    g = load_golden("matern_5d")
    for kk, ls, n_ls in (("ard", g["length_scale"], 5), ("iso", float(g["iso_length_scale"]), 1)):
        pre = f"m{tag}_{kk}_"
        gp = GaussianProcess(kernel=C(0.3) * Matern(ls, nu=nu) + WhiteKernel(0.01), optimizer=None, verbose=False)
        gp.fit(g["X"], g["Y"])
        assert_parity(gp.gp.alpha_, g[pre + "alpha_"], RTOL, "alpha_")
        assert_parity(np.diag(gp.gp.L_), g[pre + "Ldiag"], RTOL, "diag L")
        m, s = gp.predict(g["Xq"], return_std=True)
        assert_parity(m, g[pre + "mean"], RTOL, "mean")
        assert_parity(s, g[pre + "std"], RTOL, "std")
        _, cov = gp.predict(g["Xq"][:12], return_cov=True)
        assert_parity(cov, g[pre + "cov"], RTOL, "cov")
        h = _lib.Handle(0)
        for th, v, gr in zip(g[pre + "lml_theta"], g[pre + "lml_value"], g[pre + "lml_grad"]):
            h.fit(g["X"], g["Y"], np.exp(th[1:1 + n_ls]), np.exp(th[0]), np.exp(th[1 + n_ls]), 1e-10, code)
            lml, grad = h.lml_gradient(n_ls)
            assert lml == pytest.approx(float(v), rel=1e-9)
            assert_parity(grad, gr, 1e-6, f"d lml / d theta ({kk})")
        # Matern x fp32: fp64 factorisation, fp32 prediction kernels; against the fp64 model at fp32's resolution
        big = np.random.default_rng(3).uniform(-0.1, 1.1, (3000, 5))
        lsv = np.atleast_1d(np.asarray(ls, dtype=float))
        h.fit(g["X"], g["Y"], lsv, 0.3, 0.01, 1e-10, code)
        o64 = h.predict_all(big, mean=True, var=True)
        h.set_dtype(_lib.GPT_F32)
        h.fit(g["X"], g["Y"], lsv, 0.3, 0.01, 1e-10, code)
        o32 = h.predict_all(big, mean=True, var=True)
        assert o32["mean"].dtype == np.float32
        assert_parity(o32["mean"], o64["mean"], 2e-4, "mean (Matern, fp32 model)")
        assert np.max(np.abs(o32["var"] - o64["var"])) < 2e-4 * 0.31
        with pytest.raises(ValueError):
            h.predict_all(big[:5], J=True)                     # RBF-only formulas (reference quirk 6): refused
        h.close()


@pytest.mark.parametrize("D", [9, 12, 15])
def test_rows_of_sixteen_layout_against_the_oracle(D):
    """Input dimension 9 .. 15: source rows of 16 (k_gram<16>, k_mean_jac<.., 16>, k_var<.., DW = 16> with NCOMP 1 and 16,
    k_lml_terms<16>).  The reference fixtures synthetic_12d_N160 / synthetic_15d_N96 pin the small case (test_golden_*); this is a
    model of several i-blocks with a tail in the variance launch: every output against the CPU oracle, the fp32 model against the
    fp64 one, a Matern kernel (mean + variance), the posterior covariance of a few queries and LML + gradient against the
    oracle's (itself pinned to sklearn's by the fixtures)."""
    from gaussian_process_transportation_amd import _lib
    from oracle import gp_oracle as orc
    rng = np.random.default_rng(100 + D)
    N, M = 1300, 2500
    X = rng.uniform(0, 1, (N, D)); Y = np.column_stack([np.sin(X.sum(1)), np.cos(2 * X[:, 0] - X[:, D - 1])]); Xq = rng.uniform(-0.05, 1.05, (M, D))
    c, ls, noise, jit = 0.6, np.linspace(0.9, 1.6, D), 1e-3, 1e-10
    h = _lib.Handle(0)
    h.fit(X, Y, ls, c, noise, jit)
    out = h.predict_all(Xq, mean=True, var=True, J=True, Jvar=True, dvar=True)
    only_var = h.predict_all(Xq, var=True)
    only_jv = h.predict_all(Xq, Jvar=True)
    L, a = orc.gpr_fit(X, Y, c, ls, noise, jit)
    idx = np.arange(0, M, 5)
    mean, var, J, Jvar = orc.posterior_all_fast(Xq[idx], X, L, a, c, ls, noise, want_jvar=True)
    assert_parity(out["mean"][idx], mean, RTOL, "mean")
    assert_parity(out["var"][idx], var, RTOL, "var")
    assert_parity(out["J"][idx], J, RTOL, "J")
    assert_parity(out["Jvar"][idx], Jvar, RTOL, "Jvar")
    assert_parity(only_var["var"], out["var"], 1e-10, "var alone vs fused launch")
    assert_parity(only_jv["Jvar"], out["Jvar"], 1e-10, "Jvar alone vs fused launch")
    eps = 1e-5
    sub = idx[:40]
    for d in (0, D // 2, D - 1):                                # d var / d x against central differences of var
        e = np.zeros(D); e[d] = eps
        vp = h.predict_all(Xq[sub] + e, var=True)["var"]; vm = h.predict_all(Xq[sub] - e, var=True)["var"]
        assert_parity(out["dvar"][d][sub], (vp - vm) / (2 * eps), 1e-4, f"dvar[{d}] vs finite difference")
    _, cov = h.predict_cov(Xq[:10])
    assert_parity(np.diag(cov), out["var"][:10], 1e-8, "diag of the posterior covariance")
    _, cov_o = orc.gpr_predict(Xq[:10], X, L, a, c, ls, noise, return_cov=True)
    assert_parity(cov, cov_o[..., 0], 1e-7, "posterior covariance")
    lml, grad = h.lml_gradient(D)
    lml_o, grad_o = orc.log_marginal_likelihood(np.log(np.concatenate([[c], ls, [noise]])), X, Y, D, jit)
    assert lml == pytest.approx(float(lml_o), rel=1e-9)
    assert_parity(grad, grad_o, 1e-6, "d lml / d theta")
    # fp32 model
    h.set_dtype(_lib.GPT_F32)
    h.fit(X, Y, ls, c, noise, jit)
    o32 = h.predict_all(Xq, mean=True, var=True, J=True, Jvar=True)
    assert o32["mean"].dtype == np.float32
    assert_parity(o32["mean"], out["mean"], 5e-4, "mean (fp32 model)")
    assert_parity(o32["J"], out["J"], 5e-4, "J (fp32 model)")
    assert np.max(np.abs(o32["var"] - out["var"])) < 5e-4 * (c + noise)
    assert np.max(np.abs(o32["Jvar"] - out["Jvar"])) < 5e-4 * np.max(out["Jvar"])
    # Matern 3/2: mean + variance
    h.set_dtype(_lib.GPT_F64)
    h.fit(X, Y, ls, c, noise, jit, 2)
    om = h.predict_all(Xq[idx], mean=True, var=True)
    Lm, am = orc.gpr_fit(X, Y, c, ls, noise, jit, kind="matern32")
    mm, sm = orc.gpr_predict(Xq[idx], X, Lm, am, c, ls, noise, return_std=True, kind="matern32")
    assert_parity(om["mean"], mm, RTOL, "mean (Matern 3/2)")
    assert_parity(om["var"], sm[:, 0] ** 2, RTOL, "var (Matern 3/2)")
    h.close()


def test_transport_orientation_on_the_recorded_robot_demo():
    """transport_orientation (policy_transportation.py:61-77) fed with the reference's own recorded demonstration
    (data/last.npz: 102 positions, quaternions and velocities of the robot, stored as arrays in robot_demo_last.npz).
    PARITY UNPINNED — the reference's `Quaternion` module is absent and its repo holds no transported orientation — so
    this stays a property test: unit norm, R(out) = polar(J_Phi) R(in) with J_Phi at the UN-rotated positions (the
    reference's quirk), and the identity map leaves the orientations unchanged.  The source / target surfaces of the
    recorded experiment are pickles and are not loaded: a synthetic surface pair under the trajectory stands in."""
    from gaussian_process_transportation_amd import GaussianProcessTransportation
    from gaussian_process_transportation_amd.quaternion import rotation_matrix_from_quaternion
    g = load_golden("robot_demo_last")
    traj, ori, vel = g["training_traj"], g["training_ori"], g["training_delta"]
    assert traj.shape == (102, 3) and ori.shape == (102, 4)
    assert_parity(np.linalg.norm(ori, axis=1), np.ones(102), 1e-5, "recorded quaternions are unit (to 2.5e-6)")
    lo, hi = traj.min(0) - 0.05, traj.max(0) + 0.05
    gx, gy = np.meshgrid(np.linspace(lo[0], hi[0], 20), np.linspace(lo[1], hi[1], 20))
    src = np.column_stack([gx.ravel(), gy.ravel(), np.full(gx.size, lo[2])])
    ang = 0.3
    Rz = np.array([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1.0]])
    span = hi - lo
    tgt = (src - src.mean(0)) @ Rz.T + src.mean(0) + np.array([0.02, -0.01, 0.03])
    tgt[:, 2] += 0.15 * span[0] * np.sin(2 * np.pi * (src[:, 0] - lo[0]) / span[0]) * np.cos(np.pi * (src[:, 1] - lo[1]) / span[1])
    kern = sk_kernel(0.05, [0.6 * float(np.max(span[:2]))], 1e-5)
    for target, identity in ((tgt, False), (src.copy(), True)):
        tr = GaussianProcessTransportation(kernel_transport=kern, optimizer=None, verbose=False)
        tr.source_distribution = src; tr.target_distribution = target
        tr.training_traj = traj; tr.training_delta = vel; tr.training_ori = ori
        tr.fit_transportation()
        tr.apply_transportation()
        out = tr.training_ori
        assert out.shape == ori.shape
        # the product with a unit quaternion keeps the norm of the recorded one (no renormalisation, as in the reference)
        assert_parity(np.linalg.norm(out, axis=1), np.linalg.norm(ori, axis=1), 1e-9, "quaternion norm preserved")
        J = tr.method.delta_map.derivative(traj)
        Jg = tr.method.affine_transform.derivative(traj)
        Jphi = Jg + J @ Jg
        U, _, Vt = np.linalg.svd(Jphi)
        polar = U @ Vt
        assert np.all(np.linalg.det(Jphi) > 0)
        Rout, Rin = rotation_matrix_from_quaternion(out), rotation_matrix_from_quaternion(ori)
        assert np.max(np.abs(Rout - polar @ Rin)) < 5e-3
        if identity:
            assert np.max(np.abs(Rout - Rin)) < 1e-6
            assert_parity(tr.training_traj, traj, 1e-6, "identity transport leaves the positions")


# ---------------------------------------------------------------------------------------------------------------
# round 4
def test_devices_argument_shards_rows_over_the_handles_of_one_process():
    """GaussianProcess(devices=[...]) / GaussianProcessTransportation(devices=[...]) (SURVEY 8b's n_devices):
    fit on the first device, one gpt_factor_copy per further device, predict / derivative / derivative_of_variance shard their
    rows over a host thread per device.  This pool has one GPU: the same device listed twice is two handles, two streams and a
    device-to-device copy of the model — everything but the peer link.  Against the single-handle class: mean and Jacobian are
    per-query contractions (bit-identical); a variance depends on its batch at the last-bit level (the work split depends on M),
    so the shards are held bit for bit to the single handle predicting the same shard alone, and to 1e-10 to the whole batch."""
    from sklearn.gaussian_process.kernels import RBF, ConstantKernel, WhiteKernel
    from gaussian_process_transportation_amd import GaussianProcess, GaussianProcessTransportation
    from gaussian_process_transportation_amd.device_group import DeviceGroup
    from gaussian_process_transportation_amd.distributed import shard_range
    g = load_golden("synthetic_3d_N1024")
    kern = ConstantKernel(0.1) * RBF([0.1, 0.1, 0.1]) + WhiteKernel(1e-4)
    one = GaussianProcess(kern, optimizer=None, verbose=False).fit(g["X"], g["Y"])
    two = GaussianProcess(kern, optimizer=None, verbose=False, devices=[0, 0]).fit(g["X"], g["Y"])
    assert isinstance(two._handle, DeviceGroup) and len(two._handle.handles) == 2
    M = 5001
    Xq = np.random.default_rng(3).uniform(-0.1, 1.1, (M, 3))
    m1, s1 = one.predict(Xq, return_std=True)
    m2, s2 = two.predict(Xq, return_std=True)
    J1, V1 = one.derivative(Xq, return_var=True)
    J2, V2 = two.derivative(Xq, return_var=True)
    assert np.array_equal(m1, m2) and np.array_equal(J1, J2)
    assert_parity(s2, s1, 1e-10, "std, sharded vs whole batch")
    assert_parity(V2, V1, 1e-10, "Jacobian variance, sharded vs whole batch")
    assert_parity(two.derivative_of_variance(Xq), one.derivative_of_variance(Xq), 1e-9, "d var, sharded vs whole batch")
    for r in range(2):
        a, b = shard_range(M, r, 2)
        ms, ss = one.predict(Xq[a:b], return_std=True)
        assert np.array_equal(ms, m2[a:b]) and np.array_equal(ss, s2[a:b]), "a shard must be what one handle predicts for it"
    # the golden vectors through the sharded class (the reference's own numbers: gaussian_process.py:46-55, 63-102)
    mg, sg = two.predict(np.tile(g["Xq"], (40, 1)), return_std=True)
    nq = len(g["Xq"])
    assert_parity(mg[-nq:], g["mean"], RTOL, "mean (second shard)")
    assert_parity(sg[:nq], g["std"], RTOL, "std (first shard)")
    # fit-side state stays with the first handle; the replica refuses what it does not hold
    assert two.gp.L_.shape == (1024, 1024) and np.isfinite(two.gp.log_marginal_likelihood_value_)
    assert two.predict(g["Xq"][:50], return_cov=True)[1].shape == (50, 50, 3)
    from gaussian_process_transportation_amd import _lib
    with pytest.raises(_lib.GptError):
        two._handle.handles[1].export(want_alpha=False)
    # the user-facing class
    rng = np.random.default_rng(0)
    src = rng.uniform(0, 1, (300, 3)); tgt = src + 0.05 * np.sin(3 * src)
    traj, vel = rng.uniform(0, 1, (2000, 3)), rng.standard_normal((2000, 3))
    res = []
    for devs in (None, [0, 0]):
        tr = GaussianProcessTransportation(kernel_transport=ConstantKernel(0.1) * RBF([0.3]) + WhiteKernel(1e-4), optimizer=None,
                                           verbose=False, devices=devs)
        tr.source_distribution, tr.target_distribution = src, tgt
        tr.training_traj, tr.training_delta = traj.copy(), vel.copy()
        tr.fit_transportation()
        tr.apply_transportation()
        res.append((tr.training_traj, tr.std, tr.training_delta, tr.var_vel_transported))
    for a, b, name in zip(res[0], res[1], ("traj", "std", "vel", "var_vel")):
        assert_parity(b, a, 1e-10, name + " (devices=[0, 0] vs one device)")
    two._handle.close()


@pytest.mark.parametrize("N", [4200, 5300])
def test_factor_and_inverse_forms_against_lapack(N, monkeypatch):
    """Every form of the factor + inverse plan (csrc/gpt_fit_plan.h) on ONE handle, the plan — and the scratch arena it needs —
    changing between fits: the shipped split form (second half's launch chain beside the first half's inverse on CU-masked
    streams), the same with a group width that moves the split OFF the half (GPT_POTRF_GROUP=4: N = 5300 -> h = 2560, r = 3072,
    the shape that wrote past a half-sized scratch region in round 3 — VERDICT r3 item 3), the one-leaf form, and the opt-in
    left-looking form for three panel widths, with streams and in the serial order.  L against LAPACK, W L = I, alpha; the pivot
    of a matrix that stops being positive definite late in the second half."""
    import scipy.linalg
    from gaussian_process_transportation_amd import _lib
    from oracle import gp_oracle as orc
    rng = np.random.default_rng(N)
    X = rng.uniform(0, 1, (N, 3))
    Y = np.sin(3 * X[:, :2])
    ls, c, noise, jit = np.array([0.25, 0.3, 0.2]), 0.7, 1e-3, 1e-10
    Kref = c * orc.rbf_gram(X / ls) + (noise + jit) * np.eye(N)
    Lref = np.linalg.cholesky(Kref)
    aref = scipy.linalg.cho_solve((Lref, True), Y)
    NP = (N + 511) // 512 * 512
    h = _lib.Handle(0)
    Sigma = 1e-3 * np.eye(N)
    Sigma[N - 300, N - 300] = -2.0
    #        form  panel   overlap  potrf group
    cases = [("1", None, "1", None), ("1", None, "1", "4"), ("0", None, "1", None), ("2", "512", "1", None), ("2", "2048", "0", None),
             ("2", "1024", "1", None), (None, None, "1", None)]
    for form, panel, overlap, group in cases:
        for k, v in (("GPT_FIT_FORM", form), ("GPT_FIT_PANEL", panel), ("GPT_FIT_OVERLAP", overlap), ("GPT_POTRF_GROUP", group)):
            if v is None:
                monkeypatch.delenv(k, raising=False)
            else:
                monkeypatch.setenv(k, v)
        pl = _lib.debug_fit_plan(NP)
        assert pl["form"] == int(form or 1)
        if group == "4" and N == 5300:
            assert [int(o[2]) for o in pl["ops"] if _lib.FIT_OP_KINDS[int(o[0])] == "T"] == [2560]      # the split is off the half
        what = f"form {form}, panel {panel}, streams {overlap}, group {group}"
        h.fit(X, Y, ls, c, noise, jit)
        L, alpha = h.export()
        assert_parity(L, Lref, 1e-11, f"L_ ({what})")
        assert_parity(alpha, aref, 1e-7, f"alpha_ ({what})")
        W = h.export_inverse_factor()
        assert np.abs(W @ Lref - np.eye(N)).max() < 1e-9, what
        h.fit(X, Y, ls, c, noise, jit)                       # streams and events are reused by the next fit
        assert np.array_equal(h.export()[0], L), what
        with pytest.raises(np.linalg.LinAlgError) as ei:
            h.fit_noise_matrix(X, Y, ls, c, Sigma)
        assert int(str(ei.value).split("pivot")[1].split()[0]) == N - 300 + 1, what
    h.close()


def test_quarter_tile_cuts_change_no_result_beyond_rounding(monkeypatch):
    """The tail's work split cuts sweeps at quarter tiles since round 4 (gpt_plan.h); GPT_VAR_CUT_TILES=1 restores whole tiles.
    Same queries, both splits, the shapes whose whole batch is a tail: variances agree to rounding (the order of the partial
    products' sum changes), and both agree with the oracle."""
    from gaussian_process_transportation_amd import _lib
    from oracle import gp_oracle as orc
    for N, M in ((1024, 4096), (2500, 4096), (2500, 900)):
        X, Y, Xq = orc.synthetic_problem(N, M)
        c, ls, noise, jit = 0.1, np.array([0.1, 0.12, 0.09]), 1e-4, 1e-10
        outs = []
        for tiles in ("0", "1"):
            monkeypatch.setenv("GPT_VAR_CUT_TILES", tiles)
            h = _lib.Handle(0)                                  # (a handle caches the plan of its last launch shape)
            h.fit(X, Y, ls, c, noise, jit)
            outs.append(h.predict_all(Xq, var=True, Jvar=True, dvar=True))
            h.close()
        for k in ("var", "Jvar", "dvar"):
            assert_parity(outs[0][k], outs[1][k], 1e-10, f"{k}: quarter-tile vs tile cuts (N={N}, M={M})")
        idx = np.arange(0, M, 7)
        L, a = orc.gpr_fit(X, Y, c, ls, noise, jit)
        _, var, _, Jvar = orc.posterior_all_fast(Xq[idx], X, L, a, c, ls, noise, want_jvar=True)
        assert_parity(outs[0]["var"][idx], var, RTOL, "var vs oracle")
        assert_parity(outs[0]["Jvar"][idx], Jvar, RTOL, "Jvar vs oracle")


def test_divided_diagonal_tiles_change_no_result_beyond_rounding(monkeypatch):
    """Lists whose shares are small cut the diagonal tile at quarters too (gpt_plan.h cut_diag; configs[1]'s 14-block tail): an item
    then runs k-steps [d_lo, d_hi) of the tile, a wave the part of its 16 (g + 1) steps inside.  GPT_VAR_CUT_DIAG=0 keeps the tile
    whole, =1 divides it wherever the equal-cost cut falls.  Every path of the tile: HALF (N <= 2560, one and several i-blocks, rounds +
    tail), the ring from the scratch image (N = 3000), the 3-column kernel (ring of 2), the wide layout, fp64 multi-task, and fp32 through
    LDS.  Both splits agree to rounding and with the oracle."""
    from gaussian_process_transportation_amd import _lib
    from oracle import gp_oracle as orc
    rng = np.random.default_rng(12)

    def both(make, Xq, tol, **flags):
        outs = []
        for v in ("0", "1"):
            monkeypatch.setenv("GPT_VAR_CUT_DIAG", v)
            h = make()                                          # (a handle caches the plan of its last launch shape)
            outs.append(h.predict_all(Xq, **flags))
            h.close()
        for k, a in outs[0].items():
            if a is not None:
                assert_parity(outs[1][k], a, tol, f"{k}: divided vs whole diagonal tiles {flags}")
        return outs[1]

    for N, M in ((1024, 50_000), (1024, 4096), (300, 700), (3000, 1500), (2500, 460)):
        X, Y, Xq = orc.synthetic_problem(N, M)
        c, ls, noise, jit = 0.1, np.array([0.1, 0.12, 0.09]), 1e-4, 1e-10

        def make():
            h = _lib.Handle(0)
            h.fit(X, Y, ls, c, noise, jit)
            return h
        o1 = both(make, Xq, 1e-10, mean=True, var=True)
        o4 = both(make, Xq, 1e-10, var=True, Jvar=True, dvar=True)
        o3 = both(make, Xq, 1e-10, Jvar=True)
        idx = np.arange(0, M, max(1, M // 300))
        L, a = orc.gpr_fit(X, Y, c, ls, noise, jit)
        _, var, _, Jvar = orc.posterior_all_fast(Xq[idx], X, L, a, c, ls, noise, want_jvar=True)
        assert_parity(o1["var"][idx], var, RTOL, f"var vs oracle (N={N}, M={M})")
        assert_parity(o4["Jvar"][idx], Jvar, RTOL, f"Jvar vs oracle (N={N}, M={M})")
        assert_parity(o3["Jvar"][idx], Jvar, RTOL, f"Jvar alone vs oracle (N={N}, M={M})")
    # wide layout
    X = rng.uniform(0, 1, (900, 6)); Y = np.sin(X[:, :2].sum(1, keepdims=True)); Xq = rng.uniform(0, 1, (1500, 6))

    def make_wide():
        h = _lib.Handle(0)
        h.fit(X, Y, np.full(6, 0.6), 0.5, 1e-3, 1e-10)
        return h
    both(make_wide, Xq, 1e-10, var=True, Jvar=True, dvar=True)
    # multi-task, fp64 and fp32 (the fp32 tile reads its image from LDS)
    Z, T, D = 1100, 3, 3
    Zp = rng.uniform(0, 1, (Z, D)); A = rng.standard_normal((T, Z, Z)); Sigma = A @ A.transpose(0, 2, 1) / Z + 1e-3 * np.eye(Z)
    y = rng.standard_normal((T, Z))
    for dtype, tol in ((_lib.GPT_F64, 1e-10), (_lib.GPT_F32, 2e-4)):
        def make_svgp():
            h = _lib.Handle(0)
            h.fit_svgp(Zp, y, Sigma, np.full(D, 0.2), np.ones(T), dtype=dtype)
            return h
        both(make_svgp, rng.uniform(0, 1, (2000, D)), tol, mean=True, var=True)


def test_half_image_diagonal_tiles_change_no_bit(monkeypatch):
    """Small models (N <= 2560) run k_var's HALF instantiation: the first 64 k-steps of a diagonal tile's B image in LDS
    (generated there by the generating sweep, copied there by the reload sweeps), the rest from the scratch image.  Every k-step
    enters the same accumulator in the same order, so the results must be IDENTICAL to the instantiation that reads the whole
    tile from the scratch image (GPT_VAR_DIAG_HALF=0) — whole rounds, tails, one i-block, the 4-column kernels, the wide layout and the
    fp64 multi-task model (whose top diagonal tile IS reloaded by the other tasks)."""
    from gaussian_process_transportation_amd import _lib
    from oracle import gp_oracle as orc
    rng = np.random.default_rng(11)

    def both(h, Xq, **flags):
        outs = []
        for v in ("1", "0"):
            monkeypatch.setenv("GPT_VAR_DIAG_HALF", v)
            outs.append(h.predict_all(Xq, **flags))
        for k, a in outs[0].items():
            if a is not None:
                assert np.array_equal(a, outs[1][k]), (k, flags)
        return outs[0]

    for N, M in ((300, 700), (1024, 20_000), (2500, 4096), (2500, 17_000), (700, 64)):
        X, Y, Xq = orc.synthetic_problem(N, M)
        c, ls, noise, jit = 0.1, np.array([0.1, 0.12, 0.09]), 1e-4, 1e-10
        h = _lib.Handle(0)
        h.fit(X, Y, ls, c, noise, jit)
        o1 = both(h, Xq, mean=True, var=True, J=True)
        o4 = both(h, Xq, var=True, Jvar=True, dvar=True)
        both(h, Xq, Jvar=True)
        idx = np.arange(0, M, max(1, M // 400))
        L, a = orc.gpr_fit(X, Y, c, ls, noise, jit)
        _, var, _, Jvar = orc.posterior_all_fast(Xq[idx], X, L, a, c, ls, noise, want_jvar=True)
        assert_parity(o1["var"][idx], var, RTOL, f"var vs oracle (N={N}, M={M})")
        assert_parity(o4["Jvar"][idx], Jvar, RTOL, f"Jvar vs oracle (N={N}, M={M})")
        h.close()
    # wide layout (D = 6) and Matern
    X = rng.uniform(0, 1, (900, 6)); Y = np.sin(X[:, :2].sum(1, keepdims=True)); Xq = rng.uniform(0, 1, (3000, 6))
    h = _lib.Handle(0)
    h.fit(X, Y, np.full(6, 0.6), 0.5, 1e-3, 1e-10)
    both(h, Xq, var=True, Jvar=True, dvar=True)
    h.fit(X, Y, np.full(6, 0.6), 0.5, 1e-3, 1e-10, 2)
    both(h, Xq, mean=True, var=True)
    h.close()
    # fp64 multi-task model: the other tasks reload the top diagonal tile's k-steps
    Z, T, D = 1100, 3, 3
    Zp = rng.uniform(0, 1, (Z, D)); A = rng.standard_normal((T, Z, Z)); Sigma = A @ A.transpose(0, 2, 1) / Z + 1e-3 * np.eye(Z)
    y = rng.standard_normal((T, Z))
    h = _lib.Handle(0)
    h.fit_svgp(Zp, y, Sigma, np.full(D, 0.2), np.ones(T), dtype=_lib.GPT_F64)
    both(h, rng.uniform(0, 1, (9000, D)), mean=True, var=True, J=True, Jvar=True)
    h.close()
