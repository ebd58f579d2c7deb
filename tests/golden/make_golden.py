"""Generate golden vectors by running the REFERENCE itself (authoring container only).

    python tests/golden/make_golden.py [case ...]      # default: all fast cases
    python tests/golden/make_golden.py surface3d n8192 # slow cases (minutes each)

The reference (/root/reference, scikit-learn 1.7.2 in this image) is imported with
an empty stub for its missing third-party `Quaternion` module
(policy_transportation.py:9).  Only inputs and outputs are written (npz, no
pickles); the reference's sources are never copied.  The GPU box has no
/root/reference: tests read the committed npz files only.
"""
import os
import sys
import types
import warnings

os.environ.setdefault("MPLBACKEND", "Agg")
import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def import_reference():
    sys.modules.setdefault("Quaternion", types.ModuleType("Quaternion"))
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import policy_transportation as pt  # noqa
    from policy_transportation.utils import resample
    return pt, resample


def kern(c, ls, noise):
    from sklearn.gaussian_process.kernels import RBF, WhiteKernel, ConstantKernel as C
    return C(c) * RBF(length_scale=ls) + WhiteKernel(noise)


def theta_of(gp):
    p = gp.kernel.get_params()
    return dict(constant_value=np.float64(p["k1__k1__constant_value"]),
                length_scale=np.atleast_1d(np.asarray(p["k1__k2__length_scale"], np.float64)),
                noise_level=np.float64(p["k2__noise_level"]))


def synthetic(N, M, seed=0, qseed=1, D=3):
    rng = np.random.default_rng(seed)
    X = rng.uniform(0, 1, (N, D))
    Y = 0.05 * np.sin(4 * X) + 0.01 * rng.standard_normal((N, D))
    Xq = np.random.default_rng(qseed).uniform(-0.1, 1.1, (M, D))
    return X, Y, Xq


def gp_outputs(gp, Xq, with_L=True, with_cov=0, with_jvar=True):
    out = {}
    out["alpha_"] = gp.gp.alpha_
    if with_L:
        out["L_"] = gp.gp.L_
    else:
        out["L_diag"] = np.diag(gp.gp.L_).copy()
        out["L_col0"] = gp.gp.L_[:, 0].copy()
        out["L_lastrow"] = gp.gp.L_[-1, :].copy()
    out["mean_only"] = gp.predict(Xq)
    m, s = gp.predict(Xq, return_std=True)
    out["mean"], out["std"] = m, s
    if with_jvar:
        J, Jv = gp.derivative(Xq, return_var=True)
        out["J"], out["Jvar"] = J, Jv
        out["dvar"] = gp.derivative_of_variance(Xq)
    else:
        out["J"] = gp.derivative(Xq)
    if with_cov:
        _, cov = gp.predict(Xq[:with_cov], return_cov=True)
        out["cov"] = cov
        out["samples"] = gp.samples(Xq[:with_cov])          # (10, with_cov, O), RandomState(0) inside sklearn
    out["noise_var_"] = np.float64(gp.noise_var_)
    out["prior_var"] = np.float64(gp.prior_var)
    return out


def case_synthetic(pt, N, M, name, ls=(0.1, 0.1, 0.1), nan_rows=(), with_L=True, with_cov=0,
                   with_jvar=True, lml=True, D=3):
    X, Y, Xq = synthetic(N, M, D=D)
    Y = Y.copy()
    for r in nan_rows:
        Y[r, r % 3] = np.nan
    gp = pt.GaussianProcess(kernel=kern(0.1, list(ls), 1e-4), optimizer=None)
    gp.fit(X, Y)
    out = dict(X=X, Y=Y, Xq=Xq, alpha=np.float64(1e-10), **theta_of(gp))
    out.update(gp_outputs(gp, Xq, with_L=with_L, with_cov=with_cov, with_jvar=with_jvar))
    out["n_samples"] = np.int64(gp.n_samples)
    if lml and not nan_rows:
        thetas = []
        vals = []
        grads = []
        base = gp.gp.kernel_.theta.copy()
        rs = np.random.default_rng(7)
        for k in range(3):
            th = base + (0.0 if k == 0 else 1.0) * rs.normal(0, 0.3, base.shape)
            v, g = gp.gp.log_marginal_likelihood(th, eval_gradient=True)
            thetas.append(th); vals.append(v); grads.append(g)
        out["lml_theta"] = np.array(thetas)
        out["lml_value"] = np.array(vals)
        out["lml_grad"] = np.array(grads)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, {k: np.shape(v) for k, v in out.items()})


def case_letterS(pt, resample):
    data = np.load(os.path.join(REF, "example/2D/data/example.npz"))
    X = resample(data["demo"], num_points=400)
    src = resample(data["floor"], num_points=20)
    tgt = resample(data["newfloor"], num_points=20)
    dX = np.zeros((len(X), 2))
    dX[:-1] = X[1:] - X[:-1]
    np.random.seed(0)            # sklearn's optimizer restarts use the global RNG (_gpr.py:248,319-330)
    tr = pt.GaussianProcessTransportation(kernel_transport=kern(10, 4 * np.ones(2), 0.01))
    tr.source_distribution = src
    tr.target_distribution = tgt
    tr.training_traj = X
    tr.training_delta = dX
    tr.fit_transportation(do_scale=False, do_rotation=True)
    tr.apply_transportation()
    gp = tr.method.delta_map
    out = dict(demo_raw=data["demo"], floor_raw=data["floor"], newfloor_raw=data["newfloor"],
               demo=X, source=src, target=tgt, delta=dX,
               rotation=tr.method.affine_transform.rotation_matrix,
               scale=np.float64(tr.method.affine_transform.scale),
               S_centroid=tr.method.affine_transform.S_centroid,
               T_centroid=tr.method.affine_transform.T_centroid,
               traj=tr.training_traj, std=tr.std, vel=tr.training_delta,
               var_vel=tr.var_vel_transported, alpha_=gp.gp.alpha_, L_=gp.gp.L_,
               gp_X=gp.X, gp_Y=gp.Y, theta0=gp.gp.kernel.theta, theta_fit=gp.gp.kernel_.theta,
               bounds=gp.gp.kernel.bounds, lml_fit=np.float64(gp.gp.log_marginal_likelihood_value_),
               noise_var_=np.float64(gp.noise_var_), alpha=np.float64(1e-10), **theta_of(gp))
    smp_all = tr.sample_transportation()                     # (10, 400, 2) = rotated positions + sklearn sample_y draws
    out["samples"] = smp_all[:, ::8, :]                      # (10, 50, 2): every 8th trajectory point of the 400
    out["samples_full_shape"] = np.array(smp_all.shape)
    # Evidence for how tightly draws CAN be compared (the posterior covariance of 400 closely spaced points has a large
    # eigenspace degenerate at ~noise_level, where the SVD basis multivariate_normal uses is arbitrary):
    #  (1) the draws projected on the leading, well-separated eigenvectors of the reference's own covariance — these
    #      coordinates are basis-independent, an implementation must reproduce them closely;
    #  (2) how far the reference's draws move when the SAME covariance is merely recomputed in another order of
    #      operations (K** - K*^T K^-1 K* by cho_solve instead of sklearn's V^T V; fp64 both): the spread any
    #      faithful implementation shows in the degenerate directions.
    pos_rot = tr.method.affine_transform.predict(X)
    mean_ref, cov_ref = gp.predict(pos_rot, return_cov=True)
    cov0 = cov_ref[:, :, 0]
    lam, V = np.linalg.eigh(cov0)
    lam, V = lam[::-1], V[:, ::-1]
    gaps = (lam[:-1] - lam[1:]) / lam[0]
    k = 1
    while k < 40 and lam[k] > 50 * np.median(lam) and gaps[k - 1] > 1e-6:
        k += 1
    Vk = V[:, :k]
    out["samples_eig_vectors"] = Vk                                       # (400, k)
    out["samples_eig_values"] = lam[:k]
    out["samples_eig_coords"] = np.einsum("nk,snt->skt", Vk, smp_all - (pos_rot + mean_ref)[None])    # (10, k, 2)
    import scipy.linalg
    Kss = gp.kernel(pos_rot)
    Ks = gp.kernel(gp.X, pos_rot)
    cov_alt = Kss - Ks.T @ scipy.linalg.cho_solve((gp.gp.L_, True), Ks)   # K** - K*^T K^-1 K*: same matrix, other roundings
    rng = np.random.RandomState(0)
    alt = np.stack([rng.multivariate_normal(mean_ref[:, t], cov_alt, 10) for t in range(mean_ref.shape[1])], axis=2)
    out["samples_recomputed_spread"] = np.float64(np.max(np.abs((pos_rot[None] + alt) - smp_all)))
    out["samples_cov_recomputed_diff"] = np.float64(np.max(np.abs(cov_alt - cov0)))
    out["samples_sd_max"] = np.float64(np.sqrt(np.max(np.diag(cov0))))
    # second transport with do_scale=True, fixed hyper-parameters (affine scale branch)
    np.random.seed(0)
    tr2 = pt.GaussianProcessTransportation(kernel_transport=kern(out["constant_value"],
                                                                 out["length_scale"], out["noise_level"]))
    tr2.method.delta_map = pt.GaussianProcess(kernel=kern(out["constant_value"], out["length_scale"],
                                                          out["noise_level"]), optimizer=None)
    tr2.source_distribution = src
    tr2.target_distribution = tgt
    tr2.training_traj = X
    tr2.training_delta = dX
    tr2.fit_transportation(do_scale=True, do_rotation=True)
    tr2.apply_transportation()
    out.update(scale2=np.float64(tr2.method.affine_transform.scale), traj2=tr2.training_traj,
               std2=tr2.std, vel2=tr2.training_delta, var_vel2=tr2.var_vel_transported)
    np.savez_compressed(os.path.join(HERE, "letterS_2d.npz"), **out)
    print("wrote letterS_2d", out["length_scale"], out["constant_value"], out["noise_var_"])


def case_matern(pt, resample):
    """The examples' dynamics GP (example/2D/surface_generalization.py:49-51: C(sqrt 0.1) * Matern(1, nu=2.5) +
    White(0.01) on the resampled letter-S demo, optimizer on) and fixed-theta fits for nu = 0.5 / 1.5 / 2.5."""
    from sklearn.gaussian_process.kernels import Matern, WhiteKernel, ConstantKernel as C
    data = np.load(os.path.join(REF, "example/2D/data/example.npz"))
    X = resample(data["demo"], num_points=400)
    dX = np.zeros((len(X), 2)); dX[:-1] = X[1:] - X[:-1]
    xg, yg = np.meshgrid(np.linspace(X[:, 0].min() - 10, X[:, 0].max() + 10, 15),
                         np.linspace(X[:, 1].min() - 10, X[:, 1].max() + 10, 15))
    grid = np.column_stack([xg.ravel(), yg.ravel()])
    out = dict(X=X, Y=dX, grid=grid, alpha=np.float64(1e-10))
    np.random.seed(0)
    gp = pt.GaussianProcess(kernel=C(constant_value=np.sqrt(0.1)) * Matern(1 * np.ones(2), nu=2.5) + WhiteKernel(0.01))
    gp.fit(X, dX)
    m, s = gp.predict(grid, return_std=True)
    out.update(opt_theta0=gp.gp.kernel.theta, opt_theta=gp.gp.kernel_.theta, opt_lml=np.float64(gp.gp.log_marginal_likelihood_value_),
               opt_mean=m, opt_std=s)
    rs = np.random.default_rng(5)
    for nu, tag in ((0.5, "12"), (1.5, "32"), (2.5, "52")):
        k = C(0.3) * Matern([1.5, 2.5], nu=nu) + WhiteKernel(0.01)
        g2 = pt.GaussianProcess(kernel=k, optimizer=None)
        g2.fit(X, dX)
        m, s = g2.predict(grid, return_std=True)
        _, cov = g2.predict(grid[:12], return_cov=True)
        ths, vals, grads = [], [], []
        base = g2.gp.kernel_.theta.copy()
        for j in range(2):
            th = base + j * rs.normal(0, 0.3, base.shape)
            v, gr = g2.gp.log_marginal_likelihood(th, eval_gradient=True)
            ths.append(th); vals.append(v); grads.append(gr)
        out.update({f"m{tag}_mean": m, f"m{tag}_std": s, f"m{tag}_cov": cov, f"m{tag}_alpha_": g2.gp.alpha_,
                    f"m{tag}_Ldiag": np.diag(g2.gp.L_).copy(), f"m{tag}_lml_theta": np.array(ths),
                    f"m{tag}_lml_value": np.array(vals), f"m{tag}_lml_grad": np.array(grads)})
        # isotropic variant gradient
        ki = C(0.3) * Matern(2.0, nu=nu) + WhiteKernel(0.01)
        g3 = pt.GaussianProcess(kernel=ki, optimizer=None)
        g3.fit(X[::4], dX[::4])
        v, gr = g3.gp.log_marginal_likelihood(g3.gp.kernel_.theta + 0.1, eval_gradient=True)
        out.update({f"m{tag}_iso_theta": g3.gp.kernel_.theta + 0.1, f"m{tag}_iso_value": np.float64(v), f"m{tag}_iso_grad": gr})
    np.savez_compressed(os.path.join(HERE, "matern_2d.npz"), **out)
    print("wrote matern_2d", np.exp(out["opt_theta"]), out["opt_lml"])


def case_matern5d(pt):
    """Matern-nu on a 5-D input (the wide source layout of the HIP path, rows of 8): the reference class with
    optimizer=None for nu = 0.5 / 1.5 / 2.5, ARD and isotropic length-scales, N = 300 (off every tile boundary):
    alpha_, diag L, mean, std, covariance, LML value + gradient at the kernel's theta and at a perturbed one."""
    from sklearn.gaussian_process.kernels import Matern, WhiteKernel, ConstantKernel as C
    rng = np.random.default_rng(21)
    N, M, D, O = 300, 90, 5, 2
    X = rng.uniform(0, 1, (N, D))
    Y = np.column_stack([np.sin(3 * X[:, 0]) * X[:, 3], np.cos(2 * X[:, 1] + X[:, 4])]) + 0.02 * rng.standard_normal((N, O))
    Xq = rng.uniform(-0.1, 1.1, (M, D))
    ls = np.array([0.6, 0.9, 0.7, 1.1, 0.8])
    out = dict(X=X, Y=Y, Xq=Xq, alpha=np.float64(1e-10), constant_value=np.float64(0.3), length_scale=ls,
               noise_level=np.float64(0.01), iso_length_scale=np.float64(0.8))
    rs = np.random.default_rng(6)
    for nu, tag in ((0.5, "12"), (1.5, "32"), (2.5, "52")):
        for kind, lsk in (("ard", ls), ("iso", 0.8)):
            gp = pt.GaussianProcess(kernel=C(0.3) * Matern(lsk, nu=nu) + WhiteKernel(0.01), optimizer=None)
            gp.fit(X, Y)
            m, s = gp.predict(Xq, return_std=True)
            _, cov = gp.predict(Xq[:12], return_cov=True)
            ths, vals, grads = [], [], []
            base = gp.gp.kernel_.theta.copy()
            for j in range(2):
                th = base + j * rs.normal(0, 0.3, base.shape)
                v, gr = gp.gp.log_marginal_likelihood(th, eval_gradient=True)
                ths.append(th); vals.append(v); grads.append(gr)
            pre = f"m{tag}_{kind}_"
            out.update({pre + "mean": m, pre + "std": s, pre + "cov": cov, pre + "alpha_": gp.gp.alpha_,
                        pre + "Ldiag": np.diag(gp.gp.L_).copy(), pre + "lml_theta": np.array(ths),
                        pre + "lml_value": np.array(vals), pre + "lml_grad": np.array(grads)})
    np.savez_compressed(os.path.join(HERE, "matern_5d.npz"), **out)
    print("wrote matern_5d", {k: np.shape(v) for k, v in out.items() if k.startswith("m52_ard")})


def case_robot_demo():
    """The recorded robot demonstration the reference ships (data/last.npz, loaded with allow_pickle=False: plain float
    arrays): positions (102,3), orientations as quaternions (102,4) and velocities (102,3) — the inputs of
    transport_orientation (policy_transportation.py:61-77) in the reference's own robot use.  Data only; the source /
    target distributions of that experiment are pickles (distributions/*.pkl) and are NOT loaded."""
    z = np.load(os.path.join(REF, "data/last.npz"), allow_pickle=False)
    out = {k: z[k] for k in ("training_traj", "training_ori", "training_delta")}
    np.savez_compressed(os.path.join(HERE, "robot_demo_last.npz"), **out)
    print("wrote robot_demo_last", {k: v.shape for k, v in out.items()})


def case_surface3d(pt, optimize=True):
    """example/3D/surface_generalization_3D.py:50-61 flow (N=2500, default kernel).
    optimizer on takes ~5 min here; the fitted theta is stored and outputs come from
    a refit with optimizer=None at that theta (identical L_/alpha_ by construction)."""
    data = np.load(os.path.join(REF, "example/3D/data/example.npz"))
    X = data["demo"]
    src = data["old_surface"].reshape(-1, 3)
    tgt = data["new_surface"].reshape(-1, 3)
    dX = np.zeros((len(X), 3))
    dX[:-1] = X[1:] - X[:-1]
    np.random.seed(0)
    tr = pt.GaussianProcessTransportation()
    if not optimize:
        tr.method.delta_map = pt.GaussianProcess(kernel=kern(0.1, [0.1], 1e-4), optimizer=None)
    tr.source_distribution = src
    tr.target_distribution = tgt
    tr.training_traj = X
    tr.training_delta = dX
    tr.fit_transportation()
    tr.apply_transportation()
    gp = tr.method.delta_map
    out = dict(demo=X, source=src, target=tgt, delta=dX,
               rotation=tr.method.affine_transform.rotation_matrix,
               traj=tr.training_traj, std=tr.std, vel=tr.training_delta,
               var_vel=tr.var_vel_transported, alpha_=gp.gp.alpha_,
               theta_fit=gp.gp.kernel_.theta, lml_fit=np.float64(gp.gp.log_marginal_likelihood_value_),
               noise_var_=np.float64(gp.noise_var_), alpha=np.float64(1e-10),
               optimized=np.bool_(optimize), **theta_of(gp))
    np.savez_compressed(os.path.join(HERE, "surface_3d.npz"), **out)
    print("wrote surface_3d", out["length_scale"], out["constant_value"], out["noise_level"])


def case_n8192(pt):
    """N=8192 spot check, 256 queries (reference: ~30 s fit + minutes for Jvar)."""
    X, Y, Xq = synthetic(8192, 256)
    gp = pt.GaussianProcess(kernel=kern(0.1, [0.1] * 3, 1e-4), optimizer=None)
    gp.fit(X, Y)
    m, s = gp.predict(Xq, return_std=True)
    J, Jv = gp.derivative(Xq, return_var=True)
    g = gp.derivative_of_variance(Xq)
    L = gp.gp.L_
    out = dict(N=np.int64(8192), Xq=Xq, mean=m, std=s, J=J, Jvar=Jv, dvar=g,
               alpha_=gp.gp.alpha_, L_diag=np.diag(L).copy(), L_col0=L[:, 0].copy(),
               L_lastrow=L[-1].copy(), alpha=np.float64(1e-10), **theta_of(gp))
    np.savez_compressed(os.path.join(HERE, "synthetic_3d_N8192.npz"), **out)
    print("wrote synthetic_3d_N8192")


def main(argv):
    warnings.filterwarnings("ignore")
    pt, resample = import_reference()
    cases = argv or ["n64", "n64iso", "n64nan", "n256", "n1024", "n200d5", "n128d8", "n160d12", "n96d15", "letterS", "matern", "matern5d", "robot"]
    for c in cases:
        if c == "n64":
            case_synthetic(pt, 64, 48, "synthetic_3d_N64", with_cov=16)
        elif c == "n64iso":
            case_synthetic(pt, 64, 48, "synthetic_3d_N64_iso", ls=(0.15,), with_cov=8)
        elif c == "n64nan":
            case_synthetic(pt, 64, 48, "synthetic_3d_N64_nan", nan_rows=(3, 17, 40))
        elif c == "n256":
            case_synthetic(pt, 256, 96, "synthetic_3d_N256")
        elif c == "n1024":
            case_synthetic(pt, 1024, 128, "synthetic_3d_N1024", with_L=False)
        elif c == "n200d5":     # input dimension beyond 3: the wide layout of the HIP path (ARD length-scales, 5 outputs)
            case_synthetic(pt, 200, 80, "synthetic_5d_N200", ls=(0.3, 0.5, 0.4, 0.6, 0.35), with_cov=12, D=5)
        elif c == "n128d8":
            case_synthetic(pt, 128, 40, "synthetic_8d_N128", ls=(0.7,), with_cov=8, D=8)
        elif c == "n160d12":    # input dimension beyond 8: source rows of 16 (ARD length-scales, 12 outputs = three passes of 4)
            case_synthetic(pt, 160, 60, "synthetic_12d_N160", ls=tuple(0.8 + 0.05 * d for d in range(12)), with_cov=8, D=12)
        elif c == "n96d15":     # the largest dimension the HIP path takes: k* and 15 derivative columns fill one 16-column tile
            case_synthetic(pt, 96, 40, "synthetic_15d_N96", ls=(1.1,), with_cov=8, D=15)
        elif c == "letterS":
            case_letterS(pt, resample)
        elif c == "matern":
            case_matern(pt, resample)
        elif c == "matern5d":
            case_matern5d(pt)
        elif c == "robot":
            case_robot_demo()
        elif c == "surface3d":
            case_surface3d(pt, optimize=True)
        elif c == "n8192":
            case_n8192(pt)
        else:
            raise SystemExit("unknown case " + c)


if __name__ == "__main__":
    main(sys.argv[1:])
