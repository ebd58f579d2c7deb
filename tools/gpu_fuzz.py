"""Randomised parity sweep of the HIP path against the CPU oracle (run on the GPU box; development aid, not a test):
random N, M, D, O, kernel family, length-scales (isotropic / ARD), amplitudes, noise levels, query ranges.
    python tools/gpu_fuzz.py [cases] [seed]
Prints the worst relative error per quantity and every case that exceeds 1e-6 (the judged bound is 1e-5)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gaussian_process_transportation_amd import _lib  # noqa: E402
from oracle import gp_oracle as orc  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)


def rel(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return float(np.max(np.abs(a - b)) / max(float(np.max(np.abs(b))), 1e-300)) if b.size else 0.0


worst = {}
bad = 0
h = _lib.Handle(0)
for it in range(cases):
    N = int(rng.choice([1, 2, 5, 17, 63, 64, 65, 130, 257, 511, 513, 777, 1025, 1500, 2100, 3000]))
    M = int(rng.choice([1, 7, 64, 65, 300, 1000, 4097, 17000, 40000]))      # cut sweeps only / whole rounds + tail
    D = int(rng.integers(1, 16)) if rng.integers(0, 3) else int(rng.integers(1, 4))      # 4 .. 8 and 9 .. 15: the wide layouts
    O = int(rng.integers(1, 7))
    kind = str(rng.choice(["rbf", "rbf", "rbf", "matern12", "matern32", "matern52"]))
    iso = bool(rng.integers(0, 2))
    ls = np.exp(rng.uniform(np.log(0.03), np.log(2.0), 1 if iso else D)) * (1.0 if D <= 3 else 2.0 * np.sqrt(D / 3.0))
    c = float(np.exp(rng.uniform(np.log(1e-2), np.log(30.0))))
    noise = float(np.exp(rng.uniform(np.log(1e-6), np.log(1e-1)))) * c
    jit = 1e-10
    span = float(rng.choice([1.0, 1.0, 10.0]))
    X = rng.uniform(0, span, (N, D))
    Y = rng.standard_normal((N, O)) * np.sqrt(c) * 0.3
    Xq = rng.uniform(-0.2 * span, 1.2 * span, (M, D))
    code = {"rbf": 0, "matern12": 1, "matern32": 2, "matern52": 3}[kind]
    tag = f"case {it}: N={N} M={M} D={D} O={O} {kind} ls={np.round(ls * span, 3)} c={c:.3g} noise={noise:.3g} span={span}"
    try:
        h.fit(X, Y, ls * span, c, noise, jit, code)
    except np.linalg.LinAlgError:
        try:
            orc.GaussianProcessOracle(c, ls * span, noise, jit, kind=kind).fit(X, Y)
            print("MISMATCH (GPU not PD, oracle PD):", tag); bad += 1
        except np.linalg.LinAlgError:
            pass
        continue
    o = orc.GaussianProcessOracle(c, ls * span, noise, jit, kind=kind).fit(X, Y)
    errs = {}
    L, a = h.export()
    errs["L"] = rel(L, o.L_)
    errs["alpha"] = rel(a, o.alpha_) / max(1.0, np.linalg.cond(o.L_) ** 2 * 1e-10)   # alpha is as ill-conditioned as K
    want_der = kind == "rbf"
    out = h.predict_all(Xq, mean=True, var=True, J=want_der, Jvar=want_der, dvar=want_der)
    alone_all = h.predict_all(Xq, J=True, Jvar=True)["Jvar"] if want_der else None
    if M > 1500:            # the GPU predicts the whole batch; the (slow) CPU oracle checks a random subset of it
        sub = np.sort(rng.choice(M, 1500, replace=False))
        Xq = Xq[sub]
        out = {k: (v[:, sub] if k == "dvar" else v[sub]) for k, v in out.items() if v is not None}
        alone_all = alone_all[sub] if alone_all is not None else None
        M = 1500
    mean, std = o.predict(Xq, return_std=True)
    std = std if std.ndim == 1 else std[:, 0]
    var = (std + np.sqrt(noise)) ** 2
    errs["mean"] = rel(out["mean"], np.reshape(mean, (M, O))) / max(1.0, np.linalg.cond(o.L_) ** 2 * 1e-10)
    errs["var"] = float(np.max(np.abs(out["var"] - var)) / (c + noise))       # absolute against the prior variance (cancellation)
    if want_der:
        J, Jv = o.derivative(Xq, return_var=True)
        errs["J"] = rel(out["J"], J) / max(1.0, np.linalg.cond(o.L_) ** 2 * 1e-10)
        scale = float(np.max(c / (ls * span) ** 2))
        errs["Jvar"] = float(np.max(np.abs(out["Jvar"] - Jv[:, 0, :])) / scale)
        errs["dvar"] = float(np.max(np.abs(out["dvar"] - o.derivative_of_variance(Xq))) / (scale ** 0.5 * (c + noise) ** 0.5 * 2))
    if want_der:                      # Jacobian variance alone: the D-columns-per-query kernel
        errs["Jvar_alone"] = float(np.max(np.abs(alone_all - Jv[:, 0, :])) / scale)
    for k, v in errs.items():
        worst[k] = max(worst.get(k, 0.0), v)
    print(f"... case {it} done ({tag[:60]})", flush=True)
    if max(errs.values()) > 1e-6 or not all(np.isfinite(list(errs.values()))):
        bad += 1
        print("LARGE:", tag, {k: f"{v:.2e}" for k, v in errs.items()}, f"cond(K)~{np.linalg.cond(o.L_) ** 2:.2e}", flush=True)
h.close()
print(f"{cases} cases, {bad} flagged; worst errors:", {k: f"{v:.2e}" for k, v in worst.items()})

# ---- multi-task (SVGP exact conversion) models, fp64 and fp32, against the CPU restatement; fp32 exact GP against the fp64 one
worst2, bad2 = {}, 0
for it in range(max(cases // 2, 6)):
    Z = int(rng.choice([3, 64, 130, 511, 513, 900, 1500, 2048]))
    T = int(rng.integers(1, 6))
    D = int(rng.integers(1, 9))
    M = int(rng.choice([1, 65, 460, 3000, 20000]))
    dtype = int(rng.integers(0, 2))
    Zp = rng.uniform(0, 1, (Z, D))
    A = rng.standard_normal((T, Z, Z))
    Sigma = A @ A.transpose(0, 2, 1) / Z * float(np.exp(rng.uniform(np.log(1e-3), 0))) + 1e-3 * np.eye(Z)
    y = rng.standard_normal((T, Z))
    osc = np.exp(rng.uniform(np.log(0.1), np.log(5.0), T))
    ls = np.exp(rng.uniform(np.log(0.08), np.log(1.0), D)) * (1.0 if D <= 3 else 2.0 * np.sqrt(D / 3.0))
    Xq = rng.uniform(-0.1, 1.1, (M, D))
    tag = f"svgp case {it}: Z={Z} T={T} D={D} M={M} {'fp32' if dtype else 'fp64'}"
    hs = _lib.Handle(0)
    hs.fit_svgp(Zp, y, Sigma, ls, osc, dtype=dtype)
    out = hs.predict_all(Xq, mean=True, var=True, J=True, Jvar=True)
    hs.close()
    if M > 1500:
        sub = np.sort(rng.choice(M, 1500, replace=False))
        Xq = Xq[sub]
        out = {k: v[sub] for k, v in out.items() if v is not None}
    print(f"... {tag}", flush=True)
    rm, rs, rJ, rJs = orc.svgp_exact_oracle_fast(Xq, Zp, Sigma, y, osc, ls)
    nq = len(Xq)
    gvar, gjvar = np.reshape(out["var"], (nq, T)), np.reshape(out["Jvar"], (nq, T, D))      # the library squeezes T = 1
    vs, js = np.max(osc), np.max(osc[:, None] / ls[None, :] ** 2)
    errs = {"mean": rel(out["mean"], rm), "var": float(np.max(np.abs(gvar - rs ** 2)) / vs), "J": rel(out["J"], rJ),
            "Jvar": float(np.max(np.abs(gjvar - rJs ** 2)) / js)}
    lim = {k: 1e-6 for k in errs}
    if dtype:
        # fp32 mean / J are sums with cancellation (alpha of an ill-conditioned K): measured against the size of the
        # terms, sum_i |k_i alpha_i|, not against a result that may have cancelled to nothing (M = 1 ...)
        Rq = np.exp(-0.5 * (((Xq[:, None, :] - Zp[None, :, :]) / ls) ** 2).sum(-1))
        Ruu = np.exp(-0.5 * (((Zp[:, None, :] - Zp[None, :, :]) / ls) ** 2).sum(-1))
        terms = max(float(np.max((osc[t] * Rq) @ np.abs(np.linalg.solve(osc[t] * Ruu + Sigma[t], y[t])))) for t in range(T))
        errs["mean"] = float(np.max(np.abs(np.reshape(out["mean"], rm.shape) - rm)) / terms)
        errs["J"] = float(np.max(np.abs(np.reshape(out["J"], rJ.shape) - rJ)) / (terms / float(np.min(ls))))
    if dtype:       # fp32: no worse than twice what the same algebra loses in numpy float32 (ill-conditioned K: cancellation)
        m32, s32, J32, Js32 = orc.svgp_exact_oracle_fast(Xq, Zp, Sigma, y, osc, ls, dtype=np.float32)
        ref32 = {"mean": float(np.max(np.abs(m32 - rm)) / terms), "var": float(np.max(np.abs(s32.astype(float) ** 2 - rs ** 2)) / vs), "J": float(np.max(np.abs(J32 - rJ)) / (terms / float(np.min(ls)))),
                 "Jvar": float(np.max(np.abs(Js32.astype(float) ** 2 - rJs ** 2)) / js)}
        lim = {k: max(2e-4, 2 * ref32[k]) for k in errs}
    key = "fp32 " if dtype else "fp64 "
    for k, v in errs.items():
        worst2[key + k] = max(worst2.get(key + k, 0.0), v)
    if any(errs[k] > lim[k] for k in errs) or not all(np.isfinite(list(errs.values()))):
        bad2 += 1
        print("LARGE:", tag, {k: f"{v:.2e}" for k, v in errs.items()}, flush=True)
print(f"svgp: {max(cases // 2, 6)} cases, {bad2} flagged (limits 1e-6 fp64; fp32: max(2e-4, 2 x the numpy-float32 restatement's loss)); worst:",
      {k: f"{v:.2e}" for k, v in sorted(worst2.items())})

