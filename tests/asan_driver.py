"""Runs INSIDE the sanitizer subprocess of tests/test_host_cpu.py::test_host_orchestration_under_sanitizers: drives the
C ABI's host orchestration (GPT_HIP_LIB = csrc/build/libgpt_host_asan.so: g++ -fsanitize=address,undefined, inert HIP
runtime and kernel stand-ins that touch the memory the real kernels would) through its argument checks, state
machine, staging, hand-off and work-plan paths.  Outputs are meaningless; AddressSanitizer / UBSan abort on a finding."""
import numpy as np

from gaussian_process_transportation_amd import _lib

rng = np.random.default_rng(0)
lib = _lib.load()
assert b"gfx950" in lib.gpt_version()
h = _lib.Handle(0)
for bad in (lambda: _lib.Handle(3), lambda: h.predict_all(np.zeros((2, 3)), mean=True), lambda: h.export(), lambda: h.reserve(10),
            lambda: h.fit(np.zeros((3, 2)), np.zeros((4, 2)), [1.0], 1.0, 0.1, 0.0),
            lambda: h.fit(np.zeros((3, 2)), np.zeros((3, 2)), [1.0, 1.0, 1.0], 1.0, 0.1, 0.0),
            lambda: h.fit(np.zeros((3, 2)), np.zeros((3, 2)), [-1.0], 1.0, 0.1, 0.0),
            lambda: h.fit(np.zeros((3, 2)), np.zeros((3, 2)), [1.0], 0.0, 0.1, 0.0),
            lambda: h.fit(np.zeros((3, 2)), np.zeros((3, 2)), [1.0], 1.0, 0.1, 0.0, kernel_type=7),
            lambda: h.set_dtype(5), lambda: h.factor_commit()):
    try:
        bad()
    except (ValueError, _lib.GptError):
        continue
    raise AssertionError("an invalid call went through")

for bad in (lambda: h.fit(np.zeros((3, 16)), np.zeros((3, 2)), [1.0], 1.0, 0.1, 0.0),):        # D beyond MAX_DIMS
    try:
        bad()
    except ValueError:
        continue
    raise AssertionError("D = 16 went through")

for N, D, O in ((1, 1, 1), (700, 3, 3), (1100, 2, 6), (530, 5, 5), (200, 8, 2), (64, 4, 1)):
    X = rng.uniform(0, 1, (N, D)); Y = rng.standard_normal((N, O))
    for dtype in (_lib.GPT_F64, _lib.GPT_F32):
        h.set_dtype(dtype)
        h.fit(X, Y, np.full(D, 0.3), 1.0, 1e-2, 1e-10)
        assert h.model_info() == (1, dtype) and h.info()[:3] == (N, D, O)
        for M in (1, 63, 460, 4097, 140_000 if D <= 3 else 20_000):
            q = rng.uniform(0, 1, (M, D))
            out = h.predict_all(q, mean=True, var=True, J=True, Jvar=True, dvar=True)
            assert out["mean"].shape == (M, O) and out["dvar"].shape == (D, M)
            h.predict_all(q, var=True)
            h.predict_all(q, Jvar=True)
        h.reserve(5000, True)
        h.export(); h.export_inverse_factor(); h.lml()
        if dtype == _lib.GPT_F64:
            h.predict_cov(rng.uniform(0, 1, (130, D)))
        h.lml_gradient(D)
        h.lml_objective(X, Y, np.full(D, 0.3), 1.0, 1e-2, 1e-10)       # leaves no model behind
        try:
            h.predict_all(rng.uniform(0, 1, (3, D)), mean=True)
            raise AssertionError("predict after gpt_lml_objective must fail")
        except _lib.GptError:
            pass
        h.fit(X, Y, np.full(D, 0.3), 1.0, 1e-2, 1e-10)
        h.lml_gradient(D)
        try:
            h.export(want_alpha=False)                      # L was overwritten by the gradient's K^-1
            raise AssertionError("export of L after lml_gradient must fail")
        except _lib.GptError:
            pass
        h.fit_timings()
h.set_dtype(_lib.GPT_F64)

# a size the multi-stream forms of the factor + inverse run at (csrc/gpt_fit_plan.h): the stand-in replays every scratch region of
# the plan inside the arena the orchestration allocated — with the plan changing under the handle between two fits
import os
for form, panel in (("2", "512"), ("1", None), ("2", "2048"), (None, None)):
    for k, v in (("GPT_FIT_FORM", form), ("GPT_FIT_PANEL", panel)):
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    Xb = rng.uniform(0, 1, (4200, 3))
    h.fit(Xb, rng.standard_normal((4200, 2)), np.full(3, 0.3), 1.0, 1e-2, 1e-10)
    assert _lib.debug_fit_plan(4608)["form"] == int(form or 1)
h2 = _lib.Handle(0)
h2.factor_copy_from(h)                      # one-process replica (gpt_factor_copy)
assert h2.info() == h.info() and h2.model_info() == h.model_info()
h2.predict_all(rng.uniform(0, 1, (300, 3)), mean=True, var=True, J=True)
for bad in (lambda: h2.export(want_alpha=False), lambda: h2.factor_copy_from(h2), lambda: _lib.Handle(0).factor_copy_from(_lib.Handle(0))):
    try:
        bad()
    except (ValueError, _lib.GptError):
        continue
    raise AssertionError("an invalid hand-off went through")
h2.close()

# multi-task model + hand-off to a second handle
Z, T, D = 600, 3, 3
Zp = rng.uniform(0, 1, (Z, D)); A = rng.standard_normal((T, Z, Z)); Sigma = A @ A.transpose(0, 2, 1) / Z + 1e-3 * np.eye(Z)
y = rng.standard_normal((T, Z))
for dtype in (_lib.GPT_F32, _lib.GPT_F64):
    h.fit_svgp(Zp, y, Sigma, np.full(D, 0.2), np.ones(T), dtype=dtype)
    assert h.model_info() == (T, dtype)
    for M in (5, 1000, 133_000):
        out = h.predict_all(rng.uniform(0, 1, (M, D)), mean=True, var=True, J=True, Jvar=True)
        assert out["var"].shape == (M, T) and out["Jvar"].shape == (M, T, D)
    for bad in (lambda: h.predict_cov(np.zeros((3, D))), lambda: h.predict_all(np.zeros((3, D)), dvar=True), h.lml):
        try:
            bad()
        except (ValueError, _lib.GptError):
            continue
        raise AssertionError("an invalid call on the multi-task model went through")
    h2 = _lib.Handle(0)
    src, nbytes = h.factor_blob()
    dst, nbytes2 = h2.factor_alloc(Z, D, T, T, dtype)
    assert nbytes == nbytes2
    import ctypes
    ctypes.memmove(dst, src, nbytes)
    h2.factor_commit()
    assert h2.model_info() == (T, dtype) and h2.info() == h.info()
    h2.predict_all(rng.uniform(0, 1, (777, D)), mean=True, var=True, J=True, Jvar=True)
    h2.export(want_L=False)
    h2.close()
try:
    h.fit_svgp(Zp, y, Sigma, np.full(D, 0.2), np.array([1.0, -1.0, 1.0]))
    raise AssertionError("negative outputscale accepted")
except ValueError:
    pass
pl = _lib.debug_var_plan(123_456, 7, 3, 256)
assert pl["n_items"] > 0
h.close()
print("ASAN_DRIVER_OK")
