"""ctypes binding of libgpt_hip.so (include/gpt_hip.h).  No CPU fallback: if the library or a
GPU is missing, every compute entry point raises."""
from __future__ import annotations

import ctypes as C
import importlib.util
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GPT_HIP_LIB") or os.path.join(_HERE, "libgpt_hip.so")   # override: A/B of two builds

GPT_OK, GPT_E_HIP, GPT_E_NOT_PD, GPT_E_ARG, GPT_E_STATE = 0, -1, -2, -3, -4
GPT_F64, GPT_F32 = 0, 1
MAX_D = 15         # input dimensions the library accepts (gpt_common.h MAX_DIMS: D <= 3 tuned layout, 4 .. 8 rows of 8, 9 .. 15 rows of 16)
_NP_DTYPE = {GPT_F64: np.float64, GPT_F32: np.float32}

_dp = C.POINTER(C.c_double)
_vp = C.c_void_p
_i64 = C.c_int64

# name -> (restype, argtypes); mirrors include/gpt_hip.h one to one
SIGNATURES = {
    "gpt_device_count": (C.c_int, []),
    "gpt_last_error": (C.c_char_p, []),
    "gpt_version": (C.c_char_p, []),
    "gpt_create": (C.c_int, [C.POINTER(_vp), C.c_int]),
    "gpt_destroy": (None, [_vp]),
    "gpt_set_stream": (C.c_int, [_vp, _vp]),
    "gpt_synchronize": (C.c_int, [_vp]),
    "gpt_set_dtype": (C.c_int, [_vp, C.c_int]),
    "gpt_fit": (C.c_int, [_vp, _dp, _dp, _i64, C.c_int, C.c_int, _dp, C.c_int, C.c_double, C.c_double, C.c_double]),
    "gpt_fit_kernel": (C.c_int, [_vp, _dp, _dp, _i64, C.c_int, C.c_int, _dp, C.c_int, C.c_double, C.c_double, C.c_double, C.c_int]),
    "gpt_fit_noise_matrix": (C.c_int, [_vp, _dp, _dp, _i64, C.c_int, C.c_int, _dp, C.c_int, C.c_double, _dp, C.c_double, C.c_int]),
    "gpt_fit_svgp": (C.c_int, [_vp, _dp, _dp, _dp, _i64, C.c_int, C.c_int, _dp, C.c_int, _dp, C.c_double, C.c_int]),
    "gpt_predict": (C.c_int, [_vp, _vp, _i64, _vp, _vp]),
    "gpt_derivative": (C.c_int, [_vp, _vp, _i64, _vp, _vp]),
    "gpt_dvariance": (C.c_int, [_vp, _vp, _i64, _vp]),
    "gpt_predict_all": (C.c_int, [_vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp]),
    "gpt_predict_all_dev": (C.c_int, [_vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp]),
    "gpt_predict_cov": (C.c_int, [_vp, _dp, _i64, _dp, _dp]),
    "gpt_export": (C.c_int, [_vp, _dp, _dp]),
    "gpt_export_inverse_factor": (C.c_int, [_vp, _dp]),
    "gpt_lml": (C.c_int, [_vp, _dp]),
    "gpt_lml_gradient": (C.c_int, [_vp, _dp, _dp]),
    "gpt_lml_objective": (C.c_int, [_vp, _dp, _dp, _i64, C.c_int, C.c_int, _dp, C.c_int, C.c_double, C.c_double, C.c_double, C.c_int,
                                    _dp, _dp]),
    "gpt_factor_blob": (C.c_int, [_vp, C.POINTER(_vp), C.POINTER(C.c_size_t)]),
    "gpt_factor_alloc": (C.c_int, [_vp, _i64, C.c_int, C.c_int, C.POINTER(_vp), C.POINTER(C.c_size_t)]),
    "gpt_factor_alloc_model": (C.c_int, [_vp, _i64, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_vp), C.POINTER(C.c_size_t)]),
    "gpt_factor_commit": (C.c_int, [_vp]),
    "gpt_factor_copy": (C.c_int, [_vp, _vp]),
    "gpt_model_info": (C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "gpt_debug_var_plan": (C.c_int, [_i64, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_i64), C.POINTER(C.c_int),
                                     C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "gpt_debug_fit_plan": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_i64), C.POINTER(_i64)]),
    "gpt_info": (C.c_int, [_vp, C.POINTER(_i64), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(_i64)]),
    "gpt_fit_timings": (C.c_int, [_vp, _dp, C.c_int]),
    "gpt_set_profiling": (C.c_int, [_vp, C.c_int]),
    "gpt_reserve": (C.c_int, [_vp, C.c_int64, C.c_int]),
    "gpt_predict_timings": (C.c_int, [_vp, _dp]),
}

_lib = None


class GptError(RuntimeError):
    pass


def _preload_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so with the same SONAME as /opt/rocm's.  The
    first copy loaded serves the whole process, and torch cannot initialise on a foreign copy
    ("No HIP GPUs are available").  When torch is installed but not yet imported, load ITS runtime
    first so that libgpt_hip.so and a later `import torch` share one HIP runtime in either order."""
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load():
    """Load libgpt_hip.so (built by `__graft_entry__.build()` / csrc/Makefile).  Fails loudly."""
    global _lib
    if _lib is not None:
        return _lib
    _preload_hip_runtime()
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()' "
            "or make -C gaussian_process_transportation_amd/csrc).  There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the ABI drifted
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error() -> str:
    return load().gpt_last_error().decode("utf-8", "replace")


def check(rc: int, what: str = ""):
    if rc == GPT_OK:
        return
    msg = last_error()
    if rc == GPT_E_NOT_PD:
        raise np.linalg.LinAlgError(msg)          # what sklearn raises (_gpr.py:348-358)
    if rc == GPT_E_ARG:
        raise ValueError(msg)
    raise GptError(f"{what or 'libgpt_hip'} failed ({rc}): {msg}")


def require_gpu() -> int:
    n = load().gpt_device_count()
    if n < 1:
        raise GptError("no HIP device visible: this package only runs on an MI355X (gfx950); there is no CPU path")
    return n


def as_f64(a, ndim=None, what="Input", dtype=np.float64) -> np.ndarray:
    """C-contiguous array of `dtype`, finite.  scikit-learn's check_array refuses NaN / infinity in fit and predict
    inputs with a ValueError (sklearn/utils/validation.py, reached from _gpr.py:262 and :415); the kernels would
    otherwise turn a NaN query into a prior-looking prediction (the table exp clamps its argument).  The
    device-pointer API (`predict_all_dev`) leaves this check to the caller."""
    a = np.ascontiguousarray(a, dtype=dtype)
    if ndim is not None and a.ndim != ndim:
        raise ValueError(f"expected a {ndim}-D array, got shape {a.shape}")
    if not np.isfinite(a).all():
        raise ValueError(f"{what} contains NaN or infinity.")
    return a


def dptr(a):
    return None if a is None else a.ctypes.data_as(_dp)


def vptr(a):
    return None if a is None else C.c_void_p(a.ctypes.data)


def debug_var_plan(n_columns, n_iblocks, n_tasks, n_workgroups, order=-1):
    """The work decomposition of the variance kernel (csrc/gpt_plan.h) as numpy arrays; host code, no GPU."""
    lib = load()
    counts = (_i64 * 12)()
    ip = C.POINTER(C.c_int)
    check(lib.gpt_debug_var_plan(n_columns, n_iblocks, n_tasks, n_workgroups, order, counts, None, None, None, None))
    item_begin = np.zeros(n_workgroups + 1, dtype=np.int32)
    items = np.zeros((max(counts[0], 1), 8), dtype=np.int32)
    fin = np.zeros((max(counts[6], 1), 2), dtype=np.int32)
    splits = np.zeros((max(counts[1], 1), 3), dtype=np.int32)
    check(lib.gpt_debug_var_plan(n_columns, n_iblocks, n_tasks, n_workgroups, order, counts, item_begin.ctypes.data_as(ip),
                                 items.ctypes.data_as(ip), fin.ctypes.data_as(ip), splits.ctypes.data_as(ip)))
    return {"n_items": counts[0], "n_splits": counts[1], "n_slots": counts[2], "n_vslots": counts[3], "ncb": counts[4],
            "nfull": counts[5], "order": counts[7], "cohorts": bool(counts[8]), "cohort_s": counts[9], "cohort_f": counts[10], "cut_diag": bool(counts[11]),
            "item_begin": item_begin, "items": items[:counts[0]],
            "fin": fin[:counts[6]], "splits": splits[:counts[1]]}


FIT_OP_KINDS = ("POTRF", "FINISH", "TRINV", "UPDATE", "TRSM", "COPY_L21", "T", "WFIN", "FACTORED")
FIT_OP_FIELDS = ("kind", "stream", "off", "n1", "n2", "k0", "kw", "row_end", "grp", "r0", "r0_size", "r1", "r1_size", "wait0", "wait1",
                 "wait2", "record")


def debug_fit_plan(n_padded, form=-1, panel=-1, streams=-1):
    """The plan of the factor + inverse (csrc/gpt_fit_plan.h) for a padded size; host code, no GPU.
    ops: one row per operation, columns FIT_OP_FIELDS (regions in doubles inside the arena; events by id, -1 = none)."""
    lib = load()
    counts = (_i64 * 6)()
    check(lib.gpt_debug_fit_plan(n_padded, form, panel, streams, counts, None))
    ops = np.zeros((max(counts[0], 1), 18), dtype=np.int64)
    check(lib.gpt_debug_fit_plan(n_padded, form, panel, streams, counts, ops.ctypes.data_as(C.POINTER(_i64))))
    return {"arena": counts[1], "form": counts[2], "n_events": counts[3], "allocated": counts[4], "side_eighths": counts[5],
            "ops": ops[:counts[0], :17]}


class Handle:
    """Owns one gpt_handle (one fitted model on one GPU)."""

    def __init__(self, device: int = 0):
        self.lib = load()
        require_gpu()
        h = _vp()
        check(self.lib.gpt_create(C.byref(h), int(device)), "gpt_create")
        self._h = h
        self.device = int(device)

    def close(self):
        if getattr(self, "_h", None):
            self.lib.gpt_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- fit / export
    def set_dtype(self, dtype):
        """Element type of the models fitted from now on: GPT_F64 (default) or GPT_F32."""
        check(self.lib.gpt_set_dtype(self._h, int(dtype)), "gpt_set_dtype")

    def fit(self, X, Y, length_scale, constant_value, noise_level, alpha, kernel_type=0):
        """kernel_type: 0 RBF, 1/2/3 Matern nu = 0.5 / 1.5 / 2.5 (GPT_KERNEL_* of include/gpt_hip.h)."""
        X = as_f64(X, 2, "X")
        Y = as_f64(Y, 2, "y")
        ls = as_f64(np.atleast_1d(length_scale), 1, "length_scale")
        N, D = X.shape
        if Y.shape[0] != N:
            raise ValueError("X and Y have different numbers of rows")
        if not 1 <= D <= MAX_D:
            raise ValueError(f"X has {D} features: this GPU path supports input dimension D = 1 .. {MAX_D} only")
        check(self.lib.gpt_fit_kernel(self._h, dptr(X), dptr(Y), N, D, Y.shape[1], dptr(ls), ls.size,
                                      float(constant_value), float(noise_level), float(alpha), int(kernel_type)), "gpt_fit")

    def fit_noise_matrix(self, X, Y, length_scale, constant_value, Sigma, alpha=0.0, kernel_type=0):
        """K = c k(X,X) + Sigma (full SPD (N,N)) + alpha I — the SVGP exact-conversion model."""
        X = as_f64(X, 2, "X")
        Y = as_f64(Y, 2, "y")
        Sigma = as_f64(Sigma, 2, "Sigma")
        ls = as_f64(np.atleast_1d(length_scale), 1, "length_scale")
        N, D = X.shape
        if Y.shape[0] != N or Sigma.shape != (N, N):
            raise ValueError("X, Y and Sigma disagree on the number of points")
        check(self.lib.gpt_fit_noise_matrix(self._h, dptr(X), dptr(Y), N, D, Y.shape[1], dptr(ls), ls.size,
                                            float(constant_value), dptr(Sigma), float(alpha), int(kernel_type)),
              "gpt_fit_noise_matrix")

    def fit_svgp(self, Z, y, Sigma, length_scale, outputscale, jitter=0.0, dtype=GPT_F64):
        """The multi-task SVGP exact-conversion model in one handle (gpt_fit_svgp): Z (N,D), y (T,N), Sigma (T,N,N),
        outputscale (T,), length_scale (1 or D)."""
        Z = as_f64(Z, 2, "Z")
        y = as_f64(y, 2, "y")
        Sigma = as_f64(Sigma, 3, "Sigma")
        ls = as_f64(np.atleast_1d(length_scale), 1, "length_scale")
        osc = as_f64(np.atleast_1d(outputscale), 1, "outputscale")
        N, D = Z.shape
        T = y.shape[0]
        if y.shape != (T, N) or Sigma.shape != (T, N, N) or osc.shape != (T,):
            raise ValueError("expected y (T,N), Sigma (T,N,N), outputscale (T,)")
        check(self.lib.gpt_fit_svgp(self._h, dptr(Z), dptr(y), dptr(Sigma), N, D, T, dptr(ls), ls.size, dptr(osc),
                                    float(jitter), int(dtype)), "gpt_fit_svgp")

    def model_info(self):
        """(n_tasks, dtype) of the fitted model."""
        nt, dt = C.c_int(), C.c_int()
        check(self.lib.gpt_model_info(self._h, C.byref(nt), C.byref(dt)), "gpt_model_info")
        return nt.value, dt.value

    def info(self):
        N, NP, D, O = _i64(), _i64(), C.c_int(), C.c_int()
        check(self.lib.gpt_info(self._h, C.byref(N), C.byref(D), C.byref(O), C.byref(NP)), "gpt_info")
        return N.value, D.value, O.value, NP.value

    def export(self, want_L=True, want_alpha=True):
        N, D, O, _ = self.info()
        L = np.empty((N, N)) if want_L else None
        a = np.empty((N, O)) if want_alpha else None
        check(self.lib.gpt_export(self._h, dptr(L), dptr(a)), "gpt_export")
        return L, a

    def export_inverse_factor(self):
        N = self.info()[0]
        W = np.empty((N, N))
        check(self.lib.gpt_export_inverse_factor(self._h, dptr(W)), "gpt_export_inverse_factor")
        return W

    def lml(self) -> float:
        v = C.c_double()
        check(self.lib.gpt_lml(self._h, C.byref(v)), "gpt_lml")
        return v.value

    def lml_gradient(self, n_ls):
        v = C.c_double()
        g = np.zeros(2 + int(n_ls))
        check(self.lib.gpt_lml_gradient(self._h, C.byref(v), dptr(g)), "gpt_lml_gradient")
        return v.value, g

    def lml_objective(self, X, Y, length_scale, constant_value, noise_level, alpha, kernel_type=0):
        """(lml, gradient w.r.t. theta = log [c, length_scale..., noise]) for these hyper-parameters in one call; the
        handle holds no model afterwards.  X, Y as for fit (validated by the caller once: the optimizer calls this
        hundreds of times on the same arrays)."""
        ls = np.ascontiguousarray(np.atleast_1d(length_scale), dtype=np.float64)
        v = C.c_double()
        g = np.zeros(2 + ls.size)
        N, D = X.shape
        check(self.lib.gpt_lml_objective(self._h, dptr(X), dptr(Y), N, D, Y.shape[1], dptr(ls), ls.size, float(constant_value),
                                         float(noise_level), float(alpha), int(kernel_type), C.byref(v), dptr(g)), "gpt_lml_objective")
        return v.value, g

    def fit_timings(self):
        t = np.zeros(6)
        check(self.lib.gpt_fit_timings(self._h, dptr(t), 6), "gpt_fit_timings")
        return dict(zip(["total", "gram", "cholesky", "inverse", "alpha", "pack"], t.tolist()))

    # ---- predict (host buffers)
    def predict_all(self, Xq, mean=False, var=False, J=False, Jvar=False, dvar=False):
        """Arrays come back in the model's element type (float64, or float32 for a GPT_F32 model); the multi-task
        model returns var (M,T) and Jvar (M,T,D)."""
        N, D, O, _ = self.info()
        nt, dt = self.model_info()
        ty = _NP_DTYPE[dt]
        Xq = as_f64(Xq, 2, "X", dtype=ty)
        if Xq.shape[1] != D:
            raise ValueError(f"query has {Xq.shape[1]} features, model was fitted with {D}")
        M = Xq.shape[0]
        vshape, jvshape = ((M,), (M, D)) if nt == 1 else ((M, nt), (M, nt, D))
        out = {
            "mean": np.empty((M, O), dtype=ty) if mean else None,
            "var": np.empty(vshape, dtype=ty) if var else None,
            "J": np.empty((M, O, D), dtype=ty) if J else None,
            "Jvar": np.empty(jvshape, dtype=ty) if Jvar else None,
            "dvar": np.empty((D, M), dtype=ty) if dvar else None,
        }
        check(self.lib.gpt_predict_all(self._h, vptr(Xq), M, vptr(out["mean"]), vptr(out["var"]), vptr(out["J"]),
                                       vptr(out["Jvar"]), vptr(out["dvar"])), "gpt_predict_all")
        return out

    def predict_cov(self, Xq):
        N, D, O, _ = self.info()
        Xq = as_f64(Xq, 2, "X")
        if Xq.shape[1] != D:
            raise ValueError(f"query has {Xq.shape[1]} features, model was fitted with {D}")
        M = Xq.shape[0]
        mean, cov = np.empty((M, O)), np.empty((M, M))
        check(self.lib.gpt_predict_cov(self._h, dptr(Xq), M, dptr(mean), dptr(cov)), "gpt_predict_cov")
        return mean, cov

    # ---- predict (device pointers, asynchronous)
    def predict_all_dev(self, xq_ptr, M, mean_ptr=0, var_ptr=0, J_ptr=0, Jvar_ptr=0, dvar_ptr=0):
        check(self.lib.gpt_predict_all_dev(self._h, _vp(xq_ptr), int(M), _vp(mean_ptr or None), _vp(var_ptr or None),
                                           _vp(J_ptr or None), _vp(Jvar_ptr or None), _vp(dvar_ptr or None)),
              "gpt_predict_all_dev")

    def reserve(self, M, jacobian_variance=False):
        """Allocate the scratch of a predict_all_dev call with M queries ahead of time."""
        check(self.lib.gpt_reserve(self._h, int(M), int(bool(jacobian_variance))), "gpt_reserve")

    def set_profiling(self, enable=True):
        check(self.lib.gpt_set_profiling(self._h, int(bool(enable))), "gpt_set_profiling")

    def predict_timings(self):
        t = np.zeros(2)
        check(self.lib.gpt_predict_timings(self._h, dptr(t)), "gpt_predict_timings")
        return {"mean_jac_ms": float(t[0]), "var_ms": float(t[1])}

    def set_stream(self, stream_ptr):
        check(self.lib.gpt_set_stream(self._h, _vp(stream_ptr or None)), "gpt_set_stream")

    def synchronize(self):
        check(self.lib.gpt_synchronize(self._h), "gpt_synchronize")

    # ---- multi-GPU hand-off
    def factor_blob(self):
        p, n = _vp(), C.c_size_t()
        check(self.lib.gpt_factor_blob(self._h, C.byref(p), C.byref(n)), "gpt_factor_blob")
        return p.value, n.value

    def factor_alloc(self, N, D, O, n_tasks=1, dtype=GPT_F64):
        p, n = _vp(), C.c_size_t()
        check(self.lib.gpt_factor_alloc_model(self._h, int(N), int(D), int(O), int(n_tasks), int(dtype), C.byref(p),
                                              C.byref(n)), "gpt_factor_alloc_model")
        return p.value, n.value

    def factor_commit(self):
        check(self.lib.gpt_factor_commit(self._h), "gpt_factor_commit")

    def factor_copy_from(self, src: "Handle"):
        """This handle becomes a replica of `src`'s fitted model (same process, any device): gpt_factor_copy."""
        check(self.lib.gpt_factor_copy(self._h, src._h), "gpt_factor_copy")
