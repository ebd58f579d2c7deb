/* libgpt_hip — C ABI of the MI355X (gfx950) Gaussian-process transportation hot path.
 *
 * The reference (vyasakash231/gaussian_process_transportation) has no FFI for this path: its
 * boundary is the Python duck type consumed by PolicyTransportation
 * (policy_transportation/transportation/policy_transportation.py:12-14, 24, 30-32, 41-43) and
 * implemented by GaussianProcess (policy_transportation/models/gaussian_process.py:16-126) on
 * top of scikit-learn's GaussianProcessRegressor.  Each entry point below names the reference
 * call it stands in for; the ctypes binding a maintainer would add is shown in INTEGRATION.md.
 *
 * Conventions
 *   - plain C, no torch / numpy types; all matrices are C-contiguous row-major fp64
 *   - "host" pointers are ordinary process memory, "dev" pointers are HIP device memory on the
 *     handle's device (e.g. torch.Tensor.data_ptr()); the library owns every other allocation
 *   - every function returns 0 on success or a negative GPT_E_* code; gpt_last_error() returns
 *     the message of the last failure on the calling thread
 *   - one handle = one fitted model on one GPU; a handle is not thread-safe
 *   - D (input dims) in 1..3; O (outputs) >= 1; length_scale has 1 (isotropic) or D entries
 */
#ifndef GPT_HIP_H
#define GPT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gpt_handle gpt_handle;

#define GPT_OK 0
#define GPT_E_HIP (-1)        /* HIP runtime / launch failure                                   */
#define GPT_E_NOT_PD (-2)     /* non-positive pivot in the Cholesky: numpy.linalg.LinAlgError,  *
                               * as sklearn/gaussian_process/_gpr.py:348-358 raises             */
#define GPT_E_ARG (-3)        /* bad argument                                                   */
#define GPT_E_STATE (-4)      /* model not fitted / factor not committed                        */

/* Number of visible HIP devices (0 when no GPU: the Python shim then refuses to run). */
int gpt_device_count(void);
/* Text of the last error on this thread ("" if none). */
const char* gpt_last_error(void);
/* Library version string. */
const char* gpt_version(void);

/* Create / destroy a model handle bound to `device`. */
int gpt_create(gpt_handle** out, int device);
void gpt_destroy(gpt_handle* h);
/* Launch all kernels of this handle on `hip_stream` (hipStream_t, e.g.
 * torch.cuda.current_stream().cuda_stream); NULL restores the handle's own stream. */
int gpt_set_stream(gpt_handle* h, void* hip_stream);
/* Block until the handle's stream is idle. */
int gpt_synchronize(gpt_handle* h);

/* fit — replaces GaussianProcess.fit (models/gaussian_process.py:25-43) for fixed hyper-
 * parameters: sklearn's K = c*RBF(X/l) + (noise_level + alpha)*I, L = cholesky(K), alpha_ =
 * cho_solve(L, Y) (sklearn/_gpr.py:346-364) plus the factor of K^-1 the derivative code needs
 * (gaussian_process.py:42-43, kept here as W = L^-1).  X is (N,D), Y is (N,O), host memory; rows
 * with NaN must already be filtered by the caller (gaussian_process.py:33-35 does it in Python).
 * Returns GPT_E_NOT_PD when a pivot is <= 0. */
int gpt_fit(gpt_handle* h, const double* X, const double* Y, int64_t N, int D, int O,
            const double* length_scale, int n_ls, double constant_value, double noise_level,
            double alpha_jitter);

/* The same for `ConstantKernel * Matern(nu) + WhiteKernel` (sklearn/gaussian_process/kernels.py:1717-1778), the
 * kernel the reference's examples use for their dynamics GP (example/2D/surface_generalization.py:49,
 * example/3D/surface_generalization_3D.py:42).  gpt_fit == kernel_type GPT_KERNEL_RBF.  Derivative entry points
 * stay RBF-only (the reference's derivative formulas, gaussian_process.py:63-126, are RBF formulas). */
#define GPT_KERNEL_RBF 0
#define GPT_KERNEL_MATERN12 1
#define GPT_KERNEL_MATERN32 2
#define GPT_KERNEL_MATERN52 3
int gpt_fit_kernel(gpt_handle* h, const double* X, const double* Y, int64_t N, int D, int O,
                   const double* length_scale, int n_ls, double constant_value, double noise_level,
                   double alpha_jitter, int kernel_type);

/* Exact-GP algebra on SVGP pseudo-points — the fit behind the reference's `convert_to_exact_gp`
 * (policy_transportation/models/torch/stocastic_variational_gaussian_process_derivatives.py:72-78):
 * K = c*k(X,X) + Sigma + alpha_jitter*I with a full SPD matrix Sigma (N,N) (the per-task pseudo-point covariance)
 * in place of the scalar noise; alpha = K^-1 Y.  One handle per task (O = 1, c = that task's outputscale).
 * Afterwards gpt_predict_all gives mean, var = c - k*^T K^-1 k* (k** carries no noise: :120-123), the Jacobian
 * and its variance c/l_d^2 - dk_d^T K^-1 dk_d (:132-153).  Host memory. */
int gpt_fit_noise_matrix(gpt_handle* h, const double* X, const double* Y, int64_t N, int D, int O,
                         const double* length_scale, int n_ls, double constant_value, const double* Sigma,
                         double alpha_jitter, int kernel_type);

/* predict — replaces GaussianProcess.predict (gaussian_process.py:46-55 -> sklearn/_gpr.py:441-494).
 * mean (M,O); var (M,) = max(c + noise_level - |L^-1 k*|^2, 0) (the caller applies sqrt, the
 * tiling over O and the reference's `- sqrt(noise_level)` quirk).  var may be NULL. Host memory. */
int gpt_predict(gpt_handle* h, const double* Xq, int64_t M, double* mean, double* var);

/* derivative — replaces GaussianProcess.derivative (gaussian_process.py:63-102).
 * J (M,O,D) with J[m,o,d] = d mean_o / d x_d; Jvar (M,D) = c/l_d^2 - dk_d^T K^-1 dk_d (the
 * reference tiles it over O).  Jvar may be NULL.  Host memory. */
int gpt_derivative(gpt_handle* h, const double* Xq, int64_t M, double* J, double* Jvar);

/* derivative_of_variance — replaces GaussianProcess.derivative_of_variance
 * (gaussian_process.py:104-126).  g is (D,M).  Host memory. */
int gpt_dvariance(gpt_handle* h, const double* Xq, int64_t M, double* g);

/* Fused metric path: any of mean (M,O) / var (M,) / J (M,O,D) / Jvar (M,D) / dvar (D,M) may be
 * NULL.  Host memory (pageable is fine); queries are streamed through the device in chunks of 131072, the outputs of
 * one chunk leaving on a copy stream while the next chunk computes.  Returns when every output is in place. */
int gpt_predict_all(gpt_handle* h, const double* Xq, int64_t M, double* mean, double* var,
                    double* J, double* Jvar, double* dvar);
/* Same with every pointer in device memory; asynchronous on the handle's stream. */
int gpt_predict_all_dev(gpt_handle* h, const double* Xq_dev, int64_t M, double* mean_dev,
                        double* var_dev, double* J_dev, double* Jvar_dev, double* dvar_dev);
/* (new) Allocates the library-owned scratch a gpt_predict_all_dev call with M queries will use (grow-only; with
 * jacobian_variance != 0 for the 4-column path), so that the first such call does not allocate. */
int gpt_reserve(gpt_handle* h, int64_t M, int jacobian_variance);

/* predict(return_cov=True) — replaces sklearn/_gpr.py:458-470: mean (M,O) (may be NULL) and the joint
 * posterior covariance cov (M,M) = k(Xq,Xq) + noise_level*I - V^T V, V = L^-1 K*^T (identical for every
 * output; the caller tiles it).  Small-M path used by GaussianProcess.samples (gaussian_process.py:57-60);
 * M <= 16384; needs the handle that ran gpt_fit.  Host memory. */
int gpt_predict_cov(gpt_handle* h, const double* Xq, int64_t M, double* mean, double* cov);

/* Parity-test export of sklearn's fitted attributes: L (N,N) lower triangular (zeros above),
 * alpha (N,O).  Either may be NULL.  Host memory. */
int gpt_export(gpt_handle* h, double* L, double* alpha);
/* W = L^-1 (N,N) lower triangular, host memory (tests only). */
int gpt_export_inverse_factor(gpt_handle* h, double* W);

/* Log-marginal likelihood of the fitted theta (sklearn/_gpr.py:598-606): sum over outputs of
 * -0.5 y^T alpha - sum(log diag L) - N/2 log(2 pi). */
int gpt_lml(gpt_handle* h, double* lml);
/* The same value plus its gradient with respect to theta = log [constant_value, length_scale (n_ls
 * entries), noise_level] — what sklearn's optimizer consumes (sklearn/_gpr.py:625-648: 0.5 * trace((alpha
 * alpha^T - K^-1) dK/dtheta) summed over outputs).  grad has 2 + n_ls entries.  Overwrites the Cholesky
 * factor held for gpt_export (a later gpt_fit restores it). */
int gpt_lml_gradient(gpt_handle* h, double* lml, double* grad);

/* Multi-GPU hand-off of a fitted model (fit on rank 0, predict shards everywhere).  The model
 * is one contiguous device blob {header, scaled X, alpha, packed L^-1}:
 *   rank 0   : gpt_fit(...); gpt_factor_blob(h, &ptr, &bytes)
 *   others   : gpt_factor_alloc(h, N, D, O, &ptr, &bytes)        (same N, D, O)
 *   all      : broadcast `bytes` bytes at `ptr` (RCCL, e.g. torch.distributed.broadcast)
 *   others   : gpt_factor_commit(h)                                (parses the header)           */
int gpt_factor_blob(gpt_handle* h, void** dev_ptr, size_t* bytes);
int gpt_factor_alloc(gpt_handle* h, int64_t N, int D, int O, void** dev_ptr, size_t* bytes);
int gpt_factor_commit(gpt_handle* h);

/* Model geometry of a fitted / committed handle. */
int gpt_info(gpt_handle* h, int64_t* N, int* D, int* O, int64_t* N_padded);

/* Per-phase device times of the last gpt_fit in milliseconds (hipEvent):
 * [0] total [1] gram [2] cholesky [3] triangular inverse [4] alpha [5] pack.  n <= 6. */
int gpt_fit_timings(gpt_handle* h, double* ms_out, int n);

/* Per-kernel device times of the last gpt_predict_all_dev call.  With profiling enabled the call
 * records hipEvents around each kernel on the handle's stream (no host synchronisation);
 * gpt_predict_timings waits for them and returns ms_out[0] = mean+Jacobian contraction kernel,
 * ms_out[1] = variance (MFMA) kernel; 0 for a kernel that was not launched. */
int gpt_set_profiling(gpt_handle* h, int enable);
int gpt_predict_timings(gpt_handle* h, double* ms_out);

#ifdef __cplusplus
}
#endif
#endif /* GPT_HIP_H */
