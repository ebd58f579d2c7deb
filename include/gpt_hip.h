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
 *   - plain C, no torch / numpy types; all matrices are C-contiguous row-major.  Fit inputs are always fp64;
 *     queries and outputs of the predict entry points are in the MODEL's element type: fp64 (double) unless the
 *     model was fitted with GPT_F32 (float) — see gpt_set_dtype / gpt_fit_svgp
 *   - "host" pointers are ordinary process memory, "dev" pointers are HIP device memory on the
 *     handle's device (e.g. torch.Tensor.data_ptr()); the library owns every other allocation
 *   - every function returns 0 on success or a negative GPT_E_* code; gpt_last_error() returns
 *     the message of the last failure on the calling thread
 *   - one handle = one fitted model on one GPU; a handle is not thread-safe, but different handles may be used from
 *     different threads at the same time (the hyper-parameter search drives its independent restarts that way)
 *   - D (input dims) in 1..15: D <= 3 (the reference's transport problems are 2-D and 3-D) is the tuned layout, D = 4..8 and
 *     9..15 run on wider source layouts (rows of 8 / 16) with the same entry points and results (the reference's regressor is
 *     dimension-agnostic; D > 15 is refused with GPT_E_ARG); O (outputs) >= 1; length_scale has 1 (isotropic) or D entries
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

/* Element type of a model's prediction side (scaled sources, alpha, packed inverse factor, queries, outputs).
 * The factorisation always runs in fp64; with GPT_F32 its results are rounded once when the model is packed and the
 * prediction kernels run on v_mfma_f32_16x16x4_f32 — the arithmetic of the reference's torch SVGP path
 * (models/torch/stocastic_variational_gaussian_process_derivatives.py computes in torch.float32). */
#define GPT_F64 0
#define GPT_F32 1

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

/* Element type of the models this handle fits from now on (default GPT_F64). */
int gpt_set_dtype(gpt_handle* h, int dtype);

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

/* The whole multi-task model of the reference's SVGP exact conversion in ONE handle — replaces
 * SVGP.convert_to_exact_gp (models/torch/stocastic_variational_gaussian_process_derivatives.py:72-78) and feeds
 * posterior_f (:113-129) / posterior_f_prime (:132-153):
 *   Z (N,D) inducing points; y (T,N) pseudo-targets; Sigma (T,N,N) SPD pseudo-point covariances; outputscale (T);
 *   length_scale (1 or D, shared by the tasks as in the reference's kernel, batch_shape [1]).
 * Per task t: K_t = outputscale_t RBF(Z,Z) + Sigma_t + jitter I, W_t = chol(K_t)^-1, alpha_t = K_t^-1 y_t (fp64).
 * The tasks' factors are stacked into one A operand (each times its outputscale) and share one generated B operand,
 * so a query pays its exps once, not T times.  Afterwards the predict entry points return, with O = T:
 *   mean (M,T); var (M,T) = outputscale_t - k*_t^T K_t^-1 k*_t; J (M,T,D); Jvar (M,T,D) = outputscale_t / l_d^2 -
 *   dk_d^T K_t^-1 dk_d   (the reference takes sqrt of both variances; the caller does).
 * dtype: GPT_F64 or GPT_F32 (BASELINE configs[4] is fp32).  T <= 32.  Host memory.  GPT_E_NOT_PD if a K_t is not PD. */
int gpt_fit_svgp(gpt_handle* h, const double* Z, const double* y, const double* Sigma, int64_t N, int D, int T,
                 const double* length_scale, int n_ls, const double* outputscale, double jitter, int dtype);

/* predict — replaces GaussianProcess.predict (gaussian_process.py:46-55 -> sklearn/_gpr.py:441-494).
 * mean (M,O); var (M,) = max(c + noise_level - |L^-1 k*|^2, 0) (the caller applies sqrt, the
 * tiling over O and the reference's `- sqrt(noise_level)` quirk).  var may be NULL. Host memory. */
int gpt_predict(gpt_handle* h, const void* Xq, int64_t M, void* mean, void* var);

/* derivative — replaces GaussianProcess.derivative (gaussian_process.py:63-102).
 * J (M,O,D) with J[m,o,d] = d mean_o / d x_d; Jvar (M,D) = c/l_d^2 - dk_d^T K^-1 dk_d (the
 * reference tiles it over O).  Jvar may be NULL.  Host memory. */
int gpt_derivative(gpt_handle* h, const void* Xq, int64_t M, void* J, void* Jvar);

/* derivative_of_variance — replaces GaussianProcess.derivative_of_variance
 * (gaussian_process.py:104-126).  g is (D,M).  Host memory. */
int gpt_dvariance(gpt_handle* h, const void* Xq, int64_t M, void* g);

/* Fused metric path: any of mean (M,O) / var (M,) / J (M,O,D) / Jvar (M,D) / dvar (D,M) may be
 * NULL (multi-task model: var (M,T), Jvar (M,T,D), no dvar).  Buffers in the model's element type.
 * Host memory (pageable is fine); queries are streamed through the device in chunks of 131072, the outputs of
 * one chunk leaving on a copy stream while the next chunk computes.  Returns when every output is in place. */
int gpt_predict_all(gpt_handle* h, const void* Xq, int64_t M, void* mean, void* var,
                    void* J, void* Jvar, void* dvar);
/* Same with every pointer in device memory; asynchronous on the handle's stream. */
int gpt_predict_all_dev(gpt_handle* h, const void* Xq_dev, int64_t M, void* mean_dev,
                        void* var_dev, void* J_dev, void* Jvar_dev, void* dvar_dev);
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

/* One evaluation of the optimizer's objective in one call: what gpt_fit_kernel + gpt_lml_gradient return for these
 * hyper-parameters, without building the prediction-side model (no packed inverse factor, no alpha in the model's
 * layout) — the inner loop of GaussianProcess.fit with optimizer='fmin_l_bfgs_b' (sklearn/_gpr.py:296-338: every
 * L-BFGS-B step evaluates the LML and its gradient).  Afterwards the handle holds NO model (predict needs a gpt_fit). */
int gpt_lml_objective(gpt_handle* h, const double* X, const double* Y, int64_t N, int D, int O,
                      const double* length_scale, int n_ls, double constant_value, double noise_level,
                      double alpha_jitter, int kernel_type, double* lml, double* grad);

/* Multi-GPU hand-off of a fitted model (fit on rank 0, predict shards everywhere).  The model
 * is one contiguous device blob {header, scaled X, alpha, packed L^-1}:
 *   rank 0   : gpt_fit(...); gpt_factor_blob(h, &ptr, &bytes)
 *   others   : gpt_factor_alloc(h, N, D, O, &ptr, &bytes)        (same N, D, O)
 *   all      : broadcast `bytes` bytes at `ptr` (RCCL, e.g. torch.distributed.broadcast)
 *   others   : gpt_factor_commit(h)                                (parses the header)           */
int gpt_factor_blob(gpt_handle* h, void** dev_ptr, size_t* bytes);
int gpt_factor_alloc(gpt_handle* h, int64_t N, int D, int O, void** dev_ptr, size_t* bytes);
/* The same for any model: n_tasks = 1 and GPT_F64 for gpt_fit*, (T, dtype) for gpt_fit_svgp — see gpt_model_info. */
int gpt_factor_alloc_model(gpt_handle* h, int64_t N, int D, int O, int n_tasks, int dtype, void** dev_ptr, size_t* bytes);
int gpt_factor_commit(gpt_handle* h);
/* The same hand-off inside ONE process (a handle per GPU, SURVEY section 8b's n_devices): copies src's fitted model
 * blob to dst's device (hipMemcpyPeerAsync over xGMI; a device-to-device copy when both handles sit on one GPU) and
 * commits it.  Afterwards dst predicts exactly what src predicts.  What GaussianProcess(devices=[...]) calls after fit so
 * that predict / derivative (gaussian_process.py:46-55, 63-102) can shard their rows over the GPUs behind the same class
 * (transportation/gaussian_process_transportation.py:19-26 never sees the devices).  dst's fit-side state (L, W) is not
 * copied: export / return_cov / LML stay with the handle that ran the fit. */
int gpt_factor_copy(gpt_handle* dst, gpt_handle* src);

/* Model geometry of a fitted / committed handle. */
int gpt_info(gpt_handle* h, int64_t* N, int* D, int* O, int64_t* N_padded);

/* Stacked tasks (1 unless fitted by gpt_fit_svgp) and element type (GPT_F64 / GPT_F32) of the model. */
int gpt_model_info(gpt_handle* h, int* n_tasks, int* dtype);

/* Per-phase device times of the last gpt_fit in milliseconds (hipEvent):
 * [0] total [1] gram [2] cholesky [3] triangular inverse [4] alpha [5] pack.  n <= 6. */
int gpt_fit_timings(gpt_handle* h, double* ms_out, int n);

/* Per-kernel device times of the last gpt_predict_all_dev call.  With profiling enabled the call
 * records hipEvents around each kernel on the handle's stream (no host synchronisation);
 * gpt_predict_timings waits for them and returns ms_out[0] = mean+Jacobian contraction kernel,
 * ms_out[1] = variance (MFMA) kernel; 0 for a kernel that was not launched. */
int gpt_set_profiling(gpt_handle* h, int enable);
int gpt_predict_timings(gpt_handle* h, double* ms_out);

/* Test hook (host only, no GPU): the work decomposition of the variance kernel for a launch of n_columns kernel
 * columns over n_iblocks 512-row blocks x n_tasks tasks on n_workgroups workgroups (csrc/gpt_plan.h).
 * order: -1 automatic, 0 block-major, 1 sweep-major.  counts[12] = {items, cut sweeps, slab slots, partial-product
 * slots, column blocks, blocks in whole rounds, tail (block, task) pairs, order used, cohort plan used (0 / 1), its first whole
 * i-block s, tiles f of sweep s - 1 taken with the long sweeps, diagonal tiles divisible in this list (0 / 1)}; item k ranges
 * count quarter tiles; a cut sweep owns 8 consecutive slab slots from splits[.][2]; the arrays may be NULL (first
 * call) or hold item_begin[n_workgroups + 1], items[counts[0]][8], fin[counts[6]][2], splits[counts[1]][3]. */
int gpt_debug_var_plan(int64_t n_columns, int n_iblocks, int n_tasks, int n_workgroups, int order, int64_t* counts,
                       int* item_begin, int* items, int* fin, int* splits);

/* Test hook (host only, no GPU): the plan of the factor + inverse for a padded size (csrc/gpt_fit_plan.h).
 * form: 0 one leaf, 1 split with the second half's chain beside the first half's inverse, 2 left-looking panels with look-ahead,
 * < 0: environment / by size; panel (form 2) and streams (1 = CU-masked side and chain streams, 0 = the serial order) < 0:
 * environment or defaults.  counts[6] = {ops, arena doubles, form, events, doubles the fit workspace allocates (>= arena),
 * eighths of the CUs given to the side stream}; ops (may be NULL) receives counts[0] rows of 18: kind, stream, off, n1, n2, k0,
 * kw, row_end, grp, r0, r0_size, r1, r1_size, wait0, wait1, wait2, record, 0 (regions in doubles inside the arena; events by id,
 * -1 = none; FitOpKind and the streams in gpt_fit_plan.h). */
int gpt_debug_fit_plan(int n_padded, int form, int panel, int streams, int64_t* counts, int64_t* ops);

#ifdef __cplusplus
}
#endif
#endif /* GPT_HIP_H */
