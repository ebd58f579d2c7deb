// Shared declarations of the gfx950 GP-transportation library (internal header).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <mutex>
#include <cstdint>
#include <cstddef>

namespace gpt {

#if defined(__clang__)      // device code only (the sanitizer build of the host orchestration is compiled by g++)
typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
#endif

// Element type of the prediction side of a model (the factorisation is always fp64).
constexpr int DT_F64 = 0, DT_F32 = 1;

// ---- geometry --------------------------------------------------------------------------
// N is padded to NP = multiple of PAD_N so that every blocked kernel sees whole tiles.
constexpr int PAD_N = 512;
// Tile edge of the packed, fragment-ordered inverse factor Wf (rows of W per i-block and
// columns per k-block; square so that the block lower triangle packs cleanly).  One workgroup
// of the variance kernel sweeps a whole 512-row i-block: 8 waves x 64 rows ("row groups").
constexpr int WT = 512;
constexpr int WT_K4 = WT / 4;                 // MFMA k-steps (depth 4) per tile            (128)
constexpr int WT_GROUPS = WT / 64;            // 64-row groups per tile, one per wave       (8)
constexpr size_t WT_STEP_DOUBLES = (size_t)WT * 4;            // doubles per k4-step of one tile (2048)
constexpr size_t WT_TILE_DOUBLES = (size_t)WT * WT;           // doubles per tile
// Cholesky panel width / diagonal block size.
constexpr int NB = 64;
// Input dimensions.  D <= 3 (the transport use: planar / spatial positions) is the tuned path: source rows of 4 elements
// (x, y, z, 0), coordinates in registers.  3 < D <= MAX_DIMS is the wide path: rows of 8 (D <= 8) or 16 elements, coordinate
// loops, query coordinates staged through LDS in the variance kernel.  MAX_DIMS = 15: a query's k* column and its D
// derivative columns share one 16-column MFMA tile in the fused variance launch.
constexpr int MAX_DIMS = 15;
constexpr int MAX_D = 16;            // widest source row = size of every per-dimension array
constexpr int WIDE_D = 8;            // the narrower of the two wide layouts
inline int xs_stride(int D) { return D <= 3 ? 4 : (D <= WIDE_D ? WIDE_D : MAX_D); }
// the DW template argument of the kernels that carry coordinates: 3 (rows of 4), 8 or 16
inline int coord_width(int D) { return D <= 3 ? 3 : xs_stride(D); }
// Columns per query of the fused variance launch (k*, dk_0 .. dk_{D-1}, zero columns up to a power of two).
inline int var_fused_cols(int D) { return D <= 3 ? 4 : (D <= 7 ? 8 : 16); }
// `ncomp` codes of var_prepare / launch_var: 1 = k* alone; 3 = Jacobian variance alone, D columns per query (D <= 3);
// 4 / 8 / 16 = the fused layout (var_fused_cols); and the Jacobian variance alone at D = 4 and D = 8, the two wide
// dimensions where dropping the k* column halves the columns per query (4 and 8 instead of 8 and 16).
constexpr int VAR_NCOMP_DERIV4 = 104, VAR_NCOMP_DERIV8 = 108;
inline int var_cols_per_query(int D, int ncomp) { return ncomp == 3 ? D : (ncomp > 100 ? ncomp - 100 : ncomp); }

// Per-device one-time setup (hipFuncSetAttribute opt-ins, CU counts): a process may hold handles on several devices
// (gpt_create takes a device), so "done once" has to mean once per device, not once per process.
constexpr int MAX_DEVICES = 64;
inline int current_device() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= MAX_DEVICES) d = 0;
    return d;
}
struct PerDeviceOnce {
    std::once_flag flags[MAX_DEVICES];
    // f runs exactly once per device; callers arriving meanwhile (handles on other threads: the optimizer's concurrent
    // restarts) wait until it has finished, so nobody launches a kernel before its attributes are set
    template <class F> void run(F&& f) { std::call_once(flags[current_device()], f); }
};

// Model parameters passed by value to the prediction kernels.
struct KernelParams {
    double c;            // constant_value (prior variance)
    double lnc;          // log(constant_value)
    double noise;        // WhiteKernel noise_level
    double inv_ls[MAX_D];   // 1/length_scale per input dimension (unused dims: 0)
    int D;               // input dims (1..MAX_DIMS)
    int O;               // outputs
    int N;               // source points
    int NP;              // padded source points
    int ktype;           // 0 RBF, 1/2/3 Matern nu = 1/2, 3/2, 5/2 (gpt_exp.h)
    int ntask;           // stacked inverse factors (1: exact GP; T: SVGP exact conversion, one per task)
    int dtype;           // DT_F64 / DT_F32: element type of Xs, A4, Wf, queries and outputs
};

// ---- launchers (defined in the .hip files) --------------------------------------------
// fit
void launch_gram(hipStream_t s, const double* Xs, int D, int N, int NP, int ktype, double c, double diag_add, double* K);
void launch_add_lower(hipStream_t s, double* K, const double* S, int N, int NP);
// raw (N, D) sources on the device -> scaled, padded rows (fp64 workspace image; Xm, may be null: the model's copy in dtype)
void launch_scale_x(hipStream_t s, const double* X, int N, int NP, int D, const double* inv_ls /* host, MAX_D */, double* Xs64, void* Xm, int dtype);
void launch_dot(hipStream_t s, const double* a, const double* b, int64_t n, double* out);
// Streams and events of the blocked factor + inverse (gpt_fit_plan.h), owned by a handle: `side` and `chain` are confined to
// disjoint sets of CUs (hipExtStreamCreateWithCUMask).  Created on first use; `ok` false = unavailable, everything runs in
// the caller's stream in the plan's serial order.
constexpr int FIT_AUX_EVENTS = 256;
struct FitAux {
    bool tried = false, ok = false;
    int side_eighths = 0;          // CUs of `side` in eighths of the chip (the split the streams were made for)
    hipStream_t side = nullptr, chain = nullptr;
    hipEvent_t events[FIT_AUX_EVENTS] = {};
};
void fit_aux_release(FitAux& aux);
size_t factor_scratch_doubles(int NP);        // gpt_fit_plan.h
// L = chol(K) in place (lower), W = L^-1; scratch >= factor_scratch_doubles(NP); ev_factored (may be null) is recorded
// in `s` when L is complete
void launch_factor_inverse(hipStream_t s, double* K, double* W, int NP, int* info, double* scratch, FitAux* aux, hipEvent_t ev_factored);
void launch_alpha(hipStream_t s, const double* W, const double* Y4, int N, int NP, double* tmp4, double* A4,
                  double* scratch /* >= (NP/512)*NP*4 doubles */);
// W (row-major fp64, lower) * scale -> tile set `task` of the fragment-ordered stream Wf (element type dtype)
void launch_pack_w(hipStream_t s, const double* W, int N, int NP, void* Wf, int dtype, int task, double scale);
// fp64 [rows][4] image -> element type dtype, `ncol` columns starting at dst column `col0`, times scale
void launch_store4(hipStream_t s, const double* src4, int rows, void* dst4, int dtype, int src_col0, int dst_col0, int ncol, double scale);
void launch_logdet(hipStream_t s, const double* K, int N, int NP, double* out);
void launch_kinv(hipStream_t s, const double* W, int NP, double* Kout);
void launch_cov(hipStream_t s, const KernelParams& p, const double* Xs, const double* W, const double* Xq_dev, int64_t M,
                int Mp, double* KsT /* NP*Mp */, double* V /* NP*Mp */, double* VtV /* Mp*Mp */, double* cov_dev /* M*M */);
// out: [d/dlog c, d/dlog l_0 .. l_{MAX_D-1}, d/dlog noise (per unit noise)] = LML_TERMS doubles
constexpr int LML_TERMS = MAX_D + 2;
constexpr int LML_PARTIAL_STRIDE = 20;      // >= LML_TERMS
void launch_lml_terms(hipStream_t s, const double* Xs, int D, const double* A4, int npass, const double* Kinv, int N, int NP,
                      int O, int ktype, double c, double* partial /* (NP/64)^2*LML_PARTIAL_STRIDE doubles */, double* out /* LML_TERMS doubles */);
// predict (buffers in the model's element type)
void launch_mean_jac(hipStream_t s, const KernelParams& p, const void* Xs, const void* A4,
                     const void* Xq, int64_t M, void* mean, void* J);

// Scratch of the variance kernel, owned by a handle (grow-only) and the cached work plan of the last launch shape.
struct VarPlanHost;
struct VarWorkspace {
    void *slab = nullptr, *vslab = nullptr, *bscratch = nullptr, *plan_dev = nullptr;
    size_t slab_bytes = 0, vslab_bytes = 0, bscratch_bytes = 0, plan_bytes = 0;
    VarPlanHost* plan = nullptr;            // host copy of the plan resident in plan_dev
    int64_t key_cols = -1; int key_nbi = 0, key_ntask = 0, key_P = 0;
    int64_t allocs = 0;                     // number of (re)allocations so far (tests / gpt_reserve)
};
// Builds (or re-uses) the plan for M queries x `ncomp` columns and makes every buffer large enough; may synchronise
// `s` and reallocate.  Call before launch_var with the same arguments.
hipError_t var_prepare(VarWorkspace& ws, hipStream_t s, const KernelParams& p, int64_t M, int ncomp);
// hdr: the model blob's header (device), hdr[16 + t] = prior variance of task t.
void launch_var(hipStream_t s, const KernelParams& p, const VarWorkspace& ws, const void* Xs, const void* Wf,
                const void* Xq, int64_t M, int ncomp, void* var, void* Jvar, void* dvar, const double* hdr);
void var_release(VarWorkspace& ws);
int var_workgroups();

size_t wf_elems(int NP);          // elements of ONE task's tile set (+ prefetch overrun after the last task: wf_overrun_elems)
size_t wf_overrun_elems();

}  // namespace gpt
