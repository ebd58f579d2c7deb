// Shared declarations of the gfx950 GP-transportation library (internal header).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstddef>

namespace gpt {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

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

// Model parameters passed by value to the prediction kernels.
struct KernelParams {
    double c;            // constant_value (prior variance)
    double lnc;          // log(constant_value)
    double noise;        // WhiteKernel noise_level
    double inv_ls[3];    // 1/length_scale per input dimension (unused dims: 0)
    int D;               // input dims (1..3)
    int O;               // outputs
    int N;               // source points
    int NP;              // padded source points
    int ktype;           // 0 RBF, 1/2/3 Matern nu = 1/2, 3/2, 5/2 (gpt_exp.h)
};

// ---- launchers (defined in the .hip files) --------------------------------------------
// fit
void launch_gram(hipStream_t s, const double* Xs, int N, int NP, int ktype, double c, double diag_add, double* K);
void launch_add_lower(hipStream_t s, double* K, const double* S, int N, int NP);
void launch_potrf(hipStream_t s, double* K, double* W, int NP, int* info);
void launch_trinv(hipStream_t s, const double* L, double* W, int NP, double* scratch /* >= NP*NP/4 doubles */);
void launch_alpha(hipStream_t s, const double* W, const double* Y4, int N, int NP, double* tmp4, double* A4,
                  double* scratch /* >= (NP/512)*NP*4 doubles */);
void launch_pack_w(hipStream_t s, const double* W, int N, int NP, double* Wf);
void launch_logdet(hipStream_t s, const double* K, int N, int NP, double* out);
void launch_kinv(hipStream_t s, const double* W, int NP, double* Kout);
void launch_cov(hipStream_t s, const KernelParams& p, const double* Xs, const double* W, const double* Xq_dev, int64_t M,
                int Mp, double* KsT /* NP*Mp */, double* V /* NP*Mp */, double* VtV /* Mp*Mp */, double* cov_dev /* M*M */);
void launch_lml_terms(hipStream_t s, const double* Xs, const double* A4, int npass, const double* Kinv, int N, int NP,
                      int O, int ktype, double c, double* partial /* (NP/64)^2*8 doubles */, double* out /* 5 doubles */);
// predict
void launch_mean_jac(hipStream_t s, const KernelParams& p, const double* Xs, const double* A4,
                     const double* Xq, int64_t M, double* mean, double* J);
// `slab`: scratch of var_slab_doubles(M, ncomp) doubles (per-piece partial column sums);
// `bscratch`: var_bscratch_doubles(NP) doubles (per-workgroup image of the generated B fragments)
void launch_var(hipStream_t s, const KernelParams& p, const double* Xs, const double* Wf,
                const double* Xq, int64_t M, int ncomp, double* var, double* Jvar, double* dvar, double* slab,
                double* bscratch);
size_t var_slab_doubles(int64_t M, int ncomp);
size_t var_bscratch_doubles(int NP);

size_t wf_doubles(int NP);

}  // namespace gpt
