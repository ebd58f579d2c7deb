// C ABI of libgpt_hip (see include/gpt_hip.h).  Host-side orchestration only: owns the device
// buffers, prepares scaled/padded inputs, sequences the kernels of gpt_fit.hip / gpt_predict.hip
// on one HIP stream and maps failures to error codes.
#include "gpt_common.h"
#include "../../include/gpt_hip.h"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

using namespace gpt;

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

#define HIPCHK(expr)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return fail(GPT_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));            \
    } while (0)

constexpr double MAGIC = 1196446769.0;   // "GPT1"
constexpr int HDR_DOUBLES = 64;
constexpr int64_t HOST_CHUNK = 1 << 17;  // queries per chunk of the host-pointer API (two chunks in flight)

struct Layout {
    int64_t N, NP;
    int D, O, npass;
    size_t off_xs, off_a4, off_wf, total;   // in doubles
};

Layout make_layout(int64_t N, int D, int O) {
    Layout l;
    l.N = N; l.D = D; l.O = O;
    l.NP = (N + PAD_N - 1) / PAD_N * PAD_N;
    l.npass = (O + 3) / 4;
    l.off_xs = HDR_DOUBLES;
    l.off_a4 = l.off_xs + (size_t)l.NP * 4;
    l.off_wf = l.off_a4 + (size_t)l.npass * l.NP * 4;
    l.total = l.off_wf + wf_doubles((int)l.NP);
    return l;
}

}  // namespace

struct gpt_handle {
    int device = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    // model blob
    double* blob = nullptr;
    Layout lay{};
    bool have_layout = false, committed = false;
    KernelParams p{};
    double jitter = 0, ls[3] = {1, 1, 1};
    int n_ls = 1;
    // fit workspace
    double *dK = nullptr, *dW = nullptr, *dY4 = nullptr, *dT4 = nullptr, *dscal = nullptr;
    int* dinfo = nullptr;
    int64_t ws_np = 0;
    int ws_npass = 0;
    bool have_factor_ws = false;   // dK/dW hold L / L^-1 of the current model
    std::vector<double> hostY;     // filtered targets (N,O) for the LML
    // staging of the host-pointer API
    // (two sets: while the results of one chunk travel to the host on `copy_stream`, the next chunk computes)
    struct Staging { double *q = nullptr, *mean = nullptr, *var = nullptr, *J = nullptr, *Jvar = nullptr, *dvar = nullptr; } st[2];
    int64_t scap = 0;
    int sD = 0, sO = 0;
    hipStream_t copy_stream = nullptr;
    hipEvent_t ev_done[2] = {}, ev_copied[2] = {};   // chunk computed / chunk's outputs copied out, per staging set
    double fit_ms[6] = {0, 0, 0, 0, 0, 0};
    hipEvent_t ev[7] = {};
    // per-kernel timing of the last predict (gpt_set_profiling)
    bool profiling = false, pred_mj = false, pred_var = false;
    double* slab = nullptr;        // partial column sums of the variance kernel (grow-only)
    size_t slab_cap = 0;
    double* bscratch = nullptr;    // per-workgroup B-fragment images of the variance kernel (grow-only)
    size_t bscratch_cap = 0;
    hipEvent_t pev[4] = {};

    double* dXs() const { return blob + lay.off_xs; }
    double* dA4() const { return blob + lay.off_a4; }
    double* dWf() const { return blob + lay.off_wf; }
};

namespace {

int set_device(gpt_handle* h) {
    HIPCHK(hipSetDevice(h->device));
    return GPT_OK;
}

void free_staging(gpt_handle* h) {
    for (auto& t : h->st) {
        double** ptrs[] = {&t.q, &t.mean, &t.var, &t.J, &t.Jvar, &t.dvar};
        for (auto pp : ptrs) { if (*pp) (void)hipFree(*pp); *pp = nullptr; }
    }
    h->scap = 0;
}

void free_workspace(gpt_handle* h) {
    double** ptrs[] = {&h->dK, &h->dW, &h->dY4, &h->dT4, &h->dscal};
    for (auto pp : ptrs) { if (*pp) (void)hipFree(*pp); *pp = nullptr; }
    if (h->dinfo) (void)hipFree(h->dinfo);
    h->dinfo = nullptr;
    h->ws_np = 0; h->ws_npass = 0; h->have_factor_ws = false;
}

int ensure_blob(gpt_handle* h, const Layout& l) {
    if (h->blob && h->have_layout && h->lay.total == l.total && h->lay.NP == l.NP && h->lay.npass == l.npass) {
        h->lay = l;
        return GPT_OK;
    }
    if (h->blob) { (void)hipFree(h->blob); h->blob = nullptr; }
    HIPCHK(hipMalloc(&h->blob, l.total * sizeof(double)));
    h->lay = l;
    h->have_layout = true;
    return GPT_OK;
}

int ensure_workspace(gpt_handle* h, int64_t NP, int npass) {
    if (h->ws_np == NP && h->ws_npass >= npass && h->dK) return GPT_OK;
    free_workspace(h);
    HIPCHK(hipMalloc(&h->dK, (size_t)NP * NP * sizeof(double)));
    HIPCHK(hipMalloc(&h->dW, (size_t)NP * NP * sizeof(double)));
    HIPCHK(hipMalloc(&h->dY4, (size_t)npass * NP * 4 * sizeof(double)));
    HIPCHK(hipMalloc(&h->dT4, (size_t)NP * 4 * sizeof(double)));
    HIPCHK(hipMalloc(&h->dscal, 8 * sizeof(double)));
    HIPCHK(hipMalloc(&h->dinfo, sizeof(int)));
    h->ws_np = NP; h->ws_npass = npass;
    return GPT_OK;
}

int ensure_staging(gpt_handle* h, int64_t cap, int D, int O) {
    if (h->scap >= cap && h->sD == D && h->sO == O) return GPT_OK;
    HIPCHK(hipStreamSynchronize(h->stream));
    if (h->copy_stream) HIPCHK(hipStreamSynchronize(h->copy_stream));
    free_staging(h);
    for (auto& t : h->st) {
        HIPCHK(hipMalloc(&t.q, (size_t)cap * D * sizeof(double)));
        HIPCHK(hipMalloc(&t.mean, (size_t)cap * O * sizeof(double)));
        HIPCHK(hipMalloc(&t.var, (size_t)cap * sizeof(double)));
        HIPCHK(hipMalloc(&t.J, (size_t)cap * O * D * sizeof(double)));
        HIPCHK(hipMalloc(&t.Jvar, (size_t)cap * D * sizeof(double)));
        HIPCHK(hipMalloc(&t.dvar, (size_t)cap * D * sizeof(double)));
    }
    if (!h->copy_stream) {
        HIPCHK(hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
        for (int i = 0; i < 2; ++i) {
            HIPCHK(hipEventCreateWithFlags(&h->ev_done[i], hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&h->ev_copied[i], hipEventDisableTiming));
        }
    }
    h->scap = cap; h->sD = D; h->sO = O;
    return GPT_OK;
}

void fill_params(gpt_handle* h, const double* hdr) {
    KernelParams& p = h->p;
    p.N = (int)hdr[1]; p.NP = (int)hdr[2]; p.D = (int)hdr[3]; p.O = (int)hdr[4];
    p.c = hdr[5]; p.noise = hdr[6];
    p.lnc = std::log(hdr[5]);
    p.ktype = (int)hdr[13];
    h->jitter = hdr[7];
    h->n_ls = (int)hdr[11];
    for (int d = 0; d < 3; ++d) {
        h->ls[d] = hdr[8 + d];
        p.inv_ls[d] = (d < p.D) ? 1.0 / hdr[8 + d] : 0.0;
    }
}

}  // namespace

extern "C" {

int gpt_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char* gpt_last_error(void) { return g_err.c_str(); }
const char* gpt_version(void) { return "gpt_hip 0.1 (gfx950)"; }

int gpt_create(gpt_handle** out, int device) {
    if (!out) return fail(GPT_E_ARG, "gpt_create: out is NULL");
    int n = gpt_device_count();
    if (device < 0 || device >= n) return fail(GPT_E_ARG, "gpt_create: no such HIP device");
    HIPCHK(hipSetDevice(device));
    gpt_handle* h = new gpt_handle();
    h->device = device;
    hipError_t e = hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete h; return fail(GPT_E_HIP, std::string("hipStreamCreate: ") + hipGetErrorString(e)); }
    h->stream = h->own_stream;
    for (auto& ev : h->ev) {
        e = hipEventCreate(&ev);
        if (e != hipSuccess) { delete h; return fail(GPT_E_HIP, std::string("hipEventCreate: ") + hipGetErrorString(e)); }
    }
    for (auto& ev : h->pev) {
        e = hipEventCreate(&ev);
        if (e != hipSuccess) { delete h; return fail(GPT_E_HIP, std::string("hipEventCreate: ") + hipGetErrorString(e)); }
    }
    *out = h;
    return GPT_OK;
}

void gpt_destroy(gpt_handle* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    if (h->copy_stream) (void)hipStreamSynchronize(h->copy_stream);
    free_staging(h);
    free_workspace(h);
    if (h->blob) (void)hipFree(h->blob);
    if (h->slab) (void)hipFree(h->slab);
    if (h->bscratch) (void)hipFree(h->bscratch);
    for (auto& ev : h->ev) if (ev) (void)hipEventDestroy(ev);
    for (auto& ev : h->pev) if (ev) (void)hipEventDestroy(ev);
    for (auto& ev : h->ev_done) if (ev) (void)hipEventDestroy(ev);
    for (auto& ev : h->ev_copied) if (ev) (void)hipEventDestroy(ev);
    if (h->copy_stream) (void)hipStreamDestroy(h->copy_stream);
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
    delete h;
}

int gpt_set_stream(gpt_handle* h, void* hip_stream) {
    if (!h) return fail(GPT_E_ARG, "gpt_set_stream: NULL handle");
    h->stream = hip_stream ? reinterpret_cast<hipStream_t>(hip_stream) : h->own_stream;
    return GPT_OK;
}

int gpt_synchronize(gpt_handle* h) {
    if (!h) return fail(GPT_E_ARG, "gpt_synchronize: NULL handle");
    if (int rc = set_device(h)) return rc;
    HIPCHK(hipStreamSynchronize(h->stream));
    return GPT_OK;
}

int gpt_fit(gpt_handle* h, const double* X, const double* Y, int64_t N, int D, int O,
            const double* length_scale, int n_ls, double constant_value, double noise_level,
            double alpha_jitter) {
    return gpt_fit_kernel(h, X, Y, N, D, O, length_scale, n_ls, constant_value, noise_level, alpha_jitter, GPT_KERNEL_RBF);
}

static int fit_impl(gpt_handle* h, const double* X, const double* Y, int64_t N, int D, int O,
                    const double* length_scale, int n_ls, double constant_value, double noise_level,
                    double alpha_jitter, int kernel_type, const double* Sigma);

int gpt_fit_kernel(gpt_handle* h, const double* X, const double* Y, int64_t N, int D, int O,
                   const double* length_scale, int n_ls, double constant_value, double noise_level,
                   double alpha_jitter, int kernel_type) {
    return fit_impl(h, X, Y, N, D, O, length_scale, n_ls, constant_value, noise_level, alpha_jitter, kernel_type, nullptr);
}

int gpt_fit_noise_matrix(gpt_handle* h, const double* X, const double* Y, int64_t N, int D, int O,
                         const double* length_scale, int n_ls, double constant_value, const double* Sigma,
                         double alpha_jitter, int kernel_type) {
    if (!Sigma) return fail(GPT_E_ARG, "gpt_fit_noise_matrix: Sigma is NULL");
    return fit_impl(h, X, Y, N, D, O, length_scale, n_ls, constant_value, 0.0, alpha_jitter, kernel_type, Sigma);
}

static int fit_impl(gpt_handle* h, const double* X, const double* Y, int64_t N, int D, int O,
                    const double* length_scale, int n_ls, double constant_value, double noise_level,
                    double alpha_jitter, int kernel_type, const double* Sigma) {
    if (!h || !X || !Y || !length_scale) return fail(GPT_E_ARG, "gpt_fit: NULL argument");
    if (kernel_type < GPT_KERNEL_RBF || kernel_type > GPT_KERNEL_MATERN52) return fail(GPT_E_ARG, "gpt_fit: unknown kernel_type");
    if (N < 1 || N > (1 << 20)) return fail(GPT_E_ARG, "gpt_fit: N out of range");
    if (D < 1 || D > 3) return fail(GPT_E_ARG, "gpt_fit: D must be 1, 2 or 3");
    if (O < 1) return fail(GPT_E_ARG, "gpt_fit: O must be >= 1");
    if (n_ls != 1 && n_ls != D) return fail(GPT_E_ARG, "gpt_fit: length_scale must have 1 or D entries");
    for (int d = 0; d < n_ls; ++d)
        if (!(length_scale[d] > 0.0)) return fail(GPT_E_ARG, "gpt_fit: length_scale must be > 0");
    if (!(constant_value > 0.0) || !(noise_level >= 0.0) || !(alpha_jitter >= 0.0))
        return fail(GPT_E_ARG, "gpt_fit: constant_value > 0, noise_level >= 0, alpha >= 0 required");
    if (int rc = set_device(h)) return rc;
    h->committed = false;
    const Layout l = make_layout(N, D, O);
    if (int rc = ensure_blob(h, l)) return rc;
    if (int rc = ensure_workspace(h, l.NP, l.npass)) return rc;
    const int NP = (int)l.NP;
    hipStream_t s = h->stream;

    // ---- host preparation: header, scaled + padded sources, padded targets
    std::vector<double> hdr(HDR_DOUBLES, 0.0);
    hdr[0] = MAGIC; hdr[1] = (double)N; hdr[2] = (double)NP; hdr[3] = D; hdr[4] = O;
    hdr[5] = constant_value; hdr[6] = noise_level; hdr[7] = alpha_jitter;
    for (int d = 0; d < 3; ++d) hdr[8 + d] = (d < D) ? length_scale[n_ls == 1 ? 0 : d] : 1.0;
    hdr[11] = n_ls; hdr[12] = l.npass; hdr[13] = kernel_type;
    fill_params(h, hdr.data());
    std::vector<double> xs((size_t)NP * 4, 0.0), y4((size_t)l.npass * NP * 4, 0.0);
    for (int64_t i = 0; i < N; ++i) {
        for (int d = 0; d < D; ++d) xs[(size_t)i * 4 + d] = X[i * D + d] * h->p.inv_ls[d];
        for (int o = 0; o < O; ++o) y4[((size_t)(o / 4) * NP + i) * 4 + (o % 4)] = Y[i * O + o];
    }
    h->hostY.assign(Y, Y + (size_t)N * O);
    HIPCHK(hipMemcpyAsync(h->blob, hdr.data(), HDR_DOUBLES * sizeof(double), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(h->dXs(), xs.data(), xs.size() * sizeof(double), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(h->dY4, y4.data(), y4.size() * sizeof(double), hipMemcpyHostToDevice, s));
    HIPCHK(hipStreamSynchronize(s));   // host vectors go out of scope below

    // ---- device pipeline
    HIPCHK(hipEventRecord(h->ev[0], s));
    HIPCHK(hipMemsetAsync(h->dinfo, 0, sizeof(int), s));
    HIPCHK(hipMemsetAsync(h->dW, 0, (size_t)NP * NP * sizeof(double), s));
    launch_gram(s, h->dXs(), (int)N, NP, kernel_type, constant_value, noise_level + alpha_jitter, h->dK);
    if (Sigma) {      // K += Sigma: staged through dW (not in use until the factorisation starts)
        HIPCHK(hipMemcpyAsync(h->dW, Sigma, (size_t)N * N * sizeof(double), hipMemcpyHostToDevice, s));
        launch_add_lower(s, h->dK, h->dW, (int)N, NP);
        HIPCHK(hipMemsetAsync(h->dW, 0, (size_t)NP * NP * sizeof(double), s));
    }
    HIPCHK(hipEventRecord(h->ev[1], s));
    launch_potrf(s, h->dK, h->dW, NP, h->dinfo);
    HIPCHK(hipEventRecord(h->ev[2], s));
    launch_trinv(s, h->dK, h->dW, NP, h->dWf());
    HIPCHK(hipEventRecord(h->ev[3], s));
    for (int ps = 0; ps < l.npass; ++ps)
        launch_alpha(s, h->dW, h->dY4 + (size_t)ps * NP * 4, (int)N, NP, h->dT4, h->dA4() + (size_t)ps * NP * 4, h->dWf());
    HIPCHK(hipEventRecord(h->ev[4], s));
    launch_pack_w(s, h->dW, (int)N, NP, h->dWf());
    HIPCHK(hipEventRecord(h->ev[5], s));
    HIPCHK(hipGetLastError());
    int info = 0;
    HIPCHK(hipMemcpyAsync(&info, h->dinfo, sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    float ms = 0;
    for (int i = 1; i <= 5; ++i) {
        HIPCHK(hipEventElapsedTime(&ms, h->ev[i - 1], h->ev[i]));
        h->fit_ms[i] = ms;
    }
    HIPCHK(hipEventElapsedTime(&ms, h->ev[0], h->ev[5]));
    h->fit_ms[0] = ms;
    if (info != 0) {
        h->have_factor_ws = false;
        char buf[160];
        snprintf(buf, sizeof buf, "gpt_fit: kernel matrix is not positive definite (pivot %d <= 0)", info);
        return fail(GPT_E_NOT_PD, buf);
    }
    h->committed = true;
    h->have_factor_ws = true;
    return GPT_OK;
}

namespace {
// grow-only scratch of the variance kernel; reallocation waits for work that may still use the old buffers
int ensure_var_scratch(gpt_handle* h, int64_t M, int ncomp) {
    hipStream_t s = h->stream;
    const size_t need = var_slab_doubles(M, ncomp);
    if (need > h->slab_cap) {
        HIPCHK(hipStreamSynchronize(s));
        if (h->slab) (void)hipFree(h->slab);
        h->slab = nullptr; h->slab_cap = 0;
        HIPCHK(hipMalloc(&h->slab, need * sizeof(double)));
        h->slab_cap = need;
    }
    const size_t needb = var_bscratch_doubles(h->p.NP);
    if (needb > h->bscratch_cap) {
        HIPCHK(hipStreamSynchronize(s));
        if (h->bscratch) (void)hipFree(h->bscratch);
        h->bscratch = nullptr; h->bscratch_cap = 0;
        HIPCHK(hipMalloc(&h->bscratch, needb * sizeof(double)));
        h->bscratch_cap = needb;
    }
    return GPT_OK;
}
}  // namespace

int gpt_reserve(gpt_handle* h, int64_t M, int jacobian_variance) {
    if (!h) return fail(GPT_E_ARG, "gpt_reserve: NULL handle");
    if (!h->committed) return fail(GPT_E_STATE, "gpt_reserve: model is not fitted");
    if (M < 0) return fail(GPT_E_ARG, "gpt_reserve: bad query count");
    if (int rc = set_device(h)) return rc;
    return M == 0 ? GPT_OK : ensure_var_scratch(h, M, jacobian_variance ? 4 : 1);
}

int gpt_predict_all_dev(gpt_handle* h, const double* Xq, int64_t M, double* mean, double* var,
                        double* J, double* Jvar, double* dvar) {
    if (!h) return fail(GPT_E_ARG, "gpt_predict_all_dev: NULL handle");
    if (!h->committed) return fail(GPT_E_STATE, "predict: model is not fitted");
    if (M < 0 || (M > 0 && !Xq)) return fail(GPT_E_ARG, "predict: bad query buffer");
    if (M == 0) return GPT_OK;
    if ((J || Jvar || dvar) && h->p.ktype != GPT_KERNEL_RBF)
        return fail(GPT_E_ARG, "derivative / Jacobian variance / d variance are defined for the RBF kernel only");
    if (int rc = set_device(h)) return rc;
    hipStream_t s = h->stream;
    const bool prof = h->profiling;
    h->pred_mj = (mean || J);
    h->pred_var = (var || Jvar || dvar);
    if (h->pred_mj) {
        if (prof) HIPCHK(hipEventRecord(h->pev[0], s));
        launch_mean_jac(s, h->p, h->dXs(), h->dA4(), Xq, M, mean, J);
        if (prof) HIPCHK(hipEventRecord(h->pev[1], s));
    }
    if (h->pred_var) {
        const int ncomp = dvar ? 4 : (Jvar ? (var ? 4 : 3) : 1);     // 3: Jacobian variance alone (no k* column)
        if (int rc = ensure_var_scratch(h, M, ncomp)) return rc;
        if (prof) HIPCHK(hipEventRecord(h->pev[2], s));
        if (ncomp == 1) launch_var(s, h->p, h->dXs(), h->dWf(), Xq, M, 1, var, nullptr, nullptr, h->slab, h->bscratch);
        else launch_var(s, h->p, h->dXs(), h->dWf(), Xq, M, ncomp, var, Jvar, dvar, h->slab, h->bscratch);
        if (prof) HIPCHK(hipEventRecord(h->pev[3], s));
    }
    HIPCHK(hipGetLastError());
    return GPT_OK;
}

int gpt_predict_all(gpt_handle* h, const double* Xq, int64_t M, double* mean, double* var,
                    double* J, double* Jvar, double* dvar) {
    if (!h) return fail(GPT_E_ARG, "gpt_predict_all: NULL handle");
    if (!h->committed) return fail(GPT_E_STATE, "predict: model is not fitted");
    if (M < 0 || (M > 0 && !Xq)) return fail(GPT_E_ARG, "predict: bad query buffer");
    if (M == 0) return GPT_OK;
    if (int rc = set_device(h)) return rc;
    const int D = h->p.D, O = h->p.O;
    const int64_t cap = M < HOST_CHUNK ? M : HOST_CHUNK;
    if (int rc = ensure_staging(h, cap, D, O)) return rc;
    hipStream_t s = h->stream, cs = h->copy_stream;
    const int64_t nchunks = (M + cap - 1) / cap;
    // chunk i computes on `s` in staging set i & 1; its outputs leave on `cs` while chunk i + 1 computes
    auto enqueue = [&](int64_t i) -> int {
        const int b = (int)(i & 1);
        const int64_t off = i * cap, m = (M - off) < cap ? (M - off) : cap;
        gpt_handle::Staging& t = h->st[b];
        if (i >= 2) HIPCHK(hipStreamWaitEvent(s, h->ev_copied[b], 0));          // set b is free again
        HIPCHK(hipMemcpyAsync(t.q, Xq + off * D, (size_t)m * D * sizeof(double), hipMemcpyHostToDevice, s));
        if (int rc = gpt_predict_all_dev(h, t.q, m, mean ? t.mean : nullptr, var ? t.var : nullptr, J ? t.J : nullptr,
                                         Jvar ? t.Jvar : nullptr, dvar ? t.dvar : nullptr))
            return rc;
        HIPCHK(hipEventRecord(h->ev_done[b], s));
        return GPT_OK;
    };
    auto copy_out = [&](int64_t i) -> int {
        const int b = (int)(i & 1);
        const int64_t off = i * cap, m = (M - off) < cap ? (M - off) : cap;
        const gpt_handle::Staging& t = h->st[b];
        HIPCHK(hipStreamWaitEvent(cs, h->ev_done[b], 0));
        if (mean) HIPCHK(hipMemcpyAsync(mean + off * O, t.mean, (size_t)m * O * sizeof(double), hipMemcpyDeviceToHost, cs));
        if (var) HIPCHK(hipMemcpyAsync(var + off, t.var, (size_t)m * sizeof(double), hipMemcpyDeviceToHost, cs));
        if (J) HIPCHK(hipMemcpyAsync(J + off * O * D, t.J, (size_t)m * O * D * sizeof(double), hipMemcpyDeviceToHost, cs));
        if (Jvar) HIPCHK(hipMemcpyAsync(Jvar + off * D, t.Jvar, (size_t)m * D * sizeof(double), hipMemcpyDeviceToHost, cs));
        if (dvar)
            for (int d = 0; d < D; ++d)   // device chunk is (D, m); host result is (D, M)
                HIPCHK(hipMemcpyAsync(dvar + (size_t)d * M + off, t.dvar + (size_t)d * m, (size_t)m * sizeof(double),
                                      hipMemcpyDeviceToHost, cs));
        HIPCHK(hipEventRecord(h->ev_copied[b], cs));
        return GPT_OK;
    };
    if (int rc = enqueue(0)) return rc;
    for (int64_t i = 0; i < nchunks; ++i) {
        if (i + 1 < nchunks) { if (int rc = enqueue(i + 1)) return rc; }   // queued before chunk i's (host-blocking) copies
        if (int rc = copy_out(i)) return rc;
    }
    HIPCHK(hipStreamSynchronize(cs));
    HIPCHK(hipStreamSynchronize(s));
    return GPT_OK;
}

int gpt_predict(gpt_handle* h, const double* Xq, int64_t M, double* mean, double* var) {
    return gpt_predict_all(h, Xq, M, mean, var, nullptr, nullptr, nullptr);
}

int gpt_derivative(gpt_handle* h, const double* Xq, int64_t M, double* J, double* Jvar) {
    return gpt_predict_all(h, Xq, M, nullptr, nullptr, J, Jvar, nullptr);
}

int gpt_dvariance(gpt_handle* h, const double* Xq, int64_t M, double* g) {
    return gpt_predict_all(h, Xq, M, nullptr, nullptr, nullptr, nullptr, g);
}

int gpt_export(gpt_handle* h, double* L, double* alpha) {
    if (!h) return fail(GPT_E_ARG, "gpt_export: NULL handle");
    if (!h->committed) return fail(GPT_E_STATE, "gpt_export: model is not fitted");
    if (int rc = set_device(h)) return rc;
    const int64_t N = h->p.N, NP = h->p.NP;
    const int O = h->p.O;
    HIPCHK(hipStreamSynchronize(h->stream));
    if (L) {
        if (!h->have_factor_ws) return fail(GPT_E_STATE, "gpt_export: L is only available on the rank that ran gpt_fit");
        HIPCHK(hipMemcpy2D(L, (size_t)N * sizeof(double), h->dK, (size_t)NP * sizeof(double), (size_t)N * sizeof(double),
                           (size_t)N, hipMemcpyDeviceToHost));
        for (int64_t i = 0; i < N; ++i)
            for (int64_t j = i + 1; j < N; ++j) L[i * N + j] = 0.0;
    }
    if (alpha) {
        std::vector<double> a4((size_t)h->lay.npass * NP * 4);
        HIPCHK(hipMemcpy(a4.data(), h->dA4(), a4.size() * sizeof(double), hipMemcpyDeviceToHost));
        for (int64_t i = 0; i < N; ++i)
            for (int o = 0; o < O; ++o) alpha[i * O + o] = a4[((size_t)(o / 4) * NP + i) * 4 + (o % 4)];
    }
    return GPT_OK;
}

int gpt_export_inverse_factor(gpt_handle* h, double* W) {
    if (!h || !W) return fail(GPT_E_ARG, "gpt_export_inverse_factor: NULL argument");
    if (!h->committed || !h->have_factor_ws) return fail(GPT_E_STATE, "gpt_export_inverse_factor: no factor on this handle");
    if (int rc = set_device(h)) return rc;
    const int64_t N = h->p.N, NP = h->p.NP;
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy2D(W, (size_t)N * sizeof(double), h->dW, (size_t)NP * sizeof(double), (size_t)N * sizeof(double),
                       (size_t)N, hipMemcpyDeviceToHost));
    return GPT_OK;
}

int gpt_lml(gpt_handle* h, double* lml) {
    if (!h || !lml) return fail(GPT_E_ARG, "gpt_lml: NULL argument");
    if (!h->committed || !h->have_factor_ws) return fail(GPT_E_STATE, "gpt_lml: needs the handle that ran gpt_fit");
    if (int rc = set_device(h)) return rc;
    const int64_t N = h->p.N, NP = h->p.NP;
    const int O = h->p.O;
    launch_logdet(h->stream, h->dK, (int)N, (int)NP, h->dscal);
    double logdet = 0;
    HIPCHK(hipMemcpyAsync(&logdet, h->dscal, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    std::vector<double> a4((size_t)h->lay.npass * NP * 4);
    HIPCHK(hipMemcpyAsync(a4.data(), h->dA4(), a4.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    double total = 0;
    for (int o = 0; o < O; ++o) {
        double ya = 0;
        for (int64_t i = 0; i < N; ++i) ya += h->hostY[i * O + o] * a4[((size_t)(o / 4) * NP + i) * 4 + (o % 4)];
        total += -0.5 * ya - logdet - 0.5 * (double)N * std::log(2.0 * M_PI);
    }
    *lml = total;
    return GPT_OK;
}

int gpt_predict_cov(gpt_handle* h, const double* Xq, int64_t M, double* mean, double* cov) {
    if (!h || !cov) return fail(GPT_E_ARG, "gpt_predict_cov: NULL argument");
    if (!h->committed) return fail(GPT_E_STATE, "predict: model is not fitted");
    if (!h->dW || h->ws_np != h->p.NP) return fail(GPT_E_STATE, "gpt_predict_cov: needs the handle that ran gpt_fit");
    if (M < 0 || (M > 0 && !Xq)) return fail(GPT_E_ARG, "predict: bad query buffer");
    if (M > 16384) return fail(GPT_E_ARG, "gpt_predict_cov: M x M covariance limited to M <= 16384");
    if (M == 0) return GPT_OK;
    if (int rc = set_device(h)) return rc;
    if (mean) { if (int rc = gpt_predict_all(h, Xq, M, mean, nullptr, nullptr, nullptr, nullptr)) return rc; }
    const int D = h->p.D;
    const int64_t NP = h->p.NP;
    const int Mp = (int)((M + 127) / 128 * 128);
    hipStream_t s = h->stream;
    double *dq = nullptr, *KsT = nullptr, *V = nullptr, *VtV = nullptr, *dcov = nullptr;
    auto cleanup = [&]() { for (double* ptr : {dq, KsT, V, VtV, dcov}) if (ptr) (void)hipFree(ptr); };
    hipError_t e = hipSuccess;
    if ((e = hipMalloc(&dq, (size_t)M * D * sizeof(double))) == hipSuccess &&
        (e = hipMalloc(&KsT, (size_t)NP * Mp * sizeof(double))) == hipSuccess &&
        (e = hipMalloc(&V, (size_t)NP * Mp * sizeof(double))) == hipSuccess &&
        (e = hipMalloc(&VtV, (size_t)Mp * Mp * sizeof(double))) == hipSuccess &&
        (e = hipMalloc(&dcov, (size_t)M * M * sizeof(double))) == hipSuccess &&
        (e = hipMemcpyAsync(dq, Xq, (size_t)M * D * sizeof(double), hipMemcpyHostToDevice, s)) == hipSuccess) {
        launch_cov(s, h->p, h->dXs(), h->dW, dq, M, Mp, KsT, V, VtV, dcov);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(cov, dcov, (size_t)M * M * sizeof(double), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
    }
    cleanup();
    if (e != hipSuccess) return fail(GPT_E_HIP, std::string("gpt_predict_cov: ") + hipGetErrorString(e));
    return GPT_OK;
}

int gpt_lml_gradient(gpt_handle* h, double* lml, double* grad) {
    if (!h || !lml || !grad) return fail(GPT_E_ARG, "gpt_lml_gradient: NULL argument");
    if (!h->committed || !h->have_factor_ws) return fail(GPT_E_STATE, "gpt_lml_gradient: needs the handle that ran gpt_fit");
    if (int rc = gpt_lml(h, lml)) return rc;                    // uses diag(L) in dK before it is overwritten
    const int64_t N = h->p.N, NP = h->p.NP;
    const int D = h->p.D, O = h->p.O;
    hipStream_t s = h->stream;
    // K^-1 (lower) into dK, partial sums into the (currently idle) slab/partial scratch
    const size_t need = (size_t)(NP / 64) * (NP / 64) * 8;
    if (need > h->slab_cap) {
        HIPCHK(hipStreamSynchronize(s));
        if (h->slab) (void)hipFree(h->slab);
        h->slab = nullptr; h->slab_cap = 0;
        HIPCHK(hipMalloc(&h->slab, need * sizeof(double)));
        h->slab_cap = need;
    }
    h->have_factor_ws = false;                                   // L is gone: gpt_export(L) needs a new gpt_fit
    launch_kinv(s, h->dW, (int)NP, h->dK);
    launch_lml_terms(s, h->dXs(), h->dA4(), h->lay.npass, h->dK, (int)N, (int)NP, O, h->p.ktype, h->p.c, h->slab, h->dscal);
    HIPCHK(hipGetLastError());
    double S[5];
    HIPCHK(hipMemcpyAsync(S, h->dscal, sizeof S, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    // theta = log [constant_value, length_scale (1 or D), noise_level]
    grad[0] = 0.5 * S[0];
    if (h->n_ls == 1) {
        double t = 0;
        for (int d = 0; d < D; ++d) t += S[1 + d];
        grad[1] = 0.5 * t;
    } else {
        for (int d = 0; d < D; ++d) grad[1 + d] = 0.5 * S[1 + d];
    }
    grad[1 + h->n_ls] = 0.5 * h->p.noise * S[4];
    return GPT_OK;
}

int gpt_factor_blob(gpt_handle* h, void** dev_ptr, size_t* bytes) {
    if (!h || !dev_ptr || !bytes) return fail(GPT_E_ARG, "gpt_factor_blob: NULL argument");
    if (!h->blob || !h->have_layout) return fail(GPT_E_STATE, "gpt_factor_blob: no model storage");
    *dev_ptr = h->blob;
    *bytes = h->lay.total * sizeof(double);
    return GPT_OK;
}

int gpt_factor_alloc(gpt_handle* h, int64_t N, int D, int O, void** dev_ptr, size_t* bytes) {
    if (!h || !dev_ptr || !bytes) return fail(GPT_E_ARG, "gpt_factor_alloc: NULL argument");
    if (N < 1 || D < 1 || D > 3 || O < 1) return fail(GPT_E_ARG, "gpt_factor_alloc: bad geometry");
    if (int rc = set_device(h)) return rc;
    h->committed = false;
    h->have_factor_ws = false;
    if (int rc = ensure_blob(h, make_layout(N, D, O))) return rc;
    *dev_ptr = h->blob;
    *bytes = h->lay.total * sizeof(double);
    return GPT_OK;
}

int gpt_factor_commit(gpt_handle* h) {
    if (!h) return fail(GPT_E_ARG, "gpt_factor_commit: NULL handle");
    if (!h->blob || !h->have_layout) return fail(GPT_E_STATE, "gpt_factor_commit: no model storage");
    if (int rc = set_device(h)) return rc;
    double hdr[HDR_DOUBLES];
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(hdr, h->blob, sizeof hdr, hipMemcpyDeviceToHost));
    if (hdr[0] != MAGIC) return fail(GPT_E_STATE, "gpt_factor_commit: blob has no model header (broadcast missing?)");
    if ((int64_t)hdr[1] != h->lay.N || (int)hdr[3] != h->lay.D || (int)hdr[4] != h->lay.O || (int64_t)hdr[2] != h->lay.NP)
        return fail(GPT_E_STATE, "gpt_factor_commit: header geometry differs from gpt_factor_alloc");
    fill_params(h, hdr);
    h->committed = true;
    return GPT_OK;
}

int gpt_info(gpt_handle* h, int64_t* N, int* D, int* O, int64_t* N_padded) {
    if (!h) return fail(GPT_E_ARG, "gpt_info: NULL handle");
    if (!h->committed) return fail(GPT_E_STATE, "gpt_info: model is not fitted");
    if (N) *N = h->p.N;
    if (D) *D = h->p.D;
    if (O) *O = h->p.O;
    if (N_padded) *N_padded = h->p.NP;
    return GPT_OK;
}

int gpt_set_profiling(gpt_handle* h, int enable) {
    if (!h) return fail(GPT_E_ARG, "gpt_set_profiling: NULL handle");
    h->profiling = enable != 0;
    return GPT_OK;
}

int gpt_predict_timings(gpt_handle* h, double* ms_out) {
    if (!h || !ms_out) return fail(GPT_E_ARG, "gpt_predict_timings: NULL argument");
    if (!h->profiling) return fail(GPT_E_STATE, "gpt_predict_timings: profiling is off (gpt_set_profiling)");
    if (int rc = set_device(h)) return rc;
    ms_out[0] = ms_out[1] = 0.0;
    float ms = 0;
    if (h->pred_mj) {
        HIPCHK(hipEventSynchronize(h->pev[1]));
        HIPCHK(hipEventElapsedTime(&ms, h->pev[0], h->pev[1]));
        ms_out[0] = ms;
    }
    if (h->pred_var) {
        HIPCHK(hipEventSynchronize(h->pev[3]));
        HIPCHK(hipEventElapsedTime(&ms, h->pev[2], h->pev[3]));
        ms_out[1] = ms;
    }
    return GPT_OK;
}

int gpt_fit_timings(gpt_handle* h, double* ms_out, int n) {
    if (!h || !ms_out) return fail(GPT_E_ARG, "gpt_fit_timings: NULL argument");
    for (int i = 0; i < n && i < 6; ++i) ms_out[i] = h->fit_ms[i];
    return GPT_OK;
}

}  // extern "C"
