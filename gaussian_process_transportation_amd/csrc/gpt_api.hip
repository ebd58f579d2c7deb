// C ABI of libgpt_hip (see include/gpt_hip.h).  Host-side orchestration only: owns the device
// buffers, prepares scaled/padded inputs, sequences the kernels of gpt_fit.hip / gpt_predict.hip
// on one HIP stream and maps failures to error codes.
#include "gpt_common.h"
#include "gpt_plan.h"
#include "gpt_fit_plan.h"
#include "../../include/gpt_hip.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
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

constexpr double MAGIC = 1196446770.0;   // "GPT2": header layout of this version
constexpr int HDR_DOUBLES = 64;
constexpr int HDR_TASK_C = 16;           // hdr[16 + t]: prior variance of task t (constant_value / outputscale)
constexpr int MAX_TASKS = 32;
constexpr int HDR_LS = 8;                // hdr[8 + d], d < 3: length-scale of dimension d
constexpr int HDR_LS_HI = 48;            // hdr[48 + d - 3], 3 <= d < MAX_DIMS: the further dimensions of the wide layouts (slots 48 .. 59)
inline int hdr_ls_slot(int d) { return d < 3 ? HDR_LS + d : HDR_LS_HI + d - 3; }
constexpr int64_t HOST_CHUNK = 1 << 17;  // queries per chunk of the host-pointer API (two chunks in flight)

// Model blob: [header: 64 doubles][Xs: NP x 4 (D <= 3) or NP x 8][A4: npass x NP x 4][Wf: ntask tile sets + overrun],
// the three arrays in the model's element type.  Offsets in bytes.
struct Layout {
    int64_t N = 0, NP = 0;
    int D = 0, O = 0, npass = 0, ntask = 1, dtype = DT_F64;
    size_t esz = 8, off_xs = 0, off_a4 = 0, off_wf = 0, total = 0;
    bool same_shape(const Layout& o) const { return total == o.total && NP == o.NP && npass == o.npass && ntask == o.ntask && dtype == o.dtype; }
};

Layout make_layout(int64_t N, int D, int O, int ntask, int dtype) {
    Layout l;
    l.N = N; l.D = D; l.O = O; l.ntask = ntask; l.dtype = dtype;
    l.esz = dtype == DT_F32 ? sizeof(float) : sizeof(double);
    l.NP = (N + PAD_N - 1) / PAD_N * PAD_N;
    l.npass = (O + 3) / 4;
    l.off_xs = HDR_DOUBLES * sizeof(double);
    l.off_a4 = l.off_xs + (size_t)l.NP * xs_stride(D) * l.esz;
    l.off_wf = l.off_a4 + (size_t)l.npass * l.NP * 4 * l.esz;
    l.total = l.off_wf + ((size_t)ntask * wf_elems((int)l.NP) + wf_overrun_elems()) * l.esz;
    return l;
}

enum { ST_Q = 0, ST_MEAN, ST_VAR, ST_J, ST_JVAR, ST_DVAR, ST_COUNT };

}  // namespace

// ---- host side of the variance kernel: workgroup count, scratch buffers and the (cached) work plan of a launch shape
namespace gpt {

int var_workgroups() {
    static std::atomic<int> n[MAX_DEVICES] = {};       // (threads may race to fill it: they compute the same value)
    const int dev = current_device();
    if (n[dev].load() == 0) {
        int v = 0;
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) == hipSuccess) v = prop.multiProcessorCount;
        if (v <= 0) v = 256;
        const char* e = getenv("GPT_VAR_WGS");
        if (e && atoi(e) > 0) v = atoi(e);
        n[dev].store(v);
    }
    return n[dev].load();
}

static int var_cols_per_query(const KernelParams& p, int ncomp) { return gpt::var_cols_per_query(p.D, ncomp); }   // codes: gpt_common.h

void var_release(VarWorkspace& ws) {
    for (void** pp : {&ws.slab, &ws.vslab, &ws.bscratch, &ws.plan_dev}) { if (*pp) (void)hipFree(*pp); *pp = nullptr; }
    ws.slab_bytes = ws.vslab_bytes = ws.bscratch_bytes = ws.plan_bytes = 0;
    delete ws.plan;
    ws.plan = nullptr;
    ws.key_cols = -1;
}

hipError_t var_prepare(VarWorkspace& ws, hipStream_t s, const KernelParams& p, int64_t M, int ncomp) {
    const size_t esz = p.dtype == DT_F32 ? sizeof(float) : sizeof(double);
    const int P = var_workgroups();
    const int nbi = p.NP / WT;
    const int64_t ncols = M * var_cols_per_query(p, ncomp);
    bool synced = false;
    auto grow = [&](void*& buf, size_t& have, size_t need) -> hipError_t {
        if (need <= have) return hipSuccess;
        if (!synced) { if (hipError_t e = hipStreamSynchronize(s)) return e; synced = true; }   // work in flight may still use the old buffers
        if (buf) (void)hipFree(buf);
        buf = nullptr; have = 0;
        if (hipError_t e = hipMalloc(&buf, need)) return e;
        have = need;
        ++ws.allocs;
        return hipSuccess;
    };
    const bool same = ws.plan && ws.key_cols == ncols && ws.key_nbi == nbi && ws.key_ntask == p.ntask && ws.key_P == P;
    if (!same) {
        int order_env = -1;                                  // diagnostic override of the tail order (README)
        if (const char* e = getenv("GPT_VAR_TAIL_ORDER")) order_env = atoi(e);
        VarPlanHost* np = new VarPlanHost(build_var_plan(ncols, nbi, p.ntask, P, order_env));
        // device image: [item_begin (P+1) | items | fin | splits], each part 16-byte aligned
        auto al = [](size_t b) { return (b + 15) / 16 * 16; };
        const size_t b0 = al((size_t)(P + 1) * sizeof(int)), b1 = al(np->items.size() * sizeof(VarItem)),
                     b2 = al(np->fin.size() * sizeof(int)), b3 = al(np->splits.size() * sizeof(VarSplit));
        const size_t total = b0 + b1 + b2 + b3 + 16;
        // the previous plan may still be read by a launch in flight: wait before overwriting it
        if (!synced) { if (hipError_t e = hipStreamSynchronize(s)) { delete np; return e; } synced = true; }
        if (hipError_t e = grow(ws.plan_dev, ws.plan_bytes, total)) { delete np; return e; }
        std::vector<unsigned char> img(total, 0);
        memcpy(img.data(), np->item_begin.data(), (size_t)(P + 1) * sizeof(int));
        if (!np->items.empty()) memcpy(img.data() + b0, np->items.data(), np->items.size() * sizeof(VarItem));
        if (!np->fin.empty()) memcpy(img.data() + b0 + b1, np->fin.data(), np->fin.size() * sizeof(int));
        if (!np->splits.empty()) memcpy(img.data() + b0 + b1 + b2, np->splits.data(), np->splits.size() * sizeof(VarSplit));
        if (hipError_t e = hipMemcpy(ws.plan_dev, img.data(), total, hipMemcpyHostToDevice)) { delete np; return e; }
        unsigned char* base = static_cast<unsigned char*>(ws.plan_dev);
        np->d.item_begin = reinterpret_cast<const int*>(base);
        np->d.items = reinterpret_cast<const VarItem*>(base + b0);
        np->d.fin = reinterpret_cast<const int*>(base + b0 + b1);
        np->d.splits = reinterpret_cast<const VarSplit*>(base + b0 + b1 + b2);
        delete ws.plan;
        ws.plan = np;
        ws.key_cols = ncols; ws.key_nbi = nbi; ws.key_ntask = p.ntask; ws.key_P = P;
    }
    if (hipError_t e = grow(ws.slab, ws.slab_bytes, (size_t)(ws.plan->n_slots + 1) * VAR_SLOT * esz)) return e;
    if (hipError_t e = grow(ws.vslab, ws.vslab_bytes, (size_t)ws.plan->n_vslots * VAR_VSLOT * esz)) return e;
    if (hipError_t e = grow(ws.bscratch, ws.bscratch_bytes, (size_t)P * p.NP * VAR_COLS * esz)) return e;
    return hipSuccess;
}

}  // namespace gpt

struct gpt_handle {
    int device = 0;
    int dtype_next = DT_F64;       // element type of models fitted from now on (gpt_set_dtype)
    hipStream_t own_stream = nullptr, stream = nullptr;
    // model blob
    unsigned char* blob = nullptr;
    Layout lay{};
    bool have_layout = false, committed = false;
    KernelParams p{};
    double jitter = 0, ls[MAX_D] = {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1};
    int n_ls = 1;
    // fit workspace (fp64)
    double *dK = nullptr, *dW = nullptr, *dY4 = nullptr, *dT4 = nullptr, *dTa = nullptr, *dXs64 = nullptr, *dA64 = nullptr, *dscal = nullptr;
    double* dXraw = nullptr;       // the raw (N, D) sources as uploaded; scaled on the device for every new set of length-scales
    double* dScr = nullptr;        // arena of the factor + inverse (gpt_fit_plan.h) and scratch of alpha's backward pass
    size_t scr_doubles = 0;        // its capacity
    int* dinfo = nullptr;
    int64_t ws_np = 0;
    int ws_npass = 0;
    bool have_L = false;           // dK holds L of the committed model (gpt_export, gpt_lml)
    bool have_W = false;           // dW holds L^-1 of the committed model (gpt_predict_cov, gpt_lml_gradient, gpt_export_inverse_factor)
    bool objective_ready = false;  // gpt_lml_objective: factor and alpha in the workspace, no committed model
    // Host mirrors of what dXraw / dY4 hold, (N, D) and (N, O): a fit with the same X / Y (every evaluation of the
    // optimizer's objective) uploads nothing.  Cleared whenever the device copies are lost or overwritten.
    std::vector<double> hostX, hostY, hostY4;
    int hostX_D = 0, hostY_O = 0;
    double host_scal[2 + LML_TERMS + 4] = {};     // landing area of the per-fit scalar read-back
    double host_hdr[HDR_DOUBLES] = {};
    int host_info = 0;
    // staging of the host-pointer API, grow-only per buffer
    // (two sets: while the results of one chunk travel to the host on `copy_stream`, the next chunk computes)
    struct Staging { void* buf[ST_COUNT] = {}; size_t bytes[ST_COUNT] = {}; } st[2];
    hipStream_t copy_stream = nullptr;
    hipEvent_t ev_done[2] = {}, ev_copied[2] = {};   // chunk computed / chunk's outputs copied out, per staging set
    double fit_ms[6] = {0, 0, 0, 0, 0, 0};
    hipEvent_t ev[7] = {};
    // per-kernel timing of the last predict (gpt_set_profiling)
    bool profiling = false, pred_mj = false, pred_var = false;
    VarWorkspace vws;              // scratch + cached plan of the variance kernel
    FitAux fit_aux;                // CU-masked streams + events of the factor + inverse plan (gpt_fit_plan.h)
    double* lml_partial = nullptr; // partial sums of the LML gradient (grow-only)
    size_t lml_partial_cap = 0;
    unsigned char* cov_buf = nullptr;   // scratch of gpt_predict_cov (grow-only)
    size_t cov_cap = 0;
    hipEvent_t pev[4] = {};

    void* dXs() const { return blob + lay.off_xs; }
    void* dA4() const { return blob + lay.off_a4; }
    void* dWf() const { return blob + lay.off_wf; }
    const double* dHdr() const { return reinterpret_cast<const double*>(blob); }
};

namespace {

int set_device(gpt_handle* h) {
    HIPCHK(hipSetDevice(h->device));
    return GPT_OK;
}

void free_staging(gpt_handle* h) {
    for (auto& t : h->st)
        for (int i = 0; i < ST_COUNT; ++i) { if (t.buf[i]) (void)hipFree(t.buf[i]); t.buf[i] = nullptr; t.bytes[i] = 0; }
}

void free_workspace(gpt_handle* h) {
    double** ptrs[] = {&h->dK, &h->dW, &h->dY4, &h->dT4, &h->dTa, &h->dXs64, &h->dA64, &h->dscal, &h->dScr, &h->dXraw};
    h->hostX.clear(); h->hostY.clear();
    for (auto pp : ptrs) { if (*pp) (void)hipFree(*pp); *pp = nullptr; }
    if (h->dinfo) (void)hipFree(h->dinfo);
    h->dinfo = nullptr;
    h->ws_np = 0; h->ws_npass = 0; h->have_L = h->have_W = false;
    h->scr_doubles = 0;
}

int ensure_blob(gpt_handle* h, const Layout& l) {
    if (h->blob && h->have_layout && h->lay.same_shape(l)) {
        h->lay = l;
        return GPT_OK;
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    if (h->blob) { (void)hipFree(h->blob); h->blob = nullptr; }
    h->have_layout = false;
    HIPCHK(hipMalloc(&h->blob, l.total));
    h->lay = l;
    h->have_layout = true;
    return GPT_OK;
}

// the factor + inverse arena: sized by the plan THIS fit will run (the plan depends on diagnostic environment knobs, which a
// process may change between two fits of one handle)
int ensure_scratch(gpt_handle* h, int64_t NP) {
    const size_t a = factor_scratch_doubles((int)NP), b = (size_t)(NP / 512) * NP * 4;
    const size_t need = a > b ? a : b;
    if (h->dScr && h->scr_doubles >= need) return GPT_OK;
    HIPCHK(hipStreamSynchronize(h->stream));
    if (h->dScr) (void)hipFree(h->dScr);
    h->dScr = nullptr; h->scr_doubles = 0;
    HIPCHK(hipMalloc(&h->dScr, need * sizeof(double)));
    h->scr_doubles = need;
    return GPT_OK;
}

int ensure_workspace(gpt_handle* h, int64_t NP, int npass) {
    if (h->ws_np == NP && h->ws_npass >= npass && h->dK) return ensure_scratch(h, NP);
    HIPCHK(hipStreamSynchronize(h->stream));
    free_workspace(h);
    HIPCHK(hipMalloc(&h->dK, (size_t)NP * NP * sizeof(double)));
    HIPCHK(hipMalloc(&h->dW, (size_t)NP * NP * sizeof(double)));
    HIPCHK(hipMalloc(&h->dY4, (size_t)npass * NP * 4 * sizeof(double)));
    HIPCHK(hipMalloc(&h->dA64, (size_t)npass * NP * 4 * sizeof(double)));
    HIPCHK(hipMalloc(&h->dT4, (size_t)NP * 4 * sizeof(double)));
    HIPCHK(hipMalloc(&h->dTa, (size_t)NP * 4 * sizeof(double)));
    HIPCHK(hipMalloc(&h->dXs64, (size_t)NP * MAX_D * sizeof(double)));
    HIPCHK(hipMalloc(&h->dXraw, (size_t)NP * MAX_D * sizeof(double)));
    HIPCHK(hipMalloc(&h->dscal, (2 + LML_TERMS + 4) * sizeof(double)));
    HIPCHK(hipMalloc(&h->dinfo, sizeof(int)));
    if (int rc = ensure_scratch(h, NP)) return rc;
    h->ws_np = NP; h->ws_npass = npass;
    return GPT_OK;
}

// one staging buffer of one set, grow-only; only what a call asks for is ever allocated
int ensure_stage(gpt_handle* h, int set, int which, size_t bytes) {
    gpt_handle::Staging& t = h->st[set];
    if (t.bytes[which] >= bytes) return GPT_OK;
    HIPCHK(hipStreamSynchronize(h->stream));
    if (h->copy_stream) HIPCHK(hipStreamSynchronize(h->copy_stream));
    if (t.buf[which]) (void)hipFree(t.buf[which]);
    t.buf[which] = nullptr; t.bytes[which] = 0;
    HIPCHK(hipMalloc(&t.buf[which], bytes));
    t.bytes[which] = bytes;
    return GPT_OK;
}

int ensure_copy_stream(gpt_handle* h) {
    if (h->copy_stream) return GPT_OK;
    HIPCHK(hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
    for (int i = 0; i < 2; ++i) {
        HIPCHK(hipEventCreateWithFlags(&h->ev_done[i], hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&h->ev_copied[i], hipEventDisableTiming));
    }
    return GPT_OK;
}

void fill_params(gpt_handle* h, const double* hdr) {
    KernelParams& p = h->p;
    p.N = (int)hdr[1]; p.NP = (int)hdr[2]; p.D = (int)hdr[3]; p.O = (int)hdr[4];
    p.c = hdr[5]; p.noise = hdr[6];
    p.lnc = std::log(hdr[5]);
    p.ktype = (int)hdr[13];
    p.ntask = (int)hdr[14];
    p.dtype = (int)hdr[15];
    h->jitter = hdr[7];
    h->n_ls = (int)hdr[11];
    for (int d = 0; d < MAX_D; ++d) {
        h->ls[d] = (d < p.D) ? hdr[hdr_ls_slot(d)] : 1.0;
        p.inv_ls[d] = (d < p.D) ? 1.0 / hdr[hdr_ls_slot(d)] : 0.0;
    }
}

int check_geometry(const char* who, int64_t N, int D, int O, const double* length_scale, int n_ls) {
    if (N < 1 || N > (1 << 20)) return fail(GPT_E_ARG, std::string(who) + ": N out of range");
    if (D < 1 || D > MAX_DIMS) return fail(GPT_E_ARG, std::string(who) + ": D must be 1 .. 15");
    if (O < 1) return fail(GPT_E_ARG, std::string(who) + ": O must be >= 1");
    if (n_ls != 1 && n_ls != D) return fail(GPT_E_ARG, std::string(who) + ": length_scale must have 1 or D entries");
    for (int d = 0; d < n_ls; ++d)
        if (!(length_scale[d] > 0.0)) return fail(GPT_E_ARG, std::string(who) + ": length_scale must be > 0");
    return GPT_OK;
}

// Header + scaled, padded sources: to the fp64 workspace image (what the fit kernels read) and, when `model`, in the
// model's element type to the blob.  hdr is complete on return (task priors at hdr[16..] are the caller's).  Nothing here
// waits for the device: the raw sources are uploaded only when they differ from what the device already holds, the scaling
// by the length-scales runs on the device.
int upload_sources(gpt_handle* h, const Layout& l, const double* X, std::vector<double>& hdr, const double* length_scale,
                   int n_ls, double constant_value, double noise_level, double alpha_jitter, int kernel_type, bool model) {
    hdr.assign(HDR_DOUBLES, 0.0);
    hdr[0] = MAGIC; hdr[1] = (double)l.N; hdr[2] = (double)l.NP; hdr[3] = l.D; hdr[4] = l.O;
    hdr[5] = constant_value; hdr[6] = noise_level; hdr[7] = alpha_jitter;
    for (int d = 0; d < 3; ++d) hdr[HDR_LS + d] = 1.0;
    for (int d = 0; d < l.D; ++d) hdr[hdr_ls_slot(d)] = length_scale[n_ls == 1 ? 0 : d];
    hdr[11] = n_ls; hdr[12] = l.npass; hdr[13] = kernel_type; hdr[14] = l.ntask; hdr[15] = l.dtype;
    fill_params(h, hdr.data());
    hipStream_t s = h->stream;
    const size_t nx = (size_t)l.N * l.D;
    if (h->hostX.size() != nx || h->hostX_D != l.D || memcmp(h->hostX.data(), X, nx * sizeof(double)) != 0) {
        HIPCHK(hipStreamSynchronize(s));     // (a copy out of the old mirror that an aborted call left in flight)
        h->hostX.assign(X, X + nx);          // the mirror is also the source of the asynchronous copy: it outlives the call
        h->hostX_D = l.D;
        HIPCHK(hipMemcpyAsync(h->dXraw, h->hostX.data(), nx * sizeof(double), hipMemcpyHostToDevice, s));
    }
    launch_scale_x(s, h->dXraw, (int)l.N, (int)l.NP, l.D, h->p.inv_ls, h->dXs64, model ? h->dXs() : nullptr, l.dtype);
    return GPT_OK;
}

// Padded targets [pass][NP][4] into dY4, unless the device already holds exactly these (N, O) values.
int upload_targets(gpt_handle* h, const Layout& l, const double* Y) {
    const size_t ny = (size_t)l.N * l.O, NP = (size_t)l.NP;
    if (h->hostY.size() == ny && h->hostY_O == l.O && memcmp(h->hostY.data(), Y, ny * sizeof(double)) == 0) return GPT_OK;
    HIPCHK(hipStreamSynchronize(h->stream));
    h->hostY.assign(Y, Y + ny);
    h->hostY_O = l.O;
    h->hostY4.assign((size_t)l.npass * NP * 4, 0.0);
    for (int64_t i = 0; i < l.N; ++i)
        for (int o = 0; o < l.O; ++o) h->hostY4[((size_t)(o / 4) * NP + i) * 4 + (o % 4)] = Y[i * l.O + o];
    HIPCHK(hipMemcpyAsync(h->dY4, h->hostY4.data(), h->hostY4.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
    return GPT_OK;
}

// K -> L (in dK), W = L^-1 (in dW); K's Gram part from dXs64, optional dense SPD addend Sigma (host, N x N)
int factorise(gpt_handle* h, int64_t N, int NP, int kernel_type, double c, double diag_add, const double* Sigma) {
    hipStream_t s = h->stream;
    HIPCHK(hipMemsetAsync(h->dinfo, 0, sizeof(int), s));
    HIPCHK(hipMemsetAsync(h->dW, 0, (size_t)NP * NP * sizeof(double), s));
    launch_gram(s, h->dXs64, h->p.D, (int)N, NP, kernel_type, c, diag_add, h->dK);
    if (Sigma) {      // K += Sigma: staged through dW (not in use until the factorisation starts)
        HIPCHK(hipMemcpyAsync(h->dW, Sigma, (size_t)N * N * sizeof(double), hipMemcpyHostToDevice, s));
        launch_add_lower(s, h->dK, h->dW, (int)N, NP);
        HIPCHK(hipMemsetAsync(h->dW, 0, (size_t)NP * NP * sizeof(double), s));
    }
    HIPCHK(hipEventRecord(h->ev[1], s));
    launch_factor_inverse(s, h->dK, h->dW, NP, h->dinfo, h->dScr, &h->fit_aux, h->ev[2]);
    HIPCHK(hipEventRecord(h->ev[3], s));
    return GPT_OK;
}

}  // namespace

extern "C" {

int gpt_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char* gpt_last_error(void) { return g_err.c_str(); }
const char* gpt_version(void) { return "gpt_hip 0.2 (gfx950)"; }

int gpt_create(gpt_handle** out, int device) {
    if (!out) return fail(GPT_E_ARG, "gpt_create: out is NULL");
    int n = gpt_device_count();
    if (device < 0 || device >= n || device >= MAX_DEVICES) return fail(GPT_E_ARG, "gpt_create: no such HIP device");
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
    var_release(h->vws);
    fit_aux_release(h->fit_aux);
    if (h->blob) (void)hipFree(h->blob);
    if (h->lml_partial) (void)hipFree(h->lml_partial);
    if (h->cov_buf) (void)hipFree(h->cov_buf);
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

int gpt_set_dtype(gpt_handle* h, int dtype) {
    if (!h) return fail(GPT_E_ARG, "gpt_set_dtype: NULL handle");
    if (dtype != GPT_F64 && dtype != GPT_F32) return fail(GPT_E_ARG, "gpt_set_dtype: dtype must be GPT_F64 or GPT_F32");
    h->dtype_next = dtype;
    return GPT_OK;
}

int gpt_fit(gpt_handle* h, const double* X, const double* Y, int64_t N, int D, int O,
            const double* length_scale, int n_ls, double constant_value, double noise_level,
            double alpha_jitter) {
    return gpt_fit_kernel(h, X, Y, N, D, O, length_scale, n_ls, constant_value, noise_level, alpha_jitter, GPT_KERNEL_RBF);
}

static int fit_impl(gpt_handle* h, const double* X, const double* Y, int64_t N, int D, int O,
                    const double* length_scale, int n_ls, double constant_value, double noise_level,
                    double alpha_jitter, int kernel_type, const double* Sigma, bool model = true);

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

static int read_fit_times(gpt_handle* h) {
    float ms = 0;
    for (int i = 1; i <= 5; ++i) {
        HIPCHK(hipEventElapsedTime(&ms, h->ev[i - 1], h->ev[i]));
        h->fit_ms[i] = ms;
    }
    HIPCHK(hipEventElapsedTime(&ms, h->ev[0], h->ev[5]));
    h->fit_ms[0] = ms;
    return GPT_OK;
}

// Everything of a fit up to the last kernel, nothing waited for.  model = false: what one evaluation of the optimizer's
// objective needs (L, W, alpha in the fp64 workspace) and nothing of the prediction-side model (no sources / alpha / W in
// the blob's layouts).
static int fit_enqueue(gpt_handle* h, const double* X, const double* Y, int64_t N, int D, int O,
                       const double* length_scale, int n_ls, double constant_value, double noise_level,
                       double alpha_jitter, int kernel_type, const double* Sigma, bool model) {
    if (!h || !X || !Y || !length_scale) return fail(GPT_E_ARG, "gpt_fit: NULL argument");
    if (kernel_type < GPT_KERNEL_RBF || kernel_type > GPT_KERNEL_MATERN52) return fail(GPT_E_ARG, "gpt_fit: unknown kernel_type");
    if (int rc = check_geometry("gpt_fit", N, D, O, length_scale, n_ls)) return rc;
    if (!(constant_value > 0.0) || !(noise_level >= 0.0) || !(alpha_jitter >= 0.0))
        return fail(GPT_E_ARG, "gpt_fit: constant_value > 0, noise_level >= 0, alpha >= 0 required");
    if (int rc = set_device(h)) return rc;
    h->committed = false;
    h->have_L = h->have_W = false;
    h->objective_ready = false;
    const Layout l = make_layout(N, D, O, 1, h->dtype_next);
    if (int rc = ensure_blob(h, l)) return rc;
    if (int rc = ensure_workspace(h, l.NP, l.npass)) return rc;
    const int NP = (int)l.NP;
    hipStream_t s = h->stream;

    // ---- inputs: header, sources (scaled on the device), padded targets; copies only of what the device does not hold yet
    std::vector<double> hdr;
    if (int rc = upload_sources(h, l, X, hdr, length_scale, n_ls, constant_value, noise_level, alpha_jitter, kernel_type, model)) return rc;
    if (int rc = upload_targets(h, l, Y)) return rc;
    if (model) {
        hdr[HDR_TASK_C] = constant_value;
        memcpy(h->host_hdr, hdr.data(), sizeof h->host_hdr);          // a member: the source of an asynchronous copy
        HIPCHK(hipMemcpyAsync(h->blob, h->host_hdr, sizeof h->host_hdr, hipMemcpyHostToDevice, s));
    }

    // ---- device pipeline
    HIPCHK(hipEventRecord(h->ev[0], s));
    if (int rc = factorise(h, N, NP, kernel_type, constant_value, noise_level + alpha_jitter, Sigma)) return rc;
    for (int ps = 0; ps < l.npass; ++ps) {
        double* a64 = h->dA64 + (size_t)ps * NP * 4;
        launch_alpha(s, h->dW, h->dY4 + (size_t)ps * NP * 4, (int)N, NP, h->dT4, a64, h->dScr);
        if (model) launch_store4(s, a64, NP, static_cast<unsigned char*>(h->dA4()) + (size_t)ps * NP * 4 * l.esz, l.dtype, 0, 0, 4, 1.0);
    }
    HIPCHK(hipEventRecord(h->ev[4], s));
    if (model) {
        launch_pack_w(s, h->dW, (int)N, NP, h->dWf(), l.dtype, 0, 1.0);
        HIPCHK(hipMemsetAsync(static_cast<unsigned char*>(h->dWf()) + wf_elems(NP) * l.esz, 0, wf_overrun_elems() * l.esz, s));
    }
    HIPCHK(hipEventRecord(h->ev[5], s));
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(&h->host_info, h->dinfo, sizeof(int), hipMemcpyDeviceToHost, s));
    return GPT_OK;
}

// After the stream has been synchronised: the pivot check and the state of the handle.
static int fit_finish(gpt_handle* h, bool model) {
    if (h->host_info != 0) {
        char buf[160];
        snprintf(buf, sizeof buf, "gpt_fit: kernel matrix is not positive definite (pivot %d <= 0)", h->host_info);
        return fail(GPT_E_NOT_PD, buf);
    }
    h->committed = model;
    h->objective_ready = !model;
    h->have_L = h->have_W = true;
    return GPT_OK;
}

static int fit_impl(gpt_handle* h, const double* X, const double* Y, int64_t N, int D, int O,
                    const double* length_scale, int n_ls, double constant_value, double noise_level,
                    double alpha_jitter, int kernel_type, const double* Sigma, bool model) {
    if (int rc = fit_enqueue(h, X, Y, N, D, O, length_scale, n_ls, constant_value, noise_level, alpha_jitter, kernel_type, Sigma, model)) return rc;
    HIPCHK(hipStreamSynchronize(h->stream));
    if (int rc = read_fit_times(h)) return rc;
    return fit_finish(h, model);
}

// ---- log-marginal likelihood and its gradient: device-side reductions into dscal, one read-back
//   dscal[0] = sum_i log L_ii, dscal[1] = sum_o y_o^T alpha_o, dscal[2 ..] = the LML_TERMS traces of the gradient
static void enqueue_lml_scalars(gpt_handle* h) {
    launch_logdet(h->stream, h->dK, h->p.N, h->p.NP, h->dscal);
    launch_dot(h->stream, h->dY4, h->dA64, (int64_t)h->lay.npass * h->p.NP * 4, h->dscal + 1);
}

static int enqueue_gradient_terms(gpt_handle* h) {
    const int64_t NP = h->p.NP;
    hipStream_t s = h->stream;
    // K^-1 (lower) into dK, per-tile partial sums into their own scratch
    const size_t need = (size_t)(NP / 64) * (NP / 64) * LML_PARTIAL_STRIDE;
    if (need > h->lml_partial_cap) {
        HIPCHK(hipStreamSynchronize(s));
        if (h->lml_partial) (void)hipFree(h->lml_partial);
        h->lml_partial = nullptr; h->lml_partial_cap = 0;
        HIPCHK(hipMalloc(&h->lml_partial, need * sizeof(double)));
        h->lml_partial_cap = need;
    }
    launch_kinv(s, h->dW, (int)NP, h->dK);
    launch_lml_terms(s, h->dXs64, h->p.D, h->dA64, h->lay.npass, h->dK, h->p.N, (int)NP, h->p.O, h->p.ktype, h->p.c, h->lml_partial, h->dscal + 2);
    HIPCHK(hipGetLastError());
    return GPT_OK;
}

static double lml_from_scalars(const gpt_handle* h, const double* sc) {
    // sum over outputs of  -1/2 y^T alpha - sum log L_ii - N/2 log 2 pi   (sklearn/_gpr.py:598-606)
    const double O = h->p.O, N = h->p.N;
    return -0.5 * sc[1] - O * sc[0] - 0.5 * O * N * std::log(2.0 * M_PI);
}

static void gradient_from_scalars(const gpt_handle* h, const double* S, double* grad) {
    // theta = log [constant_value, length_scale (1 or D), noise_level]
    const int D = h->p.D;
    grad[0] = 0.5 * S[0];
    if (h->n_ls == 1) {
        double t = 0;
        for (int d = 0; d < D; ++d) t += S[1 + d];
        grad[1] = 0.5 * t;
    } else {
        for (int d = 0; d < D; ++d) grad[1 + d] = 0.5 * S[1 + d];
    }
    grad[1 + h->n_ls] = 0.5 * h->p.noise * S[LML_TERMS - 1];
}

// One evaluation of the optimizer's objective: everything enqueued back to back, ONE wait, one read-back.
int gpt_lml_objective(gpt_handle* h, const double* X, const double* Y, int64_t N, int D, int O,
                      const double* length_scale, int n_ls, double constant_value, double noise_level,
                      double alpha_jitter, int kernel_type, double* lml, double* grad) {
    if (!lml || !grad) return fail(GPT_E_ARG, "gpt_lml_objective: NULL argument");
    if (int rc = fit_enqueue(h, X, Y, N, D, O, length_scale, n_ls, constant_value, noise_level, alpha_jitter, kernel_type, nullptr, false))
        return rc;
    hipStream_t s = h->stream;
    enqueue_lml_scalars(h);                                   // reads diag(L) in dK before the gradient overwrites it
    if (int rc = enqueue_gradient_terms(h)) return rc;
    HIPCHK(hipMemcpyAsync(h->host_scal, h->dscal, (2 + LML_TERMS) * sizeof(double), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (int rc = fit_finish(h, false)) return rc;             // not positive definite: whatever ran behind the factor is discarded
    h->have_L = false;                                        // dK holds K^-1 now
    *lml = lml_from_scalars(h, h->host_scal);
    gradient_from_scalars(h, h->host_scal + 2, grad);
    return GPT_OK;
}

int gpt_fit_svgp(gpt_handle* h, const double* Z, const double* y, const double* Sigma, int64_t N, int D, int T,
                 const double* length_scale, int n_ls, const double* outputscale, double jitter, int dtype) {
    if (!h || !Z || !y || !Sigma || !length_scale || !outputscale) return fail(GPT_E_ARG, "gpt_fit_svgp: NULL argument");
    if (T < 1 || T > MAX_TASKS) return fail(GPT_E_ARG, "gpt_fit_svgp: number of tasks must be 1..32");
    if (dtype != GPT_F64 && dtype != GPT_F32) return fail(GPT_E_ARG, "gpt_fit_svgp: dtype must be GPT_F64 or GPT_F32");
    if (int rc = check_geometry("gpt_fit_svgp", N, D, T, length_scale, n_ls)) return rc;
    for (int t = 0; t < T; ++t)
        if (!(outputscale[t] > 0.0)) return fail(GPT_E_ARG, "gpt_fit_svgp: outputscale must be > 0");
    if (!(jitter >= 0.0)) return fail(GPT_E_ARG, "gpt_fit_svgp: jitter >= 0 required");
    if (int rc = set_device(h)) return rc;
    h->committed = false;
    h->have_L = h->have_W = false;
    h->objective_ready = false;
    const Layout l = make_layout(N, D, T, T, dtype);
    if (int rc = ensure_blob(h, l)) return rc;
    if (int rc = ensure_workspace(h, l.NP, l.npass)) return rc;
    const int NP = (int)l.NP;
    hipStream_t s = h->stream;
    // the kernel columns are generated WITHOUT the outputscale (c = 1): it is folded into each task's inverse factor
    // and alpha, so that all tasks share one B operand
    std::vector<double> hdr;
    if (int rc = upload_sources(h, l, Z, hdr, length_scale, n_ls, 1.0, 0.0, jitter, GPT_KERNEL_RBF, true)) return rc;
    for (int t = 0; t < T; ++t) hdr[HDR_TASK_C + t] = outputscale[t];
    HIPCHK(hipMemcpyAsync(h->blob, hdr.data(), HDR_DOUBLES * sizeof(double), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemsetAsync(h->dA4(), 0, (size_t)l.npass * NP * 4 * l.esz, s));
    HIPCHK(hipMemsetAsync(h->dA64, 0, (size_t)l.npass * NP * 4 * sizeof(double), s));
    HIPCHK(hipStreamSynchronize(s));
    h->hostY.clear();
    std::vector<double> y4((size_t)NP * 4);
    HIPCHK(hipEventRecord(h->ev[0], s));
    int info = 0, bad_task = -1;
    for (int t = 0; t < T && info == 0; ++t) {
        std::fill(y4.begin(), y4.end(), 0.0);
        for (int64_t i = 0; i < N; ++i) y4[(size_t)i * 4] = y[(size_t)t * N + i];
        HIPCHK(hipMemcpyAsync(h->dY4, y4.data(), y4.size() * sizeof(double), hipMemcpyHostToDevice, s));
        if (int rc = factorise(h, N, NP, GPT_KERNEL_RBF, outputscale[t], jitter, Sigma + (size_t)t * N * N)) return rc;
        launch_alpha(s, h->dW, h->dY4, (int)N, NP, h->dT4, h->dTa, h->dScr);
        const size_t col_off = (size_t)(t / 4) * NP * 4;
        launch_store4(s, h->dTa, NP, h->dA64 + col_off, DT_F64, 0, t % 4, 1, 1.0);
        launch_store4(s, h->dTa, NP, static_cast<unsigned char*>(h->dA4()) + col_off * l.esz, l.dtype, 0, t % 4, 1, outputscale[t]);
        launch_pack_w(s, h->dW, (int)N, NP, h->dWf(), l.dtype, t, outputscale[t]);
        HIPCHK(hipMemcpyAsync(&info, h->dinfo, sizeof(int), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));           // y4 is reused; the pivot check of this task
        if (info != 0) bad_task = t;
    }
    HIPCHK(hipMemsetAsync(static_cast<unsigned char*>(h->dWf()) + (size_t)T * wf_elems(NP) * l.esz, 0, wf_overrun_elems() * l.esz, s));
    HIPCHK(hipEventRecord(h->ev[4], s));
    HIPCHK(hipEventRecord(h->ev[5], s));
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(s));
    if (int rc = read_fit_times(h)) return rc;
    if (info != 0) {
        char buf[200];
        snprintf(buf, sizeof buf, "gpt_fit_svgp: K_uu + Sigma of task %d is not positive definite (pivot %d <= 0)", bad_task, info);
        return fail(GPT_E_NOT_PD, buf);
    }
    h->committed = true;           // dK / dW hold the LAST task's factors only: no export, covariance or LML for this model
    return GPT_OK;
}

int gpt_reserve(gpt_handle* h, int64_t M, int jacobian_variance) {
    if (!h) return fail(GPT_E_ARG, "gpt_reserve: NULL handle");
    if (!h->committed) return fail(GPT_E_STATE, "gpt_reserve: model is not fitted");
    if (M < 0) return fail(GPT_E_ARG, "gpt_reserve: bad query count");
    if (int rc = set_device(h)) return rc;
    if (M == 0) return GPT_OK;
    HIPCHK(var_prepare(h->vws, h->stream, h->p, M, jacobian_variance ? var_fused_cols(h->p.D) : 1));
    return GPT_OK;
}

int gpt_predict_all_dev(gpt_handle* h, const void* Xq, int64_t M, void* mean, void* var,
                        void* J, void* Jvar, void* dvar) {
    if (!h) return fail(GPT_E_ARG, "gpt_predict_all_dev: NULL handle");
    if (!h->committed) return fail(GPT_E_STATE, "predict: model is not fitted");
    if (M < 0 || (M > 0 && !Xq)) return fail(GPT_E_ARG, "predict: bad query buffer");
    if (M == 0) return GPT_OK;
    if ((J || Jvar || dvar) && h->p.ktype != GPT_KERNEL_RBF)
        return fail(GPT_E_ARG, "derivative / Jacobian variance / d variance are defined for the RBF kernel only");
    if (dvar && h->p.ntask != 1) return fail(GPT_E_ARG, "d variance is not defined for the multi-task (SVGP) model");
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
        // fused: k* and the D derivative columns of a query side by side; 3: Jacobian variance alone (no k* column, D <= 3)
        const int fused = var_fused_cols(h->p.D);
        const int alone = h->p.D <= 3 ? 3 : (h->p.D == 4 ? VAR_NCOMP_DERIV4 : (h->p.D == 8 ? VAR_NCOMP_DERIV8 : fused));
        const int ncomp = dvar ? fused : (Jvar ? (var ? fused : alone) : 1);
        HIPCHK(var_prepare(h->vws, s, h->p, M, ncomp));
        if (prof) HIPCHK(hipEventRecord(h->pev[2], s));
        launch_var(s, h->p, h->vws, h->dXs(), h->dWf(), Xq, M, ncomp, var, ncomp == 1 ? nullptr : Jvar, ncomp == 1 ? nullptr : dvar, h->dHdr());
        if (prof) HIPCHK(hipEventRecord(h->pev[3], s));
    }
    HIPCHK(hipGetLastError());
    return GPT_OK;
}

int gpt_predict_all(gpt_handle* h, const void* Xq_, int64_t M, void* mean_, void* var_,
                    void* J_, void* Jvar_, void* dvar_) {
    if (!h) return fail(GPT_E_ARG, "gpt_predict_all: NULL handle");
    if (!h->committed) return fail(GPT_E_STATE, "predict: model is not fitted");
    if (M < 0 || (M > 0 && !Xq_)) return fail(GPT_E_ARG, "predict: bad query buffer");
    if (M == 0) return GPT_OK;
    if (int rc = set_device(h)) return rc;
    const size_t esz = h->lay.esz;
    const size_t D = h->p.D, O = h->p.O, NT = h->p.ntask;
    const unsigned char* Xq = static_cast<const unsigned char*>(Xq_);
    unsigned char *mean = static_cast<unsigned char*>(mean_), *var = static_cast<unsigned char*>(var_), *J = static_cast<unsigned char*>(J_),
                  *Jvar = static_cast<unsigned char*>(Jvar_), *dvar = static_cast<unsigned char*>(dvar_);
    const int64_t cap = M < HOST_CHUNK ? M : HOST_CHUNK;
    const int64_t nchunks = (M + cap - 1) / cap;
    // elements per query of each staged array
    const size_t per[ST_COUNT] = {D, O, NT, O * D, NT * D, D};
    void* const want[ST_COUNT] = {(void*)Xq, mean, var, J, Jvar, dvar};
    for (int b = 0; b < (nchunks > 1 ? 2 : 1); ++b)
        for (int i = 0; i < ST_COUNT; ++i)
            if (want[i]) { if (int rc = ensure_stage(h, b, i, (size_t)cap * per[i] * esz)) return rc; }
    // one chunk (the reference's own batch sizes): queries in, kernels, results out, all in the handle's stream — no second
    // stream, no events; several chunks: the results of chunk i leave on the copy stream while chunk i + 1 computes
    const bool single = nchunks == 1;
    if (!single) { if (int rc = ensure_copy_stream(h)) return rc; }
    hipStream_t s = h->stream, cs = single ? h->stream : h->copy_stream;
    // chunk i computes on `s` in staging set i & 1; its outputs leave on `cs` while chunk i + 1 computes
    auto enqueue = [&](int64_t i) -> int {
        const int b = (int)(i & 1);
        const int64_t off = i * cap, m = (M - off) < cap ? (M - off) : cap;
        gpt_handle::Staging& t = h->st[b];
        if (i >= 2) HIPCHK(hipStreamWaitEvent(s, h->ev_copied[b], 0));          // set b is free again
        HIPCHK(hipMemcpyAsync(t.buf[ST_Q], Xq + (size_t)off * D * esz, (size_t)m * D * esz, hipMemcpyHostToDevice, s));
        if (int rc = gpt_predict_all_dev(h, t.buf[ST_Q], m, mean ? t.buf[ST_MEAN] : nullptr, var ? t.buf[ST_VAR] : nullptr,
                                         J ? t.buf[ST_J] : nullptr, Jvar ? t.buf[ST_JVAR] : nullptr, dvar ? t.buf[ST_DVAR] : nullptr))
            return rc;
        if (!single) HIPCHK(hipEventRecord(h->ev_done[b], s));
        return GPT_OK;
    };
    auto copy_out = [&](int64_t i) -> int {
        const int b = (int)(i & 1);
        const int64_t off = i * cap, m = (M - off) < cap ? (M - off) : cap;
        const gpt_handle::Staging& t = h->st[b];
        if (!single) HIPCHK(hipStreamWaitEvent(cs, h->ev_done[b], 0));
        unsigned char* const dst[ST_COUNT] = {nullptr, mean, var, J, Jvar, nullptr};
        for (int k = ST_MEAN; k <= ST_JVAR; ++k)
            if (dst[k]) HIPCHK(hipMemcpyAsync(dst[k] + (size_t)off * per[k] * esz, t.buf[k], (size_t)m * per[k] * esz, hipMemcpyDeviceToHost, cs));
        if (dvar)
            for (size_t d = 0; d < D; ++d)   // device chunk is (D, m); host result is (D, M)
                HIPCHK(hipMemcpyAsync(dvar + (d * (size_t)M + (size_t)off) * esz, static_cast<unsigned char*>(t.buf[ST_DVAR]) + d * (size_t)m * esz,
                                      (size_t)m * esz, hipMemcpyDeviceToHost, cs));
        if (!single) HIPCHK(hipEventRecord(h->ev_copied[b], cs));
        return GPT_OK;
    };
    if (int rc = enqueue(0)) return rc;
    for (int64_t i = 0; i < nchunks; ++i) {
        if (i + 1 < nchunks) { if (int rc = enqueue(i + 1)) return rc; }   // queued before chunk i's (host-blocking) copies
        if (int rc = copy_out(i)) return rc;
    }
    if (!single) HIPCHK(hipStreamSynchronize(cs));
    HIPCHK(hipStreamSynchronize(s));
    return GPT_OK;
}

int gpt_predict(gpt_handle* h, const void* Xq, int64_t M, void* mean, void* var) {
    return gpt_predict_all(h, Xq, M, mean, var, nullptr, nullptr, nullptr);
}

int gpt_derivative(gpt_handle* h, const void* Xq, int64_t M, void* J, void* Jvar) {
    return gpt_predict_all(h, Xq, M, nullptr, nullptr, J, Jvar, nullptr);
}

int gpt_dvariance(gpt_handle* h, const void* Xq, int64_t M, void* g) {
    return gpt_predict_all(h, Xq, M, nullptr, nullptr, nullptr, nullptr, g);
}

// alpha as (N,O) fp64 host array: from the fp64 workspace when this handle ran the fit, else converted from the blob
static int fetch_alpha(gpt_handle* h, std::vector<double>& a4) {
    const int64_t NP = h->p.NP;
    const size_t n = (size_t)h->lay.npass * NP * 4;
    a4.resize(n);
    if (h->have_W) {
        HIPCHK(hipMemcpyAsync(a4.data(), h->dA64, n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
    } else if (h->lay.dtype == DT_F32) {
        std::vector<float> af(n);
        HIPCHK(hipMemcpyAsync(af.data(), h->dA4(), n * sizeof(float), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        for (size_t i = 0; i < n; ++i) a4[i] = af[i];
    } else {
        HIPCHK(hipMemcpyAsync(a4.data(), h->dA4(), n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    return GPT_OK;
}

int gpt_export(gpt_handle* h, double* L, double* alpha) {
    if (!h) return fail(GPT_E_ARG, "gpt_export: NULL handle");
    if (!h->committed) return fail(GPT_E_STATE, "gpt_export: model is not fitted");
    if (int rc = set_device(h)) return rc;
    const int64_t N = h->p.N, NP = h->p.NP;
    const int O = h->p.O;
    HIPCHK(hipStreamSynchronize(h->stream));
    if (L) {
        if (!h->have_L) return fail(GPT_E_STATE, "gpt_export: L is only available on the handle that ran gpt_fit (and before gpt_lml_gradient)");
        HIPCHK(hipMemcpy2D(L, (size_t)N * sizeof(double), h->dK, (size_t)NP * sizeof(double), (size_t)N * sizeof(double),
                           (size_t)N, hipMemcpyDeviceToHost));
        for (int64_t i = 0; i < N; ++i)
            for (int64_t j = i + 1; j < N; ++j) L[i * N + j] = 0.0;
    }
    if (alpha) {
        // (for the multi-task model: alpha of task t in column t, times its outputscale when read back from the blob)
        std::vector<double> a4;
        if (int rc = fetch_alpha(h, a4)) return rc;
        for (int64_t i = 0; i < N; ++i)
            for (int o = 0; o < O; ++o) alpha[i * O + o] = a4[((size_t)(o / 4) * NP + i) * 4 + (o % 4)];
    }
    return GPT_OK;
}

int gpt_export_inverse_factor(gpt_handle* h, double* W) {
    if (!h || !W) return fail(GPT_E_ARG, "gpt_export_inverse_factor: NULL argument");
    if (!h->committed || !h->have_W) return fail(GPT_E_STATE, "gpt_export_inverse_factor: no factor on this handle");
    if (int rc = set_device(h)) return rc;
    const int64_t N = h->p.N, NP = h->p.NP;
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy2D(W, (size_t)N * sizeof(double), h->dW, (size_t)NP * sizeof(double), (size_t)N * sizeof(double),
                       (size_t)N, hipMemcpyDeviceToHost));
    return GPT_OK;
}

int gpt_lml(gpt_handle* h, double* lml) {
    if (!h || !lml) return fail(GPT_E_ARG, "gpt_lml: NULL argument");
    if (!(h->committed || h->objective_ready) || !h->have_L || !h->have_W) return fail(GPT_E_STATE, "gpt_lml: needs the handle that ran gpt_fit");
    if (int rc = set_device(h)) return rc;
    enqueue_lml_scalars(h);
    HIPCHK(hipMemcpyAsync(h->host_scal, h->dscal, 2 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    *lml = lml_from_scalars(h, h->host_scal);
    return GPT_OK;
}

int gpt_predict_cov(gpt_handle* h, const double* Xq, int64_t M, double* mean, double* cov) {
    if (!h || !cov) return fail(GPT_E_ARG, "gpt_predict_cov: NULL argument");
    if (!h->committed) return fail(GPT_E_STATE, "predict: model is not fitted");
    if (!h->have_W) return fail(GPT_E_STATE, "gpt_predict_cov: needs the handle that ran gpt_fit for this model");
    if (h->lay.dtype != DT_F64) return fail(GPT_E_STATE, "gpt_predict_cov: fp64 models only");
    if (M < 0 || (M > 0 && !Xq)) return fail(GPT_E_ARG, "predict: bad query buffer");
    if (M > 16384) return fail(GPT_E_ARG, "gpt_predict_cov: M x M covariance limited to M <= 16384");
    if (M == 0) return GPT_OK;
    if (int rc = set_device(h)) return rc;
    if (mean) { if (int rc = gpt_predict_all(h, Xq, M, mean, nullptr, nullptr, nullptr, nullptr)) return rc; }
    const int D = h->p.D;
    const int64_t NP = h->p.NP;
    const int Mp = (int)((M + 127) / 128 * 128);
    hipStream_t s = h->stream;
    // one grow-only scratch: [dq | KsT | V | VtV | dcov]
    auto al = [](size_t b) { return (b + 255) / 256 * 256; };
    const size_t b_q = al((size_t)M * D * sizeof(double)), b_k = al((size_t)NP * Mp * sizeof(double)),
                 b_vv = al((size_t)Mp * Mp * sizeof(double)), b_c = al((size_t)M * M * sizeof(double));
    const size_t need = b_q + 2 * b_k + b_vv + b_c;
    if (need > h->cov_cap) {
        HIPCHK(hipStreamSynchronize(s));
        if (h->cov_buf) (void)hipFree(h->cov_buf);
        h->cov_buf = nullptr; h->cov_cap = 0;
        HIPCHK(hipMalloc(&h->cov_buf, need));
        h->cov_cap = need;
    }
    double* dq = reinterpret_cast<double*>(h->cov_buf);
    double* KsT = reinterpret_cast<double*>(h->cov_buf + b_q);
    double* V = reinterpret_cast<double*>(h->cov_buf + b_q + b_k);
    double* VtV = reinterpret_cast<double*>(h->cov_buf + b_q + 2 * b_k);
    double* dcov = reinterpret_cast<double*>(h->cov_buf + b_q + 2 * b_k + b_vv);
    HIPCHK(hipMemcpyAsync(dq, Xq, (size_t)M * D * sizeof(double), hipMemcpyHostToDevice, s));
    launch_cov(s, h->p, h->dXs64, h->dW, dq, M, Mp, KsT, V, VtV, dcov);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(cov, dcov, (size_t)M * M * sizeof(double), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return GPT_OK;
}

int gpt_lml_gradient(gpt_handle* h, double* lml, double* grad) {
    if (!h || !lml || !grad) return fail(GPT_E_ARG, "gpt_lml_gradient: NULL argument");
    if (!(h->committed || h->objective_ready) || !h->have_L || !h->have_W) return fail(GPT_E_STATE, "gpt_lml_gradient: needs the handle that ran gpt_fit");
    if (int rc = set_device(h)) return rc;
    enqueue_lml_scalars(h);                                      // uses diag(L) in dK before it is overwritten
    h->have_L = false;                                           // L is gone: gpt_export(L) needs a new gpt_fit (W stays valid)
    if (int rc = enqueue_gradient_terms(h)) return rc;
    HIPCHK(hipMemcpyAsync(h->host_scal, h->dscal, (2 + LML_TERMS) * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    *lml = lml_from_scalars(h, h->host_scal);
    gradient_from_scalars(h, h->host_scal + 2, grad);
    return GPT_OK;
}

int gpt_factor_blob(gpt_handle* h, void** dev_ptr, size_t* bytes) {
    if (!h || !dev_ptr || !bytes) return fail(GPT_E_ARG, "gpt_factor_blob: NULL argument");
    if (!h->blob || !h->have_layout) return fail(GPT_E_STATE, "gpt_factor_blob: no model storage");
    *dev_ptr = h->blob;
    *bytes = h->lay.total;
    return GPT_OK;
}

int gpt_factor_alloc_model(gpt_handle* h, int64_t N, int D, int O, int n_tasks, int dtype, void** dev_ptr, size_t* bytes) {
    if (!h || !dev_ptr || !bytes) return fail(GPT_E_ARG, "gpt_factor_alloc: NULL argument");
    if (N < 1 || D < 1 || D > MAX_DIMS || O < 1 || n_tasks < 1 || n_tasks > MAX_TASKS || (dtype != GPT_F64 && dtype != GPT_F32))
        return fail(GPT_E_ARG, "gpt_factor_alloc: bad geometry");
    if (int rc = set_device(h)) return rc;
    h->committed = false;
    h->have_L = h->have_W = false;
    h->objective_ready = false;
    if (int rc = ensure_blob(h, make_layout(N, D, O, n_tasks, dtype))) return rc;
    *dev_ptr = h->blob;
    *bytes = h->lay.total;
    return GPT_OK;
}

int gpt_factor_alloc(gpt_handle* h, int64_t N, int D, int O, void** dev_ptr, size_t* bytes) {
    return gpt_factor_alloc_model(h, N, D, O, 1, GPT_F64, dev_ptr, bytes);
}

int gpt_factor_commit(gpt_handle* h) {
    if (!h) return fail(GPT_E_ARG, "gpt_factor_commit: NULL handle");
    if (!h->blob || !h->have_layout) return fail(GPT_E_STATE, "gpt_factor_commit: no model storage");
    if (int rc = set_device(h)) return rc;
    double hdr[HDR_DOUBLES];
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(hdr, h->blob, sizeof hdr, hipMemcpyDeviceToHost));
    if (hdr[0] != MAGIC) return fail(GPT_E_STATE, "gpt_factor_commit: blob has no model header (broadcast missing?)");
    if ((int64_t)hdr[1] != h->lay.N || (int)hdr[3] != h->lay.D || (int)hdr[4] != h->lay.O || (int64_t)hdr[2] != h->lay.NP ||
        (int)hdr[14] != h->lay.ntask || (int)hdr[15] != h->lay.dtype)
        return fail(GPT_E_STATE, "gpt_factor_commit: header geometry differs from gpt_factor_alloc");
    fill_params(h, hdr);
    h->committed = true;
    return GPT_OK;
}

int gpt_factor_copy(gpt_handle* dst, gpt_handle* src) {
    if (!dst || !src) return fail(GPT_E_ARG, "gpt_factor_copy: NULL handle");
    if (dst == src) return fail(GPT_E_ARG, "gpt_factor_copy: source and destination are the same handle");
    if (!src->committed || !src->blob) return fail(GPT_E_STATE, "gpt_factor_copy: source handle holds no fitted model");
    // the source's kernels (fit, pack) have to be done before another device reads the blob
    if (int rc = set_device(src)) return rc;
    HIPCHK(hipStreamSynchronize(src->stream));
    double hdr[HDR_DOUBLES];
    HIPCHK(hipMemcpy(hdr, src->blob, sizeof hdr, hipMemcpyDeviceToHost));
    if (hdr[0] != MAGIC) return fail(GPT_E_STATE, "gpt_factor_copy: source blob has no model header");
    const Layout l = src->lay;
    if (int rc = set_device(dst)) return rc;
    dst->committed = false;
    dst->have_L = dst->have_W = false;
    dst->objective_ready = false;
    if (int rc = ensure_blob(dst, l)) return rc;
    if (dst->device == src->device) {
        HIPCHK(hipMemcpyAsync(dst->blob, src->blob, l.total, hipMemcpyDeviceToDevice, dst->stream));
    } else {
        // device to device over xGMI (peer access is enabled on demand; without it the runtime stages through the host)
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, dst->device, src->device) == hipSuccess && can) {
            hipError_t e = hipDeviceEnablePeerAccess(src->device, 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
            else (void)hipGetLastError();
        }
        HIPCHK(hipMemcpyPeerAsync(dst->blob, dst->device, src->blob, src->device, l.total, dst->stream));
    }
    HIPCHK(hipStreamSynchronize(dst->stream));
    fill_params(dst, hdr);
    dst->committed = true;
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

int gpt_model_info(gpt_handle* h, int* n_tasks, int* dtype) {
    if (!h) return fail(GPT_E_ARG, "gpt_model_info: NULL handle");
    if (!h->committed) return fail(GPT_E_STATE, "gpt_model_info: model is not fitted");
    if (n_tasks) *n_tasks = h->p.ntask;
    if (dtype) *dtype = h->p.dtype;
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

int gpt_debug_var_plan(int64_t n_columns, int n_iblocks, int n_tasks, int n_workgroups, int order, int64_t* counts,
                       int* item_begin, int* items, int* fin, int* splits) {
    if (n_columns < 0 || n_iblocks < 1 || n_tasks < 1 || n_workgroups < 1 || !counts) return fail(GPT_E_ARG, "gpt_debug_var_plan: bad argument");
    const VarPlanHost pl = build_var_plan(n_columns, n_iblocks, n_tasks, n_workgroups, order);
    counts[0] = (int64_t)pl.items.size(); counts[1] = (int64_t)pl.splits.size(); counts[2] = pl.n_slots; counts[3] = pl.n_vslots;
    counts[4] = pl.d.ncb; counts[5] = pl.d.nfull; counts[6] = (int64_t)pl.fin.size() / 2; counts[7] = pl.order;
    counts[8] = pl.cohorts; counts[9] = pl.cohort_s; counts[10] = pl.cohort_f; counts[11] = pl.cut_diag;
    if (item_begin) memcpy(item_begin, pl.item_begin.data(), pl.item_begin.size() * sizeof(int));
    if (items && !pl.items.empty()) memcpy(items, pl.items.data(), pl.items.size() * sizeof(VarItem));
    if (fin && !pl.fin.empty()) memcpy(fin, pl.fin.data(), pl.fin.size() * sizeof(int));
    if (splits && !pl.splits.empty()) memcpy(splits, pl.splits.data(), pl.splits.size() * sizeof(VarSplit));
    return GPT_OK;
}

int gpt_debug_fit_plan(int n_padded, int form, int panel, int streams, int64_t* counts, int64_t* ops) {
    if (n_padded < 64 || n_padded % 64 != 0 || !counts) return fail(GPT_E_ARG, "gpt_debug_fit_plan: bad argument");
    const FitPlan pl = fit_plan(n_padded, form, panel, streams);
    counts[0] = (int64_t)pl.ops.size(); counts[1] = (int64_t)pl.arena; counts[2] = pl.form; counts[3] = pl.n_events;
    counts[4] = (int64_t)factor_scratch_doubles_of(n_padded); counts[5] = pl.side_eighths;
    if (ops)
        for (size_t i = 0; i < pl.ops.size(); ++i) {
            const FitOp& o = pl.ops[i];
            int64_t* r = ops + 18 * i;
            r[0] = o.kind; r[1] = o.stream; r[2] = o.off; r[3] = o.n1; r[4] = o.n2; r[5] = o.k0; r[6] = o.kw; r[7] = o.row_end; r[8] = o.grp;
            r[9] = (int64_t)o.r0; r[10] = (int64_t)o.r0_size; r[11] = (int64_t)o.r1; r[12] = (int64_t)o.r1_size;
            r[13] = o.wait[0]; r[14] = o.wait[1]; r[15] = o.wait[2]; r[16] = o.record; r[17] = 0;
        }
    return GPT_OK;
}

}  // extern "C"
