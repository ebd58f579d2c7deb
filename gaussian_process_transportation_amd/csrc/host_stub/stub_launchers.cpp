// Kernel launchers of the sanitizer build (see host_stub/hip/hip_runtime.h): no arithmetic — each stand-in TOUCHES the
// memory its kernel would read and write, with the sizes the kernel derives from its arguments, so that AddressSanitizer
// checks the buffer sizing of the host orchestration (staging, model blob, slab / vslab / scratch of the variance plan).
#include "../gpt_common.h"
#include "../gpt_plan.h"
#include "../gpt_fit_plan.h"

namespace gpt {

static size_t esz(int dtype) { return dtype == DT_F32 ? sizeof(float) : sizeof(double); }
static void touch_w(void* p, size_t bytes) { if (p && bytes) memset(p, 0, bytes); }
static void touch_r(const void* p, size_t bytes) {
    if (!p || !bytes) return;
    volatile unsigned char acc = 0;
    const unsigned char* b = static_cast<const unsigned char*>(p);
    acc = acc + b[0]; acc = acc + b[bytes - 1]; acc = acc + b[bytes / 2];
}

size_t wf_elems(int NP) { const size_t nb = NP / WT; return nb * (nb + 1) / 2 * WT_TILE_DOUBLES; }
size_t wf_overrun_elems() { return WT_STEP_DOUBLES; }

void launch_gram(hipStream_t, const double* Xs, int D, int, int NP, int, double, double, double* K) {
    touch_r(Xs, (size_t)NP * xs_stride(D) * 8); touch_w(K, (size_t)NP * NP * 8);
}
void launch_scale_x(hipStream_t, const double* X, int N, int NP, int D, const double* inv_ls, double* Xs64, void* Xm, int dtype) {
    touch_r(X, (size_t)N * D * 8); touch_r(inv_ls, MAX_D * 8); touch_w(Xs64, (size_t)NP * xs_stride(D) * 8);
    touch_w(Xm, (size_t)NP * xs_stride(D) * esz(dtype));
}
void launch_dot(hipStream_t, const double* a, const double* b, int64_t n, double* out) { touch_r(a, (size_t)n * 8); touch_r(b, (size_t)n * 8); *out = 0.0; }
void launch_add_lower(hipStream_t, double* K, const double* S, int N, int NP) { touch_r(S, (size_t)N * N * 8); touch_w(K, (size_t)NP * NP * 8); }
void fit_aux_release(FitAux&) {}
size_t factor_scratch_doubles(int NP) { return factor_scratch_doubles_of(NP); }
// replays the plan's scratch regions (gpt_fit_plan.h) inside the buffer the orchestration allocated with factor_scratch_doubles
void launch_factor_inverse(hipStream_t, double* K, double* W, int NP, int* info, double* scratch, FitAux*, hipEvent_t) {
    touch_w(K, (size_t)NP * NP * 8); touch_w(W, (size_t)NP * NP * 8); *info = 0;
    const FitPlan pl = fit_plan(NP);
    for (const FitOp& op : pl.ops) {
        if (op.r0_size) touch_w(scratch + op.r0, op.r0_size * 8);
        if (op.r1_size) touch_w(scratch + op.r1, op.r1_size * 8);
    }
}
void launch_alpha(hipStream_t, const double* W, const double* Y4, int, int NP, double* tmp4, double* A4, double* scratch) {
    touch_r(W, (size_t)NP * NP * 8); touch_r(Y4, (size_t)NP * 32); touch_w(tmp4, (size_t)NP * 32); touch_w(A4, (size_t)NP * 32);
    touch_w(scratch, (size_t)(NP / 512) * NP * 32);
}
void launch_pack_w(hipStream_t, const double* W, int, int NP, void* Wf, int dtype, int task, double) {
    touch_r(W, (size_t)NP * NP * 8);
    touch_w(static_cast<unsigned char*>(Wf) + (size_t)task * wf_elems(NP) * esz(dtype), wf_elems(NP) * esz(dtype));
}
void launch_store4(hipStream_t, const double* src4, int rows, void* dst4, int dtype, int, int, int, double) {
    touch_r(src4, (size_t)rows * 32); touch_w(dst4, (size_t)rows * 4 * esz(dtype));
}
void launch_logdet(hipStream_t, const double* K, int, int NP, double* out) { touch_r(K, (size_t)NP * NP * 8); *out = 0.0; }
void launch_kinv(hipStream_t, const double* W, int NP, double* Kout) { touch_r(W, (size_t)NP * NP * 8); touch_w(Kout, (size_t)NP * NP * 8); }
void launch_cov(hipStream_t, const KernelParams& p, const double* Xs, const double* W, const double* Xq, int64_t M, int Mp, double* KsT,
                double* V, double* VtV, double* cov) {
    touch_r(Xs, (size_t)p.NP * xs_stride(p.D) * 8); touch_r(W, (size_t)p.NP * p.NP * 8); touch_r(Xq, (size_t)M * p.D * 8);
    touch_w(KsT, (size_t)p.NP * Mp * 8); touch_w(V, (size_t)p.NP * Mp * 8); touch_w(VtV, (size_t)Mp * Mp * 8); touch_w(cov, (size_t)M * M * 8);
}
void launch_lml_terms(hipStream_t, const double* Xs, int D, const double* A4, int npass, const double* Kinv, int, int NP, int, int, double,
                      double* partial, double* out) {
    touch_r(Xs, (size_t)NP * xs_stride(D) * 8); touch_r(A4, (size_t)npass * NP * 32); touch_r(Kinv, (size_t)NP * NP * 8);
    touch_w(partial, (size_t)(NP / 64) * (NP / 64) * LML_PARTIAL_STRIDE * 8); touch_w(out, LML_TERMS * 8);
}
void launch_mean_jac(hipStream_t, const KernelParams& p, const void* Xs, const void* A4, const void* Xq, int64_t M, void* mean, void* J) {
    const size_t e = esz(p.dtype);
    touch_r(Xs, (size_t)p.NP * xs_stride(p.D) * e); touch_r(A4, (size_t)((p.O + 3) / 4) * p.NP * 4 * e); touch_r(Xq, (size_t)M * p.D * e);
    touch_w(mean, (size_t)M * p.O * e); touch_w(J, (size_t)M * p.O * p.D * e);
}
void launch_var(hipStream_t, const KernelParams& p, const VarWorkspace& ws, const void* Xs, const void* Wf, const void* Xq, int64_t M,
                int ncomp, void* var, void* Jvar, void* dvar, const double* hdr) {
    if (M <= 0 || !ws.plan) return;
    const size_t e = esz(p.dtype);
    const VarPlanHost& pl = *ws.plan;
    touch_r(Xs, (size_t)p.NP * xs_stride(p.D) * e); touch_r(Xq, (size_t)M * p.D * e); touch_r(hdr, (16 + p.ntask) * 8);
    touch_r(Wf, ((size_t)p.ntask * wf_elems(p.NP) + wf_overrun_elems()) * e);
    unsigned char* base = static_cast<unsigned char*>(ws.plan_dev);                   // the uploaded image, as the kernels index it
    const int* item_begin = pl.d.item_begin;
    touch_r(item_begin, (size_t)(pl.d.P + 1) * 4);
    (void)base;
    for (int q = 0; q < pl.d.P; ++q) {
        touch_w(static_cast<unsigned char*>(ws.bscratch) + (size_t)q * p.NP * VAR_COLS * e, (size_t)p.NP * VAR_COLS * e);
        for (int i = item_begin[q]; i < item_begin[q + 1]; ++i) {
            const VarItem it = pl.d.items[i];
            if (it.vslot >= 0) touch_w(static_cast<unsigned char*>(ws.vslab) + (size_t)it.vslot * VAR_VSLOT * e, (size_t)VAR_VSLOT * e);
            if (it.slot >= 0) touch_w(static_cast<unsigned char*>(ws.slab) + (size_t)it.slot * VAR_SLOT * e, (size_t)VAR_SLOT * e);
        }
    }
    touch_w(ws.slab, (size_t)pl.d.nfull * p.ntask * VAR_SLOT * e);
    for (int i = 0; i < pl.d.n_splits; ++i) {
        const VarSplit sp = pl.d.splits[i];
        touch_r(static_cast<unsigned char*>(ws.vslab) + (size_t)sp.v_begin * VAR_VSLOT * e, (size_t)(sp.v_end - sp.v_begin) * VAR_VSLOT * e);
        touch_w(static_cast<unsigned char*>(ws.slab) + (size_t)sp.slot * VAR_SLOT * e, (size_t)VAR_SPLIT_SLOTS * VAR_SLOT * e);
    }
    for (int64_t c = 0; c < pl.d.ncb - pl.d.nfull; ++c)
        for (int t = 0; t < p.ntask; ++t) {
            const int b = pl.d.fin[2 * (c * p.ntask + t)], en = pl.d.fin[2 * (c * p.ntask + t) + 1];
            touch_r(static_cast<unsigned char*>(ws.slab) + (size_t)b * VAR_SLOT * e, (size_t)(en - b) * VAR_SLOT * e);
        }
    (void)var_cols_per_query(p.D, ncomp);
    touch_w(var, (size_t)M * p.ntask * e); touch_w(Jvar, (size_t)M * p.ntask * p.D * e); touch_w(dvar, (size_t)M * p.D * e);
}

}  // namespace gpt
