// extern "C" entry points of libwc_hip.so (declared in include/wc_hip.h): argument checks,
// workspace carving and the launch sequence of each stage.  Nothing here allocates, synchronises
// or keeps state, so every entry point is stream-ordered and graph-capturable.
#include "../../include/wc_hip.h"
#include "wc_common.h"

namespace {

inline bool bad_channels(int C) { return C < 32 || C > 1024 || (C % 32) != 0; }

struct Carver {
    char* p; size_t left;
    Carver(void* ws, size_t bytes) : p(static_cast<char*>(ws)), left(bytes) {}
    template <typename T> T* take(size_t n) {
        const size_t b = wc_align_up(n * sizeof(T), 256);
        T* r = reinterpret_cast<T*>(p);
        p += b; left = (left >= b) ? left - b : 0;
        return r;
    }
};
inline size_t slot_bytes(size_t n, size_t elem) { return wc_align_up(n * elem, 256); }

#define WC_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return (int)e_; } while (0)

}  // namespace

extern "C" {

int wc_abi_version(void) { return WC_ABI_VERSION; }

const char* wc_error_string(int code)
{
    switch (code) {
        case WC_OK: return "ok";
        case WC_ERR_NULL: return "required pointer is NULL";
        case WC_ERR_SHAPE: return "bad shape (M/N/HW/Kc must be positive)";
        case WC_ERR_CHANNELS: return "C must be a multiple of 32 in [32, 1024]";
        case WC_ERR_WORKSPACE: return "workspace too small";
        case WC_ERR_ARG: return "eps/momentum/ddof out of range";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown error";
    }
}

// ---------------------------------------------------------------------------------------------
namespace {
struct XtyPlan { int fast; int nslab; int nsplit; int64_t rps; int ntypes; };
// one plan per call shape: the fast kernel and its gated exact redo must write the same slab layout
XtyPlan plan_xty(int64_t N, int64_t HW, int C, int per_sample, int sym)
{
    XtyPlan p = {};
    p.nslab = wc_fast_xty_plan(N, HW, C, per_sample, !sym, &p.nsplit, &p.rps, &p.ntypes);
    p.fast = p.nslab > 0;
    if (!p.fast) p.nslab = wc_xty_plan(N, HW, C, per_sample, sym, &p.nsplit, &p.rps);
    return p;
}
}  // namespace

size_t wc_stats_workspace_bytes(int64_t M, int C, int groups)
{
    if (M <= 0 || groups <= 0 || (M % groups) != 0 || bad_channels(C)) return 0;
    const XtyPlan p = plan_xty(groups, M / groups, C, groups > 1, 1);
    return 256 + 2 * slot_bytes(C, 4) + slot_bytes((size_t)2 * groups * C, 8) + slot_bytes((size_t)p.nslab * C, 4) + slot_bytes((size_t)p.nslab * C, 8) +
           slot_bytes((size_t)p.nslab * C * C, 8);
}

int wc_stats_f32(const float* x, int64_t M, int C, int groups, double* sum, double* xtx,
                 void* ws, size_t ws_bytes, wc_stream_t stream)
{
    if (!x || !sum || !xtx || !ws) return WC_ERR_NULL;
    if (M <= 0 || groups <= 0 || (M % groups) != 0) return WC_ERR_SHAPE;
    if (bad_channels(C)) return WC_ERR_CHANNELS;
    if (ws_bytes < wc_stats_workspace_bytes(M, C, groups)) return WC_ERR_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    // a statistic group is a run of M/groups consecutive rows; slabs never cross groups (the per-sample machinery)
    const int per_seg = groups > 1;
    const int64_t Ns = groups, HWs = M / groups;
    const XtyPlan p = plan_xty(Ns, HWs, C, per_seg, 1);
    Carver cv(ws, ws_bytes);
    int* gate = cv.take<int>(64);
    float* shift = cv.take<float>(C);
    float* scale = cv.take<float>(C);
    double* Sp = cv.take<double>((size_t)2 * groups * C);        // column sums | the diagonal's slab sums (bias compensation)
    float* colsum = cv.take<float>((size_t)p.nslab * C);
    double* dfix = cv.take<double>((size_t)p.nslab * C);
    double* P = cv.take<double>((size_t)p.nslab * C * C);

    if (p.fast) WC_TRY(wc_launch_subsample_mean_scale(x, M, C, shift, scale, gate, st));    // shift, scale, gate := 0
    else WC_TRY(wc_launch_subsample_mean(x, M, C, shift, st));
    WcXtyArgs a = {};
    a.X = x; a.Y = x; a.cx = shift; a.cy = shift; a.N = Ns; a.HW = HWs; a.per_sample = per_seg; a.nsplit = p.nsplit;
    a.rows_per_slab = p.rps; a.C = C; a.sym = 1; a.P = P; a.colsum = colsum;
    if (p.fast) {
        WC_TRY(wc_launch_fast_xty(x, x, shift, shift, scale, scale, Ns, HWs, C, per_seg, p.nsplit, p.rps, p.nslab, p.ntypes,
                                  P, colsum, dfix, gate, st));
        a.gate = gate;                       // exact redo, a no-op unless the fp16 range was exceeded
    }
    WC_TRY(wc_launch_xty(a, p.nslab, st));
    WC_TRY(wc_launch_stats_finalize(P, colsum, shift, p.nslab / groups, HWs, C, groups, Sp, sum, xtx,
                                    p.fast ? dfix : nullptr, p.fast ? gate : nullptr, st, p.fast ? wc_fast_xty_offdiag_bias() : 0.0));
    return WC_OK;
}

// K1 + K2 as one call (training mode, per-replica statistics): the moments never leave the workspace
size_t wc_whiten_workspace_bytes(int64_t M, int C, int groups)
{
    const size_t a = wc_stats_workspace_bytes(M, C, groups);
    if (a == 0) return 0;
    return a + slot_bytes((size_t)groups * C, 8) + slot_bytes((size_t)groups * C * C, 8);
}

int wc_whiten_f32(const float* x, int64_t M, int C, int groups, double eps, double momentum, int ddof,
                  float* moving_mean, float* moving_cov, float* mu, float* chan_scale, double* L, double* W,
                  void* ws, size_t ws_bytes, wc_stream_t stream)
{
    if (!x || !mu || !L || !W || !ws) return WC_ERR_NULL;
    if ((moving_mean == nullptr) != (moving_cov == nullptr)) return WC_ERR_NULL;
    if (M <= 0 || groups <= 0 || (M % groups) != 0 || M / groups <= ddof) return WC_ERR_SHAPE;
    if (bad_channels(C)) return WC_ERR_CHANNELS;
    if (!(eps > 0.0) || eps >= 1.0 || momentum < 0.0 || momentum > 1.0 || ddof < 0 || ddof > 1) return WC_ERR_ARG;
    if (ws_bytes < wc_whiten_workspace_bytes(M, C, groups)) return WC_ERR_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int per_seg = groups > 1;
    const int64_t Ns = groups, HWs = M / groups;
    const XtyPlan p = plan_xty(Ns, HWs, C, per_seg, 1);
    Carver cv(ws, ws_bytes);
    int* gate = cv.take<int>(64);
    float* shift = cv.take<float>(C);
    float* scale = cv.take<float>(C);
    double* Sp = cv.take<double>((size_t)2 * groups * C);        // column sums | the diagonal's slab sums (bias compensation)
    float* colsum = cv.take<float>((size_t)p.nslab * C);
    double* dfix = cv.take<double>((size_t)p.nslab * C);
    double* P = cv.take<double>((size_t)p.nslab * C * C);
    double* sum_scratch = cv.take<double>((size_t)groups * C);
    double* tmp = cv.take<double>((size_t)groups * C * C);

    if (p.fast) WC_TRY(wc_launch_subsample_mean_scale(x, M, C, shift, scale, gate, st));
    else WC_TRY(wc_launch_subsample_mean(x, M, C, shift, st));
    WcXtyArgs a = {};
    a.X = x; a.Y = x; a.cx = shift; a.cy = shift; a.N = Ns; a.HW = HWs; a.per_sample = per_seg; a.nsplit = p.nsplit;
    a.rows_per_slab = p.rps; a.C = C; a.sym = 1; a.P = P; a.colsum = colsum;
    if (p.fast) {
        WC_TRY(wc_launch_fast_xty(x, x, shift, shift, scale, scale, Ns, HWs, C, per_seg, p.nsplit, p.rps, p.nslab, p.ntypes,
                                  P, colsum, dfix, gate, st));
        a.gate = gate;
    }
    WC_TRY(wc_launch_xty(a, p.nslab, st));
    WC_TRY(wc_launch_stats_prepare(P, colsum, shift, p.nslab / groups, HWs, C, groups, Sp, sum_scratch, p.fast ? dfix : nullptr,
                                   p.fast ? gate : nullptr, eps, momentum, ddof, moving_mean, moving_cov, mu, chan_scale, L, st, tmp,
                                   p.fast ? wc_fast_xty_offdiag_bias() : 0.0));
    if (wc_factor_is_fused(C)) {
        WC_TRY(wc_launch_factor_fused(L, W, tmp, C, groups, st));
        return WC_OK;
    }
    WC_TRY(wc_launch_cholesky(L, C, groups, st));
    WC_TRY(wc_launch_tri_inverse(L, W, tmp, C, groups, st));
    return WC_OK;
}

// ---------------------------------------------------------------------------------------------
size_t wc_factor_workspace_bytes(int C, int groups)
{
    if (bad_channels(C) || groups <= 0) return 0;
    return slot_bytes((size_t)groups * C * C, 8);
}

// K2 in one launch (128 <= C <= 256): the inverse's workgroups wait for the factorising one with a bounded spin; a wait that ran
// out poisons W with a NaN and sets word [16 g + 1] of the row-block counters (group g) -- `groups` words 64 bytes apart from the
// returned byte offset into the call's workspace, zero after a clean call.  0: this shape has no such launch (nothing to check).
size_t wc_factor_error_offset(int C, int groups)
{
    if (bad_channels(C) || groups <= 0 || !wc_factor_is_fused(C) || C < 128 || C > 256) return 0;
    return (size_t)groups * C * 16 * 8 + 4;
}

size_t wc_whiten_error_offset(int64_t M, int C, int groups)
{
    const size_t in_tmp = wc_factor_error_offset(C, groups);
    const size_t a = wc_stats_workspace_bytes(M, C, groups);
    if (in_tmp == 0 || a == 0) return 0;
    return a + slot_bytes((size_t)groups * C, 8) + in_tmp;           // tmp is the last block of wc_whiten_f32's workspace
}

int wc_factor_f64(const double* sum, const double* xtx, int64_t M, int C, int groups, double eps, double momentum, int ddof,
                  int training, float* moving_mean, float* moving_cov, float* mu, float* chan_scale, double* L, double* W,
                  void* ws, size_t ws_bytes, wc_stream_t stream)
{
    if (!mu || !L || !W || !ws) return WC_ERR_NULL;
    if (training && (!sum || !xtx)) return WC_ERR_NULL;
    if (!training && (!moving_mean || !moving_cov)) return WC_ERR_NULL;
    if (bad_channels(C)) return WC_ERR_CHANNELS;
    if (groups <= 0 || (training && (M <= ddof || M <= 0))) return WC_ERR_SHAPE;
    if (!(eps > 0.0) || eps >= 1.0 || momentum < 0.0 || momentum > 1.0 || ddof < 0 || ddof > 1) return WC_ERR_ARG;
    if (ws_bytes < wc_factor_workspace_bytes(C, groups)) return WC_ERR_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    Carver cv(ws, ws_bytes);
    double* tmp = cv.take<double>((size_t)groups * C * C);
    WC_TRY(wc_launch_factor_prepare(sum, xtx, M, C, eps, momentum, ddof, training, groups, moving_mean, moving_cov, mu, chan_scale, L, st, tmp));
    if (wc_factor_is_fused(C)) {
        WC_TRY(wc_launch_factor_fused(L, W, tmp, C, groups, st));
        return WC_OK;
    }
    WC_TRY(wc_launch_cholesky(L, C, groups, st));
    WC_TRY(wc_launch_tri_inverse(L, W, tmp, C, groups, st));
    return WC_OK;
}

// ---------------------------------------------------------------------------------------------
size_t wc_color_workspace_bytes(int C, int Kc)
{
    (void)Kc;
    if (bad_channels(C)) return 0;
    return 256;     // none needed today; kept in the ABI so a caller never has to change
}

size_t wc_apply_plan_bytes(int C, int Kc)
{
    if (Kc <= 0 || bad_channels(C)) return 0;
    return wc_fast_affine_workspace(C, Kc);
}

int wc_color_f32(const double* W, const float* gamma, int Kc, int C, int groups, int per_group, float* A, float* At,
                 const float* chan_scale, void* plan, void* ws, size_t ws_bytes, wc_stream_t stream)
{
    (void)ws; (void)ws_bytes;
    if (!W || !A) return WC_ERR_NULL;
    if (bad_channels(C)) return WC_ERR_CHANNELS;
    if (Kc <= 0 || groups <= 0 || (!gamma && Kc != 1)) return WC_ERR_SHAPE;
    const int slots = groups * Kc;                    // table index = group * Kc + class slot
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool want_plan = plan && chan_scale && (C == 32 || C == 64 || C == 128 || C == 256);
    if (!gamma) {
        WC_TRY(wc_launch_transpose_to_f32(W, C, groups, A, At, st));        // all groups in one launch
        if (want_plan) {
            WC_TRY(wc_launch_fast_plan_tables(A, slots, C, plan, st, chan_scale));      // + the scales into the plan
        }
        return WC_OK;
    }
    const int64_t CC = (int64_t)C * C;
    WcGemm g = {};
    g.A = W; g.a_rs = 1; g.a_cs = C; g.a_bs = 0;                         // W^T
    g.B = gamma; g.b_is_f32 = 1; g.b_rs = C; g.b_cs = 1; g.b_bs = CC;
    g.Cm = A; g.c_is_f32 = 1; g.c_rs = C; g.c_cs = 1; g.c_bs = CC;
    g.Cm2 = At; g.c2_rs = 1; g.c2_cs = C; g.c2_bs = CC;
    g.m = C; g.n = C; g.k = C; g.batch = Kc; g.nred = 1; g.alpha = 1.0; g.epi = WC_EPI_NONE;
    g.batch2 = groups; g.a_b2s = CC; g.b_b2s = per_group ? (int64_t)Kc * CC : 0; g.c_b2s = (int64_t)Kc * CC;      // A[g*Kc + k] = W_g^T Gamma_k (Gamma_{g*Kc+k} if per_group)
    WC_TRY(wc_launch_gemm(g, st));
    if (want_plan) {     // the apply's fp16 tables, built once here instead of inside every wc_apply_f32 call
        WC_TRY(wc_launch_fast_plan_tables(A, slots, C, plan, st, chan_scale));      // + the scales into the plan
    }
    return WC_OK;
}

// grouped forward: one common centre + per-slot bias so that y = (x - center) A[s] + bias[s] equals
// (x - mu_g) A[g*Kc+k] + beta_k for every sample of group g and class slot k
int wc_group_bias_f32(const float* mu, const float* A, const float* beta, int groups, int Kc, int C, int per_group,
                      float* center, float* bias, wc_stream_t stream)
{
    if (!mu || !A || !center || !bias) return WC_ERR_NULL;
    if (groups <= 0 || Kc <= 0) return WC_ERR_SHAPE;
    if (bad_channels(C)) return WC_ERR_CHANNELS;
    WC_TRY(wc_launch_group_bias(mu, A, beta, groups, Kc, C, per_group, center, bias, static_cast<hipStream_t>(stream)));
    return WC_OK;
}

// The planes route's glue (ABI 5): as wc_color_f32 / wc_group_bias_f32, with the additive term of an apply on a pre-split input
// produced by the same launches (no wc_split_bias_f32 launch in front of K3).
int wc_color_split_f32(const double* W, const float* gamma, int Kc, int C, float* A, float* At, const float* xs_scale, const float* xs_center,
                       const float* mu, const float* beta, void* plan, float* bias_eff, void* ws, size_t ws_bytes, wc_stream_t stream)
{
    (void)ws; (void)ws_bytes;
    if (!W || !A || !xs_scale || !plan || !bias_eff) return WC_ERR_NULL;
    if (bad_channels(C)) return WC_ERR_CHANNELS;
    if (!(C == 32 || C == 64 || C == 128 || C == 256)) return WC_ERR_CHANNELS;
    if (Kc <= 0 || (!gamma && Kc != 1)) return WC_ERR_SHAPE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (!gamma) WC_TRY(wc_launch_transpose_to_f32(W, C, 1, A, At, st));
    else {
        const int64_t CC = (int64_t)C * C;
        WcGemm g = {};
        g.A = W; g.a_rs = 1; g.a_cs = C; g.a_bs = 0;                         // W^T
        g.B = gamma; g.b_is_f32 = 1; g.b_rs = C; g.b_cs = 1; g.b_bs = CC;
        g.Cm = A; g.c_is_f32 = 1; g.c_rs = C; g.c_cs = 1; g.c_bs = CC;
        g.Cm2 = At; g.c2_rs = 1; g.c2_cs = C; g.c2_bs = CC;
        g.m = C; g.n = C; g.k = C; g.batch = Kc; g.nred = 1; g.alpha = 1.0; g.epi = WC_EPI_NONE;
        g.batch2 = 1; g.a_b2s = CC; g.b_b2s = 0; g.c_b2s = (int64_t)Kc * CC;
        WC_TRY(wc_launch_gemm(g, st));
    }
    // the tables for the planes' scales AND bias_eff[k] = beta[k] + (center - mu) A[k] in ONE launch
    WC_TRY(wc_launch_fast_plan_tables_bias(A, Kc, C, plan, st, xs_scale, beta, xs_center, mu, bias_eff));
    return WC_OK;
}

int wc_group_bias_centered_f32(const float* mu, const float* A, const float* beta, const float* center, int groups, int Kc, int C,
                               int per_group, float* bias, wc_stream_t stream)
{
    if (!mu || !A || !center || !bias) return WC_ERR_NULL;
    if (groups <= 0 || Kc <= 0) return WC_ERR_SHAPE;
    if (bad_channels(C)) return WC_ERR_CHANNELS;
    WC_TRY(wc_launch_group_bias(mu, A, beta, groups, Kc, C, per_group, nullptr, bias, static_cast<hipStream_t>(stream), center));
    return WC_OK;
}

// ---------------------------------------------------------------------------------------------
// the dictionary mix of the soft-assignment coloring (wc_mix.hip)
int wc_factor_mix_supported(int E, int C) { return wc_mix_supported(E, C) ? 1 : 0; }

int wc_factor_mix_f32(const float* dict, const float* alpha, const int32_t* idx, const float* base, int E, int C, int K, int Kc,
                      float* out, wc_stream_t stream)
{
    if (!dict || !alpha || !out) return WC_ERR_NULL;
    if (K <= 0 || Kc <= 0 || (!idx && Kc != K)) return WC_ERR_SHAPE;
    if (!wc_mix_supported(E, C)) return WC_ERR_SHAPE;
    WC_TRY(wc_launch_mix_fwd(dict, alpha, idx, base, E, C, Kc, out, static_cast<hipStream_t>(stream)));
    return WC_OK;
}

size_t wc_factor_mix_bwd_workspace_bytes(int E, int Kc) { return (E <= 0 || Kc <= 0) ? 0 : wc_mix_bwd_workspace(E, Kc); }

int wc_factor_mix_bwd_f32(const float* dict, const float* alpha, const int32_t* idx, const float* dout, int E, int C, int K, int Kc,
                          float* ddict, float* dalpha, float* dbase, void* ws, size_t ws_bytes, wc_stream_t stream)
{
    if (!dict || !alpha || !dout) return WC_ERR_NULL;
    if (K <= 0 || Kc <= 0 || (!idx && Kc != K)) return WC_ERR_SHAPE;
    if (!wc_mix_supported(E, C)) return WC_ERR_SHAPE;
    if (dalpha && (!ws || ws_bytes < wc_mix_bwd_workspace(E, Kc))) return WC_ERR_WORKSPACE;
    WC_TRY(wc_launch_mix_bwd(dict, alpha, idx, dout, E, C, K, Kc, ddict, dalpha, dbase, ws, static_cast<hipStream_t>(stream)));
    return WC_OK;
}

// ---------------------------------------------------------------------------------------------
size_t wc_apply_workspace_bytes(int64_t N, int64_t HW, int C, int Kc)
{
    if (N <= 0 || HW <= 0 || Kc <= 0 || bad_channels(C)) return 0;
    if (!wc_fast_affine_supported(N, HW, C, Kc > 1)) return 256;
    return wc_fast_affine_workspace(C, Kc);
}

int wc_apply_f32(const float* x, const float* mu, const float* A, const float* bias, const int32_t* slot,
                 int64_t N, int64_t HW, int C, int Kc, float* y, const void* plan,
                 void* ws, size_t ws_bytes, wc_stream_t stream)
{
    return wc_apply_act_f32(x, mu, A, bias, slot, N, HW, C, Kc, 0, y, plan, ws, ws_bytes, stream);
}

int wc_apply_act_f32(const float* x, const float* mu, const float* A, const float* bias, const int32_t* slot,
                     int64_t N, int64_t HW, int C, int Kc, int relu, float* y, const void* plan,
                     void* ws, size_t ws_bytes, wc_stream_t stream)
{
    if (!x || !A || !y) return WC_ERR_NULL;
    if (relu != 0 && relu != 1) return WC_ERR_ARG;
    if (N <= 0 || HW <= 0 || Kc <= 0) return WC_ERR_SHAPE;
    if (bad_channels(C)) return WC_ERR_CHANNELS;
    hipStream_t st = static_cast<hipStream_t>(stream);
    WcRowsGemmArgs a = {};
    a.in[0] = x; a.center[0] = mu; a.B[0] = A; a.B_slot_stride[0] = (int64_t)C * C;
    a.bias = bias; a.sub = nullptr; a.slot = slot; a.N = N; a.HW = HW; a.C = C; a.nstreams = 1; a.out = y; a.relu = relu;
    const bool eligible = wc_fast_affine_supported(N, HW, C, slot != nullptr);
    if (eligible && plan) {                  // tables prepared by wc_color_f32: one launch
        WC_TRY(wc_launch_fast_affine_planned(x, mu, A, Kc, false, bias, nullptr, slot, N, HW, C, 2 * relu, y, plan, st));
        return WC_OK;
    }
    if (eligible && ws && ws_bytes >= wc_fast_affine_workspace(C, Kc)) {
        WC_TRY(wc_launch_fast_affine(x, mu, A, Kc, false, bias, nullptr, slot, N, HW, C, 2 * relu, y, ws, st));
        return WC_OK;
    }
    WC_TRY(wc_launch_rows_gemm(a, st));
    return WC_OK;
}

// K3 + ReLU + the ReLU's one-bit gradient mask (ABI 4)
size_t wc_relu_mask_bytes(int64_t M, int C)
{
    if (M <= 0 || (M % 32) != 0 || bad_channels(C)) return 0;
    return (size_t)(M / 32) * C * 4;
}

int wc_relu_mask_apply_f32(const float* gy, const void* relu_mask, int64_t M, int C, float* out, wc_stream_t stream)
{
    if (!gy || !relu_mask || !out) return WC_ERR_NULL;
    if (M <= 0 || (M % 32) != 0) return WC_ERR_SHAPE;
    if (bad_channels(C)) return WC_ERR_CHANNELS;
    WC_TRY(wc_launch_relu_mask_bits(gy, static_cast<const unsigned*>(relu_mask), out, M, C, static_cast<hipStream_t>(stream)));
    return WC_OK;
}

int wc_apply_mask_f32(const float* x, const float* mu, const float* A, const float* bias, const int32_t* slot,
                      int64_t N, int64_t HW, int C, int Kc, float* y, void* relu_mask, const void* plan,
                      void* ws, size_t ws_bytes, wc_stream_t stream)
{
    if (!relu_mask) return WC_ERR_NULL;
    if (N <= 0 || HW <= 0 || ((N * HW) % 32) != 0) return WC_ERR_SHAPE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    unsigned* mask = static_cast<unsigned*>(relu_mask);
    if (x && A && y && !bad_channels(C) && Kc > 0 && plan && wc_fast_affine_writes_mask(N, HW, C)) {      // the ring kernel writes the bits from its epilogue
        WC_TRY(wc_launch_fast_affine_planned(x, mu, A, Kc, false, bias, nullptr, slot, N, HW, C, 2, y, plan, st, mask));
        return WC_OK;
    }
    const int rc = wc_apply_act_f32(x, mu, A, bias, slot, N, HW, C, Kc, 1, y, plan, ws, ws_bytes, stream);
    if (rc != WC_OK) return rc;
    WC_TRY(wc_launch_mask_from_y(y, N * HW, C, mask, st));          // every other path: one pass over y
    return WC_OK;
}

// K3 -> convolution hand-off (ABI 4): the site's output as the next convolution's operand
int wc_apply_planes_supported(int64_t N, int64_t HW, int C)
{
    return (N > 0 && HW > 0 && !bad_channels(C) && wc_fast_affine_writes_planes(N, HW, C)) ? 1 : 0;
}

size_t wc_apply_planes_scale_floats(void) { return 2 + 2 * 1024; }

int wc_out_scale_f32(const float* gamma, const float* beta, int K, int C, float* oscale, wc_stream_t stream)
{
    if (!oscale) return WC_ERR_NULL;
    if (K <= 0 || K > 1024) return WC_ERR_ARG;
    if (bad_channels(C)) return WC_ERR_CHANNELS;
    WC_TRY(wc_launch_out_scale(gamma, beta, K, C, oscale, static_cast<hipStream_t>(stream)));
    return WC_OK;
}

int wc_apply_planes_f32(const float* x, const float* mu, const float* A, const float* bias, const int32_t* slot,
                        int64_t N, int64_t HW, int C, int Kc, int relu, void* planes, float* oscale, void* relu_mask,
                        const void* plan, wc_stream_t stream)
{
    if (!x || !A || !planes || !oscale || !plan) return WC_ERR_NULL;
    if (relu != 0 && relu != 1) return WC_ERR_ARG;
    if (relu_mask && !relu) return WC_ERR_ARG;
    if (N <= 0 || HW <= 0 || Kc <= 0) return WC_ERR_SHAPE;
    if (bad_channels(C)) return WC_ERR_CHANNELS;
    if (!wc_fast_affine_writes_planes(N, HW, C)) return WC_ERR_SHAPE;
    WC_TRY(wc_launch_fast_affine_planned(x, mu, A, Kc, false, bias, nullptr, slot, N, HW, C, relu ? 2 : 0, nullptr, plan,
                                         static_cast<hipStream_t>(stream), static_cast<unsigned*>(relu_mask), planes, oscale));
    return WC_OK;
}

// ---------------------------------------------------------------------------------------------
// pre-split activations (ABI 4)
size_t wc_split_bytes(int64_t M, int C)
{
    if (M <= 0 || bad_channels(C)) return 0;
    return (size_t)M * C * 4;          // two fp16 planes: the bytes of the fp32 tensor
}

int wc_split_scales_f32(const float* x, int64_t M, int C, float* center, float* scale, int* flag, wc_stream_t stream)
{
    if (!x || !center || !scale || !flag) return WC_ERR_NULL;
    if (M <= 0) return WC_ERR_SHAPE;
    if (bad_channels(C)) return WC_ERR_CHANNELS;
    WC_TRY(wc_launch_subsample_mean_scale(x, M, C, center, scale, flag, static_cast<hipStream_t>(stream)));
    return WC_OK;
}

int wc_split_f32(const float* x, const float* center, const float* scale, int64_t M, int C, int relu, void* xs, int* flag,
                 wc_stream_t stream)
{
    if (!x || !scale || !xs) return WC_ERR_NULL;
    if (relu != 0 && relu != 1) return WC_ERR_ARG;
    if (M <= 0) return WC_ERR_SHAPE;
    if (bad_channels(C)) return WC_ERR_CHANNELS;
    WC_TRY(wc_launch_split_rows(x, center, scale, M, C, relu, xs, flag, static_cast<hipStream_t>(stream)));
    return WC_OK;
}

int wc_unsplit_f32(const void* xs, const float* center, const float* scale, int64_t M, int C, float* x, wc_stream_t stream)
{
    if (!x || !scale || !xs) return WC_ERR_NULL;
    if (M <= 0) return WC_ERR_SHAPE;
    if (bad_channels(C)) return WC_ERR_CHANNELS;
    WC_TRY(wc_launch_unsplit_rows(xs, center, scale, M, C, x, static_cast<hipStream_t>(stream)));
    return WC_OK;
}

// K1 on a pre-split input
int wc_stats_split_supported(int64_t M, int C, int groups)
{
    if (M <= 0 || groups <= 0 || (M % groups) != 0 || bad_channels(C)) return 0;
    int nsplit, ntypes; int64_t rps;
    return wc_split_xtx_plan(groups, M / groups, C, groups > 1, &nsplit, &rps, &ntypes) > 0 ? 1 : 0;
}

size_t wc_stats_split_workspace_bytes(int64_t M, int C, int groups)
{
    if (!wc_stats_split_supported(M, C, groups)) return 0;
    int nsplit, ntypes; int64_t rps;
    const int nslab = wc_split_xtx_plan(groups, M / groups, C, groups > 1, &nsplit, &rps, &ntypes);
    return 256 + slot_bytes((size_t)2 * groups * C, 8) + slot_bytes((size_t)nslab * C, 4) + slot_bytes((size_t)nslab * C, 8) +
           slot_bytes((size_t)nslab * C * C, 8);
}

int wc_stats_split_f16x2(const void* xs, const float* xs_center, const float* xs_scale, int64_t M, int C, int groups,
                         double* sum, double* xtx, void* ws, size_t ws_bytes, wc_stream_t stream)
{
    if (!xs || !xs_center || !xs_scale || !sum || !xtx || !ws) return WC_ERR_NULL;
    if (M <= 0 || groups <= 0 || (M % groups) != 0) return WC_ERR_SHAPE;
    if (bad_channels(C)) return WC_ERR_CHANNELS;
    if (!wc_stats_split_supported(M, C, groups)) return WC_ERR_SHAPE;
    if (ws_bytes < wc_stats_split_workspace_bytes(M, C, groups)) return WC_ERR_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int per_seg = groups > 1;
    const int64_t Ns = groups, HWs = M / groups;
    int nsplit, ntypes; int64_t rps;
    const int nslab = wc_split_xtx_plan(Ns, HWs, C, per_seg, &nsplit, &rps, &ntypes);
    Carver cv(ws, ws_bytes);
    (void)cv.take<int>(64);
    double* Sp = cv.take<double>((size_t)2 * groups * C);        // column sums | the diagonal's slab sums (bias compensation)
    float* colsum = cv.take<float>((size_t)nslab * C);
    double* dfix = cv.take<double>((size_t)nslab * C);
    double* P = cv.take<double>((size_t)nslab * C * C);
    WC_TRY(wc_launch_split_xtx(xs, xs_scale, Ns, HWs, C, per_seg, nsplit, rps, nslab, ntypes, P, colsum, dfix, st));
    // the planes hold g = (x - center) scale: the tail adds the centre's terms back (it is the "shift" of wc_stats_f32's tail)
    // (the planes kernel's off-diagonal bias measures the same as the fp32-input kernel's: tools/k1_bias_survey.py --split)
    WC_TRY(wc_launch_stats_finalize(P, colsum, xs_center, nslab / groups, HWs, C, groups, Sp, sum, xtx, dfix, nullptr, st,
                                    wc_fast_xty_offdiag_bias()));
    return WC_OK;
}

// K1 + K2 on a pre-split input in one call (ABI 5): wc_stats_split_f16x2 followed by wc_factor_f64(training = 1), the K1 tail and
// the K2 head as ONE launch (wc_whiten_f32's merge); the moments never leave the workspace
size_t wc_whiten_split_workspace_bytes(int64_t M, int C, int groups)
{
    const size_t a = wc_stats_split_workspace_bytes(M, C, groups);
    if (a == 0) return 0;
    return a + slot_bytes((size_t)groups * C, 8) + slot_bytes((size_t)groups * C * C, 8);
}

size_t wc_whiten_split_error_offset(int64_t M, int C, int groups)
{
    const size_t in_tmp = wc_factor_error_offset(C, groups);
    const size_t a = wc_stats_split_workspace_bytes(M, C, groups);
    if (in_tmp == 0 || a == 0) return 0;
    return a + slot_bytes((size_t)groups * C, 8) + in_tmp;
}

int wc_whiten_split_f16x2(const void* xs, const float* xs_center, const float* xs_scale, int64_t M, int C, int groups,
                          double eps, double momentum, int ddof, float* moving_mean, float* moving_cov,
                          float* mu, double* L, double* W, void* ws, size_t ws_bytes, wc_stream_t stream)
{
    if (!xs || !xs_center || !xs_scale || !mu || !L || !W || !ws) return WC_ERR_NULL;
    if ((moving_mean == nullptr) != (moving_cov == nullptr)) return WC_ERR_NULL;
    if (M <= 0 || groups <= 0 || (M % groups) != 0 || M / groups <= ddof) return WC_ERR_SHAPE;
    if (bad_channels(C)) return WC_ERR_CHANNELS;
    if (!(eps > 0.0) || eps >= 1.0 || momentum < 0.0 || momentum > 1.0 || ddof < 0 || ddof > 1) return WC_ERR_ARG;
    if (!wc_stats_split_supported(M, C, groups)) return WC_ERR_SHAPE;
    if (ws_bytes < wc_whiten_split_workspace_bytes(M, C, groups)) return WC_ERR_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int per_seg = groups > 1;
    const int64_t Ns = groups, HWs = M / groups;
    int nsplit, ntypes; int64_t rps;
    const int nslab = wc_split_xtx_plan(Ns, HWs, C, per_seg, &nsplit, &rps, &ntypes);
    Carver cv(ws, ws_bytes);
    (void)cv.take<int>(64);
    double* Sp = cv.take<double>((size_t)2 * groups * C);
    float* colsum = cv.take<float>((size_t)nslab * C);
    double* dfix = cv.take<double>((size_t)nslab * C);
    double* P = cv.take<double>((size_t)nslab * C * C);
    double* sum_scratch = cv.take<double>((size_t)groups * C);
    double* tmp = cv.take<double>((size_t)groups * C * C);
    WC_TRY(wc_launch_split_xtx(xs, xs_scale, Ns, HWs, C, per_seg, nsplit, rps, nslab, ntypes, P, colsum, dfix, st));
    // (chan_scale = NULL: the apply's input scales are the planes' own, xs_scale -- the caller gives them to wc_color_f32)
    WC_TRY(wc_launch_stats_prepare(P, colsum, xs_center, nslab / groups, HWs, C, groups, Sp, sum_scratch, dfix, nullptr, eps, momentum,
                                   ddof, moving_mean, moving_cov, mu, nullptr, L, st, tmp, wc_fast_xty_offdiag_bias()));
    if (wc_factor_is_fused(C)) {
        WC_TRY(wc_launch_factor_fused(L, W, tmp, C, groups, st));
        return WC_OK;
    }
    WC_TRY(wc_launch_cholesky(L, C, groups, st));
    WC_TRY(wc_launch_tri_inverse(L, W, tmp, C, groups, st));
    return WC_OK;
}

size_t wc_apply_split_workspace_bytes(int C, int Kc)
{
    if (Kc <= 0 || bad_channels(C)) return 0;
    return slot_bytes((size_t)Kc * C, 4) + wc_fast_affine_workspace(C, Kc);      // bias2 | a plan (when the caller has none)
}

int wc_split_bias_f32(const float* A, const float* bias, const float* xs_center, const float* mu, int Kc, int C, float* bias_eff,
                      wc_stream_t stream)
{
    if (!A || !bias_eff) return WC_ERR_NULL;
    if (Kc <= 0) return WC_ERR_SHAPE;
    if (bad_channels(C)) return WC_ERR_CHANNELS;
    WC_TRY(wc_launch_split_bias(A, bias, xs_center, mu, Kc, C, bias_eff, static_cast<hipStream_t>(stream)));
    return WC_OK;
}

int wc_apply_split_supported(int64_t N, int64_t HW, int C)
{
    return (N > 0 && HW > 0 && wc_split_apply_supported(N, HW, C)) ? 1 : 0;
}

int wc_apply_split_f16x2(const void* xs, const float* xs_center, const float* xs_scale, const float* mu, const float* A,
                         const float* bias, const int32_t* slot, int64_t N, int64_t HW, int C, int Kc, int relu,
                         float* y, const void* plan, void* ws, size_t ws_bytes, wc_stream_t stream)
{
    if (!y) return WC_ERR_NULL;
    return wc_apply_split_ex_f16x2(xs, xs_center, xs_scale, mu, A, bias, slot, N, HW, C, Kc, relu, y, nullptr, nullptr, nullptr,
                                   plan, ws, ws_bytes, stream);
}

// K3 on a pre-split input with the epilogues of wc_apply_mask_f32 / wc_apply_planes_f32 (ABI 5)
int wc_apply_split_ex_f16x2(const void* xs, const float* xs_center, const float* xs_scale, const float* mu, const float* A,
                            const float* bias, const int32_t* slot, int64_t N, int64_t HW, int C, int Kc, int relu,
                            float* y, void* relu_mask, void* planes, float* oscale,
                            const void* plan, void* ws, size_t ws_bytes, wc_stream_t stream)
{
    if (!xs || !xs_scale || !A || !ws) return WC_ERR_NULL;
    if ((y == nullptr) == (planes == nullptr)) return WC_ERR_NULL;            // exactly one destination
    if (planes && !oscale) return WC_ERR_NULL;
    if (relu != 0 && relu != 1) return WC_ERR_ARG;
    if (relu_mask && !relu) return WC_ERR_ARG;
    if (N <= 0 || HW <= 0 || Kc <= 0) return WC_ERR_SHAPE;
    if (bad_channels(C)) return WC_ERR_CHANNELS;
    if (!wc_split_apply_supported(N, HW, C)) return WC_ERR_SHAPE;
    if (relu_mask && ((N * HW) % 32) != 0) return WC_ERR_SHAPE;
    if (planes && (N * HW) / (8192 / C) > 1024 * 1024) return WC_ERR_SHAPE;
    const size_t need = slot_bytes((size_t)Kc * C, 4) + (plan ? 0 : wc_fast_affine_workspace(C, Kc));
    if (ws_bytes < need) return WC_ERR_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    Carver cv(ws, ws_bytes);
    float* bias2 = cv.take<float>((size_t)Kc * C);
    if (!plan) {
        void* own = cv.take<char>(wc_fast_affine_workspace(C, Kc));
        WC_TRY(wc_launch_fast_plan_tables(A, Kc, C, own, st, xs_scale));
        plan = own;
    }
    const float *pscale, *pcol; const void *phi, *plo;
    wc_fast_plan_parts(plan, C, Kc, &pscale, &pcol, &phi, &plo);
    // the kernel's additive term is beta + (center - mu) A; a caller that has folded it already (wc_split_bias_f32) passes it
    // as `bias` with mu = xs_center = NULL, and the call is ONE launch
    const float* eff = bias;
    if (xs_center || mu || !bias) {
        WC_TRY(wc_launch_split_bias(A, bias, xs_center, mu, Kc, C, bias2, st));
        eff = bias2;
    }
    WC_TRY(wc_launch_apply_split(xs, xs_scale, A, Kc, eff, slot, N, HW, C, relu, y, phi, plo, pcol, nullptr, st,
                                 static_cast<unsigned*>(relu_mask), planes, oscale));
    return WC_OK;
}

// ---------------------------------------------------------------------------------------------
// the residual add as the producer of the next site's input (ABI 5; wc_resadd.hip)
int wc_resadd_split_supported(int64_t N, int64_t H, int64_t W, int C)
{
    return (N > 0 && H > 0 && W > 0 && (C == 128 || C == 256) && N * H * W < ((int64_t)1 << 31)) ? 1 : 0;
}

static int resadd_check(const float* h, int64_t N, int64_t H, int64_t W, int C, int up)
{
    if (!h) return WC_ERR_NULL;
    if (up != 0 && up != 1) return WC_ERR_ARG;
    if (N <= 0 || H <= 0 || W <= 0 || N * H * W >= ((int64_t)1 << 31)) return WC_ERR_SHAPE;
    if (up && ((H % 2) != 0 || (W % 2) != 0)) return WC_ERR_SHAPE;
    if (bad_channels(C)) return WC_ERR_CHANNELS;
    return WC_OK;
}

int wc_resadd_f32(const float* h, const float* s, int64_t N, int64_t H, int64_t W, int C, int up, float* out, wc_stream_t stream)
{
    const int rc = resadd_check(h, N, H, W, C, up);
    if (rc != WC_OK) return rc;
    if (!out) return WC_ERR_NULL;
    WC_TRY(wc_launch_resadd(h, s, N, H, W, C, up, nullptr, nullptr, nullptr, nullptr, out, static_cast<hipStream_t>(stream)));
    return WC_OK;
}

int wc_resadd_split_f32(const float* h, const float* s, int64_t N, int64_t H, int64_t W, int C, int up,
                        void* xs, float* center, float* scale, int* flag, float* x32, wc_stream_t stream)
{
    const int rc = resadd_check(h, N, H, W, C, up);
    if (rc != WC_OK) return rc;
    if (!xs || !center || !scale || !flag) return WC_ERR_NULL;
    if (!wc_resadd_split_supported(N, H, W, C)) return WC_ERR_SHAPE;
    WC_TRY(wc_launch_resadd(h, s, N, H, W, C, up, xs, center, scale, flag, x32, static_cast<hipStream_t>(stream)));
    return WC_OK;
}

// ---------------------------------------------------------------------------------------------
// the producer feeding K1 (ABI 7; resadd_xtx_kernel of wc_resadd.hip): sample + pass with the covariance partials + gate, then the
// tails of wc_whiten_split_f16x2 / wc_stats_split_f16x2 on those partials
namespace {
struct PresumLayout { int nslab, nsplit, ntypes; int64_t rps; double *Sp, *dfix, *P, *sum_scratch, *tmp; float* colsum; int* wgflag; float* wgmax; };
// one carve for the producer and both tails (the layout of wc_whiten_split_f16x2's workspace with the fp32-input kernel's slab plan)
bool presum_layout(int64_t M, int C, int groups, void* ws, size_t ws_bytes, PresumLayout* o)
{
    o->nslab = wc_fast_xty_plan(groups, M / groups, C, groups > 1, 0, &o->nsplit, &o->rps, &o->ntypes);
    if (o->nslab <= 0) return false;
    Carver cv(ws, ws_bytes);
    (void)cv.take<int>(64);
    o->Sp = cv.take<double>((size_t)2 * groups * C);
    o->colsum = cv.take<float>((size_t)o->nslab * C);
    o->dfix = cv.take<double>((size_t)o->nslab * C);
    o->P = cv.take<double>((size_t)o->nslab * C * C);
    o->sum_scratch = cv.take<double>((size_t)groups * C);
    const int grid = wc_resadd_xtx_grid(o->nslab, o->ntypes);
    o->wgflag = cv.take<int>((size_t)grid);                                   // (in front of tmp: tmp stays the LAST block, wc_whiten_presummed_error_offset)
    o->wgmax = cv.take<float>((size_t)grid * C);
    o->tmp = cv.take<double>((size_t)groups * C * C);
    return true;
}
size_t presum_bytes(int64_t M, int C, int groups)
{
    if (M <= 0 || groups <= 0 || (M % groups) != 0 || bad_channels(C)) return 0;
    int nsplit, ntypes; int64_t rps;
    const int nslab = wc_fast_xty_plan(groups, M / groups, C, groups > 1, 0, &nsplit, &rps, &ntypes);
    if (nslab <= 0) return 0;
    const int grid = wc_resadd_xtx_grid(nslab, ntypes);
    return 256 + slot_bytes((size_t)2 * groups * C, 8) + slot_bytes((size_t)nslab * C, 4) + slot_bytes((size_t)nslab * C, 8) +
           slot_bytes((size_t)nslab * C * C, 8) + slot_bytes((size_t)groups * C, 8) + slot_bytes((size_t)grid, 4) + slot_bytes((size_t)grid * C, 4) +
           slot_bytes((size_t)groups * C * C, 8);
}
}  // namespace

int wc_resadd_stats_supported(int64_t N, int64_t H, int64_t W, int C, int up, int groups)
{
    if (!wc_resadd_split_supported(N, H, W, C) || groups <= 0) return 0;
    return wc_resadd_xtx_supported(N, H, W, C, up, groups) ? 1 : 0;
}

size_t wc_resadd_stats_workspace_bytes(int64_t N, int64_t H, int64_t W, int C, int groups)
{
    if (N <= 0 || H <= 0 || W <= 0) return 0;
    return presum_bytes(N * H * W, C, groups);
}

int wc_resadd_stats_split_f32(const float* h, const float* s, int64_t N, int64_t H, int64_t W, int C, int up, int groups,
                              void* xs, float* center, float* scale, int* flag, float* x32, void* ws, size_t ws_bytes, wc_stream_t stream)
{
    const int rc = resadd_check(h, N, H, W, C, up);
    if (rc != WC_OK) return rc;
    if (!s || !xs || !center || !scale || !flag || !ws) return WC_ERR_NULL;
    if (!wc_resadd_stats_supported(N, H, W, C, up, groups)) return WC_ERR_SHAPE;
    if (ws_bytes < wc_resadd_stats_workspace_bytes(N, H, W, C, groups)) return WC_ERR_WORKSPACE;
    PresumLayout l;
    if (!presum_layout(N * H * W, C, groups, ws, ws_bytes, &l)) return WC_ERR_SHAPE;
    WC_TRY(wc_launch_resadd_xtx(h, s, N, H, W, C, up, groups, xs, center, scale, flag, x32, l.nsplit, l.rps, l.nslab, l.ntypes,
                                l.P, l.colsum, l.dfix, l.wgflag, l.wgmax, static_cast<hipStream_t>(stream)));
    return WC_OK;
}

size_t wc_whiten_presummed_error_offset(int64_t M, int C, int groups)
{
    const size_t in_tmp = wc_factor_error_offset(C, groups);
    const size_t all = presum_bytes(M, C, groups);
    if (in_tmp == 0 || all == 0) return 0;
    return all - slot_bytes((size_t)groups * C * C, 8) + in_tmp;          // tmp is the last block of the workspace
}

int wc_whiten_presummed_f16x2(const float* xs_center, int64_t M, int C, int groups, double eps, double momentum, int ddof,
                              float* moving_mean, float* moving_cov, float* mu, double* L, double* W, void* ws, size_t ws_bytes,
                              wc_stream_t stream)
{
    if (!xs_center || !mu || !L || !W || !ws) return WC_ERR_NULL;
    if ((moving_mean == nullptr) != (moving_cov == nullptr)) return WC_ERR_NULL;
    if (M <= 0 || groups <= 0 || (M % groups) != 0 || M / groups <= ddof) return WC_ERR_SHAPE;
    if (bad_channels(C)) return WC_ERR_CHANNELS;
    if (!(eps > 0.0) || eps >= 1.0 || momentum < 0.0 || momentum > 1.0 || ddof < 0 || ddof > 1) return WC_ERR_ARG;
    const size_t need = presum_bytes(M, C, groups);
    if (need == 0) return WC_ERR_SHAPE;
    if (ws_bytes < need) return WC_ERR_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    PresumLayout l;
    if (!presum_layout(M, C, groups, ws, ws_bytes, &l)) return WC_ERR_SHAPE;
    // (the partials are those of xty_f16x3_kernel's accumulation scheme: its off-diagonal compensation applies)
    WC_TRY(wc_launch_stats_prepare(l.P, l.colsum, xs_center, l.nslab / groups, M / groups, C, groups, l.Sp, l.sum_scratch, l.dfix, nullptr, eps,
                                   momentum, ddof, moving_mean, moving_cov, mu, nullptr, L, st, l.tmp, wc_fast_xty_offdiag_bias()));
    if (wc_factor_is_fused(C)) {
        WC_TRY(wc_launch_factor_fused(L, W, l.tmp, C, groups, st));
        return WC_OK;
    }
    WC_TRY(wc_launch_cholesky(L, C, groups, st));
    WC_TRY(wc_launch_tri_inverse(L, W, l.tmp, C, groups, st));
    return WC_OK;
}

int wc_stats_presummed_f16x2(const float* xs_center, int64_t M, int C, int groups, double* sum, double* xtx, void* ws, size_t ws_bytes,
                             wc_stream_t stream)
{
    if (!xs_center || !sum || !xtx || !ws) return WC_ERR_NULL;
    if (M <= 0 || groups <= 0 || (M % groups) != 0) return WC_ERR_SHAPE;
    if (bad_channels(C)) return WC_ERR_CHANNELS;
    const size_t need = presum_bytes(M, C, groups);
    if (need == 0) return WC_ERR_SHAPE;
    if (ws_bytes < need) return WC_ERR_WORKSPACE;
    PresumLayout l;
    if (!presum_layout(M, C, groups, ws, ws_bytes, &l)) return WC_ERR_SHAPE;
    WC_TRY(wc_launch_stats_finalize(l.P, l.colsum, xs_center, l.nslab / groups, M / groups, C, groups, l.Sp, sum, xtx, l.dfix, nullptr,
                                    static_cast<hipStream_t>(stream), wc_fast_xty_offdiag_bias()));
    return WC_OK;
}

int wc_patch_sum_f32(const float* g, int64_t N, int64_t Hs, int64_t Ws, int C, float* out, wc_stream_t stream)
{
    if (!g || !out) return WC_ERR_NULL;
    if (N <= 0 || Hs <= 0 || Ws <= 0) return WC_ERR_SHAPE;
    if (bad_channels(C)) return WC_ERR_CHANNELS;
    WC_TRY(wc_launch_patch_sum(g, N, Hs, Ws, C, out, static_cast<hipStream_t>(stream)));
    return WC_OK;
}

int wc_fold_channel_scale_f32(const float* w, int64_t stride_o, int64_t stride_c, int Cout, int Cin, const float* bias,
                              const float* scale, const float* center, float* wf, float* bf, wc_stream_t stream)
{
    if (!w || !scale || !center || !wf || !bf) return WC_ERR_NULL;
    if (Cout <= 0 || Cin <= 0) return WC_ERR_SHAPE;
    WC_TRY(wc_launch_fold_channel_scale(w, stride_o, stride_c, Cout, Cin, bias, scale, center, wf, bf, static_cast<hipStream_t>(stream)));
    return WC_OK;
}

int wc_unfold_channel_scale_f32(const float* D, const float* db, int64_t stride_o, int64_t stride_c, int Cout, int Cin,
                                const float* scale, const float* center, float* dW, wc_stream_t stream)
{
    if (!D || !db || !scale || !center || !dW) return WC_ERR_NULL;
    if (Cout <= 0 || Cin <= 0) return WC_ERR_SHAPE;
    WC_TRY(wc_launch_unfold_channel_scale(D, db, stride_o, stride_c, Cout, Cin, scale, center, dW, static_cast<hipStream_t>(stream)));
    return WC_OK;
}

// ---------------------------------------------------------------------------------------------
size_t wc_bwd_reduce_workspace_bytes(int64_t N, int64_t HW, int C, int Kc, int has_slot)
{
    (void)Kc;
    if (N <= 0 || HW <= 0 || bad_channels(C)) return 0;
    const XtyPlan p = has_slot ? plan_xty(N, HW, C, 1, 0) : plan_xty(1, N * HW, C, 0, 0);
    return 256 + 2 * slot_bytes(C, 4) + slot_bytes((size_t)p.nslab * C, 4) + slot_bytes((size_t)p.nslab * C * C, 8);
}

int wc_bwd_reduce_f32(const float* x, const float* mu, const float* gy, const int32_t* slot,
                      int64_t N, int64_t HW, int C, int Kc, double* R, double* gsum,
                      void* ws, size_t ws_bytes, wc_stream_t stream)
{
    return wc_bwd_reduce_scaled_f32(x, mu, gy, slot, N, HW, C, Kc, R, gsum, nullptr, ws, ws_bytes, stream);
}

int wc_bwd_reduce_scaled_f32(const float* x, const float* mu, const float* gy, const int32_t* slot,
                             int64_t N, int64_t HW, int C, int Kc, double* R, double* gsum, float* scales_out,
                             void* ws, size_t ws_bytes, wc_stream_t stream)
{
    return wc_bwd_reduce_relu_f32(x, mu, gy, nullptr, slot, N, HW, C, Kc, R, gsum, nullptr, scales_out, ws, ws_bytes, stream);
}

static int bwd_reduce_masked(const float* x, const float* mu, const float* gy, const float* relu_y, const unsigned* relu_mask,
                             const int32_t* slot, int64_t N, int64_t HW, int C, int Kc, double* R, double* gsum, float* gy_masked,
                             float* scales_out, void* ws, size_t ws_bytes, wc_stream_t stream)
{
    const bool masked = relu_y || relu_mask;
    if (!x || !gy || !R || !gsum || !ws) return WC_ERR_NULL;
    if (N <= 0 || HW <= 0 || Kc <= 0 || (!slot && Kc != 1)) return WC_ERR_SHAPE;
    if (relu_mask && ((N * HW) % 32) != 0) return WC_ERR_SHAPE;
    if (bad_channels(C)) return WC_ERR_CHANNELS;
    if (ws_bytes < wc_bwd_reduce_workspace_bytes(N, HW, C, Kc, slot != nullptr)) return WC_ERR_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int per_sample = slot != nullptr;
    const int64_t Ns = per_sample ? N : 1, HWs = per_sample ? HW : N * HW;
    const XtyPlan p = plan_xty(Ns, HWs, C, per_sample, 0);
    Carver cv(ws, ws_bytes);
    int* gate = cv.take<int>(64);
    float* sx = cv.take<float>(C);
    float* sy = cv.take<float>(C);
    if (scales_out) { sx = scales_out; sy = scales_out + C; }      // the caller keeps them for wc_bwd_apply_scaled_f32
    float* colsum = cv.take<float>((size_t)p.nslab * C);
    double* P = cv.take<double>((size_t)p.nslab * C * C);
    WcXtyArgs a = {};
    a.X = x; a.Y = gy; a.cx = mu; a.cy = nullptr; a.N = Ns; a.HW = HWs;
    a.per_sample = per_sample; a.nsplit = p.nsplit; a.rows_per_slab = p.rps; a.C = C; a.sym = 0; a.P = P; a.colsum = colsum;
    // the ReLU mask: inside the staging of the quadrant kernel (C = 256 on the fast path), else one elementwise pass in front
    const bool mask_in_kernel = masked && p.fast && C == 256;
    // gy_masked == NULL with the bit mask: the masked gradient is not written at all (wc_bwd_reduce_bits_f32: K6 applies the bits
    // itself) -- only where the mask rides in the kernel's staging; the gated exact redo then masks while it loads
    const bool nowrite = relu_mask && !gy_masked;
    if (nowrite && !mask_in_kernel) return WC_ERR_SHAPE;
    if (masked && !gy_masked && !nowrite) return WC_ERR_NULL;
    if (masked && !mask_in_kernel) {
        if (relu_mask) WC_TRY(wc_launch_relu_mask_bits(gy, relu_mask, gy_masked, N * HW, C, st));
        else WC_TRY(wc_launch_relu_mask(gy, relu_y, gy_masked, N * HW * C, st));
        gy = gy_masked; a.Y = gy_masked;
    }
    if (!p.fast && scales_out) WC_TRY(wc_launch_channel_scale2(x, mu, sx, gy, nullptr, sy, N * HW, C, gate, st));   // asked for: sampled anyway
    if (p.fast) {
        // (masked in the kernel: the scales are sampled from the unmasked gy -- a superset of the masked values' range)
        WC_TRY(wc_launch_channel_scale2(x, mu, sx, gy, nullptr, sy, N * HW, C, gate, st));      // both scales, gate := 0
        WC_TRY(wc_launch_fast_xty(x, gy, mu, nullptr, sx, sy, Ns, HWs, C, per_sample, p.nsplit, p.rps, p.nslab, p.ntypes,
                                  P, colsum, nullptr, gate, st, mask_in_kernel ? relu_y : nullptr, mask_in_kernel ? gy_masked : nullptr,
                                  mask_in_kernel ? relu_mask : nullptr));
        a.gate = gate;
        if (mask_in_kernel && !nowrite) a.Y = gy_masked;          // the gated exact redo reads what the fast kernel wrote
        if (nowrite) a.ymask = relu_mask;                         // ... or masks gy itself
    }
    WC_TRY(wc_launch_xty(a, p.nslab, st));
    WC_TRY(wc_launch_bwd_combine(P, colsum, slot, N, p.nsplit, per_sample, C, Kc, R, gsum, st));
    return WC_OK;
}

int wc_bwd_reduce_relu_f32(const float* x, const float* mu, const float* gy, const float* relu_y, const int32_t* slot,
                           int64_t N, int64_t HW, int C, int Kc, double* R, double* gsum, float* gy_masked, float* scales_out,
                           void* ws, size_t ws_bytes, wc_stream_t stream)
{
    if ((relu_y != nullptr) != (gy_masked != nullptr)) return WC_ERR_NULL;
    return bwd_reduce_masked(x, mu, gy, relu_y, nullptr, slot, N, HW, C, Kc, R, gsum, gy_masked, scales_out, ws, ws_bytes, stream);
}

int wc_bwd_reduce_mask_f32(const float* x, const float* mu, const float* gy, const void* relu_mask, const int32_t* slot,
                           int64_t N, int64_t HW, int C, int Kc, double* R, double* gsum, float* gy_masked, float* scales_out,
                           void* ws, size_t ws_bytes, wc_stream_t stream)
{
    if (!relu_mask || !gy_masked) return WC_ERR_NULL;
    return bwd_reduce_masked(x, mu, gy, nullptr, static_cast<const unsigned*>(relu_mask), slot, N, HW, C, Kc, R, gsum, gy_masked,
                             scales_out, ws, ws_bytes, stream);
}

// K4 / K6 behind a ReLU'd site with the mask as bits and NO masked copy of the gradient in between (ABI 4): K4 applies the bits
// while it stages gy and writes nothing back, K6 applies them again while it converts gy.  Only where both kernels have the mask in
// their staging: C = 256 on the fast paths (wc_bwd_bits_supported); elsewhere wc_bwd_reduce_mask_f32 + wc_bwd_apply_scaled_f32.
int wc_bwd_bits_supported(int64_t N, int64_t HW, int C, int has_slot)
{
    if (N <= 0 || HW <= 0 || C != 256 || ((N * HW) % 32) != 0) return 0;
    const int per_sample = has_slot != 0;
    const XtyPlan p = plan_xty(per_sample ? N : 1, per_sample ? HW : N * HW, C, per_sample, 0);
    return (p.fast && wc_fast_affine_supported(N, HW, C, has_slot != 0) && wc_bwd_apply_onepass_supported(N, HW, C)) ? 1 : 0;
}

int wc_bwd_reduce_bits_f32(const float* x, const float* mu, const float* gy, const void* relu_mask, const int32_t* slot,
                           int64_t N, int64_t HW, int C, int Kc, double* R, double* gsum, float* scales_out,
                           void* ws, size_t ws_bytes, wc_stream_t stream)
{
    if (!relu_mask || !scales_out) return WC_ERR_NULL;
    if (!wc_bwd_bits_supported(N, HW, C, slot != nullptr)) return WC_ERR_SHAPE;
    return bwd_reduce_masked(x, mu, gy, nullptr, static_cast<const unsigned*>(relu_mask), slot, N, HW, C, Kc, R, gsum, nullptr,
                             scales_out, ws, ws_bytes, stream);
}

// ---- K4 / K6 on a pre-split x (ABI 5): the backward of a site whose input the residual add wrote as planes ---------------------
int wc_bwd_xsplit_supported(int64_t N, int64_t HW, int C, int has_slot)
{
    if (N <= 0 || HW <= 0 || (C != 256 && C != 128) || ((N * HW) % 32) != 0) return 0;
    const int per_sample = has_slot != 0;
    const XtyPlan p = plan_xty(per_sample ? N : 1, per_sample ? HW : N * HW, C, per_sample, 0);
    if (!p.fast || !wc_fast_affine_supported(N, HW, C, has_slot != 0)) return 0;
    // C = 256: K6 in one pass; C = 128 (round 5): the planes kernel for (x - mu) S - sub, then the accumulating fp32 kernel for + gy At
    return (C == 256 ? wc_bwd_apply_onepass_supported(N, HW, C) : wc_split_apply_supported(N, HW, C)) ? 1 : 0;
}

int wc_bwd_reduce_xsplit_f32(const void* xs, const float* xs_center, const float* xs_scale, const float* mu, const float* gy,
                             const void* relu_mask, const int32_t* slot, int64_t N, int64_t HW, int C, int Kc,
                             double* R, double* gsum, float* scales_out, void* ws, size_t ws_bytes, wc_stream_t stream)
{
    if (!xs || !xs_center || !xs_scale || !mu || !gy || !R || !gsum || !scales_out || !ws) return WC_ERR_NULL;
    if (N <= 0 || HW <= 0 || Kc <= 0 || (!slot && Kc != 1)) return WC_ERR_SHAPE;
    if (bad_channels(C)) return WC_ERR_CHANNELS;
    if (!wc_bwd_xsplit_supported(N, HW, C, slot != nullptr)) return WC_ERR_SHAPE;
    if (relu_mask && C != 256) return WC_ERR_SHAPE;          // (the masked form of the reduction exists for the quadrant scheme: mask gy in front at C = 128)
    if (ws_bytes < wc_bwd_reduce_workspace_bytes(N, HW, C, Kc, slot != nullptr)) return WC_ERR_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int per_sample = slot != nullptr;
    const int64_t Ns = per_sample ? N : 1, HWs = per_sample ? HW : N * HW;
    const XtyPlan p = plan_xty(Ns, HWs, C, per_sample, 0);
    Carver cv(ws, ws_bytes);
    int* gate = cv.take<int>(64);
    (void)cv.take<float>(C); (void)cv.take<float>(C);
    float* sy = scales_out + C;                              // the caller keeps it for wc_bwd_apply_xsplit_f32 ([0, C) is not used: x's scales are the planes')
    float* colsum = cv.take<float>((size_t)p.nslab * C);
    double* P = cv.take<double>((size_t)p.nslab * C * C);
    const unsigned* mask = static_cast<const unsigned*>(relu_mask);
    WC_TRY(wc_launch_channel_scale_gate(gy, nullptr, sy, N * HW, C, gate, st));       // gy's scales (sampled from the unmasked gradient), gate := 0
    WC_TRY(wc_launch_fast_xty(nullptr, gy, nullptr, nullptr, xs_scale, sy, Ns, HWs, C, per_sample, p.nsplit, p.rps, p.nslab, p.ntypes,
                              P, colsum, nullptr, gate, st, nullptr, nullptr, mask, xs));
    WcXtyArgs a = {};       // the gated exact redo (gy beyond the fp16 range): the same reduction in fp32 / float64, x from the planes
    a.X = nullptr; a.Y = gy; a.cx = nullptr; a.cy = nullptr; a.N = Ns; a.HW = HWs;
    a.per_sample = per_sample; a.nsplit = p.nsplit; a.rows_per_slab = p.rps; a.C = C; a.sym = 0; a.P = P; a.colsum = colsum;
    a.gate = gate; a.ymask = mask;
    a.Xhi = static_cast<const _Float16*>(xs); a.Xlo = a.Xhi + N * HW * C; a.xscale = xs_scale;
    WC_TRY(wc_launch_xty(a, p.nslab, st));
    WC_TRY(wc_launch_bwd_combine(P, colsum, slot, N, p.nsplit, per_sample, C, Kc, R, gsum, st));
    WC_TRY(wc_launch_rank1_add(R, gsum, xs_center, mu, C, Kc, st));      // (g / scale = x - center; f = x - mu)
    return WC_OK;
}

size_t wc_bwd_apply_xsplit_workspace_bytes(int C, int Kc)
{
    if (Kc <= 0 || bad_channels(C)) return 0;
    return wc_fast_affine_workspace(C, Kc) + wc_fast_affine_workspace(C, 1) + slot_bytes(C, 4);
}

int wc_bwd_apply_xsplit_f32(const float* gy, const void* relu_mask, const void* xs, const float* xs_center, const float* xs_scale,
                            const float* mu, const float* At, const float* S, const float* gmean, const int32_t* slot,
                            int64_t N, int64_t HW, int C, int Kc, const float* scales, float* dx,
                            void* ws, size_t ws_bytes, wc_stream_t stream)
{
    if (!gy || !xs || !xs_center || !xs_scale || !mu || !At || !S || !gmean || !scales || !dx || !ws) return WC_ERR_NULL;
    if (N <= 0 || HW <= 0 || Kc <= 0) return WC_ERR_SHAPE;
    if (bad_channels(C)) return WC_ERR_CHANNELS;
    if (!wc_bwd_xsplit_supported(N, HW, C, slot != nullptr)) return WC_ERR_SHAPE;
    if (ws_bytes < wc_bwd_apply_xsplit_workspace_bytes(C, Kc)) return WC_ERR_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    char* w = static_cast<char*>(ws);
    void* plan0 = w;
    void* plan1 = w + wc_fast_affine_workspace(C, Kc);
    float* subf = reinterpret_cast<float*>(w + wc_fast_affine_workspace(C, Kc) + wc_fast_affine_workspace(C, 1));
    // the tables of both halves in one launch: At for gy's scales, S for the planes' scales
    // ... and, in the same launch:  dx = gy At + (x - mu) S - gmean with x - mu = g / scale + (center - mu):  sub = gmean + (mu - center) S
    if (C != 256) {
        // C = 128 (round 5): dx = (g / scale) S - sub by the planes kernel (K3's: the additive term carries -sub = -gmean + (center - mu) S),
        // then dx += gy At by the accumulating fp32 kernel -- x is never needed in fp32, so the producer writes no copy of it
        if (relu_mask) return WC_ERR_SHAPE;
        WC_TRY(wc_launch_fast_plan_tables2_bias(At, Kc, plan0, scales + C, S, plan1, xs_scale, C, st, gmean, xs_center, mu, subf, 1));
        const float *pscale, *pcol; const void *phi, *plo;
        wc_fast_plan_parts(plan1, C, 1, &pscale, &pcol, &phi, &plo);
        WC_TRY(wc_launch_apply_split(xs, xs_scale, S, 1, subf, nullptr, N, HW, C, 0, dx, phi, plo, pcol, nullptr, st, nullptr, nullptr, nullptr));
        WC_TRY(wc_launch_fast_affine_planned(gy, nullptr, At, Kc, false, nullptr, nullptr, slot, N, HW, C, 1, dx, plan0, st, nullptr, nullptr, nullptr));
        return WC_OK;
    }
    WC_TRY(wc_launch_fast_plan_tables2_bias(At, Kc, plan0, scales + C, S, plan1, xs_scale, C, st, gmean, mu, xs_center, subf));
    WC_TRY(wc_launch_bwd_apply_onepass(gy, nullptr, mu, At, Kc, S, subf, slot, N, HW, scales, dx, plan0, plan1, st,
                                       static_cast<const unsigned*>(relu_mask), xs, xs_scale));
    return WC_OK;
}

// ---------------------------------------------------------------------------------------------
// Wbar = sum_k Gamma_k R_k^T is one workgroup grid walking all Kc terms; with many tables (per-sample tables: Kc = N) the sum
// is cut into WC_WBAR_PARTS batches that run side by side and are added in a fixed order afterwards
constexpr int WC_WBAR_PARTS = 16;
static int wbar_parts(int Kc) { return Kc >= 2 * WC_WBAR_PARTS ? WC_WBAR_PARTS : 1; }

size_t wc_bwd_factor_workspace_bytes(int C, int Kc)
{
    if (bad_channels(C)) return 0;
    return 3 * slot_bytes((size_t)C * C, 8) + (wbar_parts(Kc) > 1 ? slot_bytes((size_t)WC_WBAR_PARTS * C * C, 8) : 0);
}

int wc_bwd_factor_f64(const double* R, const double* gsum, const double* W, const double* L,
                      const float* gamma, const float* A, int Kc, int C, int64_t M, double eps, int ddof, int training,
                      float* dgamma, float* dbeta, float* S, float* gmean,
                      void* ws, size_t ws_bytes, wc_stream_t stream)
{
    if (!R || !gsum || !W || !ws) return WC_ERR_NULL;
    if (training && (!L || !A || !S || !gmean)) return WC_ERR_NULL;        // (L: part of the contract, not read since round 6 -- see the chain below)
    if (bad_channels(C)) return WC_ERR_CHANNELS;
    if (Kc <= 0 || (!gamma && Kc != 1) || (training && M <= ddof)) return WC_ERR_SHAPE;
    if (ws_bytes < wc_bwd_factor_workspace_bytes(C, Kc)) return WC_ERR_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t CC = (int64_t)C * C;
    Carver cv(ws, ws_bytes);
    double* buf0 = cv.take<double>(CC);
    double* buf1 = cv.take<double>(CC);
    double* buf2 = cv.take<double>(CC);

    auto sq = [&](const void* Am, int a32, int64_t ars, int64_t acs, const void* Bm, int b32, int64_t brs, int64_t bcs,
                  void* Cm, int c32, double alpha, int epi) {
        WcGemm g = {};
        g.A = Am; g.a_is_f32 = a32; g.a_rs = ars; g.a_cs = acs;
        g.B = Bm; g.b_is_f32 = b32; g.b_rs = brs; g.b_cs = bcs;
        g.Cm = Cm; g.c_is_f32 = c32; g.c_rs = C; g.c_cs = 1;
        g.m = C; g.n = C; g.k = C; g.batch = 1; g.nred = 1; g.alpha = alpha; g.epi = epi;
        return g;
    };

    WcGemm gd = {};
    const bool want_dgamma = dgamma && gamma;
    if (want_dgamma) {                              // dgamma[k] = W R[k]
        gd = sq(W, 0, C, 1, R, 0, C, 1, dgamma, 1, 1.0, WC_EPI_NONE);
        gd.a_bs = 0; gd.b_bs = CC; gd.c_bs = CC; gd.batch = Kc;
        if (!training) WC_TRY(wc_launch_gemm(gd, st));          // training: in one launch with Wbar below
    }
    if (dbeta && !training) WC_TRY(wc_launch_f64_to_f32(gsum, dbeta, (int64_t)Kc * C, st));      // training: in the tail launch
    if (!training) return WC_OK;

    const double* Wbar; int64_t wb_rs, wb_cs;
    if (gamma) {                                    // Wbar = sum_k Gamma_k R_k^T
        WcGemm g = sq(gamma, 1, C, 1, R, 0, 1, C, buf0, 0, 1.0, WC_EPI_NONE);
        g.a_red = CC; g.b_red = CC; g.nred = Kc;
        const int parts = wbar_parts(Kc);
        double* partial = nullptr;
        if (parts > 1) {                            // many tables: `parts` partial sums side by side, added below
            partial = cv.take<double>((size_t)parts * CC);
            g.nred = (Kc + parts - 1) / parts; g.red_total = Kc; g.batch = parts;
            g.a_bs = (int64_t)g.nred * CC; g.b_bs = (int64_t)g.nred * CC; g.Cm = partial; g.c_bs = CC;
        }
        if (want_dgamma) WC_TRY(wc_launch_gemm_pair_dd_fd(gd, g, st));      // dgamma and Wbar: neither waits for the other
        else WC_TRY(wc_launch_gemm(g, st));
        if (parts > 1) WC_TRY(wc_launch_sum_partials(partial, parts, CC, buf0, st));
        Wbar = buf0; wb_rs = C; wb_cs = 1;
    } else {                                        // Gamma = I: Wbar = R^T, read through swapped strides
        Wbar = R; wb_rs = 1; wb_cs = C;
    }
    // The Cholesky step of the chain in ONE product (round 6).  The textbook form -- Lbar = -tril(W^T Wbar W^T), P = Phi(L^T Lbar): three dependent
    // products -- collapses: L^T is upper triangular, so the strictly upper part of Y = W^T Wbar W^T contributes nothing on or below the diagonal of
    // L^T Y, i.e. Phi(L^T tril(Y)) = Phi(L^T Y), and L^T W^T = (W L)^T = I: P = -Phi(Wbar W^T).  Same quantity (to the 6e-13 of K2's W L = I), two
    // launches of ~8 us fewer on the backward's critical path of every site; L is not read any more.
    {   // P = -Phi(Wbar W^T)
        WcGemm g = sq(Wbar, 0, wb_rs, wb_cs, W, 0, 1, C, buf1, 0, -1.0, WC_EPI_PHI);
        WC_TRY(wc_launch_gemm(g, st));
    }
    {   // Q1 = W^T P
        WcGemm g = sq(W, 0, 1, C, buf1, 0, C, 1, buf2, 0, 1.0, WC_EPI_NONE);
        WC_TRY(wc_launch_gemm(g, st));
    }
    {   // Q2 = Q1 W
        WcGemm g = sq(buf2, 0, C, 1, W, 0, C, 1, buf0, 0, 1.0, WC_EPI_NONE);
        WC_TRY(wc_launch_gemm(g, st));
    }
    const double scale = 2.0 * (1.0 - eps) / (double)(M - ddof);
    WC_TRY(wc_launch_bwd_tail(buf0, C, scale, S, gsum, A, Kc, M, gmean, dbeta, st));      // S, gmean, dbeta
    return WC_OK;
}

// ---------------------------------------------------------------------------------------------
size_t wc_bwd_apply_workspace_bytes(int64_t N, int64_t HW, int C, int Kc)
{
    // two plans side by side (At's tables, S's table): with given scales both are built in one launch
    const size_t one = wc_apply_workspace_bytes(N, HW, C, Kc);
    if (N <= 0 || HW <= 0 || Kc <= 0 || bad_channels(C) || !wc_fast_affine_supported(N, HW, C, Kc > 1)) return one;
    return one + wc_fast_affine_workspace(C, 1);
}

int wc_bwd_apply_f32(const float* gy, const float* x, const float* mu, const float* At, const float* S,
                     const float* gmean, const int32_t* slot, int64_t N, int64_t HW, int C, int Kc,
                     float* dx, void* ws, size_t ws_bytes, wc_stream_t stream)
{
    return wc_bwd_apply_scaled_f32(gy, x, mu, At, S, gmean, slot, N, HW, C, Kc, nullptr, dx, ws, ws_bytes, stream);
}

static int bwd_apply_impl(const float* gy, const float* x, const float* mu, const float* At, const float* S,
                          const float* gmean, const int32_t* slot, int64_t N, int64_t HW, int C, int Kc,
                          const float* scales, float* dx, void* ws, size_t ws_bytes, wc_stream_t stream, const unsigned* relu_mask);

int wc_bwd_apply_scaled_f32(const float* gy, const float* x, const float* mu, const float* At, const float* S,
                            const float* gmean, const int32_t* slot, int64_t N, int64_t HW, int C, int Kc,
                            const float* scales, float* dx, void* ws, size_t ws_bytes, wc_stream_t stream)
{
    return bwd_apply_impl(gy, x, mu, At, S, gmean, slot, N, HW, C, Kc, scales, dx, ws, ws_bytes, stream, nullptr);
}

int wc_bwd_apply_bits_f32(const float* gy, const void* relu_mask, const float* x, const float* mu, const float* At, const float* S,
                          const float* gmean, const int32_t* slot, int64_t N, int64_t HW, int C, int Kc,
                          const float* scales, float* dx, void* ws, size_t ws_bytes, wc_stream_t stream)
{
    if (!relu_mask || !scales || !S || !mu || !x) return WC_ERR_NULL;
    if (!wc_bwd_bits_supported(N, HW, C, slot != nullptr)) return WC_ERR_SHAPE;
    if (!ws || ws_bytes < wc_fast_affine_workspace(C, Kc) + wc_fast_affine_workspace(C, 1)) return WC_ERR_WORKSPACE;
    return bwd_apply_impl(gy, x, mu, At, S, gmean, slot, N, HW, C, Kc, scales, dx, ws, ws_bytes, stream, static_cast<const unsigned*>(relu_mask));
}

static int bwd_apply_impl(const float* gy, const float* x, const float* mu, const float* At, const float* S,
                          const float* gmean, const int32_t* slot, int64_t N, int64_t HW, int C, int Kc,
                          const float* scales, float* dx, void* ws, size_t ws_bytes, wc_stream_t stream, const unsigned* relu_mask)
{
    if (!gy || !At || !dx) return WC_ERR_NULL;
    if (S && !x) return WC_ERR_NULL;
    if (N <= 0 || HW <= 0 || Kc <= 0) return WC_ERR_SHAPE;
    if (bad_channels(C)) return WC_ERR_CHANNELS;
    hipStream_t st = static_cast<hipStream_t>(stream);
    WcRowsGemmArgs a = {};
    a.in[0] = gy; a.center[0] = nullptr; a.B[0] = At; a.B_slot_stride[0] = (int64_t)C * C;
    a.nstreams = 1;
    if (S) {
        a.in[1] = x; a.center[1] = mu; a.B[1] = S; a.B_slot_stride[1] = 0;
        a.nstreams = 2;
    }
    a.bias = nullptr; a.sub = gmean; a.slot = slot; a.N = N; a.HW = HW; a.C = C; a.out = dx;
    const bool fast = ws && wc_fast_affine_supported(N, HW, C, slot != nullptr) && ws_bytes >= wc_fast_affine_workspace(C, Kc);
    if (fast && scales && S && ws_bytes >= wc_fast_affine_workspace(C, Kc) + wc_fast_affine_workspace(C, 1)) {
        // the scales of both inputs are the caller's (K4 sampled the same two tensors): no sampling launches, and the tables
        // of both passes from one launch -- three launches instead of six
        void* plan0 = ws;
        void* plan1 = static_cast<char*>(ws) + wc_fast_affine_workspace(C, Kc);
        WC_TRY(wc_launch_fast_plan_tables2(At, Kc, plan0, scales + C, S, 1, plan1, scales, C, st));
        if (mu && wc_bwd_apply_onepass_supported(N, HW, C)) {       // C = 256: one pass over K = 512 (wc_fast.hip)
            WC_TRY(wc_launch_bwd_apply_onepass(gy, x, mu, At, Kc, S, gmean, slot, N, HW, scales, dx, plan0, plan1, st, relu_mask));
            return WC_OK;
        }
        if (relu_mask) return WC_ERR_SHAPE;
        WC_TRY(wc_launch_fast_affine_planned(gy, nullptr, At, Kc, false, nullptr, gmean, slot, N, HW, C, 0, dx, plan0, st));
        WC_TRY(wc_launch_fast_affine_planned(x, mu, S, 1, true, nullptr, nullptr, nullptr, N, HW, C, 1, dx, plan1, st));
        return WC_OK;
    }
    if (relu_mask) return WC_ERR_SHAPE;          // (only the one-pass kernel applies the bits)
    if (fast) {
        // two passes over dx (the B' fragments of both streams do not fit one wave's registers at C = 256):
        //   dx  = gy At[slot] - gmean ;   dx += (x - mu) S
        WC_TRY(wc_launch_fast_affine(gy, nullptr, At, Kc, false, nullptr, gmean, slot, N, HW, C, 0, dx, ws, st));
        if (S) WC_TRY(wc_launch_fast_affine(x, mu, S, 1, true, nullptr, nullptr, nullptr, N, HW, C, 1, dx, ws, st));
        return WC_OK;
    }
    WC_TRY(wc_launch_rows_gemm(a, st));
    return WC_OK;
}

int wc_stream_copy_f32(const float* src, float* dst, int64_t n, wc_stream_t stream)
{
    if (!src || !dst) return WC_ERR_NULL;
    if (n <= 0 || (n % 4) != 0) return WC_ERR_SHAPE;
    WC_TRY(wc_launch_stream_copy(src, dst, n, static_cast<hipStream_t>(stream)));
    return WC_OK;
}

size_t wc_spectral_norm_workspace_bytes(int rows, int cols)
{
    if (rows <= 0 || cols <= 0) return 0;
    return wc_sn_workspace_bytes(rows, cols);
}

size_t wc_spectral_norm_amax_offset(int rows, int cols)
{
    if (rows <= 0 || cols <= 0) return 0;
    return wc_sn_amax_offset(rows, cols);
}

size_t wc_spectral_norm_error_offset(int rows, int cols)
{
    if (rows <= 0 || cols <= 0) return 0;
    return wc_sn_error_offset(rows, cols);
}


int wc_spectral_norm_f32(const float* W, int rows, int cols, float* u, float* v, int iterations, float eps,
                         float* w_sn, float* sigma, float* u_used, float* v_used,
                         void* ws, size_t ws_bytes, wc_stream_t stream)
{
    if (!W || !u || !v || !w_sn || !sigma || !ws) return WC_ERR_NULL;
    if (rows <= 0 || cols <= 0 || (int64_t)rows * cols > (int64_t)1 << 28) return WC_ERR_SHAPE;
    if (iterations < 0 || !(eps >= 0.f)) return WC_ERR_ARG;
    if (wc_sn_lds_bytes(rows, cols) > 150 * 1024) return WC_ERR_SHAPE;        // u and v live in LDS
    if (ws_bytes < wc_sn_workspace_bytes(rows, cols)) return WC_ERR_WORKSPACE;
    WC_TRY(wc_launch_spectral_norm(W, rows, cols, u, v, iterations, eps, w_sn, sigma, u_used, v_used, ws, static_cast<hipStream_t>(stream)));
    return WC_OK;
}

int wc_spectral_norm_bwd_f32(const float* g, const float* w_sn, const float* u, const float* v, const float* sigma,
                             int rows, int cols, int fully_diff, float* dW, void* ws, size_t ws_bytes, wc_stream_t stream)
{
    if (!g || !w_sn || !u || !v || !sigma || !dW || !ws) return WC_ERR_NULL;
    if (rows <= 0 || cols <= 0 || (int64_t)rows * cols > (int64_t)1 << 28) return WC_ERR_SHAPE;
    if (ws_bytes < wc_sn_workspace_bytes(rows, cols)) return WC_ERR_WORKSPACE;
    WC_TRY(wc_launch_spectral_norm_bwd(g, w_sn, u, v, sigma, rows, cols, fully_diff, dW, ws, static_cast<hipStream_t>(stream)));
    return WC_OK;
}

int wc_spectral_norm_batched_f32(const wc_sn_item* items, int count, int iterations, float eps, wc_stream_t stream)
{
    if (!items) return WC_ERR_NULL;
    if (count <= 0) return WC_ERR_SHAPE;
    if (iterations < 0 || !(eps >= 0.f)) return WC_ERR_ARG;
    for (int i = 0; i < count; ++i) {
        const wc_sn_item& it = items[i];
        if (!it.W || !it.u || !it.v || !it.w_sn || !it.sigma || !it.ws) return WC_ERR_NULL;
        if (it.rows <= 0 || it.cols <= 0 || (int64_t)it.rows * it.cols > (int64_t)1 << 28) return WC_ERR_SHAPE;
        if (wc_sn_lds_bytes(it.rows, it.cols) > 150 * 1024) return WC_ERR_SHAPE;
    }
    WC_TRY(wc_launch_spectral_norm_batched(items, count, iterations, eps, static_cast<hipStream_t>(stream)));
    return WC_OK;
}

int wc_spectral_norm_bwd_batched_f32(const wc_sn_bwd_item* items, int count, int fully_diff, wc_stream_t stream)
{
    if (!items) return WC_ERR_NULL;
    if (count <= 0) return WC_ERR_SHAPE;
    for (int i = 0; i < count; ++i) {
        const wc_sn_bwd_item& it = items[i];
        if (!it.g || !it.w_sn || !it.u || !it.v || !it.sigma || !it.dW || !it.ws) return WC_ERR_NULL;
        if (it.rows <= 0 || it.cols <= 0 || (int64_t)it.rows * it.cols > (int64_t)1 << 28) return WC_ERR_SHAPE;
    }
    WC_TRY(wc_launch_spectral_norm_bwd_batched(items, count, fully_diff, static_cast<hipStream_t>(stream)));
    return WC_OK;
}

}  // extern "C"
