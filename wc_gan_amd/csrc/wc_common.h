// Shared declarations for the gfx950 WC kernels (internal; the public ABI is include/wc_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

typedef float  f32x4  __attribute__((ext_vector_type(4)));

// Row-subsample statistics that survive an outlier ON a sampled row (the fp16 paths take their per-channel scales and
// the covariance's shift from <= 256 sampled rows in 16 groups; one 1e7-sigma element there used to set the channel's scale
// by itself -- every ordinary value of the channel then sat in fp16's subnormal range, silently: only overflow raises
// the exact path).  The median of the 16 group values is the reference: untouched by up to seven bad groups.
__device__ __forceinline__ float wc_median16(const float (&v)[16])
{
    float s[16];          // bitonic sorting network: 80 compare-exchanges, every index a compile-time constant
#pragma unroll
    for (int i = 0; i < 16; ++i) s[i] = v[i];
#pragma unroll
    for (int k = 2; k <= 16; k <<= 1)
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int l = i ^ j;
                if (l > i) {
                    const float lo = fminf(s[i], s[l]), hi = fmaxf(s[i], s[l]);
                    const bool up = (i & k) == 0;
                    s[i] = up ? lo : hi; s[l] = up ? hi : lo;
                }
            }
    return 0.5f * (s[7] + s[8]);
}
// the sampled maximum, unless it is more than 64 x the median of the 16 group maxima: then 4 x that median
// (ordinary data: unchanged, bit for bit; a group maximum of 16 normal rows is ~1.9 sigma, the maximum of 256 ~2.8)
__device__ __forceinline__ float wc_robust_max16(const float (&gmax)[16])
{
    float m = 0.f;
#pragma unroll
    for (int p = 0; p < 16; ++p) m = fmaxf(m, gmax[p]);
    const float med = wc_median16(gmax);
    return (med > 0.f && m > 64.f * med) ? 4.f * med : m;
}

// Row of sample r of the <= 256-row subsample every fp16 path takes its centre / scales from (nsamp = min(M, 256), stride = M / nsamp).
// Rounds 1-4 sampled rows r * stride: for 128x32x32 that stride is 512 and for 128x16x16 it is 128, both multiples of W, so EVERY
// sampled pixel sat in the image's x = 0 border column, whose values come out of zero-padded 3x3 convolutions -- a border
// maximum can understate a channel's interior maximum.  Now sample r lies in [r stride, (r + 1) stride) at a hashed offset: still one
// row per stride-long run of rows (every sample of the batch is visited), spread over all (y, x).  One definition for every sampler
// (subsample_mean[_scale]_kernel, channel_scale_kernel, resadd_sample_kernel): the planes producer must pick the rows K1's own
// subsample would (bit-identical centre and scales, tests/test_producer_gpu.py).
__host__ __device__ __forceinline__ int64_t wc_sample_row(int64_t r, int64_t stride)
{
    return r * stride + (int64_t)((((uint32_t)r * 0x9E3779B1u) >> 8) % (uint32_t)stride);
}

typedef float  f32x16 __attribute__((ext_vector_type(16)));
typedef double f64x4  __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

#define WC_WAVE 64

static inline size_t wc_align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ----- big-tensor kernels (wc_rows.hip) ------------------------------------------------------

// out[m,:] = sum_s (in_s[m,:] - center_s) * B_s[slot(m)] + bias[slot(m)] - sub        (K3: 1 stream, K6: 2)
struct WcRowsGemmArgs {
    const float* in[2];
    const float* center[2];     // [C] or nullptr
    const float* B[2];          // [slots][C][C] row-major (k, n)
    int64_t      B_slot_stride[2];  // elements between slots (0 = shared)
    const float* bias;          // [Kc][C] or nullptr
    const float* sub;           // [C] or nullptr
    const int32_t* slot;        // [N] or nullptr
    int64_t N, HW;              // rows = N*HW; a sample is HW consecutive rows
    int C;
    int nstreams;
    float* out;
    const int* gate;            // optional device flag: when non-null and *gate == 0 the kernel does nothing
    int relu;                   // epilogue: out = max(out, 0)
};
hipError_t wc_launch_rows_gemm(const WcRowsGemmArgs& a, hipStream_t st);

// P[z] = sum_{m in slab z} (X[m]-cx)^T (Y[m]-cy)   float64 partials; sums of X (sym) or Y (non-sym) columns
struct WcXtyArgs {
    const float* X; const float* Y;      // Y == X for the symmetric (covariance) case
    const float* cx; const float* cy;    // centers [C] or nullptr
    int64_t N, HW;                       // slabs never cross a sample when per_sample != 0
    int per_sample;                      // 0: slabs tile the whole M = N*HW rows; 1: nsplit slabs per sample
    int nsplit;                          // per_sample: slabs per sample; else total slabs
    int64_t rows_per_slab;
    int C;
    int sym;                             // 1: only tiles jb >= ib, column sums of X; 0: all tiles, column sums of Y
    double* P;                           // [nslab][C][C] float64 partials
    float* colsum;                       // [nslab][C]
    const int* gate;                     // optional: run only when *gate != 0 (exact redo of a fast-path call)
    const unsigned* ymask;               // optional: Y is a gradient in front of a ReLU whose one-bit mask this is (wc_apply_mask_f32 layout): applied while Y is loaded
    const _Float16* Xhi; const _Float16* Xlo; const float* xscale;      // optional: X as pre-split planes, X := (hi + lo) / xscale (cx ignored)
};
int  wc_xty_plan(int64_t N, int64_t HW, int C, int per_sample, int sym, int* nsplit, int64_t* rows_per_slab);  // returns nslab
hipError_t wc_launch_xty(const WcXtyArgs& a, int nslab, hipStream_t st);

hipError_t wc_launch_subsample_mean(const float* x, int64_t M, int C, float* shift, hipStream_t st);
hipError_t wc_launch_stream_copy(const float* src, float* dst, int64_t n, hipStream_t st);
hipError_t wc_launch_relu_mask(const float* gy, const float* y, float* out, int64_t n, hipStream_t st);      // out = gy where y > 0, else 0
hipError_t wc_launch_subsample_mean_scale(const float* x, int64_t M, int C, float* shift, float* scale, int* gate, hipStream_t st);
// wc_sn.hip
#include "../../include/wc_hip.h"
typedef wc_sn_item WcSnItem;
typedef wc_sn_bwd_item WcSnBwdItem;
hipError_t wc_launch_spectral_norm_batched(const WcSnItem* items, int count, int iterations, float eps, hipStream_t st);
hipError_t wc_launch_spectral_norm_bwd_batched(const WcSnBwdItem* items, int count, int fully_diff, hipStream_t st);
size_t wc_sn_lds_bytes(int R, int K);
size_t wc_sn_workspace_bytes(int R, int K);
size_t wc_sn_amax_offset(int R, int K);
size_t wc_sn_error_offset(int R, int K);
hipError_t wc_launch_spectral_norm(const float* W, int R, int K, float* u, float* v, int iterations, float eps,
                                   float* w_sn, float* sigma, float* u_used, float* v_used, void* ws, hipStream_t st);
hipError_t wc_launch_spectral_norm_bwd(const float* g, const float* w_sn, const float* u, const float* v, const float* sigma,
                                       int R, int K, int fully_diff, float* dW, void* ws, hipStream_t st);

// ----- fast split-fp16 affine (wc_fast.hip) ---------------------------------------------------
constexpr int64_t WC_FAST_MIN_ROWS = 20480;     // reductions (K1/K4): below this the exact float64-MFMA kernel runs (development: WC_XTY_MIN_ROWS)
// apply (K3/K6): below this the f32-MFMA kernel runs.  1024 since round 2: with the plan built by wc_color_f32 the split-fp16
// apply is ONE launch whatever the size -- 8 against 35-43 us at the 4x4 / 8x8 sites of the generator (M = 2048..8192)
constexpr int64_t WC_AFFINE_MIN_ROWS = 1024;
int64_t wc_fast_xty_min_rows();
int64_t wc_fast_affine_min_rows();
bool   wc_fast_affine_supported(int64_t N, int64_t HW, int C, bool has_slot);
size_t wc_fast_affine_workspace(int C, int Kc);
hipError_t wc_launch_fast_affine(const float* in, const float* center, const float* B, int Kc, bool shared_table,
                                 const float* bias, const float* sub, const int32_t* slot,
                                 int64_t N, int64_t HW, int C, int accumulate, float* out,
                                 void* ws, hipStream_t st);

// fast reductions (wc_fast_xty.hip)
int wc_fast_xty_plan(int64_t N, int64_t HW, int C, int per_sample, int two, int* nsplit, int64_t* rows_per_slab, int* ntypes);
hipError_t wc_launch_fast_xty(const float* X, const float* Y, const float* cx, const float* cy,
                              const float* sx, const float* sy, int64_t N, int64_t HW, int C,
                              int per_sample, int nsplit, int64_t rows_per_slab, int nslab, int ntypes,
                              double* P, float* colsum, double* dfix /*[nslab][C], covariance only, nullable*/, int* gate, hipStream_t st,
                              const float* yrelu = nullptr, float* yout = nullptr,       // yrelu (C = 256, two operands): Y masked by yrelu > 0, written to yout
                              const unsigned* ymask = nullptr,                           // ... or by K3's bit mask ([M/32][C] words)
                              const void* xs = nullptr);                                 // X as pre-split planes (C = 256, two operands; X may be NULL, sx = the planes' scales, cx NULL)
hipError_t wc_launch_fast_plan_tables(const float* B, int Kc, int C, void* plan, hipStream_t st, const float* scale = nullptr);
bool wc_bwd_apply_onepass_supported(int64_t N, int64_t HW, int C);
hipError_t wc_launch_bwd_apply_onepass(const float* gy, const float* x, const float* mu, const float* At, int Kc, const float* S,
                                       const float* gmean, const int32_t* slot, int64_t N, int64_t HW, const float* scales /*[2C]: x | gy*/,
                                       float* dx, const void* plan0, const void* plan1, hipStream_t st,
                                       const unsigned* relu_mask = nullptr /*gy is the gradient before the site's ReLU: masked while it is converted*/,
                                       const void* xs = nullptr, const float* xs_scale = nullptr /*x as pre-split planes (x may then be NULL); plan1 built
                                                                                                   for xs_scale, gmean folded: gmean - (center - mu) S*/);
hipError_t wc_launch_fast_plan_tables2(const float* B0, int Kc0, void* plan0, const float* scale0,
                                       const float* B1, int Kc1, void* plan1, const float* scale1, int C, hipStream_t st);
hipError_t wc_launch_fast_plan_tables_bias(const float* B, int Kc, int C, void* plan, hipStream_t st, const float* scale,
                                           const float* bias, const float* center, const float* mu, float* bias_out);
hipError_t wc_launch_fast_plan_tables2_bias(const float* B0, int Kc0, void* plan0, const float* scale0,
                                            const float* B1, void* plan1, const float* scale1, int C, hipStream_t st,
                                            const float* bias, const float* center, const float* mu, float* bias_out, int neg_bias = 0);
float* wc_fast_plan_scale(void* plan);
hipError_t wc_launch_fast_affine_planned(const float* in, const float* center, const float* B, int Kc, bool shared_table,
                                         const float* bias, const float* sub, const int32_t* slot,
                                         int64_t N, int64_t HW, int C, int accumulate, float* out,
                                         const void* plan, hipStream_t st, unsigned* relu_mask = nullptr,      // relu_mask: see wc_fast_affine_writes_mask
                                         void* planes = nullptr, float* oscale = nullptr);                    // planes: see wc_fast_affine_writes_planes
bool wc_fast_affine_writes_planes(int64_t N, int64_t HW, int C);     // the planned ring kernel can leave its output as the next convolution's fp16 planes
hipError_t wc_launch_out_scale(const float* gamma, const float* beta, int K, int C, float* oscale, hipStream_t st);   // predicted output scale -> oscale[1]
bool wc_fast_affine_writes_mask(int64_t N, int64_t HW, int C);      // the planned ring kernel leaves the ReLU's bit mask itself (else: wc_launch_mask_from_y)
hipError_t wc_launch_mask_from_y(const float* y, int64_t M, int C, unsigned* mask, hipStream_t st);
hipError_t wc_launch_relu_mask_bits(const float* gy, const unsigned* mask, float* out, int64_t M, int C, hipStream_t st);
hipError_t wc_launch_channel_scale(const float* in, const float* center, int64_t M, int C, float* scale, hipStream_t st);
hipError_t wc_launch_channel_scale2(const float* in, const float* center, float* scale, const float* in2, const float* center2,
                                    float* scale2, int64_t M, int C, int* gate, hipStream_t st);
hipError_t wc_launch_channel_scale_gate(const float* in, const float* center, float* scale, int64_t M, int C, int* gate, hipStream_t st);   // one operand + the gate's clearing
// wc_mix.hip: the soft-assignment coloring's dictionary mix and its gradients
bool wc_mix_supported(int E, int C);
hipError_t wc_launch_mix_fwd(const float* dict, const float* alpha, const int32_t* idx, const float* base, int E, int C, int Kc, float* out,
                             hipStream_t st);
size_t wc_mix_bwd_workspace(int E, int Kc);
hipError_t wc_launch_mix_bwd(const float* dict, const float* alpha, const int32_t* idx, const float* dout, int E, int C, int K, int Kc,
                             float* ddict, float* dalpha, float* dbase, void* ws, hipStream_t st);
hipError_t wc_launch_rank1_add(double* R, const double* gsum, const float* u, const float* v, int C, int Kc, hipStream_t st);   // R[k][i][j] += (u[i] - v[i]) gsum[k][j]  (wc_small.hip)

void wc_fast_plan_parts(const void* plan, int C, int Kc, const float** scale, const float** colscale, const void** hi, const void** lo);

// ----- pre-split activations (wc_split.hip) ---------------------------------------------------
bool wc_split_apply_supported(int64_t N, int64_t HW, int C);
hipError_t wc_launch_split_rows(const float* x, const float* center, const float* scale, int64_t M, int C, int relu,
                                void* xs, int* flag, hipStream_t st);
hipError_t wc_launch_unsplit_rows(const void* xs, const float* center, const float* scale, int64_t M, int C, float* x, hipStream_t st);
hipError_t wc_launch_split_bias(const float* A, const float* bias, const float* center, const float* mu, int Kc, int C,
                                float* out, hipStream_t st);
hipError_t wc_launch_apply_split(const void* xs, const float* xs_scale, const float* A, int Kc, const float* bias2,
                                 const int32_t* slot, int64_t N, int64_t HW, int C, int relu, float* y,
                                 const void* plan_hi, const void* plan_lo, const float* plan_colscale, void* dbg, hipStream_t st,
                                 unsigned* relu_mask = nullptr /*the ReLU's bit mask (relu != 0)*/,
                                 void* planes = nullptr, float* oscale = nullptr /*the output as the next convolution's planes*/);

// ----- the residual add as the producer of pre-split activations (wc_resadd.hip) --------------
hipError_t wc_launch_resadd(const float* h, const float* s, int64_t N, int64_t H, int64_t W, int C, int up,
                            void* xs /*nullable: planes out*/, float* center, float* scale, int* flag, float* x32 /*nullable: fp32 out*/,
                            hipStream_t st);
// the same pass with the next site's covariance partials accumulated in it (resadd_xtx_kernel; slab plan: wc_fast_xty_plan, two = 0)
bool wc_resadd_xtx_supported(int64_t N, int64_t H, int64_t W, int C, int up, int groups);
hipError_t wc_launch_resadd_xtx(const float* h, const float* s, int64_t N, int64_t H, int64_t W, int C, int up, int groups,
                                void* xs, float* center, float* scale, int* flag, float* x32,
                                int nsplit, int64_t rows_per_slab, int nslab, int ntypes, double* P, float* colsum, double* dfix,
                                int* wgflag /*[grid]*/, float* wgmax /*[grid][C]*/, hipStream_t st);
int wc_resadd_xtx_grid(int nslab, int ntypes);
hipError_t wc_launch_patch_sum(const float* g, int64_t N, int64_t Hs, int64_t Ws, int C, float* out, hipStream_t st);
hipError_t wc_launch_fold_channel_scale(const float* w, int64_t so, int64_t sc, int Cout, int Cin, const float* bias,
                                        const float* scale, const float* center, float* wf, float* bf, hipStream_t st);
hipError_t wc_launch_unfold_channel_scale(const float* D, const float* db, int64_t so, int64_t sc, int Cout, int Cin,
                                          const float* scale, const float* center, float* dW, hipStream_t st);

int wc_split_xtx_plan(int64_t N, int64_t HW, int C, int per_sample, int* nsplit, int64_t* rows_per_slab, int* ntypes);      // wc_split_xty.hip
hipError_t wc_launch_split_xtx(const void* xs, const float* scale, int64_t N, int64_t HW, int C, int per_sample, int nsplit,
                               int64_t rows_per_slab, int nslab, int ntypes, double* P, float* colsum, double* dfix, hipStream_t st);

// ----- small-matrix stage (wc_small.hip) -----------------------------------------------------

// K1 tail: shifted fp32 partials -> raw float64 moments
hipError_t wc_launch_stats_finalize(const double* P, const float* colsum, const float* shift, int nslab,
                                    int64_t M, int C, int groups, double* Sp /*[groups*C] scratch*/, double* sum, double* xtx,
                                    const double* dfix /*[groups*nslab][C] the fast path's VALU diagonal, nullable*/,
                                    const int* gate /*dfix is void when *gate != 0 (the exact redo ran)*/, hipStream_t st,
                                    double kappa = 0.0 /*off-diagonal bias compensation of the kernel that wrote P (wc_fast_xty_offdiag_bias); Sp then holds 2*groups*C doubles*/);
double wc_fast_xty_offdiag_bias(void);       // kappa of xty_f16x3_kernel's covariance partials (wc_fast_xty.hip)
// K4 tail: per-slab partials -> per-slot float64 R, gsum
hipError_t wc_launch_bwd_combine(const double* P, const float* colsum, const int32_t* slot, int64_t N, int nsplit,
                                 int per_sample, int C, int Kc, double* R, double* gsum, hipStream_t st);

hipError_t wc_launch_stats_prepare(const double* P, const float* colsum, const float* shift, int nslab, int64_t M, int C, int groups,
                                   double* Sp, double* sum_scratch, const double* dfix, const int* gate, double eps, double momentum,
                                   int ddof, float* moving_mean, float* moving_cov, float* mu, float* chan_scale, double* T,
                                   hipStream_t st, double* tmp, double kappa = 0.0);      // K1 tail + K2 head in two launches (wc_whiten_f32)
hipError_t wc_launch_factor_prepare(const double* sum, const double* xtx, int64_t M, int C, double eps, double momentum,
                                    int ddof, int training, int groups, float* moving_mean, float* moving_cov, float* mu,
                                    float* chan_scale, double* T, hipStream_t st, double* tmp = nullptr);
// C <= 256: L (in place on T, upper zeroed) and W = L^-1 in one launch (two for many groups); tmp: [groups][C*16] doubles + 64 B
// per group at least, the SAME tmp that wc_launch_factor_prepare was given (it zeroes the launch's row-block counters there)
bool wc_factor_is_fused(int C);
hipError_t wc_launch_factor_fused(double* T, double* W, double* tmp, int C, int groups, hipStream_t st);
hipError_t wc_launch_cholesky(double* T, int C, int groups, hipStream_t st);                     // in place: lower factor, upper zeroed
hipError_t wc_launch_tri_inverse(const double* L, double* W, double* tmp, int C, int groups, hipStream_t st);   // W = L^-1 (lower), upper zeroed

// generic small batched GEMM in float64 on the f64 MFMA:  Cm[b] = alpha * sum_r opA[b,r] opB[b,r]  (+ epilogue)
enum WcEpi { WC_EPI_NONE = 0, WC_EPI_TRIL = 1, WC_EPI_PHI = 2 };
struct WcGemm {
    const void* A; int a_is_f32; int64_t a_rs, a_cs, a_bs, a_red;   // element strides: row, col, batch, reduce
    const void* B; int b_is_f32; int64_t b_rs, b_cs, b_bs, b_red;
    void* Cm; int c_is_f32; int64_t c_rs, c_cs, c_bs;
    void* Cm2; int64_t c2_rs, c2_cs, c2_bs;                          // optional second (float) output, e.g. the transpose
    int m, n, k;            // multiples of 32
    int batch, nred;
    int red_total;          // > 0: batch b reduces the terms r with b * nred + r < red_total (a long reduction cut into batches)
    int batch2; int64_t a_b2s, b_b2s, c_b2s;        // optional outer batch level (0 = none); c_b2s also strides Cm2
    double alpha;
    int epi;
};
hipError_t wc_launch_gemm(const WcGemm& g, hipStream_t st);
hipError_t wc_launch_sum_partials(const double* part, int nparts, int64_t n, double* out, hipStream_t st);      // out[e] = sum_p part[p][e], fixed order
hipError_t wc_launch_gemm_pair_dd_fd(const WcGemm& g0, const WcGemm& g1, hipStream_t st);      // two independent products (double x double, float x double) in one launch

hipError_t wc_launch_transpose_to_f32(const double* W, int C, int groups, float* A, float* At, hipStream_t st);  // A = W^T, At = W
hipError_t wc_launch_group_bias(const float* mu, const float* A, const float* beta, int G, int Kc, int C, int per_group,
                                float* center, float* bias, hipStream_t st, const float* center_in = nullptr /*given: the common centre (center is not written)*/);
hipError_t wc_launch_f64_to_f32(const double* src, float* dst, int64_t n, hipStream_t st);
hipError_t wc_launch_bwd_tail(const double* Q, int C, double scale, float* S, const double* gsum, const float* A, int Kc,
                              int64_t M, float* gmean, float* dbeta, hipStream_t st);      // S = scale sym(Q), gmean, dbeta = float(gsum) in one launch
