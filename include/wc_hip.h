/*
 * wc_hip.h -- C ABI of libwc_hip.so: the MI355X (gfx950) whitening-and-coloring hot path.
 *
 * The reference has no FFI boundary for this path: it sits behind the Keras 2.0.8 Layer
 * protocol.  Each entry point below names the reference call site /
 * op group it replaces:
 *
 *   wc_stats_f32        DecorelationNormalization.call, train mode, the transpose + mean +
 *                       centre + f f^T/(M-1) part            (class imported generator.py:9,
 *                       instantiated generator.py:24,26; body in the un-vendored gan/ submodule)
 *   wc_factor_f64       same call: (1-eps)Sigma+eps I, tf.cholesky, tf.matrix_triangular_solve
 *                       against I, and the moving_mean / moving_cov add_update ops; eval mode
 *                       (scorer.py:60,72) takes the moving statistics instead
 *   wc_color_f32        folds the coloring kernels into the whitening matrix:
 *                       Conv2D 1x1 (generator.py:50-51), ConditionalConv11 (generator.py:52-60),
 *                       FactorizedConv11 (generator.py:69-78), CenterScale variants
 *                       (generator.py:28-40)  ->  A_k = W^T Gamma_k
 *   wc_apply_f32        W f, the transpose back, and the 1x1-conv coloring + bias + Add
 *                       (generator.py:83-87 `stack`)            ->  y = (x - mu) A_k + beta_k
 *   wc_bwd_reduce_f32 / wc_bwd_factor_f64 / wc_bwd_apply_f32
 *                       the TF graph gradients of all of the above (training via
 *                       Trainer.train(), run.py:93-94); closed form in SURVEY.md row a10
 *
 * Contract (all entry points):
 *   - return 0 on success, a negative WC_ERR_* on a rejected argument, or a positive
 *     hipError_t from the launch; nothing throws, nothing aborts.
 *   - every pointer is DEVICE memory owned by the caller, including the workspace; the
 *     library allocates nothing and keeps no state between calls.
 *   - work is enqueued on `stream` (a hipStream_t passed as void*) and never synchronises
 *     with the host, so calls are graph-capturable and may run concurrently from several
 *     host threads on different streams/buffers.
 *   - x, y, gy, dx are row-major (M, C) float32 with C contiguous: an NHWC tensor viewed
 *     as M = N*H*W rows.  C must be a multiple of 32 with 32 <= C <= 1024 (callers pad
 *     other widths with zero channels, which leaves the real channels' result unchanged).
 *   - matrices are row-major, symmetric ones stored in full.  "slot" is an int32 (N,)
 *     array giving, per SAMPLE, the index into the leading axis of A / bias / gamma
 *     (NULL = every sample uses slot 0); a sample is HW consecutive rows.
 */
#ifndef WC_HIP_H
#define WC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WC_ABI_VERSION 8

#define WC_OK                 0
#define WC_ERR_NULL          -1   /* a required pointer is NULL                     */
#define WC_ERR_SHAPE         -2   /* M/N/HW/Kc not positive, or N*HW overflows       */
#define WC_ERR_CHANNELS      -3   /* C not a multiple of 32 in [32, 1024]            */
#define WC_ERR_WORKSPACE     -4   /* workspace smaller than wc_*_workspace_bytes()   */
#define WC_ERR_ARG           -5   /* eps/momentum/ddof out of range                  */

typedef void* wc_stream_t;        /* hipStream_t */

int         wc_abi_version(void);
const char* wc_error_string(int code);

/* ==== THE BOUNDARY (round 6) ==========================================================================================
 * SURVEY.md section 8b sketches the C ABI a replacement must export as `stats / factor / apply / bwd x 3 / workspace`.  That is this list:
 * ten functions (the seven stages + colouring glue) and their workspace sizers.  A binding that implements only these (INTEGRATION.md
 * section 3, the TF1/Keras-side stub) runs every layer of generator.py:13-90 -- forward, backward, train and eval mode, all colouring
 * variants -- to the parity bar; oracle/wc_cpu.cpp exports each of them with a `_cpu` suffix (the conformance twin SURVEY asks for),
 * and tests/test_abi.py checks both.
 *
 *   wc_stats_f32        K1   raw moments                         DecorelationNormalization.call (generator.py:24, 26)
 *   wc_factor_f64       K2   mu, L = chol((1-eps) Sigma + eps I), W = L^-1, moving statistics
 *   wc_color_f32             A_k = W^T Gamma_k                   Conv2D 1x1 / ConditionalConv11 / FactorizedConv11 (generator.py:49-78)
 *   wc_group_bias_f32        b_k = beta_k - mu A_k (grouped batches)
 *   wc_apply_f32        K3   y = (x - mu) A[slot] + b[slot]       (wc_apply_act_f32: the same with the block's ReLU in the epilogue)
 *   wc_bwd_reduce_f32   K4   R = f^T gy, column sums             autodiff of the above
 *   wc_bwd_factor_f64   K5   the C x C adjoint chain
 *   wc_bwd_apply_f32    K6   dx = [gy | f] [A^T ; S] - mean
 *   wc_*_workspace_bytes     the sizers of the eight above
 *
 * EVERYTHING ELSE in this header is an EXTENSION: a fusion or a data-format route of the same stages that the shipped torch layers take
 * where the shape allows (pre-split fp16 planes, the residual add as producer, bit masks, K1 + K2 in one call, the harness' convolutions
 * and spectral norm).  Each extension's comment names the core sequence it equals, each is parity-tested against that sequence or the
 * oracle, and none is needed for a drop-in: a caller that never heard of them loses speed, not results.  No new extension enters without
 * one leaving (ABI 8 retired nothing and added none).  The list below is what tests/test_abi.py parses. */
#define WC_CORE_API "wc_stats_f32 wc_factor_f64 wc_color_f32 wc_group_bias_f32 wc_apply_f32 wc_apply_act_f32 " \
                    "wc_bwd_reduce_f32 wc_bwd_factor_f64 wc_bwd_apply_f32 " \
                    "wc_stats_workspace_bytes wc_factor_workspace_bytes wc_color_workspace_bytes wc_apply_workspace_bytes " \
                    "wc_bwd_reduce_workspace_bytes wc_bwd_factor_workspace_bytes wc_bwd_apply_workspace_bytes"

/* K1 + K2 in one call (ABI 4): wc_stats_f32 followed by wc_factor_f64(training = 1) for the caller that does not need the moments
 * themselves -- per-replica statistics, the reference's behaviour (sync-WC all-reduces (sum, xtx) between the two and keeps the
 * separate entries).  Same arguments and results as that pair; the K1 tail's slab reduction and the K2 head's bookkeeping run as ONE
 * launch (bit-identical mu, L, W, chan_scale and moving statistics: the same float64 expressions in the same order), sum / xtx are
 * never stored.  M = all rows (groups * rows per group).  DecorelationNormalization.call, generator.py:24. */
size_t wc_whiten_workspace_bytes(int64_t M, int C, int groups);
/* Error words of K2's one-launch form (128 <= C <= 256: the inverse's workgroups wait, with a bounded spin, for the factorising
 * workgroup of the same launch): byte offset into the workspace given to wc_factor_f64 / wc_whiten_f32 of `groups` uint32 words, 64
 * bytes apart, that are 0 after a clean call and 1 when a wait ran out (W then also holds a NaN).  Read them back behind the
 * call (stream order) when the GPU is shared or time-sliced; WC_K2_TWO_LAUNCH=1 selects the form without such a wait.  0: no such
 * launch for this shape. */
size_t wc_factor_error_offset(int C, int groups);
size_t wc_whiten_error_offset(int64_t M, int C, int groups);
int    wc_whiten_f32(const float* x, int64_t M, int C, int groups, double eps, double momentum, int ddof,
                     float* moving_mean /*[C] in/out, nullable*/, float* moving_cov /*[C,C] in/out, nullable*/,
                     float* mu /*[groups,C]*/, float* chan_scale /*[C], nullable*/, double* L /*[groups,C,C]*/, double* W /*[groups,C,C]*/,
                     void* ws, size_t ws_bytes, wc_stream_t stream);

/* Bytes of scratch each stage needs for the given problem (16-byte aligned carve inside). */
size_t wc_stats_workspace_bytes(int64_t M, int C, int groups);
size_t wc_factor_workspace_bytes(int C, int groups);
size_t wc_color_workspace_bytes(int C, int Kc);
size_t wc_bwd_reduce_workspace_bytes(int64_t N, int64_t HW, int C, int Kc, int has_slot);
size_t wc_bwd_factor_workspace_bytes(int C, int Kc);
size_t wc_apply_workspace_bytes(int64_t N, int64_t HW, int C, int Kc);
size_t wc_apply_plan_bytes(int C, int Kc);
size_t wc_bwd_apply_workspace_bytes(int64_t N, int64_t HW, int C, int Kc);

/* K1: raw additive moments of the rows of x:  sum[c] = sum_m x[m,c],  xtx = x^T x  (float64).
 * These are what a sync-WC data-parallel run all-reduces before wc_factor_f64.
 * groups > 1: the M rows are `groups` independent batches of M/groups consecutive rows, each with its own
 * statistics (sum [groups,C], xtx [groups,C,C]) -- several forward passes of the same layer in one call. */
int wc_stats_f32(const float* x, int64_t M, int C, int groups,
                 double* sum /*[groups,C]*/, double* xtx /*[groups,C,C]*/,
                 void* ws, size_t ws_bytes, wc_stream_t stream);

/* K2: moments -> mu, Sigma; T = (1-eps)Sigma + eps I; L = chol(T); W = L^-1 (float64).
 * training != 0: statistics come from (sum, xtx, M); if moving_mean/moving_cov are non-NULL they
 *                are updated in place, moving <- momentum*moving + (1-momentum)*batch (un-shrunk Sigma).
 * training == 0: mu = moving_mean, Sigma = moving_cov (sum/xtx ignored, may be NULL).
 * groups > 1: mu [groups,C], L/W [groups,C,C]; the moving statistics receive the groups' updates one after the
 * other (as separate calls would); chan_scale stays [C] (the largest variance over the groups decides). */
int wc_factor_f64(const double* sum, const double* xtx, int64_t M /*rows per group*/, int C, int groups,
                  double eps, double momentum, int ddof, int training,
                  float* moving_mean /*[C]*/, float* moving_cov /*[C*C]*/,
                  float* mu /*[C] out*/, float* chan_scale /*[C] out, nullable: power-of-two 1/sigma for the fp16 path*/,
                  double* L /*[C*C] out*/, double* W /*[C*C] out*/,
                  void* ws, size_t ws_bytes, wc_stream_t stream);

/* A_k = W^T Gamma_k for k < Kc (float32 out), and its transpose At_k = A_k^T (what the backward
 * apply multiplies by; At may be NULL).  gamma == NULL means Gamma = I (whitening only, Kc = 1).
 * With chan_scale (from wc_factor_f64) and a buffer of wc_apply_plan_bytes(C, Kc) it also prepares the
 * "plan" of A -- split-fp16 tables of the fast apply -- so that wc_apply_f32 is a single kernel launch. */
/* per_group != 0: gamma holds groups*Kc tables and group g takes its own run, A[g*Kc+k] = W_g^T Gamma[g*Kc+k]
 * (per-SAMPLE coloring tables of a grouped batch: ConditionalConv11 / FactorizedConv11 with more classes than
 * samples, generator.py:52-60,69-78 at run.py:172-173's 200 / 1000 classes); 0: every group takes the same Kc tables. */
int wc_color_f32(const double* W /*[groups,C,C]*/, const float* gamma /*[Kc,C,C] ([groups*Kc,C,C] if per_group) or NULL*/,
                 int Kc, int C, int groups, int per_group,
                 float* A /*[groups*Kc,C,C] out, index g*Kc+k*/, float* At /*same shape, nullable*/,
                 const float* chan_scale /*[C], nullable*/, void* plan /*out, nullable*/,
                 void* ws, size_t ws_bytes, wc_stream_t stream);

/* Grouped forward glue: center[c] = mean_g mu[g,c] and bias[g*Kc+k] = beta[k] - (mu[g] - center) A[g*Kc+k], so that
 * wc_apply_f32(x, center, A, bias, slot = g*Kc + k) equals (x - mu_g) A[g*Kc+k] + beta[k] for every group.
 * per_group != 0: beta is [groups*Kc,C] and slot g*Kc+k takes beta[g*Kc+k] (see wc_color_f32). */
int wc_group_bias_f32(const float* mu /*[groups,C]*/, const float* A /*[groups*Kc,C,C]*/,
                      const float* beta /*[Kc,C] ([groups*Kc,C] if per_group), nullable*/,
                      int groups, int Kc, int C, int per_group, float* center /*[C] out*/, float* bias /*[groups*Kc,C] out*/,
                      wc_stream_t stream);

/* K3: y[n] = (x[n] - mu) A[slot[n]] + bias[slot[n]]   (bias NULL = 0; mu NULL = 0).
 * With a workspace of wc_apply_workspace_bytes() the split-fp16 MFMA fast path runs when the shape allows
 * (C in {32,64,128,256}, N*HW >= 1024 and a multiple of 16384/C rows); ws == NULL (and no plan) always takes the
 * exact f32-MFMA kernel.  Both give fp32-GEMM accuracy; a row tile holding an element outside the fp16 range -- or,
 * with slot != NULL and HW not a multiple of the 8192/C-row tile, a tile that straddles samples of different slots --
 * is detected on the device and recomputed in fp32 by the same kernel. */
int wc_apply_f32(const float* x, const float* mu, const float* A, const float* bias,
                 const int32_t* slot, int64_t N, int64_t HW, int C, int Kc,
                 float* y, const void* plan /*from wc_color_f32, nullable*/,
                 void* ws, size_t ws_bytes, wc_stream_t stream);

/* K3 with the activation that follows every WC site of the generator folded into the epilogue (SURVEY.md section 8f
 * row N2; generator.py:144-151, 154: `Activation('relu')` after each norm stack): relu = 1 writes max(y, 0) (NaN stays
 * NaN), relu = 0 is wc_apply_f32.  The gradient of the pair is the caller's mask (gy where y > 0, else 0) in front of
 * the unchanged backward. */
int wc_apply_act_f32(const float* x, const float* mu, const float* A, const float* bias,
                     const int32_t* slot, int64_t N, int64_t HW, int C, int Kc, int relu,
                     float* y, const void* plan /*from wc_color_f32, nullable*/,
                     void* ws, size_t ws_bytes, wc_stream_t stream);

/* ---- Pre-split activations (ABI 4, additive) -------------------------------------------------------------------------
 * The fp16 MFMA paths of this library compute with x = hi + lo (two fp16 terms, three products).  A tensor that several
 * stages read (K1, K3, K4, K6 all read the site's input x: generator.py:83-87 `stack` and its gradient) is converted by
 * each of them; the PRODUCER can hand it over already split instead -- the same 4 bytes per element:
 *     x[m][c] ~= center[c] + (hi[m][c] + lo[m][c]) / scale[c],   hi = fp16(g), lo = fp16(g - hi), g = (x - center) scale
 * `xs` = 2*M*C halves: the hi plane [M][C], then the lo plane [M][C].  scale[c]: a power of two that puts the channel's
 * sampled maximum into [8, 16] (>= 3700 x of headroom below fp16's range); center[c]: any value near the channel mean
 * (NULL = 0).  wc_split_f32 (the stand-in producer of tests and benches: scales given by the caller) clamps an element beyond +-60000
 * after scaling and sets *flag (device int, nullable) to 1; the producer the layers use, wc_resadd_split_f32, never clamps (below). */
size_t wc_split_bytes(int64_t M, int C);
/* int32 words of the producer's `flag` area (ABI 7): [0] status, [64, 64 + C) per-channel maxima, [64 + C, 64 + 2C) the sampled scales */
#define WC_SPLIT_FLAG_WORDS (64 + 2 * 1024)
/* center, scale from <= 256 sampled rows of x (outlier-proof, as K1's own shift / scales); zeroes flag[0..63]. */
int wc_split_scales_f32(const float* x, int64_t M, int C, float* center /*[C] out*/, float* scale /*[C] out*/,
                        int* flag /*[64] out: zeroed*/, wc_stream_t stream);
/* x (M, C) fp32 -> xs; relu != 0 clamps at zero first (NaN stays NaN). */
int wc_split_f32(const float* x, const float* center /*nullable*/, const float* scale, int64_t M, int C, int relu,
                 void* xs /*out*/, int* flag /*nullable*/, wc_stream_t stream);
int wc_unsplit_f32(const void* xs, const float* center /*nullable*/, const float* scale, int64_t M, int C, float* x /*out*/,
                   wc_stream_t stream);

/* K1 on a pre-split input: the raw moments of wc_stats_f32 (same outputs, same meaning of `groups`) of the tensor the planes
 * stand for.  No conversion in the kernel: LDS-DMA staging, transposing LDS reads (ds_read_b64_tr_b16) for the [channel][row]
 * MFMA operands.  C in {128, 256}, rows per group a multiple of 64 and N*HW >= 16384 (wc_stats_split_supported; WC_ERR_SHAPE
 * otherwise: convert with wc_unsplit_f32 and call wc_stats_f32).  Replaces the same reference call site as wc_stats_f32. */
int    wc_stats_split_supported(int64_t M, int C, int groups);
size_t wc_stats_split_workspace_bytes(int64_t M, int C, int groups);
int wc_stats_split_f16x2(const void* xs, const float* xs_center, const float* xs_scale, int64_t M, int C, int groups,
                         double* sum /*[groups,C]*/, double* xtx /*[groups,C,C]*/, void* ws, size_t ws_bytes, wc_stream_t stream);

/* K3 on a pre-split input:  y[n] = (x[n] - mu) A[slot[n]] + bias[slot[n]]  with x given as (xs, xs_center, xs_scale).
 * The staging is pure LDS-DMA into the fp16 image the MFMAs read (no conversion in the kernel).  `plan` = the tables
 * wc_color_f32 builds when it is given chan_scale = xs_scale (NULL: built here, in ws).  C in {128, 256} and N*HW a multiple
 * of 8192/C rows (wc_apply_split_supported; WC_ERR_SHAPE otherwise: convert with wc_unsplit_f32 and call wc_apply_f32).
 * relu as in wc_apply_act_f32.  Replaces the same reference call site as wc_apply_f32 (generator.py:83-87). */
/* bias_eff[k] = bias[k] + (xs_center - mu) A[k]  ([Kc, C]; bias, xs_center, mu nullable = 0): the additive term of the split
 * apply.  A caller that passes bias = bias_eff with mu = xs_center = NULL to wc_apply_split_f16x2 gets a single launch. */
int    wc_split_bias_f32(const float* A /*[Kc,C,C]*/, const float* bias, const float* xs_center, const float* mu, int Kc, int C,
                         float* bias_eff /*[Kc,C] out*/, wc_stream_t stream);
int    wc_apply_split_supported(int64_t N, int64_t HW, int C);
size_t wc_apply_split_workspace_bytes(int C, int Kc);
int wc_apply_split_f16x2(const void* xs, const float* xs_center /*nullable*/, const float* xs_scale,
                         const float* mu /*nullable*/, const float* A, const float* bias /*nullable*/,
                         const int32_t* slot, int64_t N, int64_t HW, int C, int Kc, int relu,
                         float* y, const void* plan /*nullable*/, void* ws, size_t ws_bytes, wc_stream_t stream);

/* ---- The residual add as the producer of pre-split activations (ABI 5, additive; SURVEY.md section 8f row N2: "residual Add
 * feeding K1") ---------------------------------------------------------------------------------------------------------------
 * reference generator.py:142-146: every `resblock(...)` ends in the Add of its convolution path h and its shortcut s, and the sum
 * is what the next block's first norm stack and Generator.BN.Final (generator.py:154) read.  With up = 1 the shortcut is given
 * BEFORE the nearest-neighbour upsample (a 1x1 convolution commutes with it) and every 2x2 output patch adds its source pixel:
 *     out[n][y][x][c] = h[n][y][x][c] + s[n][y >> up][x >> up][c]      h (N, H, W, C), s (N, H >> up, W >> up, C) or NULL (= 0)
 * wc_resadd_f32 writes the fp32 sum.  wc_resadd_split_f32 writes the sum in the pre-split format above instead (xs, center,
 * scale, flag: exactly what wc_split_scales_f32 + wc_split_f32 make of the fp32 sum -- the same <= 256 sampled rows, the same
 * centre and scales) in ONE pass over h and s behind one small sampling launch, so that K1 / K3 of the next site
 * (wc_whiten_split_f16x2, wc_apply_split_ex_f16x2) and the next block's shortcut convolution (wc_fold_channel_scale_f32 +
 * wc_conv_f16x3 on the same planes) read it without a conversion; x32 (nullable) also receives the fp32 sum, for a reader without
 * a planes path.  C in {128, 256} (wc_resadd_split_supported), N*H*W < 2^31.
 * ABI 7: NOTHING SATURATES.  A sampled scale can be too tight (a channel nearly constant on the sampled rows that spikes elsewhere);
 * the pass then records the true maximum of every channel that met an element beyond +-60000 and a second, gated launch (it leaves
 * at once otherwise; no host synchronisation, graph-capturable) rewrites the planes with those channels' scales lowered to fit, and
 * scale[] with them -- every consumer reads scale[] from device memory behind this call, so K1, K3, the shortcut convolution and
 * the backward see a consistent, exact tensor.  flag[0] = 1 afterwards says that this happened (informational).  A non-finite
 * element stays non-finite in the planes.  `flag`: WC_SPLIT_FLAG_WORDS int32 words, all scratch but [0].
 * wc_patch_sum_f32: the gradient of the up = 1 form with respect to s: out[n][y][x] = the sum of g over the 2x2 patch. */
int wc_resadd_split_supported(int64_t N, int64_t H, int64_t W, int C);
int wc_resadd_f32(const float* h, const float* s /*nullable*/, int64_t N, int64_t H, int64_t W, int C, int up, float* out,
                  wc_stream_t stream);
int wc_resadd_split_f32(const float* h, const float* s /*nullable*/, int64_t N, int64_t H, int64_t W, int C, int up,
                        void* xs /*out: 2*N*H*W*C halves*/, float* center /*[C] out*/, float* scale /*[C] out*/,
                        int* flag /*[WC_SPLIT_FLAG_WORDS] out: [0] = 1 when the gated rescaling pass ran*/, float* x32 /*out, nullable*/,
                        wc_stream_t stream);
int wc_patch_sum_f32(const float* g /*(N, 2 Hs, 2 Ws, C)*/, int64_t N, int64_t Hs, int64_t Ws, int C, float* out /*(N, Hs, Ws, C)*/,
                     wc_stream_t stream);
/* The producer FEEDING K1 (ABI 7): wc_resadd_split_f32 whose pass also accumulates the next site's covariance
 * moments -- the rows are summed, centred, scaled and split once, the words go to the planes AND, transposed, through the LDS into the
 * block triangle of X^T X on the matrix pipe (the covariance kernel of wc_stats_f32's fast path with the add in front and the planes'
 * stores behind its conversion), so the next site's K1 pass over the tensor (wc_stats_split_f16x2's kernel: 134 MB re-read at
 * 128x32x32x256) does not exist.  xs / center / scale / flag / x32: exactly wc_resadd_split_f32's outputs (bit for bit, gated rescaling
 * pass included -- it then redoes the partials as well).  `groups`: the statistic groups of the CONSUMING site (runs of N/groups
 * samples; slabs never cross them).  ws (wc_resadd_stats_workspace_bytes) receives the per-slab partials; it is what
 * wc_whiten_presummed_f16x2 (K1 tail + K2: = wc_whiten_split_f16x2 without its pass) or wc_stats_presummed_f16x2 (the raw moments, for
 * sync-WC's all-reduce: = wc_stats_split_f16x2 without its pass) take -- same M, C, groups, the same buffer.
 * wc_resadd_stats_supported: C in {128, 256}, up = 1 (every block of the shipped generators), W % 8 == 0, N*H*W/groups >= 20480 rows
 * in whole stages (64 rows at C = 256, 128 at C = 128); elsewhere WC_ERR_SHAPE: wc_resadd_split_f32 + wc_whiten_split_f16x2.
 * Replaces: the Add that ends `resblock` (generator.py:142-146) + the moments part of DecorelationNormalization.call (generator.py:24). */
int    wc_resadd_stats_supported(int64_t N, int64_t H, int64_t W, int C, int up, int groups);
size_t wc_resadd_stats_workspace_bytes(int64_t N, int64_t H, int64_t W, int C, int groups);
int    wc_resadd_stats_split_f32(const float* h, const float* s /*required*/, int64_t N, int64_t H, int64_t W, int C, int up, int groups,
                                 void* xs /*out*/, float* center /*[C] out*/, float* scale /*[C] out*/, int* flag /*[WC_SPLIT_FLAG_WORDS]*/,
                                 float* x32 /*out, nullable*/, void* ws /*out: the partials*/, size_t ws_bytes, wc_stream_t stream);
size_t wc_whiten_presummed_error_offset(int64_t M, int C, int groups);
int    wc_whiten_presummed_f16x2(const float* xs_center, int64_t M, int C, int groups, double eps, double momentum, int ddof,
                                 float* moving_mean /*nullable*/, float* moving_cov /*nullable*/, float* mu /*[groups,C]*/,
                                 double* L /*[groups,C,C]*/, double* W /*[groups,C,C]*/, void* ws /*from wc_resadd_stats_split_f32*/,
                                 size_t ws_bytes, wc_stream_t stream);
int    wc_stats_presummed_f16x2(const float* xs_center, int64_t M, int C, int groups, double* sum /*[groups,C]*/,
                                double* xtx /*[groups,C,C]*/, void* ws /*from wc_resadd_stats_split_f32*/, size_t ws_bytes, wc_stream_t stream);
/* A 1x1 convolution (the block's shortcut, generator.py:142-146) on a pre-split input: with x[c] = center[c] + g[c] / scale[c],
 *     sum_c x[c] w[o][c] + b[o] = sum_c g[c] wf[o][c] + bf[o],   wf[o][c] = w[o][c] / scale[c],  bf[o] = b[o] + <center, w[o]>
 * so wc_conv_f16x3 runs on the planes themselves (x scale 1) with the folded weight and bias; its weight gradient D (of wf, from
 * wc_conv_wrw_bias_f16x3 on the same planes) and bias gradient db unfold to dW[o][c] = D[o][c] / scale[c] + center[c] db[o].
 * w / wf / D / dW: element (o, c) at o * stride_o + c * stride_c. */
int wc_fold_channel_scale_f32(const float* w, int64_t stride_o, int64_t stride_c, int Cout, int Cin, const float* bias /*nullable*/,
                              const float* scale, const float* center, float* wf /*out*/, float* bf /*[Cout] out*/, wc_stream_t stream);
int wc_unfold_channel_scale_f32(const float* D, const float* db, int64_t stride_o, int64_t stride_c, int Cout, int Cin,
                                const float* scale, const float* center, float* dW /*out*/, wc_stream_t stream);

/* K1 + K2 on a pre-split input in one call (ABI 5): wc_stats_split_f16x2 + wc_factor_f64(training = 1) as wc_whiten_f32 is for the
 * fp32 input -- same results as that pair, the K1 tail and the K2 head as one launch, the moments never stored.  No chan_scale
 * output: the apply's input scales are the planes' own (pass xs_scale to wc_color_f32).  DecorelationNormalization.call,
 * generator.py:24. */
size_t wc_whiten_split_workspace_bytes(int64_t M, int C, int groups);
size_t wc_whiten_split_error_offset(int64_t M, int C, int groups);
int    wc_whiten_split_f16x2(const void* xs, const float* xs_center, const float* xs_scale, int64_t M, int C, int groups,
                             double eps, double momentum, int ddof, float* moving_mean /*nullable*/, float* moving_cov /*nullable*/,
                             float* mu /*[groups,C]*/, double* L /*[groups,C,C]*/, double* W /*[groups,C,C]*/,
                             void* ws, size_t ws_bytes, wc_stream_t stream);

/* The planes route's glue (ABI 5).  wc_color_split_f32: wc_color_f32 (one statistic group) for a site whose input is pre-split -- A, At,
 * the apply's tables for the PLANES' scales, and in the same launch bias_eff[k] = beta[k] + (xs_center - mu) A[k], the additive term
 * wc_apply_split_ex_f16x2 takes as `bias` with mu = xs_center = NULL (wc_split_bias_f32 as a launch of its own cost the forward site
 * 5 us).  C in {32, 64, 128, 256}.  wc_group_bias_centered_f32: wc_group_bias_f32 with the common centre GIVEN (the planes' centre):
 * bias[g*Kc+k] = beta[k] - (mu[g] - center) A[g*Kc+k] is then the grouped planes route's additive term directly.
 * Replace the same reference call sites as wc_color_f32 / wc_group_bias_f32 (generator.py:28-80 coloring layers folded into W). */
int wc_color_split_f32(const double* W /*[C,C]*/, const float* gamma /*[Kc,C,C] or NULL*/, int Kc, int C, float* A /*[Kc,C,C] out*/,
                       float* At /*nullable*/, const float* xs_scale, const float* xs_center, const float* mu /*[C]*/,
                       const float* beta /*[Kc,C], nullable*/, void* plan /*out: wc_apply_plan_bytes(C, Kc)*/, float* bias_eff /*[Kc,C] out*/,
                       void* ws, size_t ws_bytes, wc_stream_t stream);
int wc_group_bias_centered_f32(const float* mu /*[groups,C]*/, const float* A /*[groups*Kc,C,C]*/, const float* beta /*nullable*/,
                               const float* center /*[C] in*/, int groups, int Kc, int C, int per_group, float* bias /*[groups*Kc,C] out*/,
                               wc_stream_t stream);

/* The dictionary mix of the soft-assignment coloring "cWC_sa" (ABI 6):
 *     out[t] = base + sum_{e < E} alpha[idx[t], e] * dict[e],   t = 0 .. Kc-1   (C x C each)
 * dict (E, C, C): the filter dictionary, alpha (K, E): the per-class coefficients, idx (Kc) int32: the class of table t (NULL: t itself,
 * then Kc must equal K), base (C, C): the unconditional 1x1 kernel the reference adds (NULL: none).  Only the Kc tables the batch uses
 * are formed (K = 200 / 1000 classes at batch 64: run.py:172-173).  E <= 32, C a multiple of 4 (wc_factor_mix_supported).
 * wc_factor_mix_bwd_f32: from dout (Kc, C, C) -> ddict (E, C, C), dalpha (K, E; written whole, zero for absent classes), dbase (C, C); each
 * nullable; dalpha needs wc_factor_mix_bwd_workspace_bytes(E, Kc) of workspace.  Deterministic (fixed summation order).
 * Replace: generator.py:69-78 -- FactorizedConv11(number_of_classes, filters, filters_emb, use_bias=False)([x, cls]) + Conv2D 1x1 -> Add,
 * i.e. the (N, E) x (E, C^2) mix + gather of the (missing) gan.conditional_layers.FactorizedConv11 and TF's gradients of it. */
int    wc_factor_mix_supported(int E, int C);
int    wc_factor_mix_f32(const float* dict, const float* alpha, const int32_t* idx /*nullable*/, const float* base /*nullable*/,
                         int E, int C, int K, int Kc, float* out /*[Kc,C,C]*/, wc_stream_t stream);
size_t wc_factor_mix_bwd_workspace_bytes(int E, int Kc);
int    wc_factor_mix_bwd_f32(const float* dict, const float* alpha, const int32_t* idx /*nullable*/, const float* dout /*[Kc,C,C]*/,
                             int E, int C, int K, int Kc, float* ddict /*nullable*/, float* dalpha /*nullable*/, float* dbase /*nullable*/,
                             void* ws, size_t ws_bytes, wc_stream_t stream);

/* K3 on a pre-split input with the epilogues the generator's sites use (ABI 5): as wc_apply_split_f16x2, and
 *   relu_mask (nullable; relu = 1, N*HW a multiple of 32): the ReLU's one-bit gradient mask, as wc_apply_mask_f32 leaves it;
 *   planes + oscale instead of y (exactly one of y / planes is given): the output as the next convolution's fp16 planes,
 *   oscale the record wc_out_scale_f32 filled -- the protocol of wc_apply_planes_f32 (two launches: the pass and its gate).
 * Replaces the same reference call site as wc_apply_f32 (generator.py:83-87) + Activation('relu') (generator.py:144-151, 154). */
int wc_apply_split_ex_f16x2(const void* xs, const float* xs_center /*nullable*/, const float* xs_scale,
                            const float* mu /*nullable*/, const float* A, const float* bias /*nullable*/,
                            const int32_t* slot, int64_t N, int64_t HW, int C, int Kc, int relu,
                            float* y /*nullable*/, void* relu_mask /*out, nullable*/, void* planes /*out, nullable*/, float* oscale,
                            const void* plan /*nullable*/, void* ws, size_t ws_bytes, wc_stream_t stream);

/* K3 + ReLU + the ReLU's gradient mask as ONE BIT per element (ABI 4).  As wc_apply_act_f32 with relu = 1, and
 * relu_mask (wc_relu_mask_bytes(N*HW, C) bytes; N*HW a multiple of 32) receives, for every 32-row block t and channel c, the word
 * mask[t * C + c] whose bit b says "y[32 t + b][c] passed the ReLU" (y > 0, or NaN).  The planned fast kernel writes the words
 * from its epilogue (32 vector instructions and one 128-byte store per 32 x 32 outputs); every other path takes one pass over
 * y.  The backward then needs neither y nor a masked copy of the gradient to know the mask: wc_bwd_reduce_mask_f32.
 * (generator.py:144-151, 154: `Activation('relu')` behind every norm stack.) */
size_t wc_relu_mask_bytes(int64_t M, int C);
/* out = gy where the mask's bit is set, else 0 (the elementwise form of the ReLU gradient; wc_bwd_reduce_mask_f32 does the same
 * while it stages gy).  M a multiple of 32, C a multiple of 32 in [32, 1024]. */
int wc_relu_mask_apply_f32(const float* gy, const void* relu_mask, int64_t M, int C, float* out, wc_stream_t stream);
int wc_apply_mask_f32(const float* x, const float* mu, const float* A, const float* bias,
                      const int32_t* slot, int64_t N, int64_t HW, int C, int Kc,
                      float* y, void* relu_mask /*out*/, const void* plan /*from wc_color_f32, nullable*/,
                      void* ws, size_t ws_bytes, wc_stream_t stream);

/* K3 -> convolution hand-off (ABI 4).  Every WC site of the generator is followed by
 * ReLU and a convolution (generator.py:144-151), and this build's convolution (wc_conv_f16x3) reads its input as two fp16 planes
 * hi | lo of s * y with ONE power-of-two scale s: wc_apply_planes_f32 is wc_apply_act_f32 whose output leaves in that form -- no
 * fp32 y, and no wc_conv_split_f32 (one pass for max |y|, one to split) in front of the convolution.
 *   planes  (out) 2 * N*HW*C fp16: hi plane, then lo plane;  y ~= (hi + lo) / oscale[0]
 *   oscale  float[wc_apply_planes_scale_floats()], the scale record: wc_out_scale_f32 fills its input part from the coloring
 *           parameters alone, no pass over data (a whitened activation has unit covariance, so channel c of table k has standard
 *           deviation |Gamma_k[:, c]|: [1] = K as an int, [2 + k] = max_c (|beta_k[c]| + 16 |Gamma_k[:, c]|), K <= 1024; the
 *           kernel puts the largest bound into [2^13, 2^14), the convolution's own target for max |y|).  A caller with a better
 *           bound writes [1] = 1 (int) and [2] = the bound.  [0] (out) = the scale the planes were written with -- the device
 *           scalar wc_conv_f16x3 takes.  The rest is scratch (per-workgroup maxima).
 *   Two launches of the same kernel: the pass, and a gated one that leaves at once unless an element of s * y left fp16's range
 *   (beyond ~64 sigma), in which case it redoes the pass with the scale the measured maximum asks for.  No host synchronisation.
 *   relu_mask (nullable, relu = 1): as in wc_apply_mask_f32.  plan: required (wc_color_f32).
 * wc_apply_planes_supported: C in {128, 256}, N*HW a multiple of 8192/C rows and of 32; otherwise WC_ERR_SHAPE -- call
 * wc_apply_act_f32 and let the convolution split its input. */
int    wc_apply_planes_supported(int64_t N, int64_t HW, int C);
size_t wc_apply_planes_scale_floats(void);
int    wc_out_scale_f32(const float* gamma /*[K,C,C], nullable = identity*/, const float* beta /*[K,C], nullable*/, int K, int C,
                        float* oscale, wc_stream_t stream);
int    wc_apply_planes_f32(const float* x, const float* mu, const float* A, const float* bias, const int32_t* slot,
                           int64_t N, int64_t HW, int C, int Kc, int relu, void* planes /*out*/, float* oscale,
                           void* relu_mask /*out, nullable*/, const void* plan, wc_stream_t stream);

/* K4: R[k] = sum_{n: slot[n]=k} (x[n]-mu)^T gy[n]  (Kc,C,C),  gsum[k] = sum_{n in k} rows of gy[n]  (Kc,C). */
int wc_bwd_reduce_f32(const float* x, const float* mu, const float* gy, const int32_t* slot,
                      int64_t N, int64_t HW, int C, int Kc,
                      double* R /*[Kc,C,C]*/, double* gsum /*[Kc,C]*/,
                      void* ws, size_t ws_bytes, wc_stream_t stream);

/* K4 / K6 sharing the fp16 paths' per-channel input scales (ABI 3).  Both stages scale (x - mu) and gy by per-channel powers
 * of two sampled from <= 256 rows; K4 samples them anyway.  `scales` = float[2*C]: the scales of (x - mu), then those of gy.
 * wc_bwd_reduce_scaled_f32 writes them (also when its own reduction takes the exact path); wc_bwd_apply_scaled_f32 takes them
 * instead of sampling the same two tensors again and builds the tables of both its passes in one launch: three launches
 * instead of six.  scales == NULL: exactly wc_bwd_reduce_f32 / wc_bwd_apply_f32.  Results are identical either way (the same
 * samples give the same scales). */
int wc_bwd_reduce_scaled_f32(const float* x, const float* mu, const float* gy, const int32_t* slot,
                             int64_t N, int64_t HW, int C, int Kc,
                             double* R /*[Kc,C,C]*/, double* gsum /*[Kc,C]*/, float* scales_out /*[2C], nullable*/,
                             void* ws, size_t ws_bytes, wc_stream_t stream);
/* K4 behind a site whose ReLU rode in K3's epilogue (wc_apply_act_f32, relu = 1): the activation's gradient mask
 * gy := gy unless y <= 0 (a NaN in y lets the gradient through, as aten::threshold_backward does; generator.py:144-154
 * `Activation('relu')` after each norm stack) applied while K4 stages gy,
 * instead of an elementwise pass over three tensors in front of it.  relu_y = the site's output y; gy_masked (out, same
 * shape as gy, must not alias it) receives the masked gradient -- what K6 then takes as its gy.  R, gsum, scales_out as in
 * wc_bwd_reduce_scaled_f32, computed from the masked gradient.  relu_y == NULL (then gy_masked must be NULL too) is that
 * function.  Shapes whose reduction kernel does not mask in its staging take one elementwise pass inside the call. */
int wc_bwd_reduce_relu_f32(const float* x, const float* mu, const float* gy, const float* relu_y /*nullable*/, const int32_t* slot,
                           int64_t N, int64_t HW, int C, int Kc,
                           double* R /*[Kc,C,C]*/, double* gsum /*[Kc,C]*/, float* gy_masked /*[N*HW*C] out, nullable*/,
                           float* scales_out /*[2C], nullable*/, void* ws, size_t ws_bytes, wc_stream_t stream);
/* The same with the mask in wc_apply_mask_f32's one-bit form: K4 reads x, gy and 1/32 of a tensor where wc_bwd_reduce_relu_f32
 * reads three tensors (the quadrant kernel at C = 256 masks while it stages: one 16-byte load of mask words per thread and
 * 64-row stage; other shapes take one elementwise pass inside the call).  gy_masked as above (required). */
int wc_bwd_reduce_mask_f32(const float* x, const float* mu, const float* gy, const void* relu_mask, const int32_t* slot,
                           int64_t N, int64_t HW, int C, int Kc,
                           double* R /*[Kc,C,C]*/, double* gsum /*[Kc,C]*/, float* gy_masked /*[N*HW*C] out*/,
                           float* scales_out /*[2C], nullable*/, void* ws, size_t ws_bytes, wc_stream_t stream);

/* The ReLU'd backward with NO masked copy of the gradient in between (ABI 4): K4 applies the one-bit mask while it stages gy and
 * writes nothing back (wc_bwd_reduce_bits_f32: x, gy and M*C/8 bytes of mask in, R / gsum / scales out), K6 applies the same bits
 * while it converts gy (wc_bwd_apply_bits_f32: gy here is the gradient BEFORE the ReLU, as K4 received it).  The backward site then
 * moves 5.03 M*C*4 bytes instead of 7 (fp32 y, masked copy) -- VERDICT r2 item 3.  Only where both kernels carry the mask in their
 * staging (wc_bwd_bits_supported: C = 256, the fast reduction and the one-pass K6, N*HW a multiple of 32; `scales` required);
 * elsewhere WC_ERR_SHAPE: use wc_bwd_reduce_mask_f32 (which writes the masked gradient) + wc_bwd_apply_scaled_f32.
 * Arguments otherwise as in wc_bwd_reduce_scaled_f32 / wc_bwd_apply_scaled_f32.  generator.py:144-151, 154. */
int wc_bwd_bits_supported(int64_t N, int64_t HW, int C, int has_slot);
int wc_bwd_reduce_bits_f32(const float* x, const float* mu, const float* gy, const void* relu_mask, const int32_t* slot,
                           int64_t N, int64_t HW, int C, int Kc, double* R /*[Kc,C,C]*/, double* gsum /*[Kc,C]*/,
                           float* scales_out /*[2C]*/, void* ws, size_t ws_bytes, wc_stream_t stream);
int wc_bwd_apply_bits_f32(const float* gy, const void* relu_mask, const float* x, const float* mu, const float* At, const float* S,
                          const float* gmean, const int32_t* slot, int64_t N, int64_t HW, int C, int Kc,
                          const float* scales /*[2C] from wc_bwd_reduce_bits_f32*/, float* dx, void* ws, size_t ws_bytes, wc_stream_t stream);
int wc_bwd_apply_scaled_f32(const float* gy, const float* x, const float* mu, const float* At,
                            const float* S, const float* gmean, const int32_t* slot,
                            int64_t N, int64_t HW, int C, int Kc, const float* scales /*[2C], nullable*/, float* dx,
                            void* ws, size_t ws_bytes, wc_stream_t stream);

/* K4 / K6 of a site whose input exists as pre-split planes (ABI 5): the backward reads x from the planes the residual add wrote for the
 * forward (wc_resadd_split_f32) -- no fp32 copy of x has to exist.  K4: the X threads take 8 bytes per plane and row and build the
 * transposed fp16 image with one byte-permute per word (no centre / scale / split instructions); K6: a chunk of x is [hi row | lo row]
 * and its conversion a copy.  (x - mu) = g / scale + (center - mu): K4 adds the rank-one term (center - mu) (sum gy)^T to R, K6 folds
 * (center - mu) S into gmean.  relu_mask (nullable): the site's one-bit ReLU mask, applied by both as in wc_bwd_reduce_bits_f32 /
 * wc_bwd_apply_bits_f32 (gy is then the gradient BEFORE the ReLU).  scales: float[2C]; K4 writes gy's scales to [C, 2C) and K6 reads
 * them there ([0, C) is unused: x's scales are xs_scale).  wc_bwd_xsplit_supported: C = 256 (the fast reduction and the one-pass K6) or, since ABI 7, C = 128 (the plain two-operand
 * reduction staging x from the planes; K6 as the planes kernel for (x - mu) S - sub followed by the accumulating fp32 kernel for + gy At; relu_mask must be
 * NULL there: mask gy in front),
 * N*HW a multiple of 32; elsewhere WC_ERR_SHAPE (keep an fp32 x: wc_resadd_split_f32's x32).  Replace the same TF graph gradients as
 * wc_bwd_reduce_f32 / wc_bwd_apply_f32 (run.py:93-94). */
int    wc_bwd_xsplit_supported(int64_t N, int64_t HW, int C, int has_slot);
int    wc_bwd_reduce_xsplit_f32(const void* xs, const float* xs_center, const float* xs_scale, const float* mu, const float* gy,
                                const void* relu_mask /*nullable*/, const int32_t* slot, int64_t N, int64_t HW, int C, int Kc,
                                double* R /*[Kc,C,C]*/, double* gsum /*[Kc,C]*/, float* scales_out /*[2C]*/,
                                void* ws /*wc_bwd_reduce_workspace_bytes*/, size_t ws_bytes, wc_stream_t stream);
size_t wc_bwd_apply_xsplit_workspace_bytes(int C, int Kc);
int    wc_bwd_apply_xsplit_f32(const float* gy, const void* relu_mask /*nullable*/, const void* xs, const float* xs_center,
                               const float* xs_scale, const float* mu, const float* At, const float* S, const float* gmean,
                               const int32_t* slot, int64_t N, int64_t HW, int C, int Kc, const float* scales /*[2C] from the K4 above*/,
                               float* dx, void* ws, size_t ws_bytes, wc_stream_t stream);

/* K5: dgamma[k] = W R[k];  and, when training != 0, the statistics path
 *     Wbar = sum_k Gamma_k R_k^T;  Lbar = -tril(W^T Wbar W^T);  P = Phi(L^T Lbar)  [computed as -Phi(Wbar W^T): L^T W^T = I and
 *     the strictly upper part of W^T Wbar W^T never reaches Phi's triangle -- one product instead of three; L itself is not read];
 *     S = 2(1-eps)/(M-ddof) sym(W^T P W)  (float32, C x C);   gmean = (1/M) sum_k gsum_k A_k^T  (C).
 * gamma == NULL means Gamma = I.  dgamma / dbeta may be NULL (that output is skipped).  S and gmean are
 * untouched when training == 0. */
int wc_bwd_factor_f64(const double* R, const double* gsum, const double* W, const double* L,
                      const float* gamma, const float* A, int Kc, int C, int64_t M,
                      double eps, int ddof, int training,
                      float* dgamma /*[Kc,C,C]*/, float* dbeta /*[Kc,C]*/,
                      float* S /*[C,C]*/, float* gmean /*[C]*/,
                      void* ws, size_t ws_bytes, wc_stream_t stream);

/* K6: dx[n] = gy[n] At[slot[n]] + (x[n]-mu) S - gmean     (S, gmean NULL in eval mode: dx = gy A^T).
 * At[k] = A[k]^T as written by wc_color_f32. */
int wc_bwd_apply_f32(const float* gy, const float* x, const float* mu, const float* At,
                     const float* S, const float* gmean, const int32_t* slot,
                     int64_t N, int64_t HW, int C, int Kc, float* dx,
                     void* ws, size_t ws_bytes, wc_stream_t stream);

/* N3: spectral normalisation of a weight matrix W (rows, cols) row-major -- SNConv2D / SNDense /
 * SNEmbeding at discriminator.py:26-33, generator.py:104-113 (a convolution kernel is passed in its memory order with
 * rows = output channels; singular values do not depend on the order of the columns).
 *   `iterations` steps of  v <- W^T u / max(|W^T u|, eps),  u <- W v / max(|W v|, eps)   (--spectral_iterations,
 *   run.py:269; 0 = inference: u, v are used as they are);  sigma = u^T W v;  w_sn = W / sigma.
 * u (rows) and v (cols) are updated in place when iterations > 0.  One launch of up to 32 co-resident workgroups that
 * meet through the last 16 bytes of `ws`: a buffer of wc_spectral_norm_workspace_bytes() that belongs to this weight,
 * ZEROED ONCE by the caller before its first use (every launch leaves the meeting words zero) and used by one launch at
 * a time.  rows + cols floats must fit the LDS (WC_ERR_SHAPE otherwise). */
size_t wc_spectral_norm_workspace_bytes(int rows, int cols);
/* After a forward call the 32 floats at this byte offset of `ws` hold per-workgroup maxima of |w_sn| (unused entries keep
 * the caller's zeros): wc_conv_weights_f32 takes them as `known_amax` and skips its own sweep over the weight. */
size_t wc_spectral_norm_amax_offset(int rows, int cols);
/* The unsigned word at this byte offset of `ws` is the weight's sticky error word: a workgroup whose wait for its peers ran out
 * (> 0.1 s: the device could not hold the launch's workgroups at once -- CUs held by another process or stream) sets it to 1 and
 * goes on; that call's outputs are then invalid.  0 after every normal call; never cleared by the library. */
size_t wc_spectral_norm_error_offset(int rows, int cols);
int wc_spectral_norm_f32(const float* W, int rows, int cols, float* u, float* v, int iterations, float eps,
                         float* w_sn /*[rows*cols] out*/, float* sigma /*[1] out*/,
                         float* u_used /*[rows] out, nullable*/, float* v_used /*[cols] out, nullable: u, v as used for sigma,
                                                                    for the backward of THIS call (u, v move on)*/,
                         void* ws, size_t ws_bytes, wc_stream_t stream);

/* Gradient of the above w.r.t. W for constant u, v:  dW = (g - fully_diff * <g, w_sn> u v^T) / sigma.
 * fully_diff == 0 treats sigma as a constant of the step (--fully_diff_spectral 0, run.py:268: dW = g / sigma).
 * Same workspace rules (the same buffer as the forward may be passed: different words are used). */
int wc_spectral_norm_bwd_f32(const float* g, const float* w_sn, const float* u, const float* v, const float* sigma,
                             int rows, int cols, int fully_diff, float* dW,
                             void* ws, size_t ws_bytes, wc_stream_t stream);

/* The same two operations for EVERY spectrally normalised layer of a network in one launch each: the layers depend on
 * the weights only, not on each other or on the activations, and one launch keeps up to 16 x 32 workgroups busy where
 * per-layer launches run back to back on a mostly idle chip.  `items` is a HOST array (its contents travel in the
 * kernel arguments); every field as in the per-layer calls, one workspace per item. */
typedef struct {
    const float* W; float* u; float* v; float* w_sn; float* sigma; float* u_used; float* v_used; void* ws;
    int rows, cols;
} wc_sn_item;
typedef struct {
    const float* g; const float* w_sn; const float* u; const float* v; const float* sigma; float* dW; void* ws;
    int rows, cols;
} wc_sn_bwd_item;
int wc_spectral_norm_batched_f32(const wc_sn_item* items, int count, int iterations, float eps, wc_stream_t stream);
int wc_spectral_norm_bwd_batched_f32(const wc_sn_bwd_item* items, int count, int fully_diff, wc_stream_t stream);

/* ---- Convolutions around the WC sites (the callers either side of the path) ------------------
 * Replaces, for the generator/critic residual blocks, Keras `Conv2D(3x3, padding='same')` (generator.py:142-158,
 * discriminator.py:41-54), its `UpSampling2D -> Conv2D` and `Conv2D -> AveragePooling2D` pairs (as ONE 4x4 stride-2
 * transposed / strided convolution: exact rewrites) and the data gradient of each, fp32-accurate
 * on the fp16 MFMA pipe with split operands (3 products, the scheme of wc_apply_f32's fast path).
 *
 * One geometry describes all of them as an implicit GEMM over a "virtual grid" (N, H, W): point (y, x) of phase p
 * reads input pixel (y*in_stride + dy[p][t], x*in_stride + dx[p][t]) for tap t (zero outside the Hin x Win plane) and
 * writes output pixel (y*out_stride + off_y[p], x*out_stride + off_x[p]) of the Hout x Wout plane; tap t of phase p
 * multiplies by a weight slice built from the source slices (wr, ws) below.  Tensors are NHWC fp32, dense.
 *   3x3 same:              H=Hin=Hout, strides 1, 9 taps dy=r-1;  1 phase
 *   4x4 stride-2 conv:     H=Hout=Hin/2, in_stride 2, 16 taps dy=r-1; 1 phase
 *   4x4 stride-2 transposed conv: H=Hin, out_stride 2, 4 phases (py,px) of 2x2 taps, r = py+1-2*dy
 * Data gradients are the same three with the channel roles swapped in the weight image (stride_k <-> stride_n). */
typedef struct {
    int N, H, W;                    /* virtual grid */
    int Hin, Win, Cin;              /* input plane, reduction channels (multiple of 32) */
    int Hout, Wout, Cout;           /* output plane, output channels (multiple of 128) */
    int in_stride, out_stride;
    int ntaps, nphase;              /* <= 16 taps, <= 4 phases */
    signed char dy[4][16], dx[4][16];
    signed char off_y[4], off_x[4];
    /* the weight slice of tap t of phase p = wcoef * the sum of nsrc[p][t] (1..4) source slices (wr[p][t][m], ws[p][t][m]):
     * one slice for a plain convolution; the 4x4 kernels of the pooled / upsampled 3x3 convolutions are sums of 3x3 taps
     *  and are formed here, from the 3x3 weight -- and the weight gradient is folded back the same way */
    signed char nsrc[4][16];
    signed char wr[4][16][4], ws[4][16][4];
    float wcoef;
} wc_conv_geom;

#define WC_CONV_SCRATCH_BYTES 2048

/* 1 when wc_conv_f16x3 takes the geometry (N*H*W a multiple of 128, channel multiples as above). */
int wc_conv_supported(const wc_conv_geom* g);

/* hi = fp16(s*x), lo = fp16(s*x - hi) over n floats (n % 4 == 0), s = the power of two that puts max|x| into
 * [2^13, 2^14) (written to *scale on the device); relu != 0 clamps at zero first.  `amax_scratch`: WC_CONV_SCRATCH_BYTES
 * device bytes (per-workgroup maxima; no atomics, nothing to clear). */
int wc_conv_split_f32(const float* x, int64_t n, int relu, void* hi, void* lo, float* scale, void* amax_scratch,
                      wc_stream_t stream);

/* The same, and -- when `colsum_partials` (WC_CONV_COLSUM_ROWS x C floats) is given -- partial column sums of x seen as
 * rows of C channels (C a multiple of 4 with 256 % (C/4) == 0), collected while the maximum is taken: the bias gradient of
 * a layer whose output gradient is being split; wc_conv_wrw_bias_f16x3 adds the rows up. */
#define WC_CONV_COLSUM_ROWS 512
int wc_conv_split_colsum_f32(const float* x, int64_t n, int relu, void* hi, void* lo, float* scale, void* amax_scratch,
                             float* colsum_partials, int C, wc_stream_t stream);

/* The same split in ONE launch, its scale taken from the call before at the same call site (ABI 7).  `hist`: WC_CONV_HIST_FLOATS floats
 * that belong to the call site (one convolution's input, or its output gradient), zero-filled once and kept across calls: two arrays of
 * 512 (maximum, tag) pairs.  A call reads both, takes the array whose tags are all equal with the larger tag -- what the previous call
 * left -- and the power of two that puts THAT maximum into [2^7, 2^8): room for a 255-fold growth before fp16 overflows, precision to
 * 2^-20 of the tensor's maximum down to a 4000-fold shrink (hi + lo carry 22 bits wherever the scaled maximum lies in [2^-5, 65504]);
 * each workgroup leaves (its maximum, tag + 1) in the other array.  No atomics, no host-side parity: the record is a function of the
 * sequence of tensors alone, so an eager run and a replayed hipGraph of the same calls give the same bits.  NOTHING clamps.
 * ABI 8: a gated second launch follows on the same stream; it reads the record (the tensor's own maximum is in it by then) and, when the
 * scaled maximum fell outside [2^-5, 65504) -- the previous tensor was all zero (a saturated hinge critic), a growth beyond 255-fold, a
 * shrink beyond 4096-fold -- splits the tensor again with the MEASURED scale (the bits of wc_conv_split_f32) and counts the event in word
 * WC_CONV_HIST_REDO of `hist`; inside the window it returns at once.  Capturable, no host synchronisation, never quietly wrong.
 * The second launch costs ~2.3 us per call (0.56 ms of the 18-ms CIFAR-10 step when every split of the step takes it: as much as the
 * history saves), so the caller chooses per call: bit 1 of `bootstrap` set = no second launch.  The shipped layers take it for the OUTPUT
 * GRADIENTS (the tensors that do go to zero and swing by orders of magnitude) and not for the layer inputs.  Without it a call is still
 * exact after all-zero tensors (the record carries the maximum a call assumed: zeros leave the site's scale where it was), loud above a
 * 255-fold growth (inf), and short of bits for ONE call after a shrink beyond 4096-fold.
 * bit 0 of `bootstrap` (the site's first call): the two-launch form with the measured maximum, which seeds the record.  Calls of one
 * site must be ordered (one stream, or events).
 * colsum_partials / C as in wc_conv_split_colsum_f32 (nullable).  Replaces the absmax pass of wc_conv_split_f32 (~130 launches of
 * 3-30 us per G+D step; reference call sites: every Conv2D of discriminator.py:41-54 / generator.py:142-158 as in wc_conv_f16x3). */
#define WC_CONV_HIST_FLOATS (4 * 512 + 16)
#define WC_CONV_HIST_REDO (4 * 512)      /* index of the uint32 count of second passes taken */
int wc_conv_split_hist_f32(const float* x, int64_t n, int relu, void* hi, void* lo, float* scale, float* colsum_partials /*nullable*/, int C,
                           float* hist, int bootstrap, wc_stream_t stream);

/* Weight fragment images for a geometry: element (k, n, r, s) of the source is w[k*stride_k + n*stride_n + r*stride_r +
 * s*stride_s] (k = reduction channel, n = output channel of the product), `n_elems` = extent of the source storage (for
 * the tensor scale).  `image`: wc_conv_weights_bytes(g) device bytes.  `known_amax` (nullable): `known_count` device
 * floats whose maximum is max|w| (e.g. the ones wc_spectral_norm_f32 leaves) -- then the weight is not swept again. */
size_t wc_conv_weights_bytes(const wc_conv_geom* g);
int wc_conv_weights_f32(const float* w, int64_t stride_k, int64_t stride_n, int64_t stride_r, int64_t stride_s,
                        int64_t n_elems, const wc_conv_geom* g, void* image, float* scale, void* amax_scratch,
                        const float* known_amax, int known_count, wc_stream_t stream);

/* The images of one weight for two geometries (the layer's forward and its data gradient) in one launch: as two calls of
 * wc_conv_weights_f32 with the same source; `scale`: TWO floats (one per image, equal). */
int wc_conv_weights_pair_f32(const float* w, int64_t stride_r, int64_t stride_s, int64_t n_elems,
                             int64_t a_stride_k, int64_t a_stride_n, const wc_conv_geom* ga, void* image_a,
                             int64_t b_stride_k, int64_t b_stride_n, const wc_conv_geom* gb, void* image_b,
                             float* scale, void* amax_scratch, const float* known_amax, int known_count,
                             wc_stream_t stream);

/* y = conv(x, w) (+ bias[Cout]) (then max(., 0) when relu != 0) for the geometry; x as split planes, `zero_line` = 64
 * device bytes of zeros (the padding). */
size_t wc_conv_workspace_bytes(const wc_conv_geom* g);     /* 0 unless the grid is small (the tap loop is then shared) */
int wc_conv_f16x3(const void* xhi, const void* xlo, const float* xscale, const void* wimage, const float* wscale,
                  const float* bias, const void* zero_line, const wc_conv_geom* g, int relu, float* y,
                  void* ws, size_t ws_bytes, wc_stream_t stream);

/* Weight gradient of the same convolution: dW(k, n, r, s) = sum over the grid of x[input pixel][k] * gy[output pixel][n]
 * for the FORWARD geometry `g` (x and gy as split planes; Cin and Cout multiples of 128), written to
 * dw[k*stride_k + n*stride_n + r*stride_r + s*stride_s] (every element of the taps the geometry names; fixed summation
 * order).  `ws`: wc_conv_wrw_workspace_bytes(g) device bytes. */
size_t wc_conv_wrw_workspace_bytes(const wc_conv_geom* g);
int wc_conv_wrw_f16x3(const void* xhi, const void* xlo, const float* xscale, const void* ghi, const void* glo,
                      const float* gscale, const void* zero_line, const wc_conv_geom* g, float* dw, int64_t stride_k,
                      int64_t stride_n, int64_t stride_r, int64_t stride_s, void* ws, size_t ws_bytes,
                      wc_stream_t stream);

/* Weight and bias gradient of a 'same' convolution (1x1 or 3x3, stride 1) with a handful of INPUT channels, ksize^2 * Cin < 32 -- the
 * critic's first block on images: Conv2D 3 -> 128 and the 1x1 shortcut 3 -> 128 (discriminator.py:41-54; ABI 7).  x [N,H,W,Cin] and
 * gy [N,H,W,Cout] in fp32 NHWC, Cout a multiple of 128; one pass over gy on the fp32 matrix pipe (v_mfma_f32_32x32x2_f32, no split, no scales):
 *     dw[c*stride_k + o*stride_n + r*stride_r + s*stride_s] = sum_p x[p + (r, s) - pad][c] gy[p][o],   db[o] = sum_p gy[p][o]  (nullable)
 * in a fixed summation order.  `ws`: wc_conv_wrw_narrow_workspace_bytes(...) device bytes.  Replaces MIOpen's fp32 weight-gradient kernels
 * for these layers (138 / 54 us per critic update at 128x32x32, 0.5 TB/s of gy) and the bias gradient's reduction. */
int    wc_conv_wrw_narrow_supported(int64_t N, int64_t H, int64_t W, int Cin, int Cout, int ksize);
size_t wc_conv_wrw_narrow_workspace_bytes(int64_t N, int64_t H, int64_t W, int Cin, int Cout, int ksize);
int    wc_conv_wrw_narrow_f32(const float* x, const float* gy, int64_t N, int64_t H, int64_t W, int Cin, int Cout, int ksize,
                              float* dw, int64_t stride_k, int64_t stride_n, int64_t stride_r, int64_t stride_s, float* db /*nullable*/,
                              void* ws, size_t ws_bytes, wc_stream_t stream);

/* Forward of the same kind of layer (ksize^2 * Cin < 32, Cout a multiple of 128; wc_conv_wrw_narrow_supported):
 *     y[p][o] = bias[o] + sum_{r,s,c} x[p + (r, s) - pad][c] w[c*stride_k + o*stride_n + r*stride_r + s*stride_s]      (relu != 0: max(., 0))
 * in one launch on the fp32 matrix pipe, the bias as a row of the product (bias nullable).  The strides may be negative: with x := gy of a layer
 * with a handful of OUTPUT channels, stride_k / stride_n exchanged and the tap strides negated (w pointing at its last tap) this is that
 * layer's data gradient (the generator's last layer, generator.py:155-157).  Replaces MIOpen's forward + two bias launches for the critic's
 * first block (discriminator.py:41-54 on images). */
int    wc_conv_fwd_narrow_f32(const float* x, const float* w, int64_t stride_k, int64_t stride_n, int64_t stride_r, int64_t stride_s,
                              const float* bias /*nullable*/, int64_t N, int64_t H, int64_t W, int Cin, int Cout, int ksize, int relu,
                              float* y, wc_stream_t stream);

/* wc_conv_wrw_f16x3 plus the bias gradient db[Cout] = column sums of gy, from the partial rows wc_conv_split_colsum_f32
 * left while gy was split (fixed summation order, no extra launch). */
int wc_conv_wrw_bias_f16x3(const void* xhi, const void* xlo, const float* xscale, const void* ghi, const void* glo,
                           const float* gscale, const void* zero_line, const wc_conv_geom* g, float* dw, int64_t stride_k,
                           int64_t stride_n, int64_t stride_r, int64_t stride_s, const float* colsum_partials, float* db,
                           void* ws, size_t ws_bytes, wc_stream_t stream);

/* Bandwidth yardstick used by bench.py: dst[i] = src[i] (float4 grid-stride copy), same stream rules. */
int wc_stream_copy_f32(const float* src, float* dst, int64_t n, wc_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* WC_HIP_H */
