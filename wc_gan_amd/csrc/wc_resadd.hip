// The residual add of a generator block as the PRODUCER of the next WC site's input ("residual Add
// feeding K1"; reference generator.py:142-146 -- `resblock(...)` ends in the Add of the convolution path and the shortcut, and
// its result is what the next block's first norm stack and Generator.BN.Final (generator.py:154) read).
//
//     out[n][y][x][c] = h[n][y][x][c] + s[n][y >> up][x >> up][c]
//
// `s` is the 1x1 shortcut taken BEFORE the nearest-neighbour upsample (a per-pixel map commutes with it: DESIGN.md section 4.3), so
// with up = 1 every 2x2 output patch adds its one source pixel and no upsampled tensor exists.
//
// Rounds 1-3 ran this as a torch broadcast add that wrote fp32, which K1 and K3 of the next site (and the next block's shortcut
// convolution) each read back and converted to fp16 hi | lo for the matrix pipe.  Here the add writes the PRE-SPLIT format of
// wc_split.hip directly -- the same 4 bytes per element --
//     out ~= center[c] + (hi + lo) / scale[c],   hi = fp16(g), lo = fp16(g - hi), g = (out - center) scale
// so xtx_split_kernel (K1) and apply_split_kernel (K3) take the tensor by LDS-DMA with no conversion instruction, and the
// shortcut convolution reads the same planes (1 / scale and center folded into its weight and bias: wc_fold_channel_scale_f32).
// centre and scale come from <= 256 sampled rows of the SUM (resadd_sample_kernel: the statistics of wc_split_scales_f32 /
// K1's own subsample -- median-of-groups centre, robust maximum into [8, 16) -- taken on h + up(s) without forming it), so the
// planes carry exactly what wc_split_f32 would have made of the fp32 sum: one small launch, then one pass over h and s.
//
// Nothing saturates silently (round 5).  A sampled scale can be too tight -- a channel that is nearly
// constant on the <= 256 sampled rows and spikes elsewhere (sparse feature maps) -- and an element beyond +-60000 after scaling
// (> 3700 x its channel's sampled maximum) does not fit fp16.  Rounds 4's pass clamped it, raised flag[0] and went on: K1, K3, the
// shortcut convolution and the backward then computed on the clamped tensor.  Now the pass records, for every channel that met such
// an element, the channel's TRUE maximum (atomicMax on the float's bits: order-independent, so deterministic), and a second, GATED
// launch of the same kernel -- it leaves at once unless flag[0] is set: no host round trip, graph-capturable, the protocol of
// wc_apply_planes_f32's scale gate -- redoes the pass with that channel's scale lowered by the power of two that puts the true
// maximum into [2^14, 2^15) and rewrites scale[c], which every consumer reads from device memory.  The planes then hold the sum
// exactly as before (hi + lo carry 22 bits of every element whose lo is a normal fp16: >= 2^-3 after scaling, i.e. down to 2^-18 of
// the channel's maximum; below that the absolute error is 2^-25 of the scaled unit = 2^-40 of the channel's maximum, far below the
// channel's standard deviation >= max / sqrt(M)).  A non-finite element stays non-finite in the planes (NaN / Inf: loud downstream).
// fp32 is written too only where a reader without a planes path exists (x32 != NULL: the backward's K4 / K6 at C = 128 today).
#include "wc_common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

typedef float f32x2r __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pk_rne2r(float a, float b)
{
    const f32x2r v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));
}

constexpr float kResGuard = 60000.0f;

// The tensors these kernels write (planes, the fp32 sum) are streams far larger than the L2s that nobody reads before the launch ends: they leave
// as NON-TEMPORAL stores (round 6: the fused producer 86.5 -> 75.9 us per call at 128 x 32 x 32 x 256 -- a wave's loads and stores retire on one
// in-order counter, so a store that is acknowledged sooner also releases the rows requested behind it).  WC_RX_NT=0: plain stores, for A/B.
#ifndef WC_RX_NT
#define WC_RX_NT 1
#endif
template <typename V>
__device__ __forceinline__ void st_stream(V* p, V v)
{
    if (WC_RX_NT) __builtin_nontemporal_store(v, p); else *p = v;
}
typedef unsigned u32x2r __attribute__((ext_vector_type(2)));
typedef unsigned u32x4r __attribute__((ext_vector_type(4)));

struct ResAddArgs {
    const float* h; const float* s;        // h [N][H][W][C]; s [N][H >> up][W >> up][C] (nullable: out = h)
    int64_t M;                             // N * H * W
    int H, W, C, up;
    unsigned magHW, shHW, magW, shW;       // row / (H*W) and rem / W by multiply-shift (row < 2^31)
    float* center; float* scale;           // [C]
    _Float16* hi; _Float16* lo;            // planes (nullable)
    float* x32;                            // fp32 sum (nullable)
    int* flag;                             // [0]: a channel's sampled scale was too tight (pass 1), the gated pass 2 then runs
    unsigned* gmax;                        // [C] (flag + 64): bits of the largest |scaled element| beyond the guard, per channel (0: none)
    float* scale0;                         // [C] (flag + 64 + C): the sampled scales, kept while pass 2 rewrites scale[]
};

__device__ __forceinline__ int64_t src_row(const ResAddArgs& a, unsigned row)
{
    if (!a.up) return row;
    const unsigned HW = (unsigned)a.H * (unsigned)a.W;
    const unsigned n = __umulhi(row, a.magHW) >> a.shHW, rem = row - n * HW;
    const unsigned y = __umulhi(rem, a.magW) >> a.shW, x = rem - y * (unsigned)a.W;
    return ((int64_t)n * (a.H >> 1) + (y >> 1)) * (a.W >> 1) + (x >> 1);
}

// centre / scale of the sum from <= 256 sampled rows: subsample_mean_scale_kernel (wc_rows.hip) on h + up(s).  Same sample
// (wc_sample_row, wc_common.h), same statistics, same results as that kernel gives on the fp32 sum -- bit for bit: the sum of two
// floats is the float the fp32 tensor would hold.
// (16 channels per 256-thread workgroup, C / 16 workgroups: the kernel is a chain of latencies -- 32 loads per thread, three meetings --
// and 16 small workgroups on 16 CUs run it in 8-9 us where 4 workgroups of 1024 threads took 12.8)
constexpr int kSampCh = 16;
__global__ __launch_bounds__(256) void resadd_sample_kernel(ResAddArgs a)
{
    __shared__ float red[16][kSampCh];
    const int C = a.C;
    const int cl = threadIdx.x & (kSampCh - 1);
    const int c = blockIdx.x * kSampCh + cl;
    const int part = threadIdx.x / kSampCh;
    const int64_t nsamp = a.M < 256 ? a.M : 256;
    const int64_t stride = a.M / nsamp;
    if (blockIdx.x == 0 && threadIdx.x < 64) a.flag[threadIdx.x] = 0;
    float v[16];
    float sacc = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int64_t r = part + 16 * i;
        v[i] = 0.f;
        if (c < C && r < nsamp) {
            const int64_t row = wc_sample_row(r, stride);
            v[i] = a.h[row * C + c];
            if (a.s) v[i] += a.s[src_row(a, (unsigned)row) * C + c];
        }
        sacc += v[i];
    }
    red[part][cl] = sacc;
    __syncthreads();
    __shared__ float centre2[2][kSampCh];        // [0] mean, [1] median of the 16 group means (the outlier-proof centre)
    if (threadIdx.x < kSampCh) {
        float t = 0.f, pm[16];
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            const float ps = red[p][threadIdx.x];
            t += ps;
            const int cnt = (int)((nsamp - p + 15) / 16);
            pm[p] = cnt > 0 ? ps / (float)cnt : 0.f;
        }
        const float mean_ = t / (float)nsamp;
        centre2[0][threadIdx.x] = mean_;
        centre2[1][threadIdx.x] = nsamp >= 16 ? wc_median16(pm) : mean_;
    }
    __syncthreads();
    const float mean = centre2[0][cl], med = centre2[1][cl];
    float mx = 0.f, mx2 = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int64_t r = part + 16 * i;
        if (c < C && r < nsamp) { mx = fmaxf(mx, fabsf(v[i] - mean)); mx2 = fmaxf(mx2, fabsf(v[i] - med)); }
    }
    __shared__ float red2[16][kSampCh];
    __syncthreads();
    red[part][cl] = mx;
    red2[part][cl] = mx2;
    __syncthreads();
    if (threadIdx.x < kSampCh && c < C) {
        float g1[16], g2[16];
#pragma unroll
        for (int p = 0; p < 16; ++p) { g1[p] = red[p][threadIdx.x]; g2[p] = red2[p][threadIdx.x]; }
        float m = 0.f;
#pragma unroll
        for (int p = 0; p < 16; ++p) m = fmaxf(m, g1[p]);
        const float med1 = wc_median16(g1);
        const bool outlier = nsamp >= 16 && med1 > 0.f && m > 64.f * med1;
        float centre = mean;
        if (outlier) { centre = med; m = wc_robust_max16(g2); }
        float sc = 1.0f;
        if (m > 0.f && m < 3.0e38f) {
            int e;
            frexpf(m, &e);
            sc = ldexpf(1.0f, 4 - e);
        }
        a.center[c] = centre;
        a.scale[c] = sc;
        a.scale0[c] = sc;
        a.gmax[c] = 0u;
    }
}

// The pass: one thread = 8 consecutive channels of an output row (two 16-byte loads of h, two of s; one 16-byte store per plane).
// Grid and block are multiples of C / 8 threads, so a thread keeps its channels -- centre and scale stay in registers.
// REDO: the gated second launch (see the head of the file): leaves at once unless pass 1 raised flag[0]; otherwise the same pass with the
// saturated channels' scales lowered to what their true maxima ask for (thread-local arithmetic on scale0 / gmax, which nothing writes
// while this launch runs; the workgroups that own the first row also store the new scale[] for the consumers).
template <bool SPLIT, bool F32, bool REDO = false>
__global__ __launch_bounds__(256) void resadd_kernel(ResAddArgs a)
{
    static_assert(!REDO || SPLIT, "only the planes are redone");
    if (REDO && __builtin_nontemporal_load(a.flag) != 1) return;
    const int C = a.C, C8 = C >> 3;
    const int64_t n8 = a.M * C8;
    const int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x;
    // (row, channel group) of element i, advanced without a division per iteration.  SPLIT: C8 divides 256 (C = 128 | 256), so the
    // channel group never moves and centre / scale are loaded once
    const int64_t step = (int64_t)gridDim.x * 256, drow = step / C8;
    const int dcg = (int)(step - drow * C8);
    int64_t row = i0 / C8;
    int cg = (int)(i0 - row * C8);
    f32x4 s0 = {1.f, 1.f, 1.f, 1.f}, s1 = s0, c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0;
    if (SPLIT) {
        const int c = cg * 8;
        const float* sc = REDO ? a.scale0 : a.scale;
        s0 = *reinterpret_cast<const f32x4*>(sc + c); s1 = *reinterpret_cast<const f32x4*>(sc + c + 4);
        c0 = *reinterpret_cast<const f32x4*>(a.center + c); c1 = *reinterpret_cast<const f32x4*>(a.center + c + 4);
        if (REDO) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float gm = __builtin_bit_cast(float, a.gmax[c + j]);      // in units of the sampled scale; 0: the channel fitted
                if (gm > 0.f && gm < 3.0e38f) {
                    int e;
                    frexpf(gm, &e);                                              // gm = f 2^e, f in [0.5, 1): gm 2^(15 - e) in [2^14, 2^15)
                    const float k = ldexpf(1.0f, 15 - e);
                    if (j < 4) s0[j] *= k; else s1[j - 4] *= k;
                }
            }
            if (i0 < C8) {
                *reinterpret_cast<f32x4*>(a.scale + c) = s0;
                *reinterpret_cast<f32x4*>(a.scale + c + 4) = s1;
            }
        }
    }
    bool over = false;
    float ov[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};        // pass 1: the largest |scaled element| beyond the guard, per channel of this thread
    // two elements per trip: the loads of both (2 x 16 B of h, 2 x 16 B of s each) are in flight before the first is converted
    auto advance = [&](int64_t& r, int& g) { r += drow; g += dcg; if (g >= C8) { g -= C8; ++r; } };
    auto emit = [&](int64_t e, f32x4 v0, f32x4 v1) __attribute__((always_inline)) {
        if (F32) {
            st_stream(reinterpret_cast<f32x4*>(a.x32 + e), v0);
            st_stream(reinterpret_cast<f32x4*>(a.x32 + e + 4), v1);
        }
        if (SPLIT) {
            float g[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) { g[j] = (v0[j] - c0[j]) * s0[j]; g[4 + j] = (v1[j] - c1[j]) * s1[j]; }
            if (!REDO) {        // (pass 2 cannot meet one: its scales come from the true maxima; a non-finite element stays non-finite)
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (fabsf(g[j]) > kResGuard) { over = true; ov[j] = fmaxf(ov[j], fabsf(g[j])); g[j] = copysignf(kResGuard, g[j]); }
            }
            unsigned hw[4], lw[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                hw[j] = pk_rne2r(g[2 * j], g[2 * j + 1]);
                float r0, r1;        // remainder = g - float(hi) in one mixed-precision FMA per element
                asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hw[j]), "v"(g[2 * j]));
                asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hw[j]), "v"(g[2 * j + 1]));
                lw[j] = pk_rne2r(r0, r1);
            }
            st_stream(reinterpret_cast<u32x4r*>(a.hi + e), u32x4r{hw[0], hw[1], hw[2], hw[3]});
            st_stream(reinterpret_cast<u32x4r*>(a.lo + e), u32x4r{lw[0], lw[1], lw[2], lw[3]});
        }
    };
    for (int64_t i = i0; i < n8; i += 2 * step) {
        const int64_t ea = i * 8, eb = (i + step) * 8;
        const bool two = i + step < n8;
        int64_t rowb = row; int cgb = cg;
        advance(rowb, cgb);
        f32x4 a0 = *reinterpret_cast<const f32x4*>(a.h + ea), a1 = *reinterpret_cast<const f32x4*>(a.h + ea + 4);
        f32x4 b0 = a0, b1 = a1;
        if (two) { b0 = *reinterpret_cast<const f32x4*>(a.h + eb); b1 = *reinterpret_cast<const f32x4*>(a.h + eb + 4); }
        if (a.s) {
            const float* sa = a.s + src_row(a, (unsigned)row) * C + cg * 8;
            const float* sb = a.s + src_row(a, (unsigned)(two ? rowb : row)) * C + (two ? cgb : cg) * 8;
            const f32x4 p0 = *reinterpret_cast<const f32x4*>(sa), p1 = *reinterpret_cast<const f32x4*>(sa + 4);
            const f32x4 q0 = *reinterpret_cast<const f32x4*>(sb), q1 = *reinterpret_cast<const f32x4*>(sb + 4);
            a0 += p0; a1 += p1; b0 += q0; b1 += q1;
        }
        emit(ea, a0, a1);
        if (two) emit(eb, b0, b1);
        row = rowb; cg = cgb;
        advance(row, cg);
    }
    if (SPLIT && !REDO && over) {
        // the channel's true maximum for pass 2: non-negative floats order as their bit patterns, and a maximum does not depend on
        // the order of its operands -- deterministic without a reduction pass
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (ov[j] > 0.f) atomicMax(a.gmax + cg * 8 + j, __builtin_bit_cast(unsigned, ov[j]));
        *a.flag = 1;
    }
}


// ---------------------------------------------------------------------------------------------------------------------------------
// The producer that FEEDS K1 LITERALLY (round 5): the residual add's pass and the next site's covariance reduction
// as ONE kernel.  Round 4 wrote the planes (resadd_kernel: a pure stream with the matrix pipe idle) and xtx_split_kernel then read all
// of them back (134 MB at 128x32x32x256, 1.18 x over-fetched) to form the moments.  Here the rows are summed, centred, scaled and split
// ONCE, the hi | lo words leave for the planes in global memory and, byte-permuted into the transposed [channel][row] fragment image,
// for the LDS -- and the block triangle of X^T X runs on them in the same stage: the structure of xty_f16x3_kernel's covariance form
// (wc_fast_xty.hip: 64-row stages at C = 256, two stage buffers, fp32 chains of one stage flushed into float64 registers, the 36 blocks
// as 18 + 18 on two workgroups of a slab that stream the same rows -- the second reads h and s from L2 --, the diagonal on the VALU, the
// column sums) with the add in front of its conversion and the planes' stores behind it (each workgroup type stores one plane).
// The partials P / colsum / dfix go where wc_whiten_split_f16x2's tail expects them (wc_whiten_presummed_f16x2 runs that tail), so the
// site's K1 launch does not exist.  Saturation: as resadd_kernel -- the channel's true maximum is recorded, the gated second launch
// redoes planes AND partials with the lowered scales.
#ifndef WC_RX_STAGGER
#define WC_RX_STAGGER 1      // 0: every wave converts first (round 5)
#endif
#ifndef WC_RX_ABL
#define WC_RX_ABL 0      // development ablation bits (wrong results, times only): 1 no plane stores, 2 no shortcut rows, 4 no MFMAs, 8 no float64 flush, 16 the planes as 16-byte stores (misplaced)
#endif

struct ResXtxArgs {
    ResAddArgs r;                          // h, s, geometry, centre / scale / flag area, planes, x32
    int64_t N, HW;                         // segments (statistic groups) x rows per segment; N * HW = r.M
    int per_sample, nsplit;
    int64_t rows_per_slab;
    int nslab, ntypes;
    double* P; float* colsum; double* dfix;
    int sample_inside;                     // 1: every workgroup takes centre / scales from the <= 256 sampled rows itself (no resadd_sample_kernel launch)
    int* wgflag; float* wgmax;             // [grid], [grid][C]: per-workgroup "an element did not fit" and that workgroup's per-channel maxima (pass 1 -> gate)
};

// centre / scale of the sum inside the fused kernel: resadd_sample_kernel's statistics, bit for bit (the same rows, the same summation
// order per row group, the same medians), computed redundantly by every workgroup from the L2-resident sample (256 rows of h and of s:
// 512 KiB) instead of by a launch of its own in front (12-14 us on 16 small workgroups, a chain of latencies, plus a launch boundary).
// Leaves scale[C] | -centre scale[C] in LDS; the publishing workgroup also stores centre / scale / scale0 for the consumers.
template <int C>
__device__ __forceinline__ void rx_sample(const ResAddArgs& r, char* scratch, float* sc_sh, float* nc_sh, int tid, bool publish)
{
    constexpr int C4 = C / 4, RGRP = 512 / C4, NP = 16 / RGRP;          // row groups ("parts") of the sample per thread: 2 (C = 256) | 1 (C = 128)
    const int c4 = tid % C4, rgrp = tid / C4;
    float* red = reinterpret_cast<float*>(scratch);                      // [16][C]
    float* red2 = red + 16 * C;                                          // [16][C]
    float* cen2 = red2 + 16 * C;                                         // [2][C]: mean | median of the 16 group means
    const int64_t stride = r.M / 256;                                    // (M >= 20480: always 256 samples)
    f32x4 v[NP][16];
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        const int p = rgrp + RGRP * q;
        f32x4 sacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int64_t row = wc_sample_row(p + 16 * i, stride);
            v[q][i] = *reinterpret_cast<const f32x4*>(r.h + row * C + 4 * c4);
            v[q][i] += *reinterpret_cast<const f32x4*>(r.s + src_row(r, (unsigned)row) * C + 4 * c4);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) sacc += v[q][i];
        *reinterpret_cast<f32x4*>(red + p * C + 4 * c4) = sacc;
    }
    __syncthreads();
    if (tid < C) {
        float t = 0.f, pm[16];
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            const float ps = red[p * C + tid];
            t += ps;
            pm[p] = ps / 16.f;
        }
        cen2[tid] = t / 256.f;
        cen2[C + tid] = wc_median16(pm);
    }
    __syncthreads();
    {
        const f32x4 mean = *reinterpret_cast<const f32x4*>(cen2 + 4 * c4), med = *reinterpret_cast<const f32x4*>(cen2 + C + 4 * c4);
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            const int p = rgrp + RGRP * q;
            f32x4 mx = {0.f, 0.f, 0.f, 0.f}, mx2 = mx;
#pragma unroll
            for (int i = 0; i < 16; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    mx[j] = fmaxf(mx[j], fabsf(v[q][i][j] - mean[j]));
                    mx2[j] = fmaxf(mx2[j], fabsf(v[q][i][j] - med[j]));
                }
            *reinterpret_cast<f32x4*>(red + p * C + 4 * c4) = mx;
            *reinterpret_cast<f32x4*>(red2 + p * C + 4 * c4) = mx2;
        }
    }
    __syncthreads();
    if (tid < C) {
        float g1[16], g2[16];
#pragma unroll
        for (int p = 0; p < 16; ++p) { g1[p] = red[p * C + tid]; g2[p] = red2[p * C + tid]; }
        float m = 0.f;
#pragma unroll
        for (int p = 0; p < 16; ++p) m = fmaxf(m, g1[p]);
        const float med1 = wc_median16(g1);
        const bool outlier = med1 > 0.f && m > 64.f * med1;
        float centre = cen2[tid];
        if (outlier) { centre = cen2[C + tid]; m = wc_robust_max16(g2); }
        float sc = 1.0f;
        if (m > 0.f && m < 3.0e38f) {
            int e;
            frexpf(m, &e);
            sc = ldexpf(1.0f, 4 - e);
        }
        sc_sh[tid] = sc;
        nc_sh[tid] = -centre * sc;
        if (publish) { r.center[tid] = centre; r.scale[tid] = sc; r.scale0[tid] = sc; }
    }
    __syncthreads();
}

template <int C, bool F32, bool REDO>
__global__ __launch_bounds__(512, 1) void resadd_xtx_kernel(ResXtxArgs a)
{
    static_assert(C == 128 || C == 256, "fused producer: C = 128 or 256");
    // The gate (REDO): did any workgroup of pass 1 meet an element that did not fit?  Pass 1 leaves one word per workgroup (no word that
    // somebody would have had to clear in front of the launch: the sampling lives inside the kernel now), the gate folds them.
    __shared__ int any_over;
    if (REDO) {
        int mine = 0;
        for (int i = threadIdx.x; i < (int)gridDim.x; i += 512) mine |= __builtin_nontemporal_load(a.wgflag + i);
        if (threadIdx.x == 0) any_over = 0;
        __syncthreads();
        if (mine) any_over = 1;
        __syncthreads();
        if (blockIdx.x == 0 && threadIdx.x == 0) a.r.flag[0] = any_over;      // the status word of the call (informational)
        if (!any_over) return;
    }
    constexpr bool BAL = C == 256;                    // 36 blocks as 18 + 18 on the two workgroup types of a slab (wc_fast_xty.hip)
    constexpr int BW = 3;
    constexpr int C4 = C / 4;
    constexpr int RGRP = 512 / C4;
    constexpr int R = RGRP * 8;                       // rows per stage: 64 (C = 256) / 128 (C = 128)
    constexpr int CPR = R / 8;
    constexpr int KS = R / 16;
    constexpr int IMG = C * R * 2;
    constexpr int NB = C / 32;
    constexpr int NBLK = NB * (NB + 1) / 2;
    static_assert(2 * IMG == 65536, "the stage buffers are 64 KiB apart");
    // LDS: [2 stage buffers][hi | lo] (128 KiB) | per-thread accumulators that would not fit the registers beside the float64 blocks, the
    // prefetched stage and the shortcut's rows (column sums 16 B, the diagonal's float64 sums 32 B per thread: 24 KiB) | scale[C] | -centre scale[C]
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const aux = smem + 4 * IMG;
    f32x4* const cs_acc = reinterpret_cast<f32x4*>(aux);                         // [512]
    double* const sq_acc = reinterpret_cast<double*>(aux + 512 * 16);             // [512][4]
    float* const sc_sh = reinterpret_cast<float*>(aux + 512 * 48);                // [C]
    float* const nc_sh = sc_sh + C;                                               // [C]
    unsigned* const ov_sh = reinterpret_cast<unsigned*>(nc_sh + C);               // [C] pass 1: bits of the largest |scaled element| beyond the guard, per channel | [C]: "any"

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int type = q % a.ntypes;
    const int64_t z = (int64_t)(q / a.ntypes) * 8 + xcd;
    if (z >= a.nslab) {
        if (!REDO && threadIdx.x == 0) a.wgflag[blockIdx.x] = 0;
        return;
    }

    int64_t r0, r1;
    if (a.per_sample) {
        const int64_t n = z / a.nsplit, qq = z % a.nsplit;
        r0 = n * a.HW + qq * a.rows_per_slab;
        r1 = r0 + a.rows_per_slab;
        const int64_t end = (n + 1) * a.HW;
        if (r1 > end) r1 = end;
    } else {
        const int64_t M = a.N * a.HW;
        r0 = z * a.rows_per_slab;
        r1 = r0 + a.rows_per_slab;
        if (r1 > M) r1 = M;
    }
    const int nst = (int)((r1 - r0) / R);

    int ib[BW], jb[BW]; bool live[BW];
#pragma unroll
    for (int b = 0; b < BW; ++b) {
        int L = (type * 8 + wave) * BW + b;
        live[b] = L < NBLK;
        if (BAL) {
            L = type * (NBLK / 2) + (wave < 2 ? wave * 3 : 6 + (wave - 2) * 2) + b;
            live[b] = b < (wave < 2 ? 3 : 2);
        }
        if (!live[b]) L = 0;
        int i = 0; while (L >= NB - i) { L -= NB - i; ++i; }
        ib[b] = i; jb[b] = i + L;
    }
    bool all_ = true, any_ = false;
#pragma unroll
    for (int b = 0; b < BW; ++b) { all_ = all_ && live[b]; any_ = any_ || live[b]; }
    const bool all_live = __builtin_amdgcn_readfirstlane(all_ ? 1 : 0) != 0;
    const bool two_live = __builtin_amdgcn_readfirstlane((live[0] && live[1] && !live[BW - 1]) ? 1 : 0) != 0;
    const bool any_live = __builtin_amdgcn_readfirstlane(any_ ? 1 : 0) != 0;

    const int c4 = tid % C4, rgrp = tid / C4;
    if (tid < C) ov_sh[tid] = 0u;
    if (tid == 0) ov_sh[C] = 0u;
    if (!REDO && a.sample_inside) {
        rx_sample<C>(a.r, smem, sc_sh, nc_sh, tid, z == 0 && type == 0);
        cs_acc[tid] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) sq_acc[tid * 4 + j] = 0.0;
    } else {
        f32x4 scl = *reinterpret_cast<const f32x4*>((REDO ? a.r.scale0 : a.r.scale) + 4 * c4);
        if (REDO) {         // the scales the true maxima ask for (resadd_kernel's rule); the first slab's type-0 workgroup stores them
            f32x4 gm4 = {0.f, 0.f, 0.f, 0.f};       // the channel's maximum over the workgroups that reported one (rare path: a plain loop)
            for (int i = 0; i < (int)gridDim.x; ++i)
                if (a.wgflag[i]) {
                    const f32x4 w = *reinterpret_cast<const f32x4*>(a.wgmax + (int64_t)i * C + 4 * c4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) gm4[j] = fmaxf(gm4[j], w[j]);
                }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float gm = gm4[j];
                if (gm > 0.f && gm < 3.0e38f) {
                    int e;
                    frexpf(gm, &e);
                    scl[j] *= ldexpf(1.0f, 15 - e);
                }
            }
            if (z == 0 && type == 0 && rgrp == 0) *reinterpret_cast<f32x4*>(a.r.scale + 4 * c4) = scl;
        }
        if (rgrp == 0) {
            *reinterpret_cast<f32x4*>(sc_sh + 4 * c4) = scl;
            *reinterpret_cast<f32x4*>(nc_sh + 4 * c4) = -(*reinterpret_cast<const f32x4*>(a.r.center + 4 * c4)) * scl;
        }
        cs_acc[tid] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) sq_acc[tid * 4 + j] = 0.0;
    }
    __syncthreads();

    auto swz = [](int c) -> int {        // xty_f16x3_kernel's conflict-free chunk swizzles (write groups AND read groups)
        if (CPR == 8) return (((c >> 1) ^ (c >> 2)) & 1) | (((c >> 3) & 1) << 1) | (((c >> 4) & 1) << 2);
        return ((c ^ (c >> 2)) & 1) | (((c >> 3) & 1) << 1) | (((c >> 4) & 1) << 2) | (((c >> 1) & 1) << 3);
    };
    int st_off[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = 4 * c4 + j;
        st_off[j] = c * (R * 2) + ((rgrp ^ swz(c)) * 16);
    }

    // rows of a stage: this thread takes 8 consecutive rows (row0 a multiple of 8: one image row, W % 8 == 0) of its 4 channels; with
    // up = 1 they add 4 consecutive source pixels of s, each twice
    f32x4 xr[8], sr[4];
    constexpr bool has_s = true;                         // (the launcher refuses a call without a shortcut: every block of the generators has one)
    // (what the stage lambdas need of the arguments, as scalars: with the stage loop instantiated per order of its halves -- `run` below -- the
    //  lambdas are captured a second time, and an argument STRUCT reached through two captures is materialised in scratch memory)
    const float* const g_h = a.r.h; const float* const g_s = a.r.s; float* const g_x32 = a.r.x32;
    _Float16* const g_hi = a.r.hi; _Float16* const g_lo = a.r.lo;
    const int g_up = a.r.up, g_H = a.r.H, g_W = a.r.W;
    const unsigned g_magHW = a.r.magHW, g_shHW = a.r.shHW, g_magW = a.r.magW, g_shW = a.r.shW;
    auto src_row_l = [&](unsigned row) __attribute__((always_inline)) -> int64_t {        // src_row() on the scalars
        if (!g_up) return row;
        const unsigned HWp = (unsigned)g_H * (unsigned)g_W;
        const unsigned n = __umulhi(row, g_magHW) >> g_shHW, rem = row - n * HWp;
        const unsigned y = __umulhi(rem, g_magW) >> g_shW, x = rem - y * (unsigned)g_W;
        return ((int64_t)n * (g_H >> 1) + (y >> 1)) * (g_W >> 1) + (x >> 1);
    };
    auto stage_load = [&](int st) __attribute__((always_inline)) {
        const int64_t row0 = r0 + (int64_t)st * R + rgrp * 8;
        const float* base = g_h + row0 * C + 4 * c4;
#pragma unroll
        for (int p = 0; p < 8; ++p) xr[p] = *reinterpret_cast<const f32x4*>(base + p * C);
        if (has_s && !(WC_RX_ABL & 2)) {
            const float* sb = g_s + src_row_l((unsigned)row0) * C + 4 * c4;
#pragma unroll
            for (int p = 0; p < 4; ++p) sr[p] = *reinterpret_cast<const f32x4*>(sb + p * C);
        }
    };
    const bool want_csum = a.colsum != nullptr && type == 0;
    const bool want_dfix = a.dfix != nullptr && type == a.ntypes - 1;
    // which plane this workgroup stores: with two types of a slab each takes one (both hold every word); one type stores both
    const bool st_hi = a.ntypes == 1 || type == 0, st_lo = a.ntypes == 1 || type == 1;
    auto stage_write = [&](int buf, int st_of_data) __attribute__((always_inline)) {
        char* img = smem + buf * (2 * IMG);
        const int64_t row0 = r0 + (int64_t)st_of_data * R + rgrp * 8;
        const f32x4 scl = *reinterpret_cast<const f32x4*>(sc_sh + 4 * c4), ncs = *reinterpret_cast<const f32x4*>(nc_sh + 4 * c4);
        f32x4 g[8];
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            f32x4 v = xr[p];
            if (has_s && !(WC_RX_ABL & 2)) v += sr[p >> 1];
            if (F32 && st_hi) st_stream(reinterpret_cast<f32x4*>(g_x32 + (row0 + p) * C + 4 * c4), v);
            g[p] = v * scl + ncs;
        }
        if (!REDO) {        // the saturation test: the stage's maximum against the guard; the per-channel maxima only behind it (rare)
            float m = 0.f;
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                m = __builtin_fmaxf(__builtin_fmaxf(m, fabsf(g[p][0])), fabsf(g[p][1]));
                m = __builtin_fmaxf(__builtin_fmaxf(m, fabsf(g[p][2])), fabsf(g[p][3]));
            }
            if (!(m <= kResGuard)) {        // (a NaN does NOT raise the gate -- fmaxf drops NaN operands, m never is one -- and needs none: it goes into the planes as it is, loud)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float mj = 0.f;
#pragma unroll
                    for (int p = 0; p < 8; ++p) mj = __builtin_fmaxf(mj, fabsf(g[p][j]));
                    if (mj > kResGuard) atomicMax(ov_sh + 4 * c4 + j, __builtin_bit_cast(unsigned, mj));      // (LDS: order-independent, deterministic)
                }
                ov_sh[C] = 1u;
            }
        }
        if (want_csum) {
            f32x4 cs = cs_acc[tid];
#pragma unroll
            for (int p = 0; p < 8; ++p) cs += g[p];
            cs_acc[tid] = cs;
        }
        if (want_dfix) {
            f32x4 sq = g[0] * g[0];
#pragma unroll
            for (int p = 1; p < 8; ++p) sq += g[p] * g[p];
#pragma unroll
            for (int j = 0; j < 4; ++j) sq_acc[tid * 4 + j] += (double)sq[j];
        }
        // row-major words (what the planes hold: H[p][w] = channels (2w, 2w + 1) of row p, the remainders likewise) leave for global
        // memory, then go transposed into the fragment image -- channel j of rows (2 pp, 2 pp + 1) = half j & 1 of word j >> 1 of the two
        // rows: one v_perm_b32 per image word, one ds_write_b128 per channel and plane (8-byte halves met two-way in the banks: 24 % of
        // the LDS cycles, first version)
        unsigned H[8][2], Lw[8][2];
#pragma unroll
        for (int p = 0; p < 8; ++p)
#pragma unroll
            for (int w = 0; w < 2; ++w) {
                const float v0 = g[p][2 * w], v1 = g[p][2 * w + 1];
                H[p][w] = pk_rne2r(v0, v1);
                float q0, q1;
                asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(q0) : "v"(H[p][w]), "v"(v0));
                asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(q1) : "v"(H[p][w]), "v"(v1));
                Lw[p][w] = pk_rne2r(q0, q1);
            }
        if (WC_RX_ABL & 16) {        // timing only: the same bytes as four 16-byte stores per plane (rows p, p + 1 are 1 KiB contiguous; data misplaced)
            if (st_hi) {
#pragma unroll
                for (int p = 0; p < 8; p += 2) *reinterpret_cast<uint4*>(g_hi + (row0 + p) * C + 8 * c4) = make_uint4(H[p][0], H[p][1], H[p + 1][0], H[p + 1][1]);
            }
            if (st_lo) {
#pragma unroll
                for (int p = 0; p < 8; p += 2) *reinterpret_cast<uint4*>(g_lo + (row0 + p) * C + 8 * c4) = make_uint4(Lw[p][0], Lw[p][1], Lw[p + 1][0], Lw[p + 1][1]);
            }
        } else if (!(WC_RX_ABL & 1)) {
            if (st_hi) {
#pragma unroll
                for (int p = 0; p < 8; ++p) st_stream(reinterpret_cast<u32x2r*>(g_hi + (row0 + p) * C + 4 * c4), u32x2r{H[p][0], H[p][1]});
            }
            if (st_lo) {
#pragma unroll
                for (int p = 0; p < 8; ++p) st_stream(reinterpret_cast<u32x2r*>(g_lo + (row0 + p) * C + 4 * c4), u32x2r{Lw[p][0], Lw[p][1]});
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned sel = (j & 1) ? 0x07060302u : 0x05040100u;
            unsigned hw[4], lw[4];
#pragma unroll
            for (int pp = 0; pp < 4; ++pp) {
                hw[pp] = __builtin_amdgcn_perm(H[2 * pp + 1][j >> 1], H[2 * pp][j >> 1], sel);
                lw[pp] = __builtin_amdgcn_perm(Lw[2 * pp + 1][j >> 1], Lw[2 * pp][j >> 1], sel);
            }
            *reinterpret_cast<uint4*>(img + st_off[j]) = make_uint4(hw[0], hw[1], hw[2], hw[3]);
            *reinterpret_cast<uint4*>(img + st_off[j] + IMG) = make_uint4(lw[0], lw[1], lw[2], lw[3]);
        }
    };

    int a_base[BW], b_base[BW];
#pragma unroll
    for (int b = 0; b < BW; ++b) {
        const int ca = ib[b] * 32 + l31, cb = jb[b] * 32 + l31;
        a_base[b] = ca * (R * 2) + ((lh ^ swz(ca)) << 4);
        b_base[b] = cb * (R * 2) + ((lh ^ swz(cb)) << 4);
    }

    if (nst > 0) {
        stage_load(0);
        stage_write(0, 0);
        if (nst > 1) stage_load(1);
    }
    __syncthreads();
    double* P = a.P + z * (int64_t)C * C;
    // The stage loop, once per ORDER of its two halves (round 6; MI355X_MICROARCH.md: "two waves that run the same program with one barrier per
    // block: try a stagger").  Waves 0-3 convert the next stage and then multiply the current one (MF = false: the loop as it was); their SIMD
    // partners 4-7 multiply FIRST and convert afterwards (MF = true) -- both orders are legal inside a stage, the conversion writes the other
    // buffer -- so a SIMD's vector pipe and its matrix pipe are wanted by different waves at the same time.  The kernel stands at 256 registers
    // with three blocks per wave; waves 4-7 never have more than NBK = 2 (C = 256: the balanced split) or any at all (C = 128), and their copy of
    // the loop is compiled for that many: that is where the registers for holding a stage's rows across the matrix half come from.
    auto run = [&](auto MF_, auto NBK_) __attribute__((always_inline)) {
        constexpr bool MF = decltype(MF_)::value;
        constexpr int NBK = decltype(NBK_)::value;              // block slots of this copy
        double acc64[NBK > 0 ? NBK : 1][16];
#pragma unroll
        for (int b = 0; b < NBK; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc64[b][r] = 0.0;
        for (int st = 0; st < nst; ++st) {
            const int cur = st & 1;
            if (!MF) {
                if (st + 1 < nst) stage_write(cur ^ 1, st + 1);
                if (st + 2 < nst) stage_load(st + 2);
            }
            if constexpr (NBK > 0) {
                f32x16 acc[NBK];
#pragma unroll
                for (int b = 0; b < NBK; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
                const int kbuf = cur << 16;
                auto frag = [&](int base, int ks, int lo) __attribute__((always_inline)) {
                    return *reinterpret_cast<const f16x8*>(smem + (base ^ ((ks << 5) | kbuf)) + lo * IMG);
                };
                auto products = [&](auto ALL_, auto NL_) __attribute__((always_inline)) {
                    constexpr bool ALL = decltype(ALL_)::value;
                    constexpr int NL = decltype(NL_)::value;
#pragma unroll 4
                    for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
                        for (int b = 0; b < NL; ++b) {
                            if (!ALL && !live[b]) continue;
                            const f16x8 ah = frag(a_base[b], ks, 0), al = frag(a_base[b], ks, 1);
                            const f16x8 bh = frag(b_base[b], ks, 0), bl = frag(b_base[b], ks, 1);
                            acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[b], 0, 0, 0);
                            acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[b], 0, 0, 0);
                            acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[b], 0, 0, 0);
                        }
                    }
                };
                const bool all_here = NBK == BW ? all_live : two_live;          // every slot of this copy live (a scalar per wave)
                if (WC_RX_ABL & 4) {}
                else if (all_here) products(std::true_type{}, std::integral_constant<int, NBK>{});
                else if (NBK == BW && two_live) products(std::true_type{}, std::integral_constant<int, NBK < 2 ? NBK : 2>{});
                else if (any_live) products(std::false_type{}, std::integral_constant<int, NBK>{});
                if (WC_RX_ABL & 8) {}
                else if (NBK == BW && two_live) {
#pragma unroll
                    for (int b = 0; b < (NBK < 2 ? NBK : 2); ++b)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc64[b][r] += (double)acc[b][r];
                } else if (any_live) {
#pragma unroll
                    for (int b = 0; b < NBK; ++b)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc64[b][r] += (double)acc[b][r];
                }
            }
            if (MF) {
                if (st + 1 < nst) stage_write(cur ^ 1, st + 1);
                if (st + 2 < nst) stage_load(st + 2);
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
#pragma unroll
        for (int b = 0; b < NBK; ++b) {
            if (!live[b]) continue;
            const int j = jb[b] * 32 + l31;
            const double isj = 1.0 / (double)sc_sh[j];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = ib[b] * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                P[(int64_t)i * C + j] = acc64[b][r] * isj / (double)sc_sh[i];
            }
        }
    };
    constexpr int NBK_B = BAL ? 2 : 0;            // waves 4-7: two blocks each (C = 256, balanced) | none ((type * 8 + wave) * 3 >= 10 at C = 128)
    static_assert(BAL || (4 * BW >= NBLK), "C = 128: waves 4-7 own no block");
    if (WC_RX_STAGGER && wave >= 4) run(std::true_type{}, std::integral_constant<int, NBK_B>{});
    else run(std::false_type{}, std::integral_constant<int, BW>{});
    // column sums / the diagonal: the row groups' per-thread sums are already in LDS, thread (rgrp, c4) at slot tid = rgrp * C4 + c4
    __syncthreads();
    if (want_csum) {
        const float* red = reinterpret_cast<const float*>(cs_acc);
        for (int c = tid; c < C; c += 512) {
            float t = 0.f;
            for (int g = 0; g < RGRP; ++g) t += red[(g * C4 + (c >> 2)) * 4 + (c & 3)];
            a.colsum[z * C + c] = t / sc_sh[c];
        }
    }
    if (want_dfix) {
        for (int c = tid; c < C; c += 512) {
            double t = 0.0;
            for (int g = 0; g < RGRP; ++g) t += sq_acc[(g * C4 + (c >> 2)) * 4 + (c & 3)];
            a.dfix[z * C + c] = t / ((double)sc_sh[c] * (double)sc_sh[c]);
        }
    }
    if (!REDO) {        // this workgroup's word for the gate, and -- only when it met such an element -- its per-channel maxima
        const bool over = ov_sh[C] != 0u;          // (behind the __syncthreads() above: every wave's last stage_write has happened)
        if (tid == 0) a.wgflag[blockIdx.x] = over ? 1 : 0;
        if (over && tid < C) a.wgmax[(int64_t)blockIdx.x * C + tid] = __builtin_bit_cast(float, ov_sh[tid]);
    }
}

// gradient of the add with respect to the pre-upsample shortcut: every source pixel collects its 2x2 output patch
__global__ __launch_bounds__(256) void patch_sum_kernel(const float* __restrict__ g, int64_t n4, int Hs, int Ws, int C4, float* __restrict__ out)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const int c4 = (int)(i % C4);
        const int64_t p = i / C4;                    // source pixel (n, y, x)
        const int x = (int)(p % Ws);
        const int64_t q = p / Ws;
        const int y = (int)(q % Hs);
        const int64_t n = q / Hs;
        const f32x4* g4 = reinterpret_cast<const f32x4*>(g);
        const int64_t W2 = 2 * (int64_t)Ws;
        const int64_t r00 = ((n * 2 * Hs + 2 * y) * W2 + 2 * x) * C4 + c4;
        const f32x4 t = (g4[r00] + g4[r00 + C4]) + (g4[r00 + W2 * C4] + g4[r00 + W2 * C4 + C4]);
        reinterpret_cast<f32x4*>(out)[i] = t;
    }
}

// The shortcut convolution on the producer's planes: y = sum_c x[c] w[o][c] + b[o] with x[c] = center[c] + g[c] / scale[c]
//     = sum_c g[c] (w[o][c] / scale[c]) + (b[o] + sum_c center[c] w[o][c])
// wf[o][c] = w[o][c] / scale[c] (exact: powers of two), bf[o] = b[o] + <center, w[o]> (float64 sum, fixed order).
// One workgroup per output channel; the weight is a 1x1 kernel (Cout, Cin) in any dense layout with the two strides given.
__global__ __launch_bounds__(256) void fold_channel_scale_kernel(const float* __restrict__ w, int64_t so, int64_t sc, int Cin,
                                                                 const float* __restrict__ bias, const float* __restrict__ scale,
                                                                 const float* __restrict__ center, float* __restrict__ wf,
                                                                 float* __restrict__ bf)
{
    __shared__ double red[256];
    const int o = blockIdx.x;
    double acc = 0.0;
    for (int c = threadIdx.x; c < Cin; c += 256) {
        const float v = w[o * so + c * sc];
        wf[o * so + c * sc] = v / scale[c];
        acc += (double)center[c] * (double)v;
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) bf[o] = (float)(red[0] + (bias ? (double)bias[o] : 0.0));
}

// ... and its weight gradient back: D[o][c] = sum_p gy[p][o] g[p][c] (the weight-gradient kernel on the planes),
// dW[o][c] = sum_p gy[p][o] x[p][c] = D[o][c] / scale[c] + center[c] db[o],  db[o] = sum_p gy[p][o]
__global__ __launch_bounds__(256) void unfold_channel_scale_kernel(const float* __restrict__ D, const float* __restrict__ db, int64_t so,
                                                                   int64_t sc, int Cin, const float* __restrict__ scale,
                                                                   const float* __restrict__ center, float* __restrict__ dW)
{
    const int o = blockIdx.x;
    const float dbo = db[o];
    for (int c = threadIdx.x; c < Cin; c += 256) dW[o * so + c * sc] = fmaf(center[c], dbo, D[o * so + c * sc] / scale[c]);
}

void magic(unsigned d, unsigned* mag, unsigned* sh)
{
    // floor(m / d) = umulhi(m, mag) >> sh for m < 2^31, d >= 2 (the multiply-shift of wc_conv.hip)
    unsigned s = 0;
    while ((1u << s) < d) ++s;
    const unsigned long long num = 1ull << (31 + s);
    *mag = (unsigned)((num + d - 1) / d);
    *sh = s - 1;
}

}  // namespace

hipError_t wc_launch_resadd(const float* h, const float* s, int64_t N, int64_t H, int64_t W, int C, int up,
                            void* xs, float* center, float* scale, int* flag, float* x32, hipStream_t st)
{
    ResAddArgs a = {};
    a.h = h; a.s = s; a.M = N * H * W; a.H = (int)H; a.W = (int)W; a.C = C; a.up = up;
    if (up) {                     // (H, W even and >= 2 there: checked by the ABI)
        magic((unsigned)(H * W), &a.magHW, &a.shHW);
        magic((unsigned)W, &a.magW, &a.shW);
    }
    a.center = center; a.scale = scale; a.flag = flag; a.x32 = x32;
    a.gmax = flag ? reinterpret_cast<unsigned*>(flag) + 64 : nullptr;
    a.scale0 = flag ? reinterpret_cast<float*>(flag) + 64 + C : nullptr;
    a.hi = static_cast<_Float16*>(xs); a.lo = a.hi ? a.hi + a.M * C : nullptr;
    const int64_t n8 = a.M * C / 8;
    int64_t blocks = (n8 + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (xs) {
        hipLaunchKernelGGL(resadd_sample_kernel, dim3((C + kSampCh - 1) / kSampCh), dim3(256), 0, st, a);
        if (x32) hipLaunchKernelGGL((resadd_kernel<true, true>), dim3((unsigned)blocks), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((resadd_kernel<true, false>), dim3((unsigned)blocks), dim3(256), 0, st, a);
        // the gate (x32, where asked for, is already exact).  A grid of its own size: it has to be a multiple of C / 8 threads
        // (256 is) and the fewer workgroups only to find the flag clear, the cheaper the launch that does nothing
        const int64_t rblocks = blocks < 1024 ? blocks : 1024;
        hipLaunchKernelGGL((resadd_kernel<true, false, true>), dim3((unsigned)rblocks), dim3(256), 0, st, a);
    } else {
        hipLaunchKernelGGL((resadd_kernel<false, true>), dim3((unsigned)blocks), dim3(256), 0, st, a);
    }
    return hipGetLastError();
}


// The fused producer (resadd_xtx_kernel): sample, the pass with the covariance's partials, its gate.  The slab plan is the fp32-input
// covariance kernel's (wc_fast_xty_plan with two = 0), the partials' layout what stats_colsum / stats_xtx_prepare read.
bool wc_resadd_xtx_supported(int64_t N, int64_t H, int64_t W, int C, int up, int groups)
{
    if (!(C == 128 || C == 256) || !up || groups <= 0 || (N % groups) != 0 || (W % 8) != 0 || (H % 2) != 0) return false;
    if (N * H * W < 256) return false;      // (rx_sample strides over M / 256 rows: ADVICE r5)
    int nsplit, ntypes; int64_t rps;
    return wc_fast_xty_plan(groups, (N / groups) * H * W, C, groups > 1, 0, &nsplit, &rps, &ntypes) > 0;
}

int wc_resadd_xtx_grid(int nslab, int ntypes) { return ((nslab + 7) / 8) * ntypes * 8; }

hipError_t wc_launch_resadd_xtx(const float* h, const float* s, int64_t N, int64_t H, int64_t W, int C, int up, int groups,
                                void* xs, float* center, float* scale, int* flag, float* x32,
                                int nsplit, int64_t rows_per_slab, int nslab, int ntypes, double* P, float* colsum, double* dfix,
                                int* wgflag, float* wgmax, hipStream_t st)
{
    if (!s || !up) return hipErrorInvalidValue;
    ResXtxArgs a = {};
    ResAddArgs& r = a.r;
    r.h = h; r.s = s; r.M = N * H * W; r.H = (int)H; r.W = (int)W; r.C = C; r.up = up;
    if (up) {
        magic((unsigned)(H * W), &r.magHW, &r.shHW);
        magic((unsigned)W, &r.magW, &r.shW);
    }
    r.center = center; r.scale = scale; r.flag = flag; r.x32 = x32;
    r.gmax = reinterpret_cast<unsigned*>(flag) + 64;
    r.scale0 = reinterpret_cast<float*>(flag) + 64 + C;
    r.hi = static_cast<_Float16*>(xs); r.lo = r.hi + r.M * C;
    a.N = groups; a.HW = r.M / groups; a.per_sample = groups > 1; a.nsplit = nsplit; a.rows_per_slab = rows_per_slab;
    a.nslab = nslab; a.ntypes = ntypes; a.P = P; a.colsum = colsum; a.dfix = dfix;
    a.wgflag = wgflag; a.wgmax = wgmax;
    // WC_RX_SAMPLE_KERNEL=1 (development, A/B): centre / scales from a launch of resadd_sample_kernel in front, as wc_resadd_split_f32 has it
    static const bool sample_launch = getenv("WC_RX_SAMPLE_KERNEL") && atoi(getenv("WC_RX_SAMPLE_KERNEL")) != 0;
    a.sample_inside = sample_launch ? 0 : 1;
    if (sample_launch) hipLaunchKernelGGL(resadd_sample_kernel, dim3((C + kSampCh - 1) / kSampCh), dim3(256), 0, st, r);
    const size_t lds = 131072 + 512 * 48 + 2 * (size_t)C * 4 + ((size_t)C + 4) * 4;
    const int grid = wc_resadd_xtx_grid(nslab, ntypes);
#define WC_LAUNCH_RX(C_, F_, R_)                                                                                                   \
    do {                                                                                                                           \
        static bool attr_set = false;                                                                                              \
        if (!attr_set) {                                                                                                           \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(resadd_xtx_kernel<C_, F_, R_>),                       \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                              \
            if (e != hipSuccess) return e;                                                                                         \
            attr_set = true;                                                                                                       \
        }                                                                                                                          \
        hipLaunchKernelGGL((resadd_xtx_kernel<C_, F_, R_>), dim3(grid), dim3(512), lds, st, a);                                    \
    } while (0)
    if (C == 256) {
        if (x32) WC_LAUNCH_RX(256, true, false); else WC_LAUNCH_RX(256, false, false);
        if (x32) WC_LAUNCH_RX(256, true, true); else WC_LAUNCH_RX(256, false, true);
    } else {
        if (x32) WC_LAUNCH_RX(128, true, false); else WC_LAUNCH_RX(128, false, false);
        if (x32) WC_LAUNCH_RX(128, true, true); else WC_LAUNCH_RX(128, false, true);
    }
#undef WC_LAUNCH_RX
    return hipGetLastError();
}

hipError_t wc_launch_patch_sum(const float* g, int64_t N, int64_t Hs, int64_t Ws, int C, float* out, hipStream_t st)
{
    const int64_t n4 = N * Hs * Ws * C / 4;
    int64_t blocks = (n4 + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(patch_sum_kernel, dim3((unsigned)blocks), dim3(256), 0, st, g, n4, (int)Hs, (int)Ws, C / 4, out);
    return hipGetLastError();
}

hipError_t wc_launch_fold_channel_scale(const float* w, int64_t so, int64_t sc, int Cout, int Cin, const float* bias,
                                        const float* scale, const float* center, float* wf, float* bf, hipStream_t st)
{
    hipLaunchKernelGGL(fold_channel_scale_kernel, dim3(Cout), dim3(256), 0, st, w, so, sc, Cin, bias, scale, center, wf, bf);
    return hipGetLastError();
}

hipError_t wc_launch_unfold_channel_scale(const float* D, const float* db, int64_t so, int64_t sc, int Cout, int Cin,
                                          const float* scale, const float* center, float* dW, hipStream_t st)
{
    hipLaunchKernelGGL(unfold_channel_scale_kernel, dim3(Cout), dim3(256), 0, st, D, db, so, sc, Cin, scale, center, dW);
    return hipGetLastError();
}
