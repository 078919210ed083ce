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

namespace {

typedef float f32x2r __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pk_rne2r(float a, float b)
{
    const f32x2r v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));
}

constexpr float kResGuard = 60000.0f;

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
            *reinterpret_cast<f32x4*>(a.x32 + e) = v0;
            *reinterpret_cast<f32x4*>(a.x32 + e + 4) = v1;
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
            *reinterpret_cast<uint4*>(a.hi + e) = make_uint4(hw[0], hw[1], hw[2], hw[3]);
            *reinterpret_cast<uint4*>(a.lo + e) = make_uint4(lw[0], lw[1], lw[2], lw[3]);
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
