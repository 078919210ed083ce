// Convolutions around the WC sites (the callers either side of the path -- the 3x3 'same'
// convolutions of generator.py:142-158 / discriminator.py:41-54, their up-/down-sampling forms and the matching
// data gradients), fp32-accurate on the 16-bit MFMA pipe: the same split-operand scheme as the WC apply kernel,
//     x = (xh + xl) / sx,  w = (wh + wl) / sw   (fp16 pairs, power-of-two tensor scales)
//     y = (xl*wh + xh*wl + xh*wh) / (sx*sw)     three v_mfma_f32_32x32x16_f16 into ONE fp32 accumulator
// as an implicit GEMM over (tap, input channel):
//   * M = the points of a "virtual grid" (N, H, W); point (y, x) reads input pixel (y*in_stride + dy_t, x*in_stride + dx_t)
//     for tap t and writes output pixel (y*out_stride + off_y, x*out_stride + off_x).  With up to four "phases" (tap
//     offsets, weights and output offset per phase) one launch covers a 3x3 / 1x1 convolution, a 4x4 stride-2
//     convolution, and a 4x4 stride-2 TRANSPOSED convolution (as four 2x2 sub-pixel convolutions) -- and, with the
//     roles of the channel axes swapped in the weight image, the data gradient of each.
//   * A operand: the activation planes are split once (conv_split_kernel); a workgroup GATHERS its pixels with LDS-DMA
//     (global_load_lds_dwordx4, 16 B per lane from any address) so that each 1-KiB chunk lands in LDS as the register
//     image of one (32 pixels x 16 channels) MFMA fragment: the k-loop reads it back with one conflict-free
//     ds_read_b128 per fragment; zero padding = lanes pointed at a zero line.
//   * B operand: the weights are pre-arranged (conv_weights_kernel) as the same kind of 1-KiB fragment images in
//     (phase, tap, 32-channel chunk, n-block, k-step, hi|lo) order: a plain linear LDS-DMA copy.
//   * a workgroup = 4 waves as 2x2, each wave (32 MB) pixels x (32 NB) outputs with 16 MB NB accumulator registers; one
//     iteration = (tap, 32 input channels) = 2 k-steps; two LDS stages (the DMA of iteration i+1 runs under the MFMAs
//     of iteration i; one barrier per iteration).
// MFMA-bound by design: 3 * 2*M*Cout*K flop on the fp16 pipe against 2*M*Cout*K on the fp32 pipe (157 TFLOP/s peak).
#include "wc_common.h"
#include "../../include/wc_hip.h"

namespace {

constexpr int kMaxTaps = 16, kMaxPhase = 4;

struct ConvArgs {
    const _Float16* xhi; const _Float16* xlo;      // [N][Hin][Win][Cin] each
    const _Float16* zero;                           // >= 64 B of zeros (the padding line)
    const char* wimg;                               // weight fragment images
    const float* xscale; const float* wscale;       // device scalars (powers of two)
    const float* bias;                              // [Cout] or nullptr
    float* y;                                       // [N][Hout][Wout][Cout]
    int N, H, W, Hin, Win, Cin, Cout, in_stride, ntaps, nphase, Hout, Wout, out_stride, relu;
    signed char dy[kMaxPhase][kMaxTaps], dx[kMaxPhase][kMaxTaps];
    signed char offy[kMaxPhase], offx[kMaxPhase];
};

__device__ __forceinline__ void lds_dma16(const void* g, unsigned lds)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(lds) : "memory");
}

template <int MB, int NB>
__global__ __launch_bounds__(256) void conv_f16x3_kernel(ConvArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TM = 64 * MB, TN = 64 * NB;
    constexpr int A_BYTES = 2 * MB * 4 * 1024, B_BYTES = 2 * NB * 4 * 1024, STAGE = A_BYTES + B_BYTES;
    constexpr int AQ = (2 * MB) / 4;               // m-blocks each wave stages
    static_assert(AQ >= 1, "MB >= 2");

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int ntn = a.Cout / TN;
    const int phase = blockIdx.y / ntn, nt = blockIdx.y - phase * ntn;
    const unsigned m0 = blockIdx.x * TM;
    const unsigned HW = a.H * a.W;
    const int koff = (lane >> 5) * 8;

    int gy[AQ], gx[AQ];
    unsigned gpix[AQ];                             // input pixel index of (n, 0, 0)
    #pragma unroll
    for (int q = 0; q < AQ; ++q) {
        const unsigned m = m0 + (wave * AQ + q) * 32 + (lane & 31);
        const unsigned n = m / HW, rem = m - n * HW;
        const unsigned yy = rem / a.W, xx = rem - yy * a.W;
        gy[q] = yy * a.in_stride; gx[q] = xx * a.in_stride; gpix[q] = n * (a.Hin * a.Win);
    }
    const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) char*)smem);
    const int nchunk = a.Cin >> 5;
    const int iters = a.ntaps * nchunk;
    const int nblk_all = a.Cout >> 5;

    auto issue = [&](int it, int stage) {
        const int tap = it / nchunk, ch = it - tap * nchunk;
        const int dy = a.dy[phase][tap], dx = a.dx[phase][tap];
        const unsigned sbase = __builtin_amdgcn_readfirstlane(lds0 + stage * STAGE);
        #pragma unroll
        for (int q = 0; q < AQ; ++q) {
            const int iy = gy[q] + dy, ix = gx[q] + dx;
            const bool ok = (unsigned)iy < (unsigned)a.Hin && (unsigned)ix < (unsigned)a.Win;
            const int64_t e = ((int64_t)(gpix[q] + iy * a.Win + ix)) * a.Cin + ch * 32 + koff;
            #pragma unroll
            for (int s = 0; s < 2; ++s) {
                const _Float16* ph = ok ? a.xhi + e + s * 16 : a.zero + koff;
                const _Float16* pl = ok ? a.xlo + e + s * 16 : a.zero + koff;
                const unsigned l = sbase + (((wave * AQ + q) * 2 + s) * 2) * 1024;
                lds_dma16(ph, l);
                lds_dma16(pl, l + 1024);
            }
        }
        const char* wsrc = a.wimg + ((((int64_t)(phase * a.ntaps + tap) * nchunk + ch) * nblk_all + nt * 2 * NB) << 12);
        #pragma unroll
        for (int c = 0; c < 2 * NB; ++c) {
            const int idx = wave * 2 * NB + c;
            lds_dma16(wsrc + idx * 1024 + lane * 16, sbase + A_BYTES + idx * 1024);
        }
    };

    f32x16 acc[MB][NB];
    #pragma unroll
    for (int i = 0; i < MB; ++i)
        #pragma unroll
        for (int j = 0; j < NB; ++j)
            #pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    issue(0, 0);
    for (int it = 0; it < iters; ++it) {
        const int stage = it & 1;
        __builtin_amdgcn_s_waitcnt(0x0F70);        // vmcnt(0): this wave's chunks of the stage have landed
        __syncthreads();                           // ... and everybody's; the other stage is free (its readers are done)
        if (it + 1 < iters) issue(it + 1, stage ^ 1);
        const char* sa = smem + stage * STAGE;
        const char* sb = sa + A_BYTES;
        #pragma unroll
        for (int s = 0; s < 2; ++s) {
            f16x8 ah[MB], al[MB], bh[NB], bl[NB];
            #pragma unroll
            for (int i = 0; i < MB; ++i) {
                const char* p = sa + (((wm * MB + i) * 2 + s) * 2) * 1024 + lane * 16;
                ah[i] = *reinterpret_cast<const f16x8*>(p);
                al[i] = *reinterpret_cast<const f16x8*>(p + 1024);
            }
            #pragma unroll
            for (int j = 0; j < NB; ++j) {
                const char* p = sb + (((wn * NB + j) * 2 + s) * 2) * 1024 + lane * 16;
                bh[j] = *reinterpret_cast<const f16x8*>(p);
                bl[j] = *reinterpret_cast<const f16x8*>(p + 1024);
            }
            #pragma unroll
            for (int i = 0; i < MB; ++i)
                #pragma unroll
                for (int j = 0; j < NB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
            #pragma unroll
            for (int i = 0; i < MB; ++i)
                #pragma unroll
                for (int j = 0; j < NB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
            #pragma unroll
            for (int i = 0; i < MB; ++i)
                #pragma unroll
                for (int j = 0; j < NB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        }
    }

    // epilogue: unscale, bias, scatter rows to their output pixels
    const float inv = 1.0f / (a.xscale[0] * a.wscale[0]);
    const int oy0 = a.offy[phase], ox0 = a.offx[phase];
    float bj[NB];
    #pragma unroll
    for (int j = 0; j < NB; ++j) bj[j] = a.bias ? a.bias[nt * TN + (wn * NB + j) * 32 + (lane & 31)] : 0.f;
    #pragma unroll
    for (int i = 0; i < MB; ++i) {
        #pragma unroll
        for (int r = 0; r < 16; ++r) {
            const unsigned row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            const unsigned m = m0 + (wm * MB + i) * 32 + row;
            const unsigned n = m / HW, rem = m - n * HW;
            const unsigned yy = rem / a.W, xx = rem - yy * a.W;
            const int64_t opix = ((int64_t)n * a.Hout + (yy * a.out_stride + oy0)) * a.Wout + (xx * a.out_stride + ox0);
            float* o = a.y + opix * a.Cout + nt * TN + wn * NB * 32 + (lane & 31);
            #pragma unroll
            for (int j = 0; j < NB; ++j) {
                float v = acc[i][j][r] * inv + bj[j];
                if (a.relu) v = fmaxf(v, 0.f);
                o[j * 32] = v;
            }
        }
    }
}

// ---- max |x|: one partial per workgroup (no atomics: deterministic, nothing to clear); the consumers fold the partials ---
constexpr int kAmaxBlocks = 512;

__global__ __launch_bounds__(256) void conv_absmax_kernel(const float* __restrict__ x, int64_t n4, int64_t n, float* __restrict__ partial)
{
    __shared__ float red[4];
    float m = 0.f;
    for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + 4 * i);
        m = fmaxf(fmaxf(m, fabsf(v[0])), fmaxf(fabsf(v[1]), fmaxf(fabsf(v[2]), fabsf(v[3]))));
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) for (int64_t i = 4 * n4; i < n; ++i) m = fmaxf(m, fabsf(x[i]));
    #pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// power-of-two scale that puts max|x| into [2^13, 2^14); every thread of a workgroup folds the partials (L2 hits)
__device__ __forceinline__ float scale_of(const float* __restrict__ partial)
{
    float amax = 0.f;
    for (int i = threadIdx.x & 63; i < kAmaxBlocks; i += 64) amax = fmaxf(amax, partial[i]);
    #pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
    if (!(amax > 0.f) || !(amax < 3.0e38f)) return 1.0f;
    int e;
    (void)frexpf(amax, &e);                        // amax = f * 2^e, f in [0.5, 1)
    return ldexpf(1.0f, 14 - e);
}

// ---- activation split: hi = fp16(x*s), lo = fp16(x*s - hi), optional ReLU first ------------------------------------
__global__ __launch_bounds__(256) void conv_split_kernel(const float* __restrict__ x, int64_t n4, const float* __restrict__ amax,
                                                         int relu, _Float16* __restrict__ hi, _Float16* __restrict__ lo,
                                                         float* __restrict__ scale_out)
{
    const float s = scale_of(amax);
    if (blockIdx.x == 0 && threadIdx.x == 0) scale_out[0] = s;
    for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        f32x4 v = *reinterpret_cast<const f32x4*>(x + 4 * i);
        if (relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
        v = v * s;
        f16x4 h, l;
        #pragma unroll
        for (int j = 0; j < 4; ++j) { h[j] = (_Float16)v[j]; l[j] = (_Float16)(v[j] - (float)h[j]); }
        *reinterpret_cast<f16x4*>(hi + 4 * i) = h;
        *reinterpret_cast<f16x4*>(lo + 4 * i) = l;
    }
}

// ---- weight fragment images ----------------------------------------------------------------------------------------
struct WeightArgs {
    const float* w; int64_t sk, sn, sr, ss;        // element (k, n, r, s) of the source = w[k*sk + n*sn + r*sr + s*ss]
    const float* amax; float* scale_out; char* img;
    int K, Nn, ntaps, nphase;                      // K = reduction channels, Nn = output channels of the product
    signed char r[kMaxPhase][kMaxTaps], s[kMaxPhase][kMaxTaps];
};

__global__ __launch_bounds__(256) void conv_weights_kernel(WeightArgs a)
{
    const float sc = scale_of(a.amax);
    if (blockIdx.x == 0 && threadIdx.x == 0) a.scale_out[0] = sc;
    const int nchunk = a.K >> 5, nblk = a.Nn >> 5;
    const int64_t groups = (int64_t)a.nphase * a.ntaps * nchunk * nblk * 2 * 64;       // one (hi, lo) pair of 16-B lane chunks each
    for (int64_t g = blockIdx.x * 256 + threadIdx.x; g < groups; g += (int64_t)gridDim.x * 256) {
        const int lane = g & 63;
        int64_t t = g >> 6;
        const int ks = t & 1; t >>= 1;
        const int nb = t % nblk; t /= nblk;
        const int ch = t % nchunk; t /= nchunk;
        const int tap = t % a.ntaps; const int ph = t / a.ntaps;
        const int n = nb * 32 + (lane & 31);
        const int k0 = ch * 32 + ks * 16 + (lane >> 5) * 8;
        const float* src = a.w + n * a.sn + a.r[ph][tap] * a.sr + a.s[ph][tap] * a.ss;
        f16x8 h, l;
        #pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v = src[(k0 + j) * a.sk] * sc;
            h[j] = (_Float16)v; l[j] = (_Float16)(v - (float)h[j]);
        }
        char* dst = a.img + (((((int64_t)(ph * a.ntaps + tap) * nchunk + ch) * nblk + nb) * 2 + ks) * 2) * 1024 + lane * 16;
        *reinterpret_cast<f16x8*>(dst) = h;
        *reinterpret_cast<f16x8*>(dst + 1024) = l;
    }
}

int grid_for(int64_t work_items)
{
    int64_t g = (work_items + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}

template <int MB, int NB>
hipError_t launch_conv(const ConvArgs& a, hipStream_t st)
{
    constexpr int LDS = 2 * (2 * MB * 4 * 1024 + 2 * NB * 4 * 1024);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_f16x3_kernel<MB, NB>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const int64_t M = (int64_t)a.N * a.H * a.W;
    dim3 grid((unsigned)(M / (64 * MB)), (unsigned)(a.nphase * (a.Cout / (64 * NB))));
    hipLaunchKernelGGL((conv_f16x3_kernel<MB, NB>), grid, dim3(256), LDS, st, a);
    return hipGetLastError();
}

}  // namespace

extern "C" {

int wc_conv_split_f32(const float* x, int64_t n, int relu, void* hi, void* lo, float* scale, void* amax_scratch, wc_stream_t stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (!x || !hi || !lo || !scale || !amax_scratch || n <= 0 || (n & 3)) return WC_ERR_ARG;
    hipLaunchKernelGGL(conv_absmax_kernel, dim3(kAmaxBlocks), dim3(256), 0, st, x, n / 4, n, (float*)amax_scratch);
    hipLaunchKernelGGL(conv_split_kernel, dim3(grid_for(n / 4)), dim3(256), 0, st, x, n / 4, (const float*)amax_scratch, relu,
                       (_Float16*)hi, (_Float16*)lo, scale);
    return (int)hipGetLastError();
}

size_t wc_conv_weights_bytes(const wc_conv_geom* g)
{
    if (!g) return 0;
    return (size_t)g->nphase * g->ntaps * g->Cin * g->Cout * 4;      // hi + lo halves
}

int wc_conv_weights_f32(const float* w, int64_t stride_k, int64_t stride_n, int64_t stride_r, int64_t stride_s, int64_t n_elems,
                        const wc_conv_geom* g, void* image, float* scale, void* amax_scratch, wc_stream_t stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (!w || !g || !image || !scale || !amax_scratch || n_elems <= 0) return WC_ERR_ARG;
    if (g->ntaps < 1 || g->ntaps > kMaxTaps || g->nphase < 1 || g->nphase > kMaxPhase || (g->Cin & 31) || (g->Cout & 31)) return WC_ERR_ARG;
    // the scale comes from the whole source tensor (n_elems covers its storage extent)
    hipLaunchKernelGGL(conv_absmax_kernel, dim3(kAmaxBlocks), dim3(256), 0, st, w, n_elems / 4, n_elems, (float*)amax_scratch);
    WeightArgs a;
    a.w = w; a.sk = stride_k; a.sn = stride_n; a.sr = stride_r; a.ss = stride_s;
    a.amax = (const float*)amax_scratch; a.scale_out = scale; a.img = (char*)image;
    a.K = g->Cin; a.Nn = g->Cout; a.ntaps = g->ntaps; a.nphase = g->nphase;
    for (int p = 0; p < kMaxPhase; ++p)
        for (int t = 0; t < kMaxTaps; ++t) { a.r[p][t] = g->wr[p][t]; a.s[p][t] = g->ws[p][t]; }
    const int64_t groups = (int64_t)g->nphase * g->ntaps * (g->Cin >> 5) * (g->Cout >> 5) * 128;
    hipLaunchKernelGGL(conv_weights_kernel, dim3(grid_for(groups)), dim3(256), 0, st, a);
    return (int)hipGetLastError();
}

int wc_conv_supported(const wc_conv_geom* g)
{
    if (!g) return 0;
    if (g->ntaps < 1 || g->ntaps > kMaxTaps || g->nphase < 1 || g->nphase > kMaxPhase) return 0;
    if ((g->Cin & 31) || (g->Cout & 127)) return 0;
    const int64_t M = (int64_t)g->N * g->H * g->W;
    if (M <= 0 || (M & 127) || M > (int64_t)1 << 31) return 0;
    if ((int64_t)g->N * g->Hin * g->Win > (int64_t)1 << 31) return 0;
    return 1;
}

int wc_conv_f16x3(const void* xhi, const void* xlo, const float* xscale, const void* wimage, const float* wscale,
                  const float* bias, const void* zero_line, const wc_conv_geom* g, int relu, float* y, wc_stream_t stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (!xhi || !xlo || !xscale || !wimage || !wscale || !zero_line || !g || !y) return WC_ERR_ARG;
    if (!wc_conv_supported(g)) return WC_ERR_SHAPE;
    ConvArgs a;
    a.xhi = (const _Float16*)xhi; a.xlo = (const _Float16*)xlo; a.zero = (const _Float16*)zero_line;
    a.wimg = (const char*)wimage; a.xscale = xscale; a.wscale = wscale; a.bias = bias; a.y = y;
    a.N = g->N; a.H = g->H; a.W = g->W; a.Hin = g->Hin; a.Win = g->Win; a.Cin = g->Cin; a.Cout = g->Cout;
    a.in_stride = g->in_stride; a.ntaps = g->ntaps; a.nphase = g->nphase; a.Hout = g->Hout; a.Wout = g->Wout;
    a.out_stride = g->out_stride; a.relu = relu;
    for (int p = 0; p < kMaxPhase; ++p) {
        a.offy[p] = g->off_y[p]; a.offx[p] = g->off_x[p];
        for (int t = 0; t < kMaxTaps; ++t) { a.dy[p][t] = g->dy[p][t]; a.dx[p][t] = g->dx[p][t]; }
    }
    const int64_t M = (int64_t)g->N * g->H * g->W;
    const bool wide = (g->Cout % 256) == 0;
    // the larger pixel tile when it still gives every CU a workgroup
    const int64_t wgs_big = (M / 256) * g->nphase * (g->Cout / (wide ? 256 : 128));
    hipError_t e;
    if ((M % 256) == 0 && wgs_big >= 256) e = wide ? launch_conv<4, 4>(a, st) : launch_conv<4, 2>(a, st);
    else                                  e = wide ? launch_conv<2, 4>(a, st) : launch_conv<2, 2>(a, st);
    return (int)e;
}

}  // extern "C"
