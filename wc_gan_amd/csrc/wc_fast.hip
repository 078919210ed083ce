// Fast path of the fused WC affine (K3 wc_apply_f32, K6 wc_bwd_apply_f32) for C in {32,64,128,256}:
//   out[m,:] (+)= ((in[m,:] - center) .* s) B'[slot(m)] .* colscale[slot(m)] + bias[slot(m)] - sub
//
// Why not the f32 MFMA: at C = 256 the contraction has 64 flop/B, so v_mfma_f32_32x32x2_f32 (157 TF/s)
// caps the kernel at ~30 % of the HBM roofline and gfx950 has no xf32 MFMA.  Each fp32 operand is split
// into two fp16 terms, v = hi + lo with lo = fp16(v - hi) (22 significant bits while lo is a normal fp16,
// an absolute error <= 2^-25 of the scaled range once it is subnormal), and the product runs as three
// v_mfma_f32_32x32x16_f16 into ONE fp32 accumulator (lo*hi + hi*lo + hi*hi; lo*lo <= 2^-22 is dropped):
// 3/16 of the f32-MFMA time at fp32-GEMM accuracy.  fp16's narrow range is handled by exact power-of-two scalings:
// per input channel (s, from a row subsample), per output column (colscale, from the table itself);
// a tile holding an element that still exceeds the fp16 range is recomputed by the same workgroup with plain fp32
// FMAs from global memory (tagged in LDS while it is staged) -- the result never depends on the scale guess.
//
// Structure: one persistent 512-thread workgroup per CU.  Each of the 8 waves keeps the B' fragments of
// its 32 output columns for ALL of K in registers (128 VGPRs at C = 256), so the only LDS traffic is the
// activation tile: fp32 rows are loaded once from HBM (16 B per lane, whole rows per wave), centred,
// scaled, split and written as XOR-swizzled fp16 hi/lo images that every wave reads with ds_read_b128.
// Tiles are double-buffered; the loads of tile t+2 are in flight while tile t is on the matrix pipe.
#include "wc_common.h"
#include <stdlib.h>
#ifndef WC_MFMA16
#define WC_MFMA16 1     // ring kernel on v_mfma_f32_16x16x32_f16 and the fp16 tables in that shape's load order (0: 32x32x16, development)
#endif
#ifndef WC_NT_STORE
#define WC_NT_STORE 1      // nontemporal stores of y in the ring kernel's epilogue (y is streamed out: K3 53.5 -> 50.1 us, the step unchanged)
#endif
#ifndef WC_NT_STORE_K6
#define WC_NT_STORE_K6 0   // the same for dx in the one-pass K6 kernel: measured (stage 138 -> 133.5 us in bench.py's loop, nothing under rocprofv3) and left off -- its stores are 64-byte pieces of a row, which leave the chip un-merged: HBM writes 134 -> 164 MB per launch (WRITE_SIZE), the step unchanged
#endif
#ifndef WC_FENCE_DEP
#define WC_FENCE_DEP 0     // the slot-read fence: 0 an explicit lgkmcnt(0), 1 a register dependency (measured the same)
#endif
#ifndef WC_M16_ORDER
#define WC_M16_ORDER 0
#endif
#ifndef WC_CONV_NUM
#define WC_CONV_NUM 2      // quarters of a tile's MFMA loop that carry the next tile's conversion (3: measured the same)
#endif
#ifndef WC_STAMPS
#define WC_STAMPS 0
#endif
#define WC_STAMP(i) do { if (WC_STAMPS && stamp_on) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); ts[i] = t_; } } while (0)
#ifndef WC_NO_PIPE
#ifndef WC_NT_STORE_PL
#define WC_NT_STORE_PL 0   // planes form of the ring kernel: nontemporal stores of its 64-byte pieces (a wave owns 32 columns = 64 bytes of an fp16 row).  Measured and left off: they leave the chip un-merged -- HBM writes 170 MB per launch for 138 MB of planes + mask (WRITE_SIZE), 61 us against the fp32 form's 50, 80 us in the layer's flow; plain stores let the L2 put the two halves of a line together
#endif
#define WC_NO_PIPE 0   // development: 1 leaves the ring kernel's k-loop to hipcc's own schedule
#endif
#ifndef WC_K6_ABL
#define WC_K6_ABL 0   // development, one-pass K6 only (results wrong; tools/k6_variants.py): 1 no dx stores, 2 no MFMA, 4 no fragment reads (with 2), 8 no conversion arithmetic / image writes, 32 no waits for the DMAs, 64 no hand-off waits (counters), 256 no DMAs of x (half of the kernel's LDS-DMA instructions)
#endif
#ifndef WC_ABL
#define WC_ABL 0      // development ablation bits: 1 no stores, 2 no MFMA, 4 no staging writes, 8 no loads, 16 no counters, 32 MFMA operands from registers only
#endif
#include <type_traits>

namespace {

typedef __fp16 h16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2_ __attribute__((ext_vector_type(2)));
// round-to-nearest pack (v_cvt_pk_f16_f32) for the low terms: a truncating convert there biases every element
// toward zero by ~2^-23 and shows up as a uniform 3e-7 shrink of the covariance
__device__ __forceinline__ unsigned pk_rne(float a, float b)
{
    const f32x2_ v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));
}

typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
constexpr int kPlaneBounds = 1024;             // planes form: per-table bounds (and per-workgroup maxima) in the scale record
constexpr float kF16Guard = 60000.0f;          // |scaled element| above this -> the tile takes the exact path

__device__ __forceinline__ f32x4 ld4f(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

// --------------------------------------------------------------------------------------------
// per-input-channel power-of-two scale from <= 256 sampled rows: s_k = 2^(4 - ceil(log2(max|in - c|)))
// --------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void channel_scale_kernel(const float* __restrict__ in, const float* __restrict__ center,
                                                             int64_t M, int C, float* __restrict__ scale,
                                                             const float* __restrict__ in2, const float* __restrict__ center2,
                                                             float* __restrict__ scale2, int* __restrict__ gate)
{
    __shared__ float red[16][64];
    if (gate && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 64) gate[threadIdx.x] = 0;     // the overflow gate of the call
    if (blockIdx.y == 1) { in = in2; center = center2; scale = scale2; }                           // second operand (K4: gy)
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int part = threadIdx.x >> 6;
    const int64_t nsamp = M < 256 ? M : 256;
    const int64_t stride = M / nsamp;
    float mx = 0.f;
    if (c < C) {
        const float ce = center ? center[c] : 0.f;
        for (int64_t r = part; r < nsamp; r += 16) mx = fmaxf(mx, fabsf(in[wc_sample_row(r, stride) * C + c] - ce));
    }
    red[part][threadIdx.x & 63] = mx;
    __syncthreads();
    if (threadIdx.x < 64 && c < C) {
        float gm[16];
#pragma unroll
        for (int p = 0; p < 16; ++p) gm[p] = red[p][threadIdx.x];
        const float m = wc_robust_max16(gm);     // the sampled max, unless an outlier sits on a sampled row (wc_common.h)
        float s = 1.0f;
        if (m > 0.f && m < 3.0e38f) {
            int e;
            frexpf(m, &e);                       // m = f * 2^e, f in [0.5, 1)  ->  m <= 2^e
            s = ldexpf(1.0f, 4 - e);             // sampled max lands in [8, 16]
        }
        scale[c] = s;
    }
}

// --------------------------------------------------------------------------------------------
// B table split: B[slot][k][n] fp32 (row-major, as wc_color_f32 writes A / At) -> fp16 hi/lo images of
// B[k][n] / s_k / colscale[n], stored in the order the apply kernels load them: for column group g = n/32 and
// k-step s = k/16 the 64 lanes' 16-byte MFMA B-fragments (lane = 32*((k/8)&1) + n%32, 8 consecutive k each) form one
// contiguous KiB.  Every workgroup of an apply kernel pulls the whole 4*C*C-byte table into registers, so with a
// plain [n][k] image each load instruction touched 32 cache lines for 32 bytes apiece and the prologue took ~9 us
// at C = 256 (measured); in this order each byte crosses the L2 -> CU path once per workgroup.
// one wave per (slot, n)
// --------------------------------------------------------------------------------------------
// (two tables in one launch: K6 builds At's and S's together; a single table passes rows1 = 0)
struct SplitJob { const float* B; const float* scale; _Float16* hi; _Float16* lo; float* colscale; float* scale_out; int rows; };
// (round 4) a third, independent job in the same launch: the additive term of an apply on pre-split planes,
//     out[slot][n] = bias[slot][n] + sum_k (center[k] - mu[k]) A[slot][k][n]          (wc_split.hip's split_bias_kernel)
// -- it needs A only, as the tables do, and as a launch of its own it cost the forward site 5 us between the tables and K3
struct BiasJob { const float* A; const float* bias; const float* center; const float* mu; float* out; int slots; int neg_bias; };      // neg_bias: out = -bias + ...  (K6's two-pass form on planes)
__global__ __launch_bounds__(64) void split_table_kernel(SplitJob j0, SplitJob j1, int C, BiasJob bj)
{
    if ((int)blockIdx.x >= j0.rows + j1.rows) {      // bias job: one workgroup per (slot, 2 columns); thread (q, n) sums the rows k = q mod 32
        // (8 terms per thread, every load of the thread in flight at once: A was written by the launch in front on other XCDs, so a
        // dependent batch of loads costs a memory round trip -- 2 x 128-term chains per column took 28 us, 32 terms per thread 11.6 us
        // and the tables' launch with them; the tables alone take 5.2)
        __shared__ double red[64];
        const int bb = (int)blockIdx.x - j0.rows - j1.rows, nch = C / 2;
        const int slot = bb / nch, n = (bb % nch) * 2 + (threadIdx.x & 1), q = threadIdx.x >> 1;
        const float* a = bj.A + (int64_t)slot * C * C + n;
        double acc = 0.0;
#pragma unroll 8
        for (int k = q; k < C; k += 32) {
            const double d = (double)(bj.center ? bj.center[k] : 0.f) - (double)(bj.mu ? bj.mu[k] : 0.f);
            acc += d * (double)a[(int64_t)k * C];
        }
        red[threadIdx.x] = acc;
        __syncthreads();
        if (q == 0) {
            double t = 0.0;
#pragma unroll
            for (int i = 0; i < 32; ++i) t += red[threadIdx.x + 2 * i];
            const double bv = bj.bias ? (double)bj.bias[(int64_t)slot * C + n] : 0.0;
            bj.out[(int64_t)slot * C + n] = (float)(t + (bj.neg_bias ? -bv : bv));
        }
        return;
    }
    const bool second = (int)blockIdx.x >= j0.rows;
    const float* __restrict__ B = second ? j1.B : j0.B;
    const float* __restrict__ scale = second ? j1.scale : j0.scale;
    _Float16* __restrict__ hi = second ? j1.hi : j0.hi;
    _Float16* __restrict__ lo = second ? j1.lo : j0.lo;
    float* __restrict__ colscale = second ? j1.colscale : j0.colscale;
    float* __restrict__ scale_out = second ? j1.scale_out : j0.scale_out;
    const int64_t row = second ? blockIdx.x - j0.rows : blockIdx.x;                 // slot * C + n
    // scale_out: the plan's own copy of the input scales (the apply kernel reads them from the plan) -- written here by the
    // first workgroups instead of by a device-to-device copy in front of this launch
    if (scale_out && row * 64 < C) {
        const int k = (int)row * 64 + threadIdx.x;
        if (k < C) scale_out[k] = scale[k];
    }
    const float* src = B + (row / C) * (int64_t)C * C + (row % C);     // column n of B[slot]
    const int lane = threadIdx.x;
    float v[16];
    float mx = 0.f;
    const int per = C / 64 > 0 ? C / 64 : 1;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        v[i] = 0.f;
        const int k = lane + 64 * i;
        if (i < per && k < C) { v[i] = src[(int64_t)k * C] / scale[k]; mx = fmaxf(mx, fabsf(v[i])); }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    float cs = 1.0f;
    if (mx > 0.f && mx < 3.0e38f) { int e; frexpf(mx, &e); cs = ldexpf(1.0f, e); }     // |v / cs| <= 1
    const float inv = 1.0f / cs;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int k = lane + 64 * i;
        if (i < per && k < C) {
            const float w = v[i] * inv;
            const _Float16 h = (_Float16)w;
            const _Float16 l = (_Float16)(w - (float)h);
            // "register image" order: the 16 bytes a lane loads for MFMA fragment (column group n/32, k-step k/16) sit
            // with the other 63 lanes' in one contiguous KiB
            const int n = (int)(row % C);
#if WC_MFMA16
            // 16x16x32 fragments: (column group n/32, k-step k/32, column half (n/16)&1): lane = 16*((k/8)&3) + n%16
            const int KS32 = C / 32;
            const int64_t idx = (row / C) * (int64_t)C * C +
                                (((((int64_t)(n >> 5) * KS32 + (k >> 5)) * 2 + ((n >> 4) & 1)) * 64 + ((k >> 3) & 3) * 16 + (n & 15)) * 8 + (k & 7));
#else
            const int KS = C / 16;
            const int64_t idx = (row / C) * (int64_t)C * C +
                                ((((int64_t)(n >> 5) * KS + (k >> 4)) * 64 + ((k >> 3) & 1) * 32 + (n & 31)) * 8 + (k & 7));
#endif
            hi[idx] = h;
            lo[idx] = l;
        }
    }
    if (lane == 0) colscale[row] = cs;
}

struct FastArgs {
    const float* in; const float* center; const float* scale;       // [M,C], [C]|null, [C]
    const _Float16* Bhi; const _Float16* Blo; const float* colscale; // [slots][C*C] in register-image order, [slots][C]
    int64_t slot_stride;                                             // C*C, or 0 when the table is shared
    const float* bias; const float* sub; const int32_t* slot;          // bias/sub may be null
    int bias_on, sub_on;                                             // ring kernel: bias/sub are then pointed at `scale` and ignored
    int64_t M, HW;
    int accumulate;
    int mixed;                                                       // slots given and HW % tile != 0: a tile may straddle samples of different slots
    int relu;                                                        // epilogue: out = max(out, 0) (NaN stays NaN)
    unsigned* maskout;                                               // ring kernel, relu != 0: the activation's 1-bit mask, [M/32][C] words (bit b of word (t, c) = out[32 t + b][c] != 0); nullable
    const float* Bf; int64_t bf_stride;                             // the fp32 table [slot][k][n] for the exact path
    float* out;
    // ring kernel, planes form (PL): the output leaves as the NEXT CONVOLUTION's operand (csrc/wc_conv.hip: fp16 planes hi | lo
    // of s * out with ONE power-of-two scale s) instead of fp32.  oscale[1..] = the bounds the predicted scale follows from (wc_launch_out_scale), oscale[0] =
    // the scale the planes were written with (what the convolution reads); oamax[wg] = max |s * out| seen by a workgroup.  gate
    // (second launch): the first launch's oamax -- every workgroup folds them and leaves at once unless the prediction overflowed
    // fp16, in which case the whole pass is redone with the scale the maximum asks for.
    _Float16* phi; _Float16* plo; float* oscale; float* oamax; const float* gate; int ngate;
    int ntiles, tiles_per_wg;
    unsigned long long* dbg;      // WC_STAMPS builds only: s_memtime stamps of one steady-state tile
};

// M must be a multiple of the row tile (the caller checks): every load and store below is unconditional, which is
// what lets hipcc keep counted vmcnt waits instead of draining the memory queue at every masked access.
template <int C, bool ACC, bool HAS_SLOT>
__global__ __launch_bounds__(512, 2) void affine_f16x3_kernel(FastArgs a)
{
    constexpr int BM = 64 * 256 / C;          // rows per tile: 64 KiB of fp32 activations
    constexpr int CPR = C / 8;                // 16-byte chunks per fp16 row
    constexpr int KS = C / 16;                // MFMA k-steps
    constexpr int CG = C / 32;                // column groups (one wave each)
    constexpr int RG = 8 / CG;                // row groups of waves
    constexpr int SUB = BM / 32 / RG;         // 32-row sub-tiles per wave per tile (= 2)
    constexpr int C4 = C / 4;                 // float4 per row
    constexpr int IMG = BM * C * 2;           // bytes of one fp16 image
    constexpr int RSTEP = 512 / C4;
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [2 buffers][hi | lo] + 2 dirty tags
    volatile int* dirty = reinterpret_cast<volatile int*>(smem + 4 * IMG);      // dirty[buf] == tile+1: exact path

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cg = wave % CG, rg = wave / CG;
    const int l31 = lane & 31, lh = lane >> 5;

    // Tile order.  Without slots the workgroups interleave (tile = block + i*grid): at any moment the chip then
    // streams one contiguous window of HBM instead of 256 windows half a megabyte apart, whose identical low
    // address bits march through the same memory channels in lockstep.  With slots a workgroup keeps a contiguous
    // run of tiles so that its B' registers are reloaded only when the sample's slot changes.
    int t_first, t_stride, t_count;
    if (HAS_SLOT) {
        t_first = blockIdx.x * a.tiles_per_wg; t_stride = 1;
        t_count = a.ntiles - t_first; if (t_count > a.tiles_per_wg) t_count = a.tiles_per_wg;
    } else {
        t_first = blockIdx.x; t_stride = gridDim.x;
        t_count = (a.ntiles - t_first + t_stride - 1) / t_stride;
    }
    if (t_count <= 0) return;
    const int t_begin = 0, t_end = t_count;              // the loops below count tiles i; tile_of(i) maps to memory
    auto tile_of = [&](int i) { return t_first + i * t_stride; };

    // staging coordinates: thread handles float4 #c4 of rows srow + RSTEP*p, p = 0..7
    const int c4 = tid % C4;
    const int srow = tid / C4;
    const f32x4 scl = ld4f(a.scale + 4 * c4);
    f32x4 ncs = {0.f, 0.f, 0.f, 0.f};
    if (a.center) { const f32x4 ce = ld4f(a.center + 4 * c4); ncs = -ce * scl; }

    auto swz = [](int row) -> int {
        if (CPR >= 16) return row & 15;
        return (row / (16 / CPR)) & (CPR - 1);
    };
    const int in_off = srow * C + 4 * c4;                 // element offset inside a tile (fits 32 bits)
    // LDS byte offsets of this thread's 8 staging stores (tile-invariant)
    auto st_off_fn = [&](int p) {
        const int row = srow + RSTEP * p;
        return row * (C * 2) + (((c4 >> 1) ^ swz(row)) * 16) + (c4 & 1) * 8;
    };
    // the swizzle term repeats with period 16/gcd(RSTEP,16) in p: keep the distinct bases, add the row stride as a constant
    constexpr int NB = (RSTEP % 16 == 0) ? 1 : 16 / RSTEP;
    int st_base[NB];
#pragma unroll
    for (int q = 0; q < NB; ++q) st_base[q] = st_off_fn(q) - q * RSTEP * (C * 2);

    f32x4 xr[8];
    // In the non-accumulating kernels the steady-state loads are inline asm, invisible to hipcc's waitcnt pass:
    // with compiler-visible loads it drains vmcnt to 0 before every use (loads and stores share the counter and
    // it treats the mix as unordered), which also waits for the previous tile's 32 stores.  The waits are counted
    // by hand instead (wait_chunk): between a chunk's load and its use one iteration later exactly
    // (7-p) loads + 32 stores + p loads = 39 younger VMEM ops are issued.
    constexpr bool ASM_LOADS = !ACC;
    const int in_off_b = in_off * 4;
    auto load_chunk = [&](int ti, int p) {
        const float* base = a.in + (int64_t)tile_of(ti) * (BM * C) + p * (RSTEP * C);     // wave-uniform
        if (ASM_LOADS) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(xr[p]) : "v"(in_off_b), "s"(base) : "memory");
        else xr[p] = ld4f(base + in_off);
    };
    auto stage_load = [&](int ti) {
#pragma unroll
        for (int p = 0; p < 8; ++p) load_chunk(ti, p);
    };
    auto write_chunk = [&](int buf, int p, int tag) {
        char* hi_img = smem + buf * 2 * IMG;
        char* lo_img = hi_img + IMG;
        {
            const f32x4 g = xr[p] * scl + ncs;
            if ((fabsf(g[0]) > kF16Guard) | (fabsf(g[1]) > kF16Guard) | (fabsf(g[2]) > kF16Guard) | (fabsf(g[3]) > kF16Guard))
                dirty[buf] = tag;                      // rare; every reader compares against its own tile's tag
            const unsigned hw01 = pk_rne(g[0], g[1]), hw23 = pk_rne(g[2], g[3]);
            const f16x2 h01 = __builtin_bit_cast(f16x2, hw01), h23 = __builtin_bit_cast(f16x2, hw23);
            const float r0 = g[0] - (float)h01[0], r1 = g[1] - (float)h01[1];
            const float r2 = g[2] - (float)h23[0], r3 = g[3] - (float)h23[1];
            const unsigned l01 = pk_rne(r0, r1);
            const unsigned l23 = pk_rne(r2, r3);
            *reinterpret_cast<uint2*>(hi_img + st_base[p % NB] + p * RSTEP * (C * 2)) = make_uint2(hw01, hw23);
            *reinterpret_cast<uint2*>(lo_img + st_base[p % NB] + p * RSTEP * (C * 2)) = make_uint2(l01, l23);
        }
    };
    auto stage_write = [&](int buf, int tag) {
#pragma unroll
        for (int p = 0; p < 8; ++p) write_chunk(buf, p, tag);
    };

    // B' fragments of this wave's 32 columns, all of K, in registers
    f16x8 bhi[KS], blo[KS];
    float cscale = 1.f, addv = 0.f;
    int cur_slot = -1;
    const int col = cg * 32 + l31;
    auto load_b = [&](int slot) {
#if WC_MFMA16
        // the tables are stored for the ring kernel's 16x16x32 fragments; this kernel's 32x32x16 fragment (k-step s, 8 k
        // per lane) is the same 16 bytes at: unit (s/2, column half l31/16), lane 16*(2*(s&1) + lh) + l31%16
        const int64_t lo16 = (((int64_t)cg * (KS / 2) * 2 + (l31 >> 4)) * 64 + lh * 16 + (l31 & 15)) * 8;
        const _Float16* ph = a.Bhi + (int64_t)slot * a.slot_stride + lo16;
        const _Float16* pl = a.Blo + (int64_t)slot * a.slot_stride + lo16;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            bhi[s] = *reinterpret_cast<const f16x8*>(ph + 1024 * (s >> 1) + 256 * (s & 1));
            blo[s] = *reinterpret_cast<const f16x8*>(pl + 1024 * (s >> 1) + 256 * (s & 1));
        }
#else
        const _Float16* ph = a.Bhi + (int64_t)slot * a.slot_stride + ((int64_t)cg * KS * 64 + lane) * 8;
        const _Float16* pl = a.Blo + (int64_t)slot * a.slot_stride + ((int64_t)cg * KS * 64 + lane) * 8;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            bhi[s] = *reinterpret_cast<const f16x8*>(ph + 512 * s);
            blo[s] = *reinterpret_cast<const f16x8*>(pl + 512 * s);
        }
#endif
        const int64_t srow_ = a.slot_stride ? slot : 0;
        cscale = a.colscale[srow_ * C + col];
        addv = 0.f;
        if (a.bias) addv += a.bias[(int64_t)slot * C + col];
        if (a.sub) addv -= a.sub[col];
        cur_slot = slot;
    };

    if (tid < 2) dirty[tid] = 0;
    __syncthreads();
    stage_load(t_begin);
    if (!HAS_SLOT) load_b(0);
    if (ASM_LOADS)          // asm loads are not tracked by the compiler: wait for the first tile by hand
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(xr[0]), "+v"(xr[1]), "+v"(xr[2]), "+v"(xr[3]),
                                            "+v"(xr[4]), "+v"(xr[5]), "+v"(xr[6]), "+v"(xr[7]) :: "memory");
    stage_write(0, t_begin + 1);
    if (t_begin + 1 < t_end) stage_load(t_begin + 1);
    __syncthreads();

    // per-lane output offset inside a tile: rows (4*lh + ...) and this wave's column; the rest is compile-time
    const int out_lane = (4 * lh) * C + col;
    const int rd_lane = l31 * (C * 2);

    unsigned long long ts[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    bool stamp_on = false;
    using T_ = std::integral_constant<bool, true>;
    using F_ = std::integral_constant<bool, false>;
    const int chunk_sb = (__builtin_amdgcn_readfirstlane(wave) >= 4) ? 1 : 0;      // wave-uniform (SGPR)
    // One tile: [convert + LDS-write tile t+1] [issue the loads of tile t+2] [MFMA + store tile t] [LDS barrier].
    // The steady-state loop calls it with both stages unconditional (the last two tiles are peeled below), so the
    // body has no control-flow merge and hipcc emits COUNTED vmcnt waits: the 32 stores of a tile stay in flight
    // across the barrier and under the next tile's staging instead of being drained.
    auto tile_body = [&](int t, auto do_write, auto do_load) {
        const int cur = (t - t_begin) & 1;
        constexpr bool W_ = decltype(do_write)::value, L_ = decltype(do_load)::value;
        stamp_on = WC_STAMPS && (t == t_begin + 3);
        WC_STAMP(0);
        if (ASM_LOADS && t == t_begin)       // the prologue's loads have no stores behind them: drain once, counts hold after
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(xr[0]), "+v"(xr[1]), "+v"(xr[2]), "+v"(xr[3]),
                                                "+v"(xr[4]), "+v"(xr[5]), "+v"(xr[6]), "+v"(xr[7]) :: "memory");

        if (HAS_SLOT) {
            const int slot = a.slot[((int64_t)tile_of(t) * BM) / a.HW];
            if (slot != cur_slot) {
                load_b(slot);
                if (ASM_LOADS)     // the table reload put extra loads in the queue: drain, after which every count is conservative again
                    asm volatile("s_waitcnt vmcnt(0)" : "+v"(xr[0]), "+v"(xr[1]), "+v"(xr[2]), "+v"(xr[3]),
                                                        "+v"(xr[4]), "+v"(xr[5]), "+v"(xr[6]), "+v"(xr[7]) :: "memory");
            }
        }
        float* out_tile = a.out + (int64_t)tile_of(t) * (BM * C);       // wave-uniform
        const char* hi_img = smem + cur * 2 * IMG;
        const char* lo_img = hi_img + IMG;
        // One 32-row sub-tile: 3*KS MFMAs, and -- in exactly one of the wave's two sub-tiles -- the staging of the
        // NEXT tile in the MFMA gaps: every other k-step one 16-B chunk is centred/split/written to the other LDS
        // buffer and its register refilled from tile t+2.  Hand count of younger VMEM ops at each use: 39.
        auto sub_tile = [&](int sb) {
            const bool CH = (sb == chunk_sb);
            const int rbase = (rg * SUB + sb) * 32;
            const int sw = swz(rbase + l31);
            const char* hrow = hi_img + rbase * (C * 2) + rd_lane;
            const char* lrow = lo_img + rbase * (C * 2) + rd_lane;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            // fragments are fetched one k-step ahead of the MFMAs that consume them
            f16x8 ah = *reinterpret_cast<const f16x8*>(hrow + ((0 + lh) ^ sw) * 16);
            f16x8 al = *reinterpret_cast<const f16x8*>(lrow + ((0 + lh) ^ sw) * 16);
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                f16x8 nh = ah, nl = al;
                if (s + 1 < KS) {
                    const int chunk = (2 * (s + 1) + lh) ^ sw;
                    nh = *reinterpret_cast<const f16x8*>(hrow + chunk * 16);
                    nl = *reinterpret_cast<const f16x8*>(lrow + chunk * 16);
                }
                constexpr int STEP = (KS >= 8) ? KS / 8 : 1;         // k-steps per chunk
                constexpr int PER = (KS >= 8) ? 1 : 8 / KS;          // chunks per k-step when K is short
                if (CH && (s % STEP) == 0 && (s / STEP) < 8) {
#pragma unroll
                    for (int q = 0; q < PER; ++q) {
                        const int p = (s / STEP) * PER + q;
                        if (W_ && ASM_LOADS && !(WC_ABL & 8)) {
                            if (L_) asm volatile("s_waitcnt vmcnt(39)" : "+v"(xr[p]) :: "memory");
                            else asm volatile("s_waitcnt vmcnt(16)" : "+v"(xr[p]) :: "memory");   // <= every tail count
                        }
                        if (W_ && !(WC_ABL & 4)) write_chunk(cur ^ 1, p, t + 2);
                        if (L_ && !(WC_ABL & 8)) load_chunk(t + 2, p);
                    }
                }
                if (WC_ABL & 2) { asm volatile("" :: "v"(ah), "v"(al)); }
                else {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bhi[s], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, blo[s], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bhi[s], acc, 0, 0, 0);
                }
                ah = nh; al = nl;
            }
            WC_STAMP(1 + 2 * sb);
            float* po = out_tile + rbase * C + out_lane;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ro = ((r & 3) + 8 * (r >> 2)) * C;
                float v = acc[r] * cscale + addv;
                if (ACC) v += po[ro];
                if (a.relu) v = v > 0.f ? v : (v == v ? 0.f : v);
                if (WC_ABL & 1) asm volatile("" :: "v"(v)); else
                po[ro] = v;
            }
            WC_STAMP(2 + 2 * sb);
        };
        // The two waves that share a SIMD (w and w+4) take the staging in opposite halves of the tile: while one is
        // exposed to its vmcnt waits the other is in a pure MFMA stretch, so the matrix pipe keeps running under
        // the memory stalls.  Both orders issue the same 39 younger VMEM ops between a chunk's load and its use.
        if (dirty[cur] == t + 1) {
            // Exact path (rare): this tile holds an element outside the fp16 range.  Stage the next tile first
            // (drain the queue: the hand counts assume the regular order), then recompute this one in fp32 straight
            // from global memory -- same rows, same columns, same epilogue as the MFMA path.
            if (W_) {
                if (ASM_LOADS)
                    asm volatile("s_waitcnt vmcnt(0)" : "+v"(xr[0]), "+v"(xr[1]), "+v"(xr[2]), "+v"(xr[3]),
                                                        "+v"(xr[4]), "+v"(xr[5]), "+v"(xr[6]), "+v"(xr[7]) :: "memory");
                stage_write(cur ^ 1, t + 2);
            }
            if (L_) stage_load(t + 2);
            const float* xin = a.in + (int64_t)tile_of(t) * (BM * C);
            const float* Bf = a.Bf + (int64_t)(cur_slot < 0 ? 0 : cur_slot) * a.bf_stride + col;
            for (int sb = 0; sb < SUB; ++sb) {
                const int rbase = (rg * SUB + sb) * 32;
                for (int i = 0; i < 16; ++i) {
                    const int row = rbase + (i & 3) + 8 * (i >> 2) + 4 * lh;
                    const float* xrow = xin + row * C;
                    float accf = 0.f;
                    for (int k = 0; k < C; ++k) {
                        const float ce = a.center ? a.center[k] : 0.f;
                        accf = fmaf(xrow[k] - ce, Bf[(int64_t)k * C], accf);
                    }
                    float* po = out_tile + row * C + col;
                    float v = accf + addv;
                    if (ACC) v += *po;
                    if (a.relu) v = v > 0.f ? v : (v == v ? 0.f : v);
                    *po = v;
                }
            }
        } else {
#pragma unroll
        for (int sb = 0; sb < SUB; ++sb) sub_tile(sb);
        }
        // LDS hand-off only: a raw barrier behind lgkmcnt(0).  __syncthreads() would also drain vmcnt.
        WC_STAMP(5);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        WC_STAMP(6);
    };
    int t = t_begin;
    for (; t + 2 < t_end; ++t) tile_body(t, T_{}, T_{});
    if (t + 1 < t_end) { tile_body(t, T_{}, F_{}); ++t; }
    tile_body(t, F_{}, F_{});
    if (WC_STAMPS && a.dbg && lane == 0 && (blockIdx.x == 0 || blockIdx.x == 100)) {
        unsigned long long* d = a.dbg + ((blockIdx.x ? 1 : 0) * 8 + wave) * 8;
        for (int i = 0; i < 8; ++i) d[i] = ts[i];
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Ring variant (non-accumulating calls).  What the register-staged kernel above cannot do:
//  * The activation rows travel HBM -> LDS by LDS-DMA (global_load_lds, no VGPR in between) in 1-KiB chunks (one
//    wave-instruction), into a ring of SEVEN chunk slots per wave: chunk i+7 is requested the moment chunk i has been
//    read out for conversion, so every wave keeps 6-7 KiB in flight ~1.75 tiles ahead of use and a raw slot needs no
//    cross-wave hand-off at all (each wave converts exactly what its own DMA brought in).
//  * There is NO workgroup barrier in the tile loop and the eight waves are NOT kept in lock-step.  Measured on the
//    two-image-buffer predecessor of this kernel: its two hand-offs ("tile t converted by all" before anyone reads it,
//    "tile t-1 read by all" before its buffer is rewritten) amounted to a barrier per tile; the k-loop itself ran at
//    ~3400 of the 3072 matrix-pipe cycles a SIMD needs per tile, but every tile then lost ~2000 cycles to the slowest
//    wave, the counter polls and the DMA issue, on all SIMDs at once.  Here there are THREE fp16 image buffers, the
//    conversion of tile t+1 rides in the FIRST half of tile t's k-loop and is published in mid-loop, so both
//    hand-offs have half a tile of slack; waves 4-7 start half a tile late, and from then on one wave of each SIMD
//    polls, issues DMA and converts while its partner is in the MFMA-only half of its loop.
//  * Inside the k-loop every MFMA is followed by its share of the other work (LDS reads two k-steps ahead, one
//    conversion stage, a store of the previous tile's rows): the order is pinned gap by gap with sched_barrier.
// Tiles are 32 KiB (8192/C rows).  An element outside the fp16 range marks the workgroup; it then recomputes all of
// its tiles in fp32 at the end.
// vmcnt by hand: before chunk i (of tile t+1, read out during tile t) only its DMA has to have landed.  That DMA was
// issued in the first half of tile t-1's loop; younger are the six DMAs of chunks i+1..i+6 (when those exist:
// t + 3 < n) and the 16 stores at the end of tile t-1: vmcnt(22); fewer DMAs near the end (see WM_).
// ---------------------------------------------------------------------------------------------------------------
#ifndef WC_STAGGER
#define WC_STAGGER 1
#endif
#ifndef WC_AHEAD
#define WC_AHEAD 1      // k-steps between a fragment's ds_read and its MFMAs (2 measured no faster, costs 8 VGPRs)
#endif

// MASK: the epilogue also leaves the ReLU's one-bit gradient mask (a.maskout; relu is then on).  A template parameter, not a
// branch: as a third epilogue inside one kernel it cost the plain form 7 VGPRs and 32 bytes of scratch.
// PL: the output leaves as the next convolution's fp16 planes (FastArgs::phi ...).
template <int C, bool HAS_SLOT, bool MASK = false, bool PL = false>
__global__ __launch_bounds__(512, 1) void affine_ring_kernel(FastArgs a)
{
    constexpr int TR = 8192 / C;              // rows per tile (32 KiB of fp32)
    constexpr int CPR = C / 8, KS = C / 16, CG = C / 32, C4 = C / 4;
    // fp16 image rows: C >= 128 pads every row by one 16-byte chunk (row r starts r chunks further round the 64 banks,
    // conflict-free for the ds_read_b128 lane groups, and every fragment address is lane base + immediate); the
    // narrower tiles have no LDS to spare for that and XOR-swizzle the chunks of a row instead
    // (C = 256 on the 16x16x32 shape: 32 bytes -- a row then starts two 16-byte slots further round the banks and the
    // fragment reads of 16 rows x 4 chunks are conflict-free; with 16 bytes every 16-lane group had one 2-way conflict,
    // SQ_LDS_BANK_CONFLICT 40 % of the LDS cycles.  C = 128 has no LDS to spare for the wider pad.)
    constexpr int PAD = (C == 256 && WC_MFMA16) ? 32 : (C >= 128 ? 16 : 0);
    constexpr int PITCH = C * 2 + PAD;
    constexpr int IMG = TR * PITCH;           // one fp16 image: 16 KiB (+ padding)
    constexpr int FBUF = 2 * IMG;             // hi | lo
    constexpr int NSLOT = 7, RAWW = NSLOT * 1024;
    extern __shared__ __attribute__((aligned(16))) char smem[];      // F0 | F1 | F2 | raw[8 waves][7 KiB] | counters
    char* const fbuf = smem;
    char* const ring = smem + 3 * FBUF;
    volatile int* const cnt = reinterpret_cast<volatile int*>(smem + 3 * FBUF + 8 * RAWW);   // [0] converted, [1] read, [2] dirty

    unsigned long long rt_in = 0;
    if (WC_STAMPS) rt_in = __builtin_amdgcn_s_memrealtime();
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cg = wave % CG, rg = wave / CG;
    const int l31 = lane & 31, lh = lane >> 5;

    int t_first, t_stride, n;
    if (HAS_SLOT) {
        t_first = blockIdx.x * a.tiles_per_wg; t_stride = 1;
        n = a.ntiles - t_first; if (n > a.tiles_per_wg) n = a.tiles_per_wg;
    } else {
        t_first = blockIdx.x; t_stride = gridDim.x;
        n = (a.ntiles - t_first + t_stride - 1) / t_stride;
    }
    float os = 1.f, omax = 0.f;      // PL: output scale; running max |scaled output|
    if (PL) {
        {       // the predicted scale: fold the per-table bounds of wc_launch_out_scale
            const int nb = __builtin_bit_cast(int, a.oscale[1]);
            float bnd = 0.f;
            for (int i = lane; i < nb; i += 64) bnd = __builtin_fmaxf(bnd, a.oscale[2 + i]);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) bnd = __builtin_fmaxf(bnd, __shfl_xor(bnd, o));
            int e = 0;
            if (bnd > 0.f && bnd < 3.0e38f) { (void)frexpf(bnd, &e); os = ldexpf(1.f, 14 - e); }
        }
        if (a.gate) {       // second launch: nothing to do unless the first one's planes overflowed
            float am = 0.f;
            for (int i = lane; i < a.ngate; i += 64) am = __builtin_fmaxf(am, a.gate[i]);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) am = __builtin_fmaxf(am, __shfl_xor(am, o));
            if (!(am > kF16Guard)) return;
            int e = 0;
            if (am < 3.0e38f) { (void)frexpf(am, &e); os = ldexpf(os, 14 - e); } else os = 1.f;
        }
        if (blockIdx.x == 0 && tid == 0) a.oscale[0] = os;
    }
    if (n <= 0) { if (PL && a.oamax && tid == 0) a.oamax[blockIdx.x] = 0.f; return; }
    auto tile_of = [&](int i) { return t_first + i * t_stride; };
    auto swz = [](int row) -> int {
        if (PAD) return 0;
        if (CPR >= 16) return row & 15;
        return (row / (16 / CPR)) & (CPR - 1);
    };

    if (tid < 3) cnt[tid] = 0;
    __syncthreads();

    // DMA of chunk p (1 KiB: lane = 16 B) of this wave's 4 KiB of tile tl into raw slot `slot`.
    // Inline asm on purpose: hipcc drains vmcnt(0) before any ds_read that may alias a pending LDS-DMA it knows of,
    // which would wait for the DMA issued a moment ago and for every store in flight.  M0 (the LDS destination base)
    // is set and restored inside the statement; completion is counted by hand (see the header).
    const unsigned ring_w = (unsigned)(size_t)((__attribute__((address_space(3))) char*)(ring + wave * RAWW));
    auto dma_chunk = [&](int tl, int p, int slot, int lane_) {
        const char* g = reinterpret_cast<const char*>(a.in + (int64_t)tile_of(tl) * (TR * C)) + wave * 4096 + p * 1024 + lane_ * 16;
        const unsigned l = __builtin_amdgcn_readfirstlane(ring_w + slot * 1024);
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(g), "s"(l) : "memory");
    };

    // conversion of this wave's own 4 KiB: float4 #e = 256*wave + lane + 64*p
    const int c4i = lane % C4;
    const f32x4 scl = ld4f(a.scale + 4 * c4i);
    f32x4 ncs = {0.f, 0.f, 0.f, 0.f};
    if (a.center) ncs = -ld4f(a.center + 4 * c4i) * scl;
    float gmax = 0.f;                           // running max |scaled element| (fp16 range guard)
    f32x4 craw, cg4;
    unsigned chw01 = 0, chw23 = 0, clw01 = 0, clw23 = 0;
    auto craw_read = [&](int slot, int lane_) {
        craw = *reinterpret_cast<const f32x4*>(ring + wave * RAWW + slot * 1024 + lane_ * 16);
    };
    auto cv_scale = [&]() {
        cg4 = craw * scl + ncs;
        gmax = __builtin_fmaxf(__builtin_fmaxf(gmax, fabsf(cg4[0])), fabsf(cg4[1]));
        gmax = __builtin_fmaxf(__builtin_fmaxf(gmax, fabsf(cg4[2])), fabsf(cg4[3]));
    };
    auto cv_hi = [&]() { chw01 = pk_rne(cg4[0], cg4[1]); chw23 = pk_rne(cg4[2], cg4[3]); };
    auto cv_lo = [&]() {
        const f16x2 h01 = __builtin_bit_cast(f16x2, chw01), h23 = __builtin_bit_cast(f16x2, chw23);
        (void)h01; (void)h23;
        // remainder = g - float(hi) in one mixed-precision FMA per element (v_fma_mix_f32 reads the fp16 half directly)
        float r0, r1, r2, r3;
        asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(chw01), "v"(cg4[0]));
        asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(chw01), "v"(cg4[1]));
        asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r2) : "v"(chw23), "v"(cg4[2]));
        asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r3) : "v"(chw23), "v"(cg4[3]));
        clw01 = pk_rne(r0, r1);
        clw23 = pk_rne(r2, r3);
    };
    // padded rows: chunk p of a lane lands a constant (64 / C4) rows below its chunk 0 -- one lane offset + immediates
    const int woff0 = ((256 * wave + lane) / C4) * PITCH + c4i * 8;
    auto cv_write = [&](int fb, int p, int lane_, int c4i_, int woff0_) {
        const int row = (int)((unsigned)(256 * wave + lane_ + 64 * p) / (unsigned)C4);
        char* dst = PAD ? fbuf + fb * FBUF + woff0_ + p * ((64 / C4) * PITCH)
                        : fbuf + fb * FBUF + row * PITCH + (((c4i_ >> 1) ^ swz(row)) * 16) + (c4i_ & 1) * 8;
        *reinterpret_cast<uint2*>(dst) = make_uint2(chw01, chw23);
        *reinterpret_cast<uint2*>(dst + IMG) = make_uint2(clw01, clw23);
    };
    // one arrival per wave on a monotonic counter, after this wave's LDS traffic has completed
    // (asm: a C++ volatile/atomic access makes hipcc drain vmcnt(0) around it, i.e. wait for the stores in flight)
    const unsigned cnt_lds = (unsigned)(size_t)((__attribute__((address_space(3))) char*)(smem + 3 * FBUF + 8 * RAWW));
    auto arrive = [&](int which) {
        if (lane == 0) {
            const unsigned one = 1u, addr = cnt_lds + 4u * which;
            asm volatile("ds_add_u32 %0, %1" :: "v"(addr), "v"(one) : "memory");
        }
    };
    auto wait_for = [&](int which, int target) {
        const unsigned addr = cnt_lds + 4u * which;
        for (;;) {
            int v;
            asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
            if (__builtin_amdgcn_readfirstlane(v) >= target) break;
            __builtin_amdgcn_s_sleep(1);
        }
    };

    // B' fragments.  WC_MFMA16: unit u = 2*s + ch is k-step s (32 deep) of column half ch of this wave's 32 columns; a lane's
    // two output columns are col + 16*ch.  (32x32x16: unit s = k-step s, 16 deep, one column per lane.)
    constexpr int NCH = WC_MFMA16 ? 2 : 1;
    f16x8 bhi[KS], blo[KS];
    float cscale[NCH], addv[NCH], addv_b[NCH], addv_s[NCH];
#pragma unroll
    for (int h = 0; h < NCH; ++h) { cscale[h] = 1.f; addv[h] = 0.f; addv_b[h] = 0.f; addv_s[h] = 0.f; }
    int cur_slot = -1;
    const int l15 = lane & 15, lq = lane >> 4;
    const int col = WC_MFMA16 ? cg * 32 + l15 : cg * 32 + l31;
    auto load_b = [&](int slot) {
        const _Float16* ph = a.Bhi + (int64_t)slot * a.slot_stride + ((int64_t)cg * KS * 64 + lane) * 8;
        const _Float16* pl = a.Blo + (int64_t)slot * a.slot_stride + ((int64_t)cg * KS * 64 + lane) * 8;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            bhi[s] = *reinterpret_cast<const f16x8*>(ph + 512 * s);
            blo[s] = *reinterpret_cast<const f16x8*>(pl + 512 * s);
        }
        const int64_t srow_ = a.slot_stride ? slot : 0;
#pragma unroll
        for (int h = 0; h < NCH; ++h) {
            cscale[h] = a.colscale[srow_ * C + col + 16 * h];
            // loaded unconditionally from a valid address and selected afterwards: a consumed-at-once conditional load
            // would make hipcc drain every load in flight right here
            addv_b[h] = a.bias[a.bias_on ? (int64_t)slot * C + col + 16 * h : 0];
            addv_s[h] = a.sub[a.sub_on ? col + 16 * h : 0];
        }
        cur_slot = slot;
    };
    auto finish_b = [&]() {
#pragma unroll
        for (int h = 0; h < NCH; ++h) addv[h] = (a.bias_on ? addv_b[h] : 0.f) - (a.sub_on ? addv_s[h] : 0.f);
        if (PL) {           // the planes hold os * out: the scale rides in the two per-column constants
#pragma unroll
            for (int h = 0; h < NCH; ++h) { cscale[h] *= os; addv[h] *= os; }
        }
    };

    // prologue: chunks 0..6 requested, then the B' table (KS*2 + <= 3 loads per lane, in flight while tile 0 is
    // converted); each read-out slot is refilled at once (chunks 7..10).  Chunk p's DMA always has six younger DMAs
    // (chunks p+1..6 and the p refills) and the table's loads behind it; a single workgroup tile (n == 1) has fewer DMAs in flight and simply drains.
#pragma unroll
    for (int i = 0; i < NSLOT; ++i) if (i / 4 < n) dma_chunk(i / 4, i % 4, i, lane);
    if (!HAS_SLOT) load_b(0);
    constexpr int TBL = HAS_SLOT ? 0 : 2 * KS + 3 * NCH;      // the table's vector loads per lane: fragments, colscale, bias, sub
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        if (n >= 2) {
            if (p == 0) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(TBL + 6) : "memory");
            if (p == 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(TBL + 6) : "memory");
            if (p == 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(TBL + 6) : "memory");
            if (p == 3) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(TBL + 6) : "memory");
        } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        craw_read(p, lane);
        cv_scale();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the slot has been read out
        if ((p + NSLOT) / 4 < n) dma_chunk((p + NSLOT) / 4, (p + NSLOT) % 4, p, lane);
        cv_hi(); cv_lo(); cv_write(0, p, lane, c4i, woff0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    arrive(0);
    // the BUILTIN wait (not asm): hipcc must see that the B' loads have completed here, or it waits for them with
    // small vmcnt counts inside the loop -- draining the DMAs and stores the loop wants to keep in flight
    __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0) only
    if (!HAS_SLOT) finish_b();

    const int rbase = rg * 32;
#if WC_MFMA16
    // A fragment of 16 rows x 32 k: lane = row l15 (+ 16 for the second half of the block), 16-byte chunk lq of the k-step
    const int sw = swz(rbase + l15);                    // the same for rows l15 and l15 + 16 (and any 32-row group)
    const int rd_off = (rbase + l15) * PITCH + (PAD ? lq * 16 : 0);
    // D register r of block (rh, ch): row 16 rh + 4 lq + r, column 16 ch + l15; after the epilogue's v_permlane16_swap a
    // lane stores rows 16 rh + 8 lh + {0, 4} + r of column l31
    const int out_lane = (rbase + 8 * lh) * C + cg * 32 + l31;
    // planes form: a lane pair exchanges its packed halves (DPP quad_perm [1,0,3,2]) so that the EVEN lane holds row +0's columns
    // (l31, l31 + 1) and the ODD lane row +4's columns (l31 - 1, l31): one dword per lane and plane, 64-byte pieces of a row
    const int pl_lane = (rbase + 8 * lh + 4 * (l31 & 1)) * C + cg * 32 + (l31 & ~1);
    const unsigned pl_sel = (l31 & 1) ? 0x03020706u : 0x05040100u;       // v_perm_b32(neighbour, own, sel)
#else
    const int sw = swz(rbase + l31);
    const int rd_off = (rbase + l31) * PITCH;
    const int out_lane = (rbase + 4 * lh) * C + col;
#endif
    unsigned long long ts[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    bool stamp_on = false;
    unsigned long long clk0 = 0, rt0 = 0;
    if (WC_STAMPS) { clk0 = __builtin_amdgcn_s_memtime(); rt0 = __builtin_amdgcn_s_memrealtime(); }
    if (WC_STAGGER && wave >= 4 && n >= 4) __builtin_amdgcn_s_sleep(24);      // ~half a tile behind waves 0-3

    unsigned long long sum_wait = 0, sum_store = 0;      // WC_STAMPS builds
    int rslot = 4;                     // raw slot of chunk 4(t+1) = chunk 0 of the tile converted during tile t
    int fcur = 0;                      // image buffer of tile t
    using T_ = std::integral_constant<bool, true>;
    using F_ = std::integral_constant<bool, false>;
    // CONV_: a next tile exists and is converted during this one (compile-time, so that no branch splits the gaps'
    // issue groups).
    // WM_ picks the hand-counted wait in front of each chunk read-out (header): 0 vmcnt(0); 1 tile 0 (six younger
    // DMAs, no stores yet); 2 steady state; 3 tile n-3 (>= 4 younger DMAs); 4 tile n-2 (3-p younger DMAs for chunk p).
    auto tile_body = [&](int t, auto conv_tag, auto wm_tag) {
        constexpr bool CONV_ = decltype(conv_tag)::value;
        constexpr int WM_ = decltype(wm_tag)::value;
        stamp_on = WC_STAMPS && (t == 6);
        WC_STAMP(0);
        const int fnext = fcur == 2 ? 0 : fcur + 1;
        // per-tile opaque copies of the lane constants that feed LDS addresses: without them hipcc hoists every
        // address of every instantiated tile body out of the loop (~60 VGPRs) and spills them
        int sw_t = sw, c4i_t = c4i, lane_t = lane, lh_t = lh, woff_t = woff0, lq_t = lq;
        asm volatile("" : "+v"(sw_t), "+v"(c4i_t), "+v"(lane_t), "+v"(lh_t), "+v"(woff_t), "+v"(lq_t));
        (void)lq_t; (void)lh_t;
        unsigned long long w0_ = 0;
        if (WC_STAMPS) w0_ = __builtin_amdgcn_s_memrealtime();
        wait_for(0, 8 * (t + 1));                  // tile t converted by all eight waves (published in mid-loop t-1)
        if (CONV_) wait_for(1, 8 * (t - 1));       // tile t-2 read by all: image buffer (t+1)%3 may be rewritten
        if (WC_STAMPS) sum_wait += __builtin_amdgcn_s_memrealtime() - w0_;
        WC_STAMP(1);
        // raw slots of the four chunks converted during this tile
        int rs[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) { const int v = rslot + p; rs[p] = v >= NSLOT ? v - NSLOT : v; }
        auto chunk_wait = [&](int p) {
            if (WM_ == 2) asm volatile("s_waitcnt vmcnt(22)" ::: "memory");
            else if (WM_ == 3) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
            else if (WM_ == 4 && p == 0) asm volatile("s_waitcnt vmcnt(19)" ::: "memory");
            else if (WM_ == 4 && p == 1) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
            else if (WM_ == 4 && p == 2) asm volatile("s_waitcnt vmcnt(17)" ::: "memory");
            else if (WM_ == 4) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else if (WM_ == 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        };
        const char* hrow = fbuf + fcur * FBUF + rd_off;
        const char* lrow = hrow + IMG;
#if WC_MFMA16
        // 16x16x32: the wave's 32 x 32 block as 2 x 2 accumulators of 16 x 16; per k-step (32 deep) two A fragments (row
        // halves) x hi/lo = 4 ds_read_b128 (as many as two 16-deep steps took) and 12 MFMAs of 16 cycles in the order
        // lo*Hi, hi*Lo, hi*Hi over the four blocks (a block's accumulator is reused every fourth MFMA).  The shape is
        // worth ~10 % of the matrix pipe's time on this power-limited chip.
        f32x4 acc[2][2];
#pragma unroll
        for (int rh = 0; rh < 2; ++rh)
#pragma unroll
            for (int ch = 0; ch < 2; ++ch) acc[rh][ch] = f32x4{0.f, 0.f, 0.f, 0.f};
        constexpr int KS32 = KS / 2 > 0 ? KS / 2 : 1;
        auto frag = [&](const char* base, int rh, int s) {
            if (PAD) return *reinterpret_cast<const f16x8*>(base + rh * (16 * PITCH) + 64 * s);      // lane base + immediate
            return *reinterpret_cast<const f16x8*>(base + rh * (16 * PITCH) + (((4 * s + lq_t) ^ sw_t) * 16));
        };
        f16x8 ah[2], al[2], nh[2], nl[2];
#pragma unroll
        for (int rh = 0; rh < 2; ++rh) { ah[rh] = frag(hrow, rh, 0); al[rh] = frag(lrow, rh, 0); nh[rh] = ah[rh]; nl[rh] = al[rh]; }
        if (CONV_) { chunk_wait(0); craw_read(rs[0], lane_t); }
        WC_STAMP(2);
        // the conversion of tile t+1 as 20 small steps in the first half of the loop (chunk p: scale+refill, next
        // chunk's read-out, split hi, split lo, write), then its publication
        auto cstep = [&](int j) {
            const int p = j / 5, st = j % 5;
            if (st == 0) {
                cv_scale();                           // consumes craw: slot rs[p] is free
                if (p == 0) { if (WC_FENCE_DEP) asm volatile("" :: "v"(cg4[0]) : "memory"); else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }   // chunk 0's ds_read was issued a few instructions ago: it must have RETURNED before the DMA that refills its slot may issue -- with the read only issued, a DMA served from L2 overtook it (one-pass K6, the pair's second workgroup: one corrupted row in ~1e5 tiles)
                const int tl = t + 1 + (p + NSLOT) / 4;
                if (tl < n) dma_chunk(tl, (p + NSLOT) % 4, rs[p], lane_t);
            } else if (st == 1) {
                if (p + 1 < 4) { chunk_wait(p + 1); craw_read(rs[p + 1], lane_t); }
            } else if (st == 2) cv_hi();
            else if (st == 3) cv_lo();
            else cv_write(fnext, p, lane_t, c4i_t, woff_t);
        };
        constexpr int G = 12 * KS32;         // MFMAs = issue gaps per tile
        constexpr int H = G * WC_CONV_NUM / 4;    // the conversion of tile t+1 rides in the first H gaps (WC_CONV_NUM quarters of the loop)
        // one issue gap = one MFMA plus whatever is listed for it; sched_barrier(0) pins the order gap by gap
#pragma unroll
        for (int s = 0; s < KS32; ++s) {
#pragma unroll
            for (int m = 0; m < 12; ++m) {
                const int g = 12 * s + m;
#if WC_M16_ORDER == 1      // row-half major: six MFMAs in a row share the A registers' row half
                const int rh = m / 6, ch = m & 1, u = 2 * s + ch, pr = (m % 6) >> 1;
#else                      // product major: a block's accumulator comes round every fourth MFMA
                const int rh = (m >> 1) & 1, ch = m & 1, u = 2 * s + ch, pr = m >> 2;
#endif
                if (pr == 0) acc[rh][ch] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[rh], bhi[u], acc[rh][ch], 0, 0, 0);
                else if (pr == 1) acc[rh][ch] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[rh], blo[u], acc[rh][ch], 0, 0, 0);
                else acc[rh][ch] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[rh], bhi[u], acc[rh][ch], 0, 0, 0);
                if (s + 1 < KS32) {
                    if (m == 0) nh[0] = frag(hrow, 0, s + 1);
                    if (m == 1) nh[1] = frag(hrow, 1, s + 1);
                    if (m == 2) nl[0] = frag(lrow, 0, s + 1);
                    if (m == 3) nl[1] = frag(lrow, 1, s + 1);
                }
                if (CONV_) {
#pragma unroll
                    for (int j = 0; j < 21; ++j) {
                        if ((j * H) / 21 != g) continue;
                        if (j < 20) cstep(j);
                        else { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); arrive(0); }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int rh = 0; rh < 2; ++rh) { ah[rh] = nh[rh]; al[rh] = nl[rh]; }
        }
#else
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        // fragments are fetched TWO k-steps ahead: with eight waves on the LDS a ds_read_b128 takes longer than the
        // 96 cycles of one k-step's three MFMAs, and an in-order wave that waits for it issues nothing else
        auto frag = [&](const char* base, int s) {
            if (PAD) return *reinterpret_cast<const f16x8*>(base + lh_t * 16 + 32 * s);      // lane base + immediate
            return *reinterpret_cast<const f16x8*>(base + (((2 * s + lh) ^ sw_t) * 16));
        };
        constexpr int AHEAD = WC_AHEAD;
        f16x8 ah = frag(hrow, 0), al = frag(lrow, 0);
        f16x8 bh_ = ah, bl_ = al;
        if (KS > 1 && AHEAD == 2) { bh_ = frag(hrow, 1); bl_ = frag(lrow, 1); }
        if (CONV_) { chunk_wait(0); craw_read(rs[0], lane_t); }
        WC_STAMP(2);
        // the conversion of tile t+1 as 20 small steps in the first half of the loop (chunk p: scale+refill, next
        // chunk's read-out, split hi, split lo, write), then its publication
        auto cstep = [&](int j) {
            const int p = j / 5, st = j % 5;
            if (st == 0) {
                cv_scale();                           // consumes craw: slot rs[p] is free
                if (p == 0) { if (WC_FENCE_DEP) asm volatile("" :: "v"(cg4[0]) : "memory"); else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }   // chunk 0's ds_read was issued a few instructions ago: it must have RETURNED before the DMA that refills its slot may issue -- with the read only issued, a DMA served from L2 overtook it (one-pass K6, the pair's second workgroup: one corrupted row in ~1e5 tiles)
                const int tl = t + 1 + (p + NSLOT) / 4;
                if (tl < n) dma_chunk(tl, (p + NSLOT) % 4, rs[p], lane_t);
            } else if (st == 1) {
                if (p + 1 < 4) { chunk_wait(p + 1); craw_read(rs[p + 1], lane_t); }
            } else if (st == 2) cv_hi();
            else if (st == 3) cv_lo();
            else cv_write(fnext, p, lane_t, c4i_t, woff_t);
        };
        constexpr int G = 3 * KS;            // MFMAs = issue gaps per tile
        constexpr int H = G / 2 > 0 ? G / 2 : 1;
        // one issue gap = one MFMA plus whatever is listed for it; sched_barrier(0) pins the order gap by gap
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            f16x8 nh = bh_, nl = bl_;
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                const int g = 3 * s + m;
                if (m == 0) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bhi[s], acc, 0, 0, 0);
                if (m == 1) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, blo[s], acc, 0, 0, 0);
                if (m == 2) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bhi[s], acc, 0, 0, 0);
                if (s + AHEAD < KS) {
                    if (m == 0) nh = frag(hrow, s + AHEAD);
                    if (m == 1) nl = frag(lrow, s + AHEAD);
                }
                if (CONV_) {
#pragma unroll
                    for (int j = 0; j < 21; ++j) {
                        if ((j * H) / 21 != g) continue;
                        if (j < 20) cstep(j);
                        else { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); arrive(0); }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (AHEAD == 2) { ah = bh_; al = bl_; bh_ = nh; bl_ = nl; }
            else { ah = nh; al = nl; }
        }
#endif
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // my reads of image t are done
        arrive(1);
        WC_STAMP(3);
        // this wave's 32 x 32 block leaves now: its SIMD partner is half a tile away, in the middle of its MFMAs
        float* po = a.out + (int64_t)tile_of(t) * (TR * C) + out_lane;
        _Float16* pph = nullptr; _Float16* ppl = nullptr;
        if (PL) { pph = a.phi + (int64_t)tile_of(t) * (TR * C) + pl_lane; ppl = a.plo + (int64_t)tile_of(t) * (TR * C) + pl_lane; }
        unsigned long long s0_ = 0;
        if (WC_STAMPS) s0_ = __builtin_amdgcn_s_memrealtime();
#if WC_MFMA16
        // A D register holds 4 rows x 16 columns (64-byte pieces of 4 rows).  v_permlane16_swap of the two column halves'
        // registers gives 2 rows x 32 columns per register again -- 128-byte pieces, the store shape of the 32x32 form.
        // One scalar branch per tile, not one per value (the flag is a kernel argument: tested inside the loop it cut the epilogue
        // into eight basic blocks).  ReLU as !(v <= 0) ? v : 0 -- one v_cmp_nle + one v_cndmask, NaN stays NaN -- instead of the
        // four instructions of (v > 0 || isnan(v)).
        auto leave = [&](auto RL_, auto MK_) __attribute__((always_inline)) {
            constexpr bool RL = decltype(RL_)::value;       // the ReLU that follows every WC site rides in the epilogue
            constexpr bool MK = decltype(MK_)::value;       // ... and leaves its gradient mask behind as ONE BIT per element
            unsigned bits = 0;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int rh = i >> 2, r = i & 3;
                float v0 = acc[rh][0][r] * cscale[0] + addv[0], v1 = acc[rh][1][r] * cscale[1] + addv[1];
                if (RL) {
                    v0 = !(v0 <= 0.f) ? v0 : 0.f;
                    v1 = !(v1 <= 0.f) ? v1 : 0.f;
                }
                // (inline asm: with __builtin_amdgcn_permlane16_swap hipcc 7.0 stored the FIRST result twice here -- the second
                // definition of the instruction got lost; s_nop: the operands were just written by VALU instructions)
                asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(v0), "+v"(v1));
                if (MK) {
                    // after the ReLU a value is +0 or passes (positive, or NaN): "not zero" is one v_min_u32 on its bits.  v0 is
                    // row 16 rh + r (+ 8 in lanes 32-63) of the wave's 32, v1 four rows below; column l31
                    const unsigned b0 = __builtin_bit_cast(unsigned, v0), b1 = __builtin_bit_cast(unsigned, v1);
                    bits |= (b0 < 1u ? b0 : 1u) << (16 * rh + r);
                    bits |= (b1 < 1u ? b1 : 1u) << (16 * rh + r + 4);
                }
                if constexpr (PL) {
                    omax = __builtin_fmaxf(omax, __builtin_fmaxf(fabsf(v0), fabsf(v1)));
                    const unsigned H = pk_rne(v0, v1);
                    float r0, r1;
                    asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(H), "v"(v0));
                    asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(H), "v"(v1));
                    const unsigned L = pk_rne(r0, r1);
                    const unsigned Hn = (unsigned)__builtin_amdgcn_update_dpp(0, (int)H, 0xB1, 0xF, 0xF, false);
                    const unsigned Ln = (unsigned)__builtin_amdgcn_update_dpp(0, (int)L, 0xB1, 0xF, 0xF, false);
                    const unsigned oh = __builtin_amdgcn_perm(Hn, H, pl_sel), ol = __builtin_amdgcn_perm(Ln, L, pl_sel);
                    unsigned* qh = reinterpret_cast<unsigned*>(pph + (16 * rh + r) * C);
                    unsigned* ql = reinterpret_cast<unsigned*>(ppl + (16 * rh + r) * C);
#if WC_NT_STORE_PL
                    __builtin_nontemporal_store(oh, qh);
                    __builtin_nontemporal_store(ol, ql);
#else
                    *qh = oh; *ql = ol;
#endif
                    continue;
                }
#if WC_NT_STORE
                __builtin_nontemporal_store(v0, &po[(16 * rh + r) * C]);
                __builtin_nontemporal_store(v1, &po[(16 * rh + r + 4) * C]);
#else
                po[(16 * rh + r) * C] = v0;          // rows r (lanes 0-31) and 8 + r: columns l31
                po[(16 * rh + r + 4) * C] = v1;      // rows 4 + r and 12 + r
#endif
            }
            if (MK) {
                // lanes l and l + 32 hold the two halves of column l31's 32 row bits (rows +0..7, +16..23 | +8..15, +24..31)
                unsigned own = bits, other = bits;
                asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(own), "+v"(other));      // own[32..63] <-> other[0..31]: lanes 0-31 now hold their partner's bits in `other`
                // SGPR base + lane offset (asm: as a 64-bit VGPR address the pointer was spilled and came back behind an
                // s_waitcnt vmcnt(0) once per tile -- the whole DMA ring drained).  One more store per tile than the header's
                // count of 16: a hand-counted wait only gets more conservative by it.
                const unsigned* pm = a.maskout + ((int64_t)tile_of(t) * (TR / 32) + rg) * C;      // wave-uniform
                const unsigned moff = (unsigned)(out_lane & (C - 1)) * 4u, word = own | (other << 8);
                if (lh == 0) asm volatile("s_nop 4\n\tglobal_store_dword %0, %1, %2" :: "v"(moff), "v"(word), "s"(pm) : "memory");
            }
        };
        if (MASK) leave(std::true_type{}, std::true_type{});
        else if (a.relu) leave(std::true_type{}, std::false_type{});
        else leave(std::false_type{}, std::false_type{});
#else
        if (a.relu) {       // the ReLU that follows every WC site rides in the epilogue
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = acc[r] * cscale[0] + addv[0];
                po[((r & 3) + 8 * (r >> 2)) * C] = v > 0.f ? v : (v == v ? 0.f : v);
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) po[((r & 3) + 8 * (r >> 2)) * C] = acc[r] * cscale[0] + addv[0];
        }
#endif
        if (WC_STAMPS) sum_store += __builtin_amdgcn_s_memrealtime() - s0_;
        rslot = rs[3] + 1 >= NSLOT ? rs[3] + 1 - NSLOT : rs[3] + 1;
        fcur = fnext;
        WC_STAMP(4);
    };
    using W0 = std::integral_constant<int, 0>; using W1 = std::integral_constant<int, 1>;
    using W2 = std::integral_constant<int, 2>; using W3 = std::integral_constant<int, 3>;
    using W4 = std::integral_constant<int, 4>;
    auto pick_table = [&](int t) {       // conditional tables: a new slot's B' fragments (the wait is a full drain: rare)
        if (HAS_SLOT) {
            const int slot = a.slot[((int64_t)tile_of(t) * TR) / a.HW];
            if (slot != cur_slot) { load_b(slot); __builtin_amdgcn_s_waitcnt(0x0F70); finish_b(); }
        }
    };
    for (int t = 0; t + 1 < n; ++t) {
        pick_table(t);
        if (t == 0) { if (t + 3 < n) tile_body(t, T_{}, W1{}); else tile_body(t, T_{}, W0{}); }
        else if (t + 3 < n) tile_body(t, T_{}, W2{});
        else if (t + 3 == n) tile_body(t, T_{}, W3{});
        else tile_body(t, T_{}, W4{});
    }
    pick_table(n - 1);
    tile_body(n - 1, F_{}, W0{});
    const bool overflow = !(gmax <= kF16Guard);
    if (WC_STAMPS) { ts[6] = __builtin_amdgcn_s_memtime() - clk0; ts[7] = __builtin_amdgcn_s_memrealtime() - rt0; }
    if (WC_STAMPS && a.dbg && lane == 0 && (blockIdx.x == 0 || blockIdx.x == 100)) {
        unsigned long long* d = a.dbg + ((blockIdx.x ? 1 : 0) * 8 + wave) * 8;
        for (int i = 0; i < 8; ++i) d[i] = ts[i];
    }
    if (WC_STAMPS && a.dbg && lane == 0 && (wave == 0 || wave == 4))
        reinterpret_cast<unsigned*>(a.dbg + 384)[blockIdx.x * 2 + (wave >> 2)] = (unsigned)(sum_wait & 0xFFFF) | ((unsigned)(sum_store & 0xFFFF) << 16);
    if (WC_STAMPS && a.dbg && tid == 0) {      // whole-launch timeline in 10-ns units (s_memrealtime is global)
        const unsigned long long rt1 = rt0 + ts[7], rt_out = __builtin_amdgcn_s_memrealtime(), big = 1ull << 62;
        atomicMax(a.dbg + 128, big - rt_in);        // first workgroup start
        atomicMax(a.dbg + 129, rt_in);              // last workgroup start
        atomicMax(a.dbg + 130, rt0 - rt_in);        // longest prologue
        atomicMax(a.dbg + 131, rt_out - rt1);       // longest epilogue
        atomicMax(a.dbg + 132, big - rt_out);       // first workgroup end
        atomicMax(a.dbg + 133, rt_out);             // last workgroup end
        atomicMax(a.dbg + 134, rt1 - rt0);          // longest tile loop
        atomicMax(a.dbg + 135, big - (rt1 - rt0));  // shortest tile loop
        // per workgroup: loop time | XCC id << 16 | HW_ID[15:8] (cu, sh, se) << 20
        const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);
        const unsigned hwid = __builtin_amdgcn_s_getreg((15 << 11) | (0 << 6) | 4);
        reinterpret_cast<unsigned*>(a.dbg + 256)[blockIdx.x] = (unsigned)((rt1 - rt0) & 0xFFFF) | (xcc << 16) | (((hwid >> 8) & 0xFF) << 20);
    }

    // Exact redo (rare): of the whole workgroup if anything it staged was outside the fp16 range; and, with slots, of
    // every tile that STRADDLES samples of different slots (HW not a multiple of the tile; the MFMA pass above used the
    // table of the tile's first sample for all of its rows).  Same thread, same element as the MFMA pass, so the
    // second store wins by program order.  Here the slot is looked up per ROW.
    if (overflow) cnt[2] = 1;
    __syncthreads();
    const bool redo_all = cnt[2] != 0;
    if (redo_all || (HAS_SLOT && a.mixed)) {
        for (int t = 0; t < n; ++t) {
            const int64_t r0 = (int64_t)tile_of(t) * TR;
            bool mixed = false;
            if (HAS_SLOT && a.mixed) {
                const int64_t n0 = r0 / a.HW, n1 = (r0 + TR - 1) / a.HW;
                const int s0 = a.slot[n0];
                for (int64_t q = n0 + 1; q <= n1; ++q) mixed |= (a.slot[q] != s0);
            }
            if (!redo_all && !mixed) continue;
            const float* xin = a.in + r0 * C;
            float* out_tile = a.out + r0 * C;
            for (int i = 0; i < 16; ++i) {
#if WC_MFMA16
                const int row = rbase + 16 * (i >> 3) + 8 * lh + 4 * ((i >> 2) & 1) + (i & 3), ecol = cg * 32 + l31;
#else
                const int row = rbase + (i & 3) + 8 * (i >> 2) + 4 * lh, ecol = col;
#endif
                int slot = 0;
                if (HAS_SLOT) slot = a.slot[(r0 + row) / a.HW];
                const float* Bf = a.Bf + (int64_t)slot * a.bf_stride + ecol;
                float add = 0.f;
                if (a.bias_on) add += a.bias[(int64_t)slot * C + ecol];
                if (a.sub_on) add -= a.sub[ecol];
                const float* xrow = xin + row * C;
                float accf = 0.f;
                for (int k = 0; k < C; ++k) accf = fmaf(xrow[k] - (a.center ? a.center[k] : 0.f), Bf[(int64_t)k * C], accf);
                {
                    const float v = accf + add;
                    const float o = (MASK || a.relu) ? (!(v <= 0.f) ? v : 0.f) : v;
                    if (PL) {
                        const float so = o * os;
                        const _Float16 h = (_Float16)so;
                        a.phi[(r0 + row) * C + ecol] = h;
                        a.plo[(r0 + row) * C + ecol] = (_Float16)(so - (float)h);
                        omax = __builtin_fmaxf(omax, fabsf(so));
                    } else out_tile[row * C + ecol] = o;
                    if (MASK) {       // the redone element's mask bit (rare path: atomics on the word it shares with 31 rows)
                        unsigned* pm = a.maskout + ((r0 + row) >> 5) * C + ecol;
                        const unsigned bit = 1u << ((r0 + row) & 31);
                        if (__builtin_bit_cast(unsigned, o) != 0u) atomicOr(pm, bit); else atomicAnd(pm, ~bit);
                    }
                }
            }
        }
    }
    if (PL && a.oamax) {        // this workgroup's max |scaled output| (one partial per workgroup: deterministic, nothing to clear)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) omax = __builtin_fmaxf(omax, __shfl_xor(omax, o));
        if (lane == 0) cnt[4 + wave] = __builtin_bit_cast(int, omax);
        __syncthreads();
        if (tid == 0) {
            float m = 0.f;
            for (int w = 0; w < 8; ++w) m = __builtin_fmaxf(m, __builtin_bit_cast(float, (int)cnt[4 + w]));
            a.oamax[blockIdx.x] = m;
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------
// K6 in ONE pass at C = 256:  dx = [gy | x - mu] . [At[slot] ; S] - gmean,  a contraction over K = 512.
// The split tables of both streams are 512 KB -- the register file of a whole CU -- so a workgroup owns HALF of the output
// columns (8 waves x 16 columns, 128 VGPRs of B' fragments each: 8 k-steps of At, 8 of S) and two workgroups share a row
// tile: the pair sits on one XCD (block ids 8 apart), the second reads the tile from L2, both convert it (the price: every
// row is converted four times over the pass instead of twice over two passes).  What it buys is the traffic: the two-pass
// form writes dx, reads it back and writes it again right behind the first pass's write-back (670 MB, the accumulating
// pass at 3.3 TB/s); this form reads gy and x and writes dx once (402 MB from HBM, plain stores).
// Geometry: 16-row tiles (32 KB of fp32 input: 16 rows of gy and of x; wave w brings in rows 2w, 2w+1 of both with its
// four 1-KiB DMA chunks), fp16 images of [16 rows][512 channels] hi | lo with a 32-byte row pad (conflict-free 16x16x32
// A-fragment reads), three image buffers, the raw ring and the hand-off counters exactly as in affine_ring_kernel; per
// tile and wave 16 k-steps x 3 v_mfma_f32_16x16x32_f16 on SIX accumulators (table x product: a dependent MFMA comes round
// every third issue), summed in the epilogue with the two tables' column scales.  vmcnt by hand as there, with 4 stores per
// tile instead of 16: 6 younger DMAs + 4 stores = vmcnt(10) in steady state.
// ---------------------------------------------------------------------------------------------------------------
struct OnePassArgs {
    const float* gy; const float* x; const float* mu;      // (M, 256) row-major; mu [256]
    const float* sg; const float* sx;                      // per-channel power-of-two scales of gy and of x - mu
    const _Float16* hi0; const _Float16* lo0; const float* cs0; int64_t slot_stride0;   // At tables (per slot) + column scales
    const _Float16* hi1; const _Float16* lo1; const float* cs1;                        // S table (shared)
    const float* sub; int sub_on;                          // gmean (pointed at a valid address and ignored when absent)
    const int32_t* slot; int64_t HW;
    const float* Bf0; int64_t bf0_stride; const float* Bf1;   // the fp32 tables, for the exact redo
    float* out;
    int ntiles, tiles_per_pair, mixed;
    const unsigned* gmask;        // MK kernels: the ReLU's one-bit mask of the site (wc_apply_mask_f32 layout); gy is masked while it is converted
    const _Float16* xhi; const _Float16* xlo;      // XPL kernels: x as pre-split planes (wc_resadd.hip / wc_split.hip), sx = their scales, `sub` folded
};

// MK: gy is the gradient BEFORE the site's ReLU and a.gmask the activation's one-bit mask: the bits are applied while the gy chunks
// are converted (K4 then has no masked copy to write).  A tile's 16 rows sit in one 32-row block of the mask =
// 1 KiB of words, the same for all eight waves: it travels by LDS-DMA into one of three shared 1-KiB buffers behind the counters,
// each wave bringing in 128 bytes of it THREE tiles ahead; a wave's piece has landed before its arrive(0) of the following tile (one
// counted wait on a DMA a tile and a half old), so the existing "converted" counter also says "the next tile's mask block is complete", and the buffer is rewritten only
// after every wave has passed the next tile's wait on that counter.  (A register-destination load was tried first: as inline asm
// its results land asynchronously in registers hipcc may have moved meanwhile -- a memory fault at 128x32x32x256 and one wrong row
// pair in 500 launches at 128x16x16x256; as a plain load hipcc waits for it with the DMAs' counter.)
// XPL (round 4): x arrives as PRE-SPLIT planes (the residual add in front of the site wrote them: wc_resadd.hip), x ~= center + (hi + lo) / sx.
// A chunk of x is then [hi row | lo row] of one row -- lanes 0-31 fetch 16 bytes of the hi plane, lanes 32-63 of the lo plane: the same
// 1 KiB per row as the fp32 form -- and its "conversion" is a copy from the raw slot into the two images (one ds_read_b128, one
// ds_write_b128: none of the scale / split instructions, half of this kernel's conversion work).  The term (center - mu) S is folded
// into `sub` by the caller, the S table is built for the planes' scales; ring, counters and the hand-counted waits are unchanged.
template <bool HAS_SLOT, bool MK = false, bool XPL = false>
__global__ __launch_bounds__(512, 1) void onepass_ring_kernel(OnePassArgs a)
{
    constexpr int C = 256, TR = 16, K2 = 2 * C;
    constexpr int PAD = 32, PITCH = K2 * 2 + PAD, IMG = TR * PITCH, FBUF = 2 * IMG;
    constexpr int NSLOT = 7, RAWW = NSLOT * 1024, KS = 16;            // k-steps of 32: 8 of gy . At, 8 of (x - mu) . S
    extern __shared__ __attribute__((aligned(16))) char smem[];      // F0 | F1 | F2 | raw[8 waves][7 KiB] | counters
    char* const fbuf = smem;
    char* const ring = smem + 3 * FBUF;
    volatile int* const cnt = reinterpret_cast<volatile int*>(smem + 3 * FBUF + 8 * RAWW);   // [0] converted, [1] read, [2] dirty

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lq = lane >> 4;
    // block id -> (pair, column half): ids q*16 + xcd and q*16 + 8 + xcd are the two halves of pair q*8 + xcd (same XCD)
    const int half = (blockIdx.x >> 3) & 1;
    const int pair = (blockIdx.x >> 4) * 8 + (blockIdx.x & 7);
    const int npairs = gridDim.x >> 1;

    int t_first, t_stride, n;
    if (HAS_SLOT) {
        t_first = pair * a.tiles_per_pair; t_stride = 1;
        n = a.ntiles - t_first; if (n > a.tiles_per_pair) n = a.tiles_per_pair;
    } else {
        t_first = pair; t_stride = npairs;
        n = (a.ntiles - t_first + t_stride - 1) / t_stride;
    }
    if (n <= 0) return;
    auto tile_of = [&](int i) { return t_first + i * t_stride; };

    if (tid < 3) cnt[tid] = 0;
    __syncthreads();

    // DMA of chunk p of this wave's share of tile tl: p = 0, 1 rows 2w, 2w+1 of gy; p = 2, 3 the same rows of x
    const unsigned ring_w = (unsigned)(size_t)((__attribute__((address_space(3))) char*)(ring + wave * RAWW));
    auto dma_chunk = [&](int tl, int p, int slot, int lane_) {
        const float* src = (p & 2) ? a.x : a.gy;
        const int64_t row_ = (int64_t)tile_of(tl) * TR + 2 * wave + (p & 1);
        const char* g = reinterpret_cast<const char*>(src + row_ * C) + lane_ * 16;
        if (XPL && ((p & 2) || (WC_K6_ABL & 512)))      // [hi row | lo row]: 512 bytes of each plane  (ABL 512, timing only: the gradient's chunks too)
            g = reinterpret_cast<const char*>(((lane_ >> 5) ? a.xlo : a.xhi) + row_ * C) + (lane_ & 31) * 16;
        const unsigned l = __builtin_amdgcn_readfirstlane(ring_w + slot * 1024);
        unsigned keep;
        if ((WC_K6_ABL & 256) && (p & 2)) return;   // (development, timing only: no x chunks at all)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(g), "s"(l) : "memory");
    };

    // conversion of this wave's own chunks: lane = float4 of channels 4*lane.. of one row
    const f32x4 scl_g = ld4f(a.sg + 4 * lane), scl_x = ld4f(a.sx + 4 * lane);
    const f32x4 ncs_x = XPL ? scl_x : -ld4f(a.mu + 4 * lane) * scl_x;      // (XPL: unused)
    float gmax = 0.f;
    f32x4 craw, cg4;
    unsigned chw01 = 0, chw23 = 0, clw01 = 0, clw23 = 0;
    auto craw_read = [&](int slot, int lane_) {
        craw = *reinterpret_cast<const f32x4*>(ring + wave * RAWW + slot * 1024 + lane_ * 16);
    };
    // the mask words of this lane's four channels for the tile being converted (MK), and the bit of row 2w of that tile
    // MK: the mask block of tile tl -> shared buffer tl & 1; this wave's 128 bytes (lanes 0-7)
    char* const mbuf = smem + 3 * FBUF + 8 * RAWW + 64;
    const unsigned mb_w = (unsigned)(size_t)((__attribute__((address_space(3))) char*)(mbuf + wave * 128));
    auto mask_dma = [&](int tl) {
        if (lane < 8) {
            const char* g = reinterpret_cast<const char*>(a.gmask + (int64_t)(tile_of(tl) >> 1) * C) + wave * 128 + lane * 16;
            const unsigned l = __builtin_amdgcn_readfirstlane(mb_w + (tl % 3) * 1024);
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(g), "s"(l) : "memory");
        }
    };
    u32x4_ mkw = {0u, 0u, 0u, 0u};          // MK: this lane's four channels' mask words of the tile being converted (read once per tile)
    auto mask_words = [&](int tl) { mkw = *reinterpret_cast<const u32x4_*>(mbuf + (tl % 3) * 1024 + lane * 16); };
    auto cv_scale = [&](int p, int tl) {
        if (XPL && ((p & 2) || (WC_K6_ABL & 512))) {       // planes: the chunk IS the image's content; held until its write (craw is the next chunk's by then)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the read has RETURNED before the slot's refill may issue (no arithmetic here would make hipcc wait)
            cg4 = craw;
            return;
        }
        if (p & 2) cg4 = craw * scl_x + ncs_x;
        else {
            cg4 = craw * scl_g;
            if (MK) {
                const u32x4_ m = mkw;
                const int bit = __builtin_amdgcn_readfirstlane(((tile_of(tl) & 1) << 4) + 2 * wave + (p & 1));
                // (elements copied to scalars first: __builtin_bit_cast on an ext-vector element read element 0 for every j, hipcc 7.0)
                const float e0 = cg4[0], e1 = cg4[1], e2 = cg4[2], e3 = cg4[3];
                const unsigned w0 = m[0], w1 = m[1], w2 = m[2], w3 = m[3];
                cg4[0] = __builtin_bit_cast(float, __builtin_bit_cast(int, e0) & __builtin_amdgcn_sbfe((int)w0, bit, 1));
                cg4[1] = __builtin_bit_cast(float, __builtin_bit_cast(int, e1) & __builtin_amdgcn_sbfe((int)w1, bit, 1));
                cg4[2] = __builtin_bit_cast(float, __builtin_bit_cast(int, e2) & __builtin_amdgcn_sbfe((int)w2, bit, 1));
                cg4[3] = __builtin_bit_cast(float, __builtin_bit_cast(int, e3) & __builtin_amdgcn_sbfe((int)w3, bit, 1));
            }
        }
        gmax = __builtin_fmaxf(__builtin_fmaxf(gmax, fabsf(cg4[0])), fabsf(cg4[1]));
        gmax = __builtin_fmaxf(__builtin_fmaxf(gmax, fabsf(cg4[2])), fabsf(cg4[3]));
    };
    auto cv_hi = [&](int p) { if (XPL && ((p & 2) || (WC_K6_ABL & 512))) return; chw01 = pk_rne(cg4[0], cg4[1]); chw23 = pk_rne(cg4[2], cg4[3]); };
    auto cv_lo = [&](int p) {
        if (XPL && ((p & 2) || (WC_K6_ABL & 512))) return;
        float r0, r1, r2, r3;
        asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(chw01), "v"(cg4[0]));
        asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(chw01), "v"(cg4[1]));
        asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r2) : "v"(chw23), "v"(cg4[2]));
        asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r3) : "v"(chw23), "v"(cg4[3]));
        clw01 = pk_rne(r0, r1);
        clw23 = pk_rne(r2, r3);
    };
    // image row 2w + (p & 1); gy in channels 0..255 of the row, x in 256..511
    const int woff0 = (2 * wave) * PITCH + lane * 8;
    const int xoff0 = (2 * wave) * PITCH + C * 2 + (lane & 31) * 16 + (lane >> 5) * IMG;      // XPL: this lane's 16 bytes of the x half (hi image | lo image)
    auto cv_write = [&](int fb, int p, int woff0_, int xoff0_) {
        if (XPL && ((p & 2) || (WC_K6_ABL & 512))) {
            *reinterpret_cast<f32x4*>(fbuf + fb * FBUF + xoff0_ - ((p & 2) ? 0 : C * 2) + (p & 1) * PITCH) = cg4;
            return;
        }
        char* dst = fbuf + fb * FBUF + woff0_ + (p & 1) * PITCH + (p >> 1) * (C * 2);
        *reinterpret_cast<uint2*>(dst) = make_uint2(chw01, chw23);
        *reinterpret_cast<uint2*>(dst + IMG) = make_uint2(clw01, clw23);
    };
    const unsigned cnt_lds = (unsigned)(size_t)((__attribute__((address_space(3))) char*)(smem + 3 * FBUF + 8 * RAWW));
    auto arrive = [&](int which) {
        if (lane == 0) {
            const unsigned one = 1u, addr = cnt_lds + 4u * which;
            asm volatile("ds_add_u32 %0, %1" :: "v"(addr), "v"(one) : "memory");
        }
    };
    auto wait_for = [&](int which, int target) {
        if (WC_K6_ABL & 64) return;         // (development: no hand-off waits at all)
        const unsigned addr = cnt_lds + 4u * which;
        for (;;) {
            int v;
            asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
            if (__builtin_amdgcn_readfirstlane(v) >= target) break;
            __builtin_amdgcn_s_sleep(1);
        }
    };

    auto wait_both = [&](int target0, int target1) {      // cnt[0] >= target0 and cnt[1] >= target1
        if (WC_K6_ABL & 64) return;
        for (;;) {
            int2 v;
            asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(cnt_lds) : "memory");
            if (__builtin_amdgcn_readfirstlane(v.x) >= target0 && __builtin_amdgcn_readfirstlane(v.y) >= target1) break;
            __builtin_amdgcn_s_sleep(1);
        }
    };

    // B' fragments of this wave's 16 columns: units 0..7 the k-steps of At[slot], 8..15 those of S (16x16x32 load order)
    f16x8 bhi[KS], blo[KS];
    float cs0v = 1.f, cs1v = 1.f, subv = 0.f;
    int cur_slot = -1;
    const int cb16 = half * 8 + wave, cgp = cb16 >> 1, chh = cb16 & 1;
    const int col = cb16 * 16 + l15;
    const int64_t unit0 = ((int64_t)(cgp * 8) * 2 + chh) * 512 + lane * 8;        // + s * 1024 halves per k-step
    auto load_b0 = [&](int slot) {
        const _Float16* ph = a.hi0 + (int64_t)slot * a.slot_stride0 + unit0;
        const _Float16* pl = a.lo0 + (int64_t)slot * a.slot_stride0 + unit0;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            bhi[s] = *reinterpret_cast<const f16x8*>(ph + 1024 * s);
            blo[s] = *reinterpret_cast<const f16x8*>(pl + 1024 * s);
        }
        cs0v = a.cs0[(a.slot_stride0 ? (int64_t)slot : 0) * C + col];
        cur_slot = slot;
    };
    auto load_b1 = [&]() {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            bhi[8 + s] = *reinterpret_cast<const f16x8*>(a.hi1 + unit0 + 1024 * s);
            blo[8 + s] = *reinterpret_cast<const f16x8*>(a.lo1 + unit0 + 1024 * s);
        }
        cs1v = a.cs1[col];
        subv = a.sub[a.sub_on ? col : 0];
    };

    if (MK) {       // the blocks of tiles 0, 1 and 2: older than every DMA below, landed at the first chunk's wait; the barrier there
        mask_dma(0);    // makes all eight waves' pieces visible before anyone converts (once per launch)
        if (n > 1) mask_dma(1);
        if (n > 2) mask_dma(2);
    }
#pragma unroll
    for (int i = 0; i < NSLOT; ++i) if (i / 4 < n) dma_chunk(i / 4, i % 4, i, lane);
    load_b1();
    if (!HAS_SLOT) load_b0(0);
    constexpr int TBL = 2 * 8 + 2 + (HAS_SLOT ? 0 : 2 * 8 + 1);      // the tables' vector loads per lane behind the first DMAs
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        if (n >= 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(TBL + 6) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (MK && p == 0) { __syncthreads(); mask_words(0); }
        craw_read(p, lane);
        cv_scale(p, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the slot has been read out
        if ((p + NSLOT) / 4 < n) dma_chunk((p + NSLOT) / 4, (p + NSLOT) % 4, p, lane);
        cv_hi(p); cv_lo(p); cv_write(0, p, woff0, xoff0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    arrive(0);
    __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0) only: hipcc must see that the table loads have completed
    if (!a.sub_on) subv = 0.f;

    const int rd_off = l15 * PITCH + lq * 16;
    const int out_lane = (4 * lq) * C + col;          // D register r: row 4 lq + r, column l15 of this wave's 16
    if (WC_STAGGER && wave >= 4 && n >= 4) __builtin_amdgcn_s_sleep(12);

    int rslot = 4;
    int fcur = 0;
    int2 pre = {0, 0};                  // counters 0 and 1 as read at the end of the previous tile's body
    using T_ = std::integral_constant<bool, true>;
    using F_ = std::integral_constant<bool, false>;
    auto tile_body = [&](int t, auto conv_tag, auto wm_tag) {
        constexpr bool CONV_ = decltype(conv_tag)::value;
        constexpr int WM_ = decltype(wm_tag)::value;
        const int fnext = fcur == 2 ? 0 : fcur + 1;
        int lane_t = lane, woff_t = woff0, rd_t = rd_off, xoff_t = xoff0;
        asm volatile("" : "+v"(lane_t), "+v"(woff_t), "+v"(rd_t), "+v"(xoff_t));
        // counter 0: tile t converted by all (MK: also "tile t + 1's mask block is complete", and every wave is done with tile t's);
        // counter 1 (CONV_): image t - 2 read by all.  ONE 8-byte read for both: a poll is an LDS round trip with nothing to hide it
        // ... and that read is issued at the END of the previous tile's body, in front of its stores: the counters only grow, so a value
        // that already satisfies both targets needs no poll at all (the common case: the arrivals are three quarters of a loop old)
        if (!(__builtin_amdgcn_readfirstlane(pre.x) >= 8 * (t + 1) && __builtin_amdgcn_readfirstlane(pre.y) >= (CONV_ ? 8 * (t - 1) : 0)))
            wait_both(8 * (t + 1), CONV_ ? 8 * (t - 1) : 0);
        // MK: this wave's piece of the block of tile t + 3 into the buffer tile t's block has just left (three buffers: the piece is
        // waited for at the NEXT tile's arrival, by when it is a tile and a half old).  One more DMA per tile than the waits below
        // count: they only get more conservative by it
        if (MK && CONV_) { if (t + 3 < n) mask_dma(t + 3); mask_words(t + 1); }
        int rs[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) { const int v = rslot + p; rs[p] = v >= NSLOT ? v - NSLOT : v; }
        auto chunk_wait = [&](int p) {      // younger vector-memory operations behind chunk p's DMA (header)
            if (WC_K6_ABL & 32) return;     // (development: what the tile chain costs when no load is ever waited for)
            if (WM_ == 2) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
            else if (WM_ == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (WM_ == 4 && p == 0) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
            else if (WM_ == 4 && p == 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else if (WM_ == 4 && p == 2) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
            else if (WM_ == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if (WM_ == 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        };
        const char* hrow = fbuf + fcur * FBUF + rd_t;
        const char* lrow = hrow + IMG;
        f32x4 acc[2][3];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        auto frag = [&](const char* base, int s) { return *reinterpret_cast<const f16x8*>(base + 64 * s); };
        f16x8 ah = frag(hrow, 0), al = frag(lrow, 0), nh = ah, nl = al;
        if (CONV_) { chunk_wait(0); craw_read(rs[0], lane_t); }
        auto cstep = [&](int j) {
            const int p = j / 5, st = j % 5;
            if (st == 0) {
                cv_scale(p, t + 1);
                if (p == 0) { if (WC_FENCE_DEP) asm volatile("" :: "v"(cg4[0]) : "memory"); else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }   // chunk 0's ds_read was issued a few instructions ago: it must have RETURNED before the DMA that refills its slot may issue -- with the read only issued, a DMA served from L2 overtook it (one-pass K6, the pair's second workgroup: one corrupted row in ~1e5 tiles)
                const int tl = t + 1 + (p + NSLOT) / 4;
                if (tl < n) dma_chunk(tl, (p + NSLOT) % 4, rs[p], lane_t);
            } else if (st == 1) {
                if (p + 1 < 4) { chunk_wait(p + 1); craw_read(rs[p + 1], lane_t); }
            } else if (WC_K6_ABL & 8) {
            } else if (st == 2) cv_hi(p);
            else if (st == 3) cv_lo(p);
            else cv_write(fnext, p, woff_t, xoff_t);
        };
        constexpr int G = 3 * KS;            // 48 MFMAs = issue gaps per tile
        constexpr int H = (G * 3) / 4;       // the next tile's conversion rides in the first three quarters
#pragma unroll
        for (int s = 0; s < KS; ++s) {
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                const int g = 3 * s + m, tb = s >> 3;
                if (WC_K6_ABL & 2) { asm volatile("" :: "v"(al), "v"(ah), "v"(bhi[s]), "v"(blo[s])); }
                else {
                if (m == 0) acc[tb][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bhi[s], acc[tb][0], 0, 0, 0);
                if (m == 1) acc[tb][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, blo[s], acc[tb][1], 0, 0, 0);
                if (m == 2) acc[tb][2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bhi[s], acc[tb][2], 0, 0, 0);
                }
                if (s + 1 < KS && !(WC_K6_ABL & 4)) {
                    if (m == 0) nh = frag(hrow, s + 1);
                    if (m == 1) nl = frag(lrow, s + 1);
                }
                if (CONV_) {
#pragma unroll
                    for (int j = 0; j < 21; ++j) {
                        if ((j * H) / 21 != g) continue;
                        if (j < 20) cstep(j);
                        else {
                            // MK: the mask DMA of the PREVIOUS tile's start (the block of tile t + 2) must have landed before this arrival,
                            // which publishes it: behind it sit that tile's 4 DMAs and 4 stores, this tile's mask DMA and its 4 DMAs (one
                            // DMA and no mask DMA at n - 3); tile 0 publishes what the prologue loaded
                            if (MK && (WM_ == 1 || WM_ == 2)) asm volatile("s_waitcnt vmcnt(13)" ::: "memory");
                            else if (MK && WM_ == 3) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
                            else if (MK && WM_ == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); arrive(0);
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            ah = nh; al = nl;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // my reads of image t are done
        arrive(1);
        {       // (a plain 8-byte load: hipcc waits for it where the next tile's body reads it)
            typedef int i32x2_ __attribute__((ext_vector_type(2)));
            const i32x2_ pv = *reinterpret_cast<const volatile i32x2_*>(cnt);
            pre.x = pv[0]; pre.y = pv[1];
        }
        float* po = a.out + (int64_t)tile_of(t) * (TR * C) + out_lane;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float v0 = (acc[0][0][r] + acc[0][1][r]) + acc[0][2][r], v1 = (acc[1][0][r] + acc[1][1][r]) + acc[1][2][r];
#if WC_K6_ABL & 1
            asm volatile("" :: "v"(v0 * cs0v + (v1 * cs1v - subv)), "v"(po));
#elif WC_NT_STORE_K6
            __builtin_nontemporal_store(v0 * cs0v + (v1 * cs1v - subv), &po[r * C]);
#else
            po[r * C] = v0 * cs0v + (v1 * cs1v - subv);
#endif
        }
        rslot = rs[3] + 1 >= NSLOT ? rs[3] + 1 - NSLOT : rs[3] + 1;
        fcur = fnext;
    };
    using W0 = std::integral_constant<int, 0>; using W1 = std::integral_constant<int, 1>;
    using W2 = std::integral_constant<int, 2>; using W3 = std::integral_constant<int, 3>;
    using W4 = std::integral_constant<int, 4>;
    auto pick_table = [&](int t) {
        if (HAS_SLOT) {
            const int slot = a.slot[((int64_t)tile_of(t) * TR) / a.HW];
            if (slot != cur_slot) { load_b0(slot); __builtin_amdgcn_s_waitcnt(0x0F70); }
        }
    };
    for (int t = 0; t + 1 < n; ++t) {
        pick_table(t);
        if (t == 0) { if (t + 3 < n) tile_body(t, T_{}, W1{}); else tile_body(t, T_{}, W0{}); }
        else if (t + 3 < n) tile_body(t, T_{}, W2{});
        else if (t + 3 == n) tile_body(t, T_{}, W3{});
        else tile_body(t, T_{}, W4{});
    }
    pick_table(n - 1);
    tile_body(n - 1, F_{}, W0{});

    // Exact redo (rare), as in affine_ring_kernel: the workgroup's tiles again in fp32 from global memory when anything it
    // staged was outside the fp16 range; tiles that straddle samples of different slots per row.  Same thread, same element.
    if (!(gmax <= kF16Guard)) cnt[2] = 1;
    __syncthreads();
    const bool redo_all = cnt[2] != 0;
    if (redo_all || (HAS_SLOT && a.mixed)) {
        for (int t = 0; t < n; ++t) {
            const int64_t r0 = (int64_t)tile_of(t) * TR;
            bool mixed = false;
            if (HAS_SLOT && a.mixed) {
                const int64_t n0 = r0 / a.HW, n1 = (r0 + TR - 1) / a.HW;
                const int s0 = a.slot[n0];
                for (int64_t q = n0 + 1; q <= n1; ++q) mixed |= (a.slot[q] != s0);
            }
            if (!redo_all && !mixed) continue;
            for (int r = 0; r < 4; ++r) {
                const int64_t row = r0 + 4 * lq + r;
                int slot = 0;
                if (HAS_SLOT) slot = a.slot[row / a.HW];
                const float* B0 = a.Bf0 + (int64_t)slot * a.bf0_stride + col;
                const float* B1 = a.Bf1 + col;
                const float* gr = a.gy + row * C;
                const float* xr = XPL ? a.gy : a.x + row * C;
                float accf = 0.f;
                if (MK) {
                    const unsigned* mr = a.gmask + (row >> 5) * C;
                    const int bit = (int)(row & 31);
                    for (int k = 0; k < C; ++k) accf = fmaf(((mr[k] >> bit) & 1u) ? gr[k] : 0.f, B0[(int64_t)k * C], accf);
                } else
                for (int k = 0; k < C; ++k) accf = fmaf(gr[k], B0[(int64_t)k * C], accf);
                if (XPL) {          // (the planes' value / sx; centre and mu are in the folded `sub`)
                    const _Float16* xh = a.xhi + row * C;
                    const _Float16* xl = a.xlo + row * C;
                    for (int k = 0; k < C; ++k) accf = fmaf(((float)xh[k] + (float)xl[k]) / a.sx[k], B1[(int64_t)k * C], accf);
                } else
                for (int k = 0; k < C; ++k) accf = fmaf(xr[k] - a.mu[k], B1[(int64_t)k * C], accf);
                a.out[row * C + col] = accf - (a.sub_on ? a.sub[col] : 0.f);
            }
        }
    }
}

template <int C>
hipError_t launch_affine_ring(const FastArgs& a, hipStream_t st)
{
    constexpr int TR = 8192 / C;
    constexpr size_t lds = 3 * 2 * (size_t)(TR * (C * 2 + ((C == 256 && WC_MFMA16) ? 32 : (C >= 128 ? 16 : 0)))) + 8 * 7 * 1024 + 64;   // 3 image buffers + 8 x 7 raw chunk slots + counters
    FastArgs b = a;
    b.bias_on = a.bias != nullptr; b.sub_on = a.sub != nullptr;
    if (!b.bias_on) b.bias = a.scale;       // any valid address: loaded and ignored
    if (!b.sub_on) b.sub = a.scale;
    b.ntiles = (int)(a.M / TR);
    int nwg = b.ntiles < 256 ? b.ntiles : 256;
    b.tiles_per_wg = (b.ntiles + nwg - 1) / nwg;
    nwg = (b.ntiles + b.tiles_per_wg - 1) / b.tiles_per_wg;
#define WC_LAUNCH_RING(SLOT_, MASK_, PL_)                                                                                \
    do {                                                                                                                \
        static bool attr_set = false;                                                                                   \
        if (!attr_set) {                                                                                                \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(affine_ring_kernel<C, SLOT_, MASK_, PL_>), \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                   \
            if (e != hipSuccess) return e;                                                                              \
            attr_set = true;                                                                                            \
        }                                                                                                               \
        hipLaunchKernelGGL((affine_ring_kernel<C, SLOT_, MASK_, PL_>), dim3(nwg), dim3(512), lds, st, b);               \
    } while (0)
    const bool mask = WC_MFMA16 && b.maskout != nullptr && b.relu;
    if (b.phi != nullptr) {
        // planes form (C = 128 | 256): the pass itself, then the same kernel behind the gate (leaves at once unless the predicted
        // scale overflowed).  oscale: see wc_launch_out_scale
        if (!WC_MFMA16 || (C != 128 && C != 256) || nwg > 1024) return hipErrorInvalidValue;
        if constexpr (WC_MFMA16 && (C == 128 || C == 256)) {
            b.oamax = b.oscale + 2 + kPlaneBounds; b.gate = nullptr; b.ngate = nwg;
            for (int pass = 0; pass < 2; ++pass) {
                if (b.slot != nullptr) { if (mask) WC_LAUNCH_RING(true, true, true); else WC_LAUNCH_RING(true, false, true); }
                else { if (mask) WC_LAUNCH_RING(false, true, true); else WC_LAUNCH_RING(false, false, true); }
                b.gate = b.oscale + 2 + kPlaneBounds; b.oamax = nullptr;
            }
        }
        return hipGetLastError();
    }
    if (b.slot != nullptr) { if (mask) WC_LAUNCH_RING(true, true, false); else WC_LAUNCH_RING(true, false, false); }
    else { if (mask) WC_LAUNCH_RING(false, true, false); else WC_LAUNCH_RING(false, false, false); }
#undef WC_LAUNCH_RING
    return hipGetLastError();
}

template <int C>
hipError_t launch_affine(const FastArgs& a, hipStream_t st)
{
    constexpr int BM = 64 * 256 / C;
    constexpr size_t lds = (size_t)4 * BM * C * 2 + 16;     // 128 KiB of images + the two dirty tags
    FastArgs b = a;
    b.ntiles = (int)(a.M / BM);
    int nwg = b.ntiles < 256 ? b.ntiles : 256;
    b.tiles_per_wg = (b.ntiles + nwg - 1) / nwg;
    nwg = (b.ntiles + b.tiles_per_wg - 1) / b.tiles_per_wg;
#define WC_LAUNCH_AFFINE(ACC_, SLOT_)                                                                                   \
    do {                                                                                                                \
        static bool attr_set = false;                                                                                   \
        if (!attr_set) {                                                                                                \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(affine_f16x3_kernel<C, ACC_, SLOT_>),      \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                   \
            if (e != hipSuccess) return e;                                                                              \
            attr_set = true;                                                                                            \
        }                                                                                                               \
        hipLaunchKernelGGL((affine_f16x3_kernel<C, ACC_, SLOT_>), dim3(nwg), dim3(512), lds, st, b);                    \
    } while (0)
    const bool has_slot = b.slot != nullptr;
    if (b.accumulate) { if (has_slot) WC_LAUNCH_AFFINE(true, true); else WC_LAUNCH_AFFINE(true, false); }
    else { if (has_slot) WC_LAUNCH_AFFINE(false, true); else WC_LAUNCH_AFFINE(false, false); }
#undef WC_LAUNCH_AFFINE
    return hipGetLastError();
}

}  // namespace

int64_t wc_fast_affine_min_rows()
{
    // development: WC_AFFINE_MIN_ROWS overrides the smallest M the split-fp16 apply takes
    static const int64_t v = getenv("WC_AFFINE_MIN_ROWS") ? atoll(getenv("WC_AFFINE_MIN_ROWS")) : WC_AFFINE_MIN_ROWS;
    return v;
}

static bool use_ring()
{
    static const bool on = getenv("WC_NO_RING") == nullptr;      // development: the register-staged kernel everywhere
    return on;
}

bool wc_fast_affine_supported(int64_t N, int64_t HW, int C, bool has_slot)
{
    if (!(C == 32 || C == 64 || C == 128 || C == 256)) return false;
    const int64_t M = N * HW;
    if (M < wc_fast_affine_min_rows()) return false;
    const int BM = 64 * 256 / C;
    // slots: the ring kernel (every non-accumulating call) redoes tiles that straddle samples of different slots; the
    // register-staged kernel cannot, so without the ring a tile must not straddle two samples
    if (has_slot && !use_ring() && (HW % BM) != 0) return false;
    if ((M % BM) != 0) return false;                         // whole tiles only (keeps the kernel free of masked accesses)
    return true;
}

// does wc_launch_fast_affine_planned(relu, relu_mask) write the mask itself?  (the planned ring kernel on the 16x16x32 shape)
bool wc_fast_affine_writes_mask(int64_t N, int64_t HW, int C)
{
    return WC_MFMA16 && use_ring() && wc_fast_affine_supported(N, HW, C, false) && ((N * HW) % (8192 / C)) == 0;
}

// ... and can it leave the output as the next convolution's planes (wc_launch_fast_affine_planned(planes, oscale))?
bool wc_fast_affine_writes_planes(int64_t N, int64_t HW, int C)
{
    return (C == 128 || C == 256) && wc_fast_affine_writes_mask(N, HW, C) && (N * HW) / (8192 / C) <= 1024 * 1024;
}

// Predicted output scale of a site (device scalar, no pass over data): the whitened activation has unit covariance, so output
// channel c of y = xhat Gamma_k + beta_k has standard deviation |Gamma_k[:, c]|; the scale puts max_c,k (|beta| + 16 sigma) into
// [2^13, 2^14) -- the convolution's own target range for max |y| (csrc/wc_conv.hip, scale_of) -- which leaves fp16 a factor of
// 4 to 8 of headroom above a 16-sigma element.  Elements beyond that are caught by the kernel's own maximum (the gate).
// Record layout (floats): [0] scale used (out) | [1] K as an int | [2, 2 + 1024) per-table bounds | [2 + 1024, 2 + 2048) the apply
// kernel's per-workgroup maxima.  One workgroup per table writes its bound; the apply kernel folds them (<= 16 L2 hits per lane).
__global__ __launch_bounds__(1024) void out_scale_kernel(const float* __restrict__ gamma, const float* __restrict__ beta, int K, int C,
                                                         float* __restrict__ oscale)
{
    // one workgroup per table; thread = (column c, row group q of 1024 / CW): the rows of a group are read coalesced across
    // the columns, four independent loads in flight per thread (as one dependent chain of C loads the launch took 26 us)
    __shared__ float part[1024];
    const int k = blockIdx.x;
    const int CW = C < 1024 ? (C < 256 ? C : 256) : 1024;     // columns handled side by side (a multiple of 32)
    const int Q = 1024 / CW, cl = threadIdx.x % CW, q = threadIdx.x / CW;
    float m = 0.f;
    for (int c0 = 0; c0 < C; c0 += CW) {
        const int c = c0 + cl;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        if (gamma && c < C && q < Q) {
            const float* g = gamma + (int64_t)k * C * C + c;
            int r = q;
            for (; r + 3 * Q < C; r += 4 * Q) {
                const float a0 = g[(int64_t)r * C], a1 = g[(int64_t)(r + Q) * C], a2 = g[(int64_t)(r + 2 * Q) * C], a3 = g[(int64_t)(r + 3 * Q) * C];
                s0 = fmaf(a0, a0, s0); s1 = fmaf(a1, a1, s1); s2 = fmaf(a2, a2, s2); s3 = fmaf(a3, a3, s3);
            }
            for (; r < C; r += Q) { const float a0 = g[(int64_t)r * C]; s0 = fmaf(a0, a0, s0); }
        }
        part[threadIdx.x] = (s0 + s1) + (s2 + s3);
        __syncthreads();
        if (q == 0 && c < C) {
            float ss = gamma ? 0.f : 1.f;
            if (gamma) for (int j = 0; j < Q; ++j) ss += part[j * CW + cl];
            m = fmaxf(m, 16.f * sqrtf(ss) + (beta ? fabsf(beta[(int64_t)k * C + c]) : 0.f));
        }
        __syncthreads();
    }
    // (only threads of row group 0 hold a bound; the rest contribute 0)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        float b = 0.f;
        for (int w = 0; w < 16; ++w) b = fmaxf(b, part[w]);
        oscale[2 + k] = b;          // this table's bound; the apply kernel folds them
        if (k == 0) oscale[1] = __builtin_bit_cast(float, K);
    }
}

hipError_t wc_launch_out_scale(const float* gamma, const float* beta, int K, int C, float* oscale, hipStream_t st)
{
    if (K < 1 || K > kPlaneBounds) return hipErrorInvalidValue;
    hipLaunchKernelGGL(out_scale_kernel, dim3(K), dim3(1024), 0, st, gamma, beta, K, C, oscale);
    return hipGetLastError();
}

size_t wc_fast_affine_workspace(int C, int Kc)
{
    // scale[C] | colscale[Kc*C] | hi[Kc*C*C] | lo[Kc*C*C]
    return wc_align_up((size_t)C * 4, 256) + wc_align_up((size_t)Kc * C * 4, 256) +
           2 * wc_align_up((size_t)Kc * C * C * 2, 256) + 8192;     // + stamp area of WC_STAMPS builds
}

hipError_t wc_launch_channel_scale(const float* in, const float* center, int64_t M, int C, float* scale, hipStream_t st)
{
    hipLaunchKernelGGL(channel_scale_kernel, dim3((C + 63) / 64), dim3(1024), 0, st, in, center, M, C, scale,
                       (const float*)nullptr, (const float*)nullptr, (float*)nullptr, (int*)nullptr);
    return hipGetLastError();
}

// two operands of equal shape and the gate's clearing in one launch (K4)
hipError_t wc_launch_channel_scale2(const float* in, const float* center, float* scale, const float* in2, const float* center2,
                                    float* scale2, int64_t M, int C, int* gate, hipStream_t st)
{
    hipLaunchKernelGGL(channel_scale_kernel, dim3((C + 63) / 64, 2), dim3(1024), 0, st, in, center, M, C, scale, in2, center2, scale2, gate);
    return hipGetLastError();
}

hipError_t wc_launch_channel_scale_gate(const float* in, const float* center, float* scale, int64_t M, int C, int* gate, hipStream_t st)
{
    hipLaunchKernelGGL(channel_scale_kernel, dim3((C + 63) / 64, 1), dim3(1024), 0, st, in, center, M, C, scale,
                       (const float*)nullptr, (const float*)nullptr, (float*)nullptr, gate);
    return hipGetLastError();
}

// Plan layout (also the layout of the per-call workspace): scale[C] | colscale[Kc*C] | hi[Kc*C*C] | lo[Kc*C*C] | stamps
struct PlanView { float* scale; float* colscale; _Float16* hi; _Float16* lo; unsigned long long* dbg; };
static PlanView plan_view(void* plan, int C, int Kc)
{
    char* p = static_cast<char*>(plan);
    PlanView v;
    v.scale = reinterpret_cast<float*>(p); p += wc_align_up((size_t)C * 4, 256);
    v.colscale = reinterpret_cast<float*>(p); p += wc_align_up((size_t)Kc * C * 4, 256);
    v.hi = reinterpret_cast<_Float16*>(p); p += wc_align_up((size_t)Kc * C * C * 2, 256);
    v.lo = reinterpret_cast<_Float16*>(p); p += wc_align_up((size_t)Kc * C * C * 2, 256);
    v.dbg = reinterpret_cast<unsigned long long*>(p);
    return v;
}

// Build the fp16 tables of B ([slot][k][n] fp32) for the per-channel scales `scale` (nullptr: already stored in
// plan.scale; otherwise they are copied into the plan by the same launch).
static SplitJob split_job(const float* B, int Kc, int C, void* plan, const float* scale)
{
    const PlanView v = plan_view(plan, C, Kc);
    SplitJob j = {B, scale ? scale : (const float*)v.scale, v.hi, v.lo, v.colscale, scale ? v.scale : (float*)nullptr, Kc * C};
    return j;
}
hipError_t wc_launch_fast_plan_tables(const float* B, int Kc, int C, void* plan, hipStream_t st, const float* scale)
{
    SplitJob none = {};
    BiasJob nob = {};
    hipLaunchKernelGGL(split_table_kernel, dim3((unsigned)(Kc * C)), dim3(64), 0, st, split_job(B, Kc, C, plan, scale), none, C, nob);
    return hipGetLastError();
}
// the tables of B and, in the same launch, bias_out[slot] = bias[slot] + (center - mu) B[slot]  (the planes route's additive term)
hipError_t wc_launch_fast_plan_tables_bias(const float* B, int Kc, int C, void* plan, hipStream_t st, const float* scale,
                                           const float* bias, const float* center, const float* mu, float* bias_out)
{
    SplitJob none = {};
    BiasJob bj = {B, bias, center, mu, bias_out, Kc};
    hipLaunchKernelGGL(split_table_kernel, dim3((unsigned)(Kc * C + Kc * (C / 2))), dim3(64), 0, st, split_job(B, Kc, C, plan, scale), none, C, bj);
    return hipGetLastError();
}
// two plans (with their input scales given) in one launch
hipError_t wc_launch_fast_plan_tables2(const float* B0, int Kc0, void* plan0, const float* scale0,
                                       const float* B1, int Kc1, void* plan1, const float* scale1, int C, hipStream_t st)
{
    BiasJob nob = {};
    hipLaunchKernelGGL(split_table_kernel, dim3((unsigned)((Kc0 + Kc1) * C)), dim3(64), 0, st,
                       split_job(B0, Kc0, C, plan0, scale0), split_job(B1, Kc1, C, plan1, scale1), C, nob);
    return hipGetLastError();
}
// ... and K6 on a pre-split x: gmean folded with (mu - center) S in the same launch (bias job on the second table)
hipError_t wc_launch_fast_plan_tables2_bias(const float* B0, int Kc0, void* plan0, const float* scale0,
                                            const float* B1, void* plan1, const float* scale1, int C, hipStream_t st,
                                            const float* bias, const float* center, const float* mu, float* bias_out, int neg_bias)
{
    BiasJob bj = {B1, bias, center, mu, bias_out, 1, neg_bias};
    hipLaunchKernelGGL(split_table_kernel, dim3((unsigned)((Kc0 + 1) * C + C / 2)), dim3(64), 0, st,
                       split_job(B0, Kc0, C, plan0, scale0), split_job(B1, 1, C, plan1, scale1), C, bj);
    return hipGetLastError();
}

float* wc_fast_plan_scale(void* plan) { return reinterpret_cast<float*>(plan); }

// the parts of a plan, for the kernels of other translation units that read the same tables (wc_split.hip)
void wc_fast_plan_parts(const void* plan, int C, int Kc, const float** scale, const float** colscale, const void** hi, const void** lo)
{
    const PlanView v = plan_view(const_cast<void*>(plan), C, Kc);
    *scale = v.scale; *colscale = v.colscale; *hi = v.hi; *lo = v.lo;
}

// The main kernel alone, on a prepared plan.  B is still needed: tiles outside the fp16 range are recomputed from it.
hipError_t wc_launch_fast_affine_planned(const float* in, const float* center, const float* B, int Kc, bool shared_table,
                                         const float* bias, const float* sub, const int32_t* slot,
                                         int64_t N, int64_t HW, int C, int accumulate, float* out,
                                         const void* plan, hipStream_t st, unsigned* relu_mask, void* planes, float* oscale)
{
    const PlanView v = plan_view(const_cast<void*>(plan), C, Kc);
    FastArgs a = {};
    if (planes) {       // the output as the next convolution's fp16 planes (hi | lo, N*HW*C halves each)
        if (!oscale || !wc_fast_affine_writes_planes(N, HW, C) || (accumulate & 1)) return hipErrorInvalidValue;
        a.phi = static_cast<_Float16*>(planes); a.plo = a.phi + N * HW * C; a.oscale = oscale;
    }
    a.in = in; a.center = center; a.scale = v.scale; a.Bhi = v.hi; a.Blo = v.lo; a.colscale = v.colscale;
    a.slot_stride = shared_table ? 0 : (int64_t)C * C;
    a.bias = bias; a.sub = sub; a.slot = shared_table ? nullptr : slot; a.M = N * HW; a.HW = HW;
    a.maskout = relu_mask;
    a.accumulate = accumulate & 1; a.relu = (accumulate >> 1) & 1; a.Bf = B; a.bf_stride = shared_table ? 0 : (int64_t)C * C; a.out = out; a.dbg = v.dbg;
    // the ring kernel takes any HW: tiles are cut from the M rows, and tiles that straddle samples of different slots
    // are redone with per-row tables at the end of the launch (rare shapes; none of the shipped recipes)
    a.mixed = (a.slot != nullptr && (HW % (8192 / C)) != 0) ? 1 : 0;
    if (!(accumulate & 1) && use_ring() && ((N * HW) % (8192 / C)) == 0) {
        switch (C) {
            case 32: return launch_affine_ring<32>(a, st);
            case 64: return launch_affine_ring<64>(a, st);
            case 128: return launch_affine_ring<128>(a, st);
            case 256: return launch_affine_ring<256>(a, st);
        }
    }
    switch (C) {
        case 32: return launch_affine<32>(a, st);
        case 64: return launch_affine<64>(a, st);
        case 128: return launch_affine<128>(a, st);
        case 256: return launch_affine<256>(a, st);
    }
    return hipErrorInvalidValue;
}

// K6 in one pass (C = 256, both inputs' scales given, plans of At and of S built): see onepass_ring_kernel
bool wc_bwd_apply_onepass_supported(int64_t N, int64_t HW, int C)
{
    static const bool off = getenv("WC_K6_TWO_PASS") != nullptr;      // development: the two-pass form
    return !off && WC_MFMA16 && C == 256 && ((N * HW) % 16) == 0 && N * HW >= 4096;
}

hipError_t wc_launch_bwd_apply_onepass(const float* gy, const float* x, const float* mu, const float* At, int Kc, const float* S,
                                       const float* gmean, const int32_t* slot, int64_t N, int64_t HW, const float* scales,
                                       float* dx, const void* plan0, const void* plan1, hipStream_t st, const unsigned* relu_mask,
                                       const void* xs, const float* xs_scale)
{
    constexpr int C = 256, TR = 16;
    const PlanView v0 = plan_view(const_cast<void*>(plan0), C, Kc), v1 = plan_view(const_cast<void*>(plan1), C, 1);
    OnePassArgs a = {};
    a.gy = gy; a.x = x; a.mu = mu; a.sg = scales + C; a.sx = scales;
    if (xs) {       // x as pre-split planes: sx = the planes' scales, gmean = the caller's folded gmean - (center - mu) S
        a.xhi = static_cast<const _Float16*>(xs); a.xlo = a.xhi + N * HW * C; a.sx = xs_scale;
    }
    a.hi0 = v0.hi; a.lo0 = v0.lo; a.cs0 = v0.colscale; a.slot_stride0 = (int64_t)C * C;
    a.hi1 = v1.hi; a.lo1 = v1.lo; a.cs1 = v1.colscale;
    a.sub_on = gmean != nullptr; a.sub = gmean ? gmean : scales;
    a.slot = slot; a.HW = HW; a.Bf0 = At; a.bf0_stride = (int64_t)C * C; a.Bf1 = S; a.out = dx;
    a.gmask = relu_mask;
    if (relu_mask && ((N * HW) % 32) != 0) return hipErrorInvalidValue;
    a.ntiles = (int)(N * HW / TR);
    a.mixed = (slot != nullptr && (HW % TR) != 0) ? 1 : 0;
    int pairs = a.ntiles < 128 ? a.ntiles : 128;
    const int groups16 = (pairs + 7) / 8;                 // block ids come in groups of 16: 8 pairs x 2 column halves
    pairs = groups16 * 8;
    a.tiles_per_pair = (a.ntiles + pairs - 1) / pairs;
    const size_t lds = 3 * 2 * (size_t)(TR * (2 * C * 2 + 32)) + 8 * 7 * 1024 + 64 + (relu_mask ? 3072 : 0);      // + the three shared mask blocks
#define WC_LAUNCH_ONEPASS(SLOT_, MK_, XPL_)                                                                            \
    do {                                                                                                                \
        static bool attr_set = false;                                                                                   \
        if (!attr_set) {                                                                                                \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(onepass_ring_kernel<SLOT_, MK_, XPL_>),    \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                   \
            if (e != hipSuccess) return e;                                                                              \
            attr_set = true;                                                                                            \
        }                                                                                                               \
        hipLaunchKernelGGL((onepass_ring_kernel<SLOT_, MK_, XPL_>), dim3(groups16 * 16), dim3(512), lds, st, a);        \
    } while (0)
    if (xs) {
        if (relu_mask) { if (slot) WC_LAUNCH_ONEPASS(true, true, true); else WC_LAUNCH_ONEPASS(false, true, true); }
        else { if (slot) WC_LAUNCH_ONEPASS(true, false, true); else WC_LAUNCH_ONEPASS(false, false, true); }
    } else if (relu_mask) { if (slot) WC_LAUNCH_ONEPASS(true, true, false); else WC_LAUNCH_ONEPASS(false, true, false); }
    else { if (slot) WC_LAUNCH_ONEPASS(true, false, false); else WC_LAUNCH_ONEPASS(false, false, false); }
#undef WC_LAUNCH_ONEPASS
    return hipGetLastError();
}

// One stream of the affine on the fast path with no prepared plan: sample the channel scales, build the tables, run.
hipError_t wc_launch_fast_affine(const float* in, const float* center, const float* B, int Kc, bool shared_table,
                                 const float* bias, const float* sub, const int32_t* slot,
                                 int64_t N, int64_t HW, int C, int accumulate, float* out,
                                 void* ws, hipStream_t st)
{
    { hipError_t e0 = wc_launch_channel_scale(in, center, N * HW, C, wc_fast_plan_scale(ws), st); if (e0 != hipSuccess) return e0; }
    hipError_t e = wc_launch_fast_plan_tables(B, Kc, C, ws, st);
    if (e != hipSuccess) return e;
    return wc_launch_fast_affine_planned(in, center, B, Kc, shared_table, bias, sub, slot, N, HW, C, accumulate, out, ws, st);
}
