// The PRE-SPLIT activation format and the K3 apply that reads it (round 3).
//
// Every split-fp16 kernel of this library (wc_fast.hip, wc_fast_xty.hip, wc_conv.hip) turns an fp32 activation into two
// fp16 terms v = hi + lo before it can use the fp16 matrix pipe, and every consumer of a tensor repeats that
// conversion: ds_read_b128 -> centre/scale -> cvt_pk -> fma_mix remainder -> cvt_pk -> ds_write_b64, 2.3 vector
// instructions per MFMA in K3 and 12 in K1.  hi/lo planes are 4 bytes per element -- the bytes of the fp32 tensor -- so
// the PRODUCER can write them instead, once:
//
//     x[m][c]  ~=  center[c] + (hi[m][c] + lo[m][c]) / scale[c]         hi = fp16(g), lo = fp16(g - hi), g = (x - center) scale
//
// with a power-of-two per-channel `scale` (exact) that puts the channel's sampled maximum near 16, i.e. >= 3700 x of
// headroom below fp16's 65504, and a rough per-channel centre (any value near the mean; the consumers fold the exact
// mean in).  22 significant bits while lo is a normal fp16, an absolute error <= 2^-25 of the scaled range below that.
// Storage: ONE buffer of 2*M*C halves, the hi plane [M][C] first, then the lo plane [M][C] (the layout wc_conv_f16x3
// takes its operands in).  An element beyond +-60000 after scaling is clamped and raises the tensor's overflow flag
// (the producer's statistics were off by more than three decades: the caller re-splits with fresh scales).
//
// apply_split_kernel (K3 on that format):   y[m] = (x[m] - mu) A[slot] + beta[slot]
//                                                = (hi + lo)[m] . (A / scale) + (beta + (center - mu) A)
// The fp16 image the MFMAs read IS the input, so the staging is pure LDS-DMA: 1-KiB pieces (2 rows of one plane at
// C = 256) go HBM -> LDS with global_load_lds_dwordx4 and nothing else touches them -- no conversion, no ds_write, no
// raw ring.  A DMA writes its 64 lanes' 16 bytes contiguously, so rows cannot be padded against bank conflicts; the
// 16-byte slots of a row are XOR-swizzled instead (slot ^= row & 15), applied to the per-lane SOURCE address of the DMA
// and to the fragment reads (4 lane constants cover every k-step: (64 s) ^ L = ((64 (s & 3)) ^ L) + 256 (s >> 2)).
// One persistent 512-thread workgroup per CU, B' fragments of the wave's 32 output columns in 128 VGPRs, 16x16x32
// MFMAs in the order of affine_ring_kernel (wc_fast.hip), four 32-KiB tile buffers, two kinds of LDS counters (pieces of a
// tile landed; reads of a buffer finished -- one counter per buffer).  Per tile t and wave:
//   top             wait "tile t landed for all eight waves"
//   gap 2           s_waitcnt vmcnt(36): my pieces of tile t+1 have landed (issued one and a half tiles ago); publish
//   k-step 1        wait "tile t-1 read by all", then the four DMAs of tile t+3 into its buffer (2.9 tiles = 90 KiB per CU ahead)
//   gaps 14, 19 ..  the 16 row-pair stores of tile t-1 (nontemporal), parked in 16 VGPRs since its epilogue: no store phase
//                   during which both waves of a SIMD leave the matrix pipe idle
// Everything the memory pipe sees from this kernel is inline asm with hand-counted vmcnt (loads, stores and LDS-DMA
// retire in issue order on gfx950), table loads included: hipcc would otherwise drain vmcnt(0) around them.
#include "wc_common.h"
#include <stdlib.h>
#include <type_traits>

#ifndef WC_SPLIT_NT_STORE
#define WC_SPLIT_NT_STORE 1
#endif
#ifndef WC_SPLIT_STAGGER
#define WC_SPLIT_STAGGER 1
#endif
#ifndef WC_SPLIT_STAMPS
#define WC_SPLIT_STAMPS 0
#endif
#ifndef WC_SPLIT_PUB_GAP
#define WC_SPLIT_PUB_GAP 2      // MFMA gap of a tile's loop in which the next tile's pieces are waited for and published
#endif
#ifndef WC_SPLIT_DMA_STEP
#define WC_SPLIT_DMA_STEP -1    // k-step whose first gap carries the DMAs of tile t+3 (-1: step 1)
#endif
#ifndef WC_SPLIT_POLL_SLEEP
#define WC_SPLIT_POLL_SLEEP 1
#endif
#ifndef WC_SPLIT_DEFER
#define WC_SPLIT_DEFER 0        // 1: a tile's stores ride in the next tile's MFMA gaps; 0: they follow the tile's own loop
#endif
#ifndef WC_SPLIT_PRE3
#define WC_SPLIT_PRE3 0     // 1 (one table for the launch, >= 6 tiles): tile 3's pieces are requested in the prologue with tiles 0-2 (its buffer is free from the start)
#endif                      // instead of from tile 0's k-step 1, which waits for the table's first fragments: 128 instead of 96 KiB per CU in flight while the table arrives
#ifndef WC_SPLIT_D0FIRST
#define WC_SPLIT_D0FIRST 0
#endif
#ifndef WC_SPLIT_TROT
#define WC_SPLIT_TROT 0
#endif
#ifndef WC_SPLIT_ABL
#define WC_SPLIT_ABL 0       // development ablation bits (wrong results, times only; tools/k3_ablations.py): 1 no stores (the hand-counted waits
                             // adjusted: the DMA waits stay real), 2 no MFMA, 4 linear (unswizzled) DMA source, 8 no table loads (a zero table),
                             // 16 no mask words (bits not formed, not stored).  (A bit that skipped the epilogue arithmetic HUNG the GPU: the column
                             // constants arrive by asm loads, and registers nobody reads are handed out again while those loads are in flight.)
#endif

namespace {

typedef float f32x2s __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pk_rne2(float a, float b)
{
    const f32x2s v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));
}

constexpr float kSplitGuard = 60000.0f;

// ---------------------------------------------------------------------------------------------------------------
// producer for tensors that exist in fp32 (tests, the bench, layers whose producer is not one of ours):
// one thread = 8 consecutive elements of a row -> one 16-byte store per plane
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void split_rows_kernel(const float* __restrict__ x, const float* __restrict__ center,
                                                         const float* __restrict__ scale, int64_t n8, int C, int relu,
                                                         _Float16* __restrict__ hi, _Float16* __restrict__ lo, int* __restrict__ flag)
{
    bool over = false;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t e = i * 8;
        const int c = (int)(e % C);
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(x + e), a1 = *reinterpret_cast<const f32x4*>(x + e + 4);
        const f32x4 s0 = *reinterpret_cast<const f32x4*>(scale + c), s1 = *reinterpret_cast<const f32x4*>(scale + c + 4);
        f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0;
        if (center) { c0 = *reinterpret_cast<const f32x4*>(center + c); c1 = *reinterpret_cast<const f32x4*>(center + c + 4); }
        float g[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) { g[j] = a0[j]; g[4 + j] = a1[j]; }
        if (relu) {
#pragma unroll
            for (int j = 0; j < 8; ++j) g[j] = !(g[j] <= 0.f) ? g[j] : 0.f;      // NaN stays NaN
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) { g[j] = (g[j] - c0[j]) * s0[j]; g[4 + j] = (g[4 + j] - c1[j]) * s1[j]; }
        unsigned h[4], l[4];
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (fabsf(g[j]) > kSplitGuard) { over = true; g[j] = copysignf(kSplitGuard, g[j]); }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            h[j] = pk_rne2(g[2 * j], g[2 * j + 1]);
            const f16x2 hv = __builtin_bit_cast(f16x2, h[j]);
            l[j] = pk_rne2(g[2 * j] - (float)hv[0], g[2 * j + 1] - (float)hv[1]);
        }
        *reinterpret_cast<uint4*>(hi + e) = make_uint4(h[0], h[1], h[2], h[3]);
        *reinterpret_cast<uint4*>(lo + e) = make_uint4(l[0], l[1], l[2], l[3]);
    }
    if (flag && over) *flag = 1;
}

// inverse (tests; fp32 consumers of a split tensor): x = center + (hi + lo) / scale
__global__ __launch_bounds__(256) void unsplit_rows_kernel(const _Float16* __restrict__ hi, const _Float16* __restrict__ lo,
                                                           const float* __restrict__ center, const float* __restrict__ scale,
                                                           int64_t n8, int C, float* __restrict__ x)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t e = i * 8;
        const int c = (int)(e % C);
        const f16x8 h = *reinterpret_cast<const f16x8*>(hi + e), l = *reinterpret_cast<const f16x8*>(lo + e);
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = ((float)h[j] + (float)l[j]) / scale[c + j] + (center ? center[c + j] : 0.f);
        *reinterpret_cast<f32x4*>(x + e) = f32x4{o[0], o[1], o[2], o[3]};
        *reinterpret_cast<f32x4*>(x + e + 4) = f32x4{o[4], o[5], o[6], o[7]};
    }
}

// bias2[slot][n] = bias[slot][n] + sum_k (center[k] - mu[k]) A[slot][k][n]     (float64 sums in a fixed order; one block per
// (slot, 32 columns): thread (q, n) sums the rows k = q mod 8 of its column)
__global__ __launch_bounds__(256) void split_bias_kernel(const float* __restrict__ A, const float* __restrict__ bias,
                                                         const float* __restrict__ center, const float* __restrict__ mu,
                                                         int C, float* __restrict__ out)
{
    __shared__ float d[1024];
    __shared__ double red[8][32];
    for (int k = threadIdx.x; k < C; k += 256) d[k] = (center ? center[k] : 0.f) - (mu ? mu[k] : 0.f);
    __syncthreads();
    const int n = blockIdx.x * 32 + (threadIdx.x & 31), q = threadIdx.x >> 5;
    const int slot = blockIdx.y;
    const float* a = A + (int64_t)slot * C * C + n;
    double acc = 0.0;
#pragma unroll 8
    for (int k = q; k < C; k += 8) acc += (double)d[k] * (double)a[(int64_t)k * C];
    red[q][threadIdx.x & 31] = acc;
    __syncthreads();
    if (q == 0) {
        double t = 0.0;
#pragma unroll
        for (int i = 0; i < 8; ++i) t += red[i][threadIdx.x];
        out[(int64_t)slot * C + n] = (float)(t + (bias ? (double)bias[(int64_t)slot * C + n] : 0.0));
    }
}

// ---------------------------------------------------------------------------------------------------------------
struct SplitApplyArgs {
    const _Float16* xs; int64_t plane;                                // hi plane at xs, lo plane at xs + plane (elements)
    const _Float16* Bhi; const _Float16* Blo; const float* colscale;  // plan tables (split_table_kernel's load order), [slots][C]
    int64_t slot_stride;                                              // C*C, or 0 when the table is shared
    const float* bias;                                                // [slots][C]: beta + (center - mu) A (split_bias_kernel)
    const int32_t* slot;
    int64_t M, HW;
    int relu, mixed;
    const float* Bf; int64_t bf_stride; const float* xscale;          // exact redo of tiles that straddle slots
    float* out;
    int ntiles, tiles_per_wg;
    unsigned long long* dbg;
    unsigned* maskout;                                                // MASK: the ReLU's one-bit gradient mask, [M/32][C] words (wc_apply_mask_f32's layout)
    _Float16* phi; _Float16* plo;                                     // PL: the output as the next convolution's planes (hi | lo of os * y)
    float* oscale; float* oamax; const float* gate; int ngate;        // PL: the scale record of wc_launch_out_scale (affine_ring_kernel's protocol)
};

constexpr int kSplitPlaneBounds = 1024;        // = kPlaneBounds of wc_fast.hip: per-table bounds / per-workgroup maxima in the scale record

// MASK / PL: the two epilogues of affine_ring_kernel (wc_fast.hip) on this kernel -- round 4: the sites fed by the residual add
// (wc_resadd.hip) read planes AND leave the ReLU's bit mask (all of them) or the next convolution's planes (bn1 of a block).
template <int C, bool HAS_SLOT, bool MASK = false, bool PL = false>
__global__ __launch_bounds__(512, 1) void apply_split_kernel(SplitApplyArgs a)
{
    constexpr int TR = 8192 / C;                      // rows per tile: 32 KiB of hi | lo
    constexpr int KS = C / 16, KS32 = C / 32, CG = C / 32;
    constexpr int ROWB = C * 2, IMG = TR * ROWB, TILE = 2 * IMG;
    constexpr int NBUF = 4;
    static_assert(C == 128 || C == 256, "split apply: C = 128 or 256");
    static_assert(!(PL && WC_SPLIT_DEFER), "the planes-out epilogue has no deferred-store form (its pend[] / pend_po stay unset)");
    extern __shared__ __attribute__((aligned(1024))) char smem[];      // 4 tiles [hi image | lo image] | counter
    const unsigned tiles_lds = (unsigned)(size_t)((__attribute__((address_space(3))) char*)smem);
    const unsigned cnt_lds = tiles_lds + NBUF * TILE;
    volatile int* const cnt = reinterpret_cast<volatile int*>(smem + NBUF * TILE);

    unsigned long long rt_in = 0;
    unsigned long long pro_[8] = {0, 0, 0, 0, 0, 0, 0, 0};      // WC_SPLIT_STAMPS: the prologue and tile 0, s_memtime
    if (WC_SPLIT_STAMPS) rt_in = __builtin_amdgcn_s_memrealtime();
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cg = wave % CG, rg = wave / CG;
    const int l15 = lane & 15, lq = lane >> 4, l31 = lane & 31, lh = lane >> 5;

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
            if (!(am > kSplitGuard)) return;
            int e = 0;
            if (am < 3.0e38f) { (void)frexpf(am, &e); os = ldexpf(os, 14 - e); } else os = 1.f;
        }
        if (blockIdx.x == 0 && tid == 0) a.oscale[0] = os;
    }
    if (n <= 0) { if (PL && a.oamax && tid == 0) a.oamax[blockIdx.x] = 0.f; return; }
    auto tile_of = [&](int i) { return t_first + i * t_stride; };

    if (tid < 16) cnt[tid] = 0;
    __syncthreads();

    // ---- DMA: this wave's pieces q = 4 wave + i of a tile; lane j lands at plane offset o = (q % 16) KiB + 16 j, i.e. row
    // o / ROWB, slot p = (o % ROWB) / 16, and fetches slot p ^ (row & 15) of that row
    unsigned src_off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int o = ((4 * wave + i) & 15) * 1024 + 16 * lane;
        const int row = o / ROWB, p = (o % ROWB) >> 4;
        src_off[i] = (WC_SPLIT_ABL & 4) ? (unsigned)o : (unsigned)(row * ROWB + ((p ^ (row & 15)) << 4));
    }
    const char* const plane_base = reinterpret_cast<const char*>(a.xs) + (int64_t)(wave >> 2) * a.plane * 2;
    auto dma_tile = [&](int tl) {
        const char* g = plane_base + (int64_t)tile_of(tl) * IMG;               // wave-uniform
        const unsigned l0 = tiles_lds + (unsigned)(tl & (NBUF - 1)) * TILE + (unsigned)wave * 4096u;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned l = __builtin_amdgcn_readfirstlane(l0 + i * 1024u);
            unsigned keep;
            // (s_nop 4 first: an SGPR operand that hipcc has just restored from a spill lane with v_readlane needs five wait
            // states before a VMEM instruction may read it, and hipcc's hazard pass does not look inside asm statements --
            // found as stores to 0xffff'xxxxxxxx: a stale high half of the row base)
            asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(src_off[i]), "s"(l), "s"(g) : "memory");
        }
    };
    // counter 0: pieces landed (8 arrivals per tile); counters 1 + b: reads of buffer b finished, 8 arrivals per
    // tile that used it -- ONE running count over all tiles would let a wave that is a tile ahead stand in for one that is not done
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
            if (WC_SPLIT_POLL_SLEEP) __builtin_amdgcn_s_sleep(WC_SPLIT_POLL_SLEEP);
        }
    };

    // ---- B' fragments of this wave's 32 output columns, all of K (unit u = 2 s + ch: k-step s, column half ch), by asm loads
    f16x8 bhi[KS], blo[KS];
    float cscale[2], addv[2];
    int cur_slot = -1;
    const unsigned tb_lane = (unsigned)lane * 16u;
    const unsigned col_b = (unsigned)(cg * 32 + l15) * 4u;
    // ASYNC_TABLE (one table for the whole launch): the table's loads are asm statements whose landing the hand-counted waits of the
    // first tile cover, k-step by k-step.  With slots the table is re-loaded inside the tile loop, and the registers then meet at the
    // loop header: hipcc may copy a table register between the asm load's ISSUE and its landing -- the data then lands in a register
    // that has been handed to something else (round 4, C = 256 with the planes epilogue: whole workgroups scaled by garbage, racily;
    // the lesson of DESIGN.md section 4.9b again).  So with slots the loads are plain loads hipcc tracks itself, drained at once.
    constexpr bool ASYNC_TABLE = !HAS_SLOT;
    auto load_b = [&](int slot) {
        if (WC_SPLIT_ABL & 8) {
#pragma unroll
            for (int s = 0; s < KS; ++s) { bhi[s] = f16x8{0, 0, 0, 0, 0, 0, 0, 0}; blo[s] = bhi[s]; }
            cscale[0] = cscale[1] = 1.f; addv[0] = addv[1] = 0.f; cur_slot = slot;
            return;
        }
        const char* ph = reinterpret_cast<const char*>(a.Bhi + (int64_t)slot * a.slot_stride + (int64_t)cg * KS * 512);
        const char* pl = reinterpret_cast<const char*>(a.Blo + (int64_t)slot * a.slot_stride + (int64_t)cg * KS * 512);
        const char* pc = reinterpret_cast<const char*>(a.colscale + (a.slot_stride ? (int64_t)slot * C : 0));
        const char* pb = reinterpret_cast<const char*>(a.bias + (int64_t)slot * C);
        if (!ASYNC_TABLE) {
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                bhi[s] = *reinterpret_cast<const f16x8*>(ph + 1024 * s + tb_lane);
                blo[s] = *reinterpret_cast<const f16x8*>(pl + 1024 * s + tb_lane);
            }
            cscale[0] = *reinterpret_cast<const float*>(pc + col_b); cscale[1] = *reinterpret_cast<const float*>(pc + col_b + 64);
            addv[0] = *reinterpret_cast<const float*>(pb + col_b); addv[1] = *reinterpret_cast<const float*>(pb + col_b + 64);
            __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0), the BUILTIN: hipcc must see that the loads have completed here
            cur_slot = slot;
            return;
        }
#if WC_SPLIT_TROT        // development (WRONG results, times only): every workgroup walks the table from a different k-step -- do 256 CUs reading the same
        // lines in the same order at the same moment queue up at the same L2 channels?
        const int rot = (int)((blockIdx.x * 5u) & (unsigned)(KS - 1));
#else
        const int rot = 0;
#endif
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "=v"(bhi[s]) : "v"(tb_lane), "s"(ph + 1024 * ((s + rot) & (KS - 1))) : "memory");
            asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "=v"(blo[s]) : "v"(tb_lane), "s"(pl + 1024 * ((s + rot) & (KS - 1))) : "memory");
        }
        asm volatile("s_nop 4\n\tglobal_load_dword %0, %1, %2" : "=v"(cscale[0]) : "v"(col_b), "s"(pc) : "memory");
        asm volatile("s_nop 4\n\tglobal_load_dword %0, %1, %2 offset:64" : "=v"(cscale[1]) : "v"(col_b), "s"(pc) : "memory");
        asm volatile("s_nop 4\n\tglobal_load_dword %0, %1, %2" : "=v"(addv[0]) : "v"(col_b), "s"(pb) : "memory");
        asm volatile("s_nop 4\n\tglobal_load_dword %0, %1, %2 offset:64" : "=v"(addv[1]) : "v"(col_b), "s"(pb) : "memory");
        cur_slot = slot;
    };
    // every register the asm loads define passes through an asm statement BEHIND the wait, so that no use is scheduled ahead of it
    auto touch_b = [&]() {
#pragma unroll
        for (int s = 0; s < KS; s += 4)
            asm volatile("" : "+v"(bhi[s]), "+v"(bhi[s + 1]), "+v"(bhi[s + 2]), "+v"(bhi[s + 3]),
                              "+v"(blo[s]), "+v"(blo[s + 1]), "+v"(blo[s + 2]), "+v"(blo[s + 3]));
        asm volatile("" : "+v"(cscale[0]), "+v"(cscale[1]), "+v"(addv[0]), "+v"(addv[1]));
    };

    // Two schedules.  A workgroup with at least six tiles (DEF): tile 0 is waited for alone and the table is taken k-step by
    // k-step inside tile 0's loop (4 C^2 bytes per workgroup from L2 are ~2.5 us at 64 B/clk/CU, all of it in front of the
    // first MFMA otherwise), and a tile's 16 output stores leave in the MFMA gaps of the NEXT tile's loop -- measured with
    // in-kernel stamps on the first version, whose stores followed each tile's loop: the two waves of a SIMD ran in phase
    // (every wave waits for the same "tile landed" event), so the matrix pipe idled through 1 500 of every 4 600 cycles
    // while both were storing.  Fewer tiles (the small sites): everything drains per tile, stores at the end of the tile.
    const bool def_mode = n >= 6;
    if (WC_SPLIT_STAMPS) pro_[0] = __builtin_amdgcn_s_memtime();      // in front of the first vector-memory instruction
    if (!ASYNC_TABLE) load_b(a.slot[((int64_t)tile_of(0) * TR) / a.HW]);      // (drained: in front of the DMAs, whose counts start here)
    dma_tile(0);
#if WC_SPLIT_D0FIRST
    // Round 6 (stamps, profiles/r6_k3_prologue_stamps.txt): with the table's 36 loads per wave queued right behind D0, a wave's pieces of
    // tile 0 land 8 000 cycles after its first vector-memory instruction -- 4 300 without the table: 256 KiB per workgroup share the CU's
    // 64 B/clk return path with the 32 KiB everybody is waiting for.  Tile 0 alone first, the table behind it.
    if (ASYNC_TABLE && def_mode) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    if (ASYNC_TABLE) load_b(0);
    if (n > 1) dma_tile(1);
    if (n > 2) dma_tile(2);
    constexpr bool PRE3 = WC_SPLIT_PRE3 && ASYNC_TABLE && !(WC_SPLIT_ABL & 8);
    if (PRE3 && def_mode) dma_tile(3);             // (def_mode: n >= 6)
    if (def_mode && ASYNC_TABLE) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(((WC_SPLIT_ABL & 8) ? 0 : 2 * KS + 4) + 8 + (PRE3 ? 4 : 0)) : "memory");      // tile 0 only
    else if (n > 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (n > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    touch_b();                                    // (unconditional: a tie in one branch only doubles the table's registers at the merge)
    if (WC_SPLIT_STAMPS) pro_[1] = __builtin_amdgcn_s_memtime();      // my pieces of tile 0 have landed
    arrive(0);                                    // my pieces of tile 0

    const int rbase = rg * 32;
    // fragment reads: row rbase + 16 rh + l15, 16-byte slot (4 s + lq) ^ l15 of the row; the four lane constants below + immediates
    int rd_base[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) rd_base[j] = (rbase + l15) * ROWB + ((64 * j) ^ ((lq ^ l15) << 4));
    unsigned ol_b = (unsigned)((rbase + 8 * lh) * C + cg * 32 + l31) * 4u;      // byte offset of this lane's outputs inside a row pair
    // planes form (affine_ring_kernel's): a lane pair exchanges its packed halves (DPP quad_perm [1,0,3,2]) so that the EVEN lane holds
    // row +0's columns (l31, l31 + 1) and the ODD lane row +4's columns (l31 - 1, l31): one dword per lane and plane
    const unsigned pl_b = (unsigned)((rbase + 8 * lh + 4 * (l31 & 1)) * C + cg * 32 + (l31 & ~1)) * 2u;
    const unsigned pl_sel = (l31 & 1) ? 0x03020706u : 0x05040100u;       // v_perm_b32(neighbour, own, sel)
    const unsigned mk_b = (unsigned)(cg * 32 + l31) * 4u;                // mask word of column l31 (lanes 0-31 store)
    if (WC_SPLIT_STAGGER && wave >= 4 && n >= 4) __builtin_amdgcn_s_sleep(24);      // ~a quarter of a tile behind waves 0-3

    unsigned long long t_wait = 0, t_loop = 0, t_store = 0;
    constexpr int G = 12 * KS32;                               // MFMAs = issue gaps per tile
    constexpr int DSTEP = WC_SPLIT_DMA_STEP < 0 ? 1 : WC_SPLIT_DMA_STEP;      // k-step whose first gap carries the DMAs of tile t+3
    constexpr int SG0 = 12 * DSTEP + 2, SGS = (G - SG0 - 1) / 16;             // gaps of the parked stores: SG0 + SGS j, all behind the DMAs
    static_assert(SGS >= 1 && WC_SPLIT_PUB_GAP < 12 * DSTEP, "publication, then DMAs, then stores");
    float pend[16];                                // the previous tile's outputs (scaled, activated, swapped), parked
#pragma unroll
    for (int i = 0; i < 16; ++i) pend[i] = 0.f;
    const float* pend_po = a.out;                  // ... and their tile's base (wave-uniform)
    auto store_pair = [&](const float* base, int i, float v) __attribute__((always_inline)) {
        // value i of a tile: row 16 (i / 8) + (i / 2) % 4 + 4 (i % 2) of the wave's 32 (lanes 32-63: 8 rows below), column l31
        const unsigned olb = ol_b + 0u;            // (an asm operand alone does not capture a variable in a nested generic lambda)
        const float* p = base + (16 * (i >> 3) + ((i >> 1) & 3) + 4 * (i & 1)) * C;
#if (WC_SPLIT_ABL & 1)
        asm volatile("" :: "v"(olb), "v"(v), "s"(p));
#elif WC_SPLIT_NT_STORE
        asm volatile("s_nop 4\n\tglobal_store_dword %0, %1, %2 nt" :: "v"(olb), "v"(v), "s"(p) : "memory");
#else
        asm volatile("s_nop 4\n\tglobal_store_dword %0, %1, %2" :: "v"(olb), "v"(v), "s"(p) : "memory");
#endif
    };
    auto store_u32 = [&](const void* base, unsigned off, unsigned v) __attribute__((always_inline)) {      // plain store (planes, mask): the L2 pairs the 64-byte halves of a line
#if (WC_SPLIT_ABL & 1)
        asm volatile("" :: "v"(off), "v"(v), "s"(base));
#else
        asm volatile("s_nop 4\n\tglobal_store_dword %0, %1, %2" :: "v"(off), "v"(v), "s"(base) : "memory");
#endif
    };
    // PUB_: the hand-counted vmcnt in front of the publication of tile t+1 (-1: no next tile); DMA_: tile t+3 exists;
    // FIRST_: tile 0 in DEF mode (table k-step by k-step); PEND_: the previous tile's stores ride in this loop; DEFER_: this
    // tile's outputs are parked for the next loop
    auto tile_body = [&](int t, auto pub_tag, auto dma_tag, auto first_tag, auto pend_tag, auto defer_tag) {
        constexpr int PUB_ = (WC_SPLIT_ABL & 1) && decltype(pub_tag)::value > 0 ? decltype(pub_tag)::value % 16 : decltype(pub_tag)::value;      // (no stores: the counts without the 16 per tile)
        constexpr bool DMA_ = decltype(dma_tag)::value;
        constexpr bool FIRST_ = decltype(first_tag)::value;
        constexpr bool PEND_ = decltype(pend_tag)::value;
        constexpr bool DEFER_ = decltype(defer_tag)::value;
        unsigned long long c0_ = 0;
        if (WC_SPLIT_STAMPS) c0_ = __builtin_amdgcn_s_memtime();
        wait_for(0, 8 * (t + 1));                  // tile t has landed for all eight waves
        unsigned long long c1_ = 0;
        if (WC_SPLIT_STAMPS) { c1_ = __builtin_amdgcn_s_memtime(); t_wait += c1_ - c0_; if (FIRST_) pro_[2] = c1_; if (t == 1) pro_[6] = c1_; if (t == 2) pro_[7] = c1_; }
        int rb[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) rb[j] = rd_base[j];
        asm volatile("" : "+v"(rb[0]), "+v"(rb[1]), "+v"(rb[2]), "+v"(rb[3]));      // per-tile opaque copies: no hoisted address tables
        const char* const tb = smem + (t & (NBUF - 1)) * TILE;
        auto frag = [&](int pl, int rh, int s) {
            return *reinterpret_cast<const f16x8*>(tb + pl * IMG + rh * (16 * ROWB) + rb[s & 3] + 256 * (s >> 2));
        };
        f32x4 acc[2][2];
#pragma unroll
        for (int rh = 0; rh < 2; ++rh)
#pragma unroll
            for (int ch = 0; ch < 2; ++ch) acc[rh][ch] = f32x4{0.f, 0.f, 0.f, 0.f};
        f16x8 ah[2], al[2], nh[2], nl[2];
#pragma unroll
        for (int rh = 0; rh < 2; ++rh) { ah[rh] = frag(0, rh, 0); al[rh] = frag(1, rh, 0); nh[rh] = ah[rh]; nl[rh] = al[rh]; }
#pragma unroll
        for (int s = 0; s < KS32; ++s) {
            if (FIRST_) {    // this k-step's four table fragments have landed (younger: the later k-steps', the 4 column constants, the
                // DMAs of tiles 1 and 2 and, behind step DSTEP, of tile 3).  No "+v" ties: redefining 128 table registers inside one
                // of the tile bodies cost 60 VGPRs and spills; the scheduling barrier keeps the k-step's MFMAs behind the wait
                asm volatile("s_waitcnt vmcnt(%0)" :: "n"(12 + 4 * (KS32 - 1 - s) + ((PRE3 || s > DSTEP) ? 4 : 0)) : "memory");
                if (WC_SPLIT_STAMPS && (s == 0 || s == KS32 - 1)) pro_[s == 0 ? 3 : 4] = __builtin_amdgcn_s_memtime();      // table k-step 0 / the last one has landed
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int m = 0; m < 12; ++m) {
                const int g = 12 * s + m;
                const int rh = (m >> 1) & 1, ch = m & 1, u = 2 * s + ch, pr = m >> 2;
                if (WC_SPLIT_ABL & 2) { asm volatile("" :: "v"(ah[rh]), "v"(al[rh]), "v"(bhi[u]), "v"(blo[u])); }
                else if (pr == 0) acc[rh][ch] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[rh], bhi[u], acc[rh][ch], 0, 0, 0);
                else if (pr == 1) acc[rh][ch] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[rh], blo[u], acc[rh][ch], 0, 0, 0);
                else acc[rh][ch] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[rh], bhi[u], acc[rh][ch], 0, 0, 0);
                if (s + 1 < KS32) {
                    if (m == 0) nh[0] = frag(0, 0, s + 1);
                    if (m == 1) nh[1] = frag(0, 1, s + 1);
                    if (m == 2) nl[0] = frag(1, 0, s + 1);
                    if (m == 3) nl[1] = frag(1, 1, s + 1);
                }
                if (PUB_ >= 0 && g == (FIRST_ ? G - 6 : WC_SPLIT_PUB_GAP)) {      // my pieces of tile t+1 have landed: publish (tile 0: behind the table)
                    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PUB_) : "memory");
                    if (WC_SPLIT_STAMPS && FIRST_) pro_[5] = __builtin_amdgcn_s_memtime();      // my pieces of tile 1 have landed (waited for behind the table)
                    arrive(0);
                }
                if (DMA_ && g == 12 * DSTEP) {      // tile t-1 read by all: its buffer takes tile t+3
                    wait_for(1 + ((t + 3) & (NBUF - 1)), 8 * ((t + 3) / NBUF));
                    dma_tile(t + 3);
                }
                if (PEND_ && g >= SG0 && (g - SG0) % SGS == 0 && (g - SG0) / SGS < 16) store_pair(pend_po, (g - SG0) / SGS, pend[(g - SG0) / SGS]);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int rh = 0; rh < 2; ++rh) { ah[rh] = nh[rh]; al[rh] = nl[rh]; }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        arrive(1 + (t & (NBUF - 1)));              // my reads of tile t have returned
        unsigned long long c2_ = 0;
        if (WC_SPLIT_STAMPS) { c2_ = __builtin_amdgcn_s_memtime(); t_loop += c2_ - c1_; }
        // epilogue (affine_ring_kernel's): scale, bias, activation; v_permlane16_swap of the two column halves' registers gives
        // 2 rows x 32 columns per register -- 128-byte row pieces.  Stores: SGPR row base + one lane offset.
        int tq = t;
        asm volatile("" : "+s"(tq));      // opaque: no store bases of the peeled first tile precomputed (and spilled) ahead of the loop
        const float* const po = a.out + (int64_t)tile_of(tq) * (TR * C);      // wave-uniform
        float res[16];
        unsigned bits = 0;
        // PL: the planes hold os * y, folded into the two per-column constants HERE, per tile: the constants arrive by asm loads whose
        // landing only the k-loop's counted waits guarantee, and a product that depends on nothing in the loop is hoisted above them
        // by hipcc (first version: tile 0 of a workgroup scaled by whatever the registers held -- 0.06 % of the outputs wrong, racily).
        // The empty asm ties the constants to this point of the program.
        if (PL) asm volatile("" : "+v"(cscale[0]), "+v"(cscale[1]), "+v"(addv[0]), "+v"(addv[1]) :: "memory");
        const float cs0 = PL ? cscale[0] * os : cscale[0], cs1 = PL ? cscale[1] * os : cscale[1];
        const float ad0 = PL ? addv[0] * os : addv[0], ad1 = PL ? addv[1] * os : addv[1];
        auto leave = [&](auto RL_) __attribute__((always_inline)) {
            constexpr bool RL = decltype(RL_)::value;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int rh = i >> 2, r = i & 3;
                float v0 = acc[rh][0][r] * cs0 + ad0, v1 = acc[rh][1][r] * cs1 + ad1;
                if (RL) {
                    v0 = !(v0 <= 0.f) ? v0 : 0.f;
                    v1 = !(v1 <= 0.f) ? v1 : 0.f;
                }
                asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(v0), "+v"(v1));
                if (MASK && !(WC_SPLIT_ABL & 16)) {      // after the ReLU a value is +0 or passes: v0 is row 16 rh + r (+ 8 in lanes 32-63), v1 four rows below; column l31
                    const unsigned b0 = __builtin_bit_cast(unsigned, v0), b1 = __builtin_bit_cast(unsigned, v1);
                    bits |= (b0 < 1u ? b0 : 1u) << (16 * rh + r);
                    bits |= (b1 < 1u ? b1 : 1u) << (16 * rh + r + 4);
                }
                if constexpr (PL) {
                    omax = __builtin_fmaxf(omax, __builtin_fmaxf(fabsf(v0), fabsf(v1)));
                    const unsigned H = pk_rne2(v0, v1);
                    float r0, r1;
                    asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(H), "v"(v0));
                    asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(H), "v"(v1));
                    const unsigned L = pk_rne2(r0, r1);
                    const unsigned Hn = (unsigned)__builtin_amdgcn_update_dpp(0, (int)H, 0xB1, 0xF, 0xF, false);
                    const unsigned Ln = (unsigned)__builtin_amdgcn_update_dpp(0, (int)L, 0xB1, 0xF, 0xF, false);
                    res[2 * i] = __builtin_bit_cast(float, __builtin_amdgcn_perm(Hn, H, pl_sel));
                    res[2 * i + 1] = __builtin_bit_cast(float, __builtin_amdgcn_perm(Ln, L, pl_sel));
                } else {
                    res[2 * i] = v0; res[2 * i + 1] = v1;      // rows r (lanes 0-31) and 8 + r | rows 4 + r and 12 + r: columns l31
                }
            }
        };
        if (MASK || a.relu) leave(std::true_type{}); else leave(std::false_type{});
        if (PL) {        // 16 dword stores as the fp32 form, in 64-byte pieces: value i = (hi, lo) word of rows 16 rh + r (+ 4 in odd lanes)
            const char* const bh = reinterpret_cast<const char*>(a.phi + (int64_t)tile_of(tq) * (TR * C));
            const char* const bl = reinterpret_cast<const char*>(a.plo + (int64_t)tile_of(tq) * (TR * C));
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int rh = i >> 2, r = i & 3;
                store_u32(bh + (16 * rh + r) * (C * 2), pl_b, __builtin_bit_cast(unsigned, res[2 * i]));
                store_u32(bl + (16 * rh + r) * (C * 2), pl_b, __builtin_bit_cast(unsigned, res[2 * i + 1]));
            }
        } else if (DEFER_) {
#pragma unroll
            for (int i = 0; i < 16; ++i) pend[i] = res[i];
            pend_po = po;
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) store_pair(po, i, res[i]);
        }
        if (MASK && !(WC_SPLIT_ABL & 16)) {
            // lanes l and l + 32 hold the two halves of column l31's 32 row bits (rows +0..7, +16..23 | +8..15, +24..31); one more
            // store per tile than the schedule's count of 16: a hand-counted wait only gets more conservative by it
            unsigned own = bits, other = bits;
            asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(own), "+v"(other));
            const unsigned* pm = a.maskout + ((int64_t)tile_of(tq) * (TR / 32) + rg) * C;      // wave-uniform
            const unsigned word = own | (other << 8);
            if (lh == 0) store_u32(pm, mk_b, word);
        }
        if (WC_SPLIT_STAMPS) t_store += __builtin_amdgcn_s_memtime() - c2_;
    };
    using N_ = std::integral_constant<int, -1>;
    using P0 = std::integral_constant<int, 0>; using P4 = std::integral_constant<int, 4>; using P8 = std::integral_constant<int, 8>;
    using P20 = std::integral_constant<int, 20>; using P32 = std::integral_constant<int, 32>; using P36 = std::integral_constant<int, 36>;
    using T_ = std::true_type; using F_ = std::false_type;
    auto pick_table = [&](int t) {       // conditional tables: a new slot's B' fragments (a full drain: every later count stays conservative)
        if (HAS_SLOT) {
            const int slot = a.slot[((int64_t)tile_of(t) * TR) / a.HW];
            if (slot != cur_slot) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); load_b(slot); touch_b(); }      // (load_b drains: plain loads with slots)
        }
    };
    unsigned long long k0_ = 0, rt_loop = 0;
    if (WC_SPLIT_STAMPS) { k0_ = __builtin_amdgcn_s_memtime(); rt_loop = __builtin_amdgcn_s_memrealtime(); }
    // The vector-memory operations of a wave in issue order (DEF): D0, table, D1, D2 | tile 0: D3 | tile 1: D4, S0 | tile 2: D5, S1 |
    // ... | tile t: D(t+3) if it exists, S(t-1) | ... | tile n-1: S(n-2), S(n-1)   (D = 4 DMAs, S = 16 stores).  Tile t+1 is
    // published in tile t AHEAD of that tile's own operations (tile 0: behind them): what is younger than D(t+1) then gives
    // the counts 8, 4, 20, 36 ... 36, 32.
    for (int t = 0; t < n; ++t) {
        if (t > 0) pick_table(t);
        const bool dma = t + 3 < n;
        if (!def_mode) {
            if (t + 1 == n) tile_body(t, N_{}, F_{}, F_{}, F_{}, F_{});
            else if (dma) tile_body(t, P0{}, T_{}, F_{}, F_{}, F_{});
            else tile_body(t, P0{}, F_{}, F_{}, F_{}, F_{});
        }
#if !WC_SPLIT_DEFER
        // D0, table, D1, D2 | tile 0: D3, S0 | tile 1: D4, S1 | ... : 8, 20, 36 ... 36, 32
        // (with slots the table is complete before tile 0 and tile 1 is published EARLY in tile 0's loop, ahead of D3: younger than D1 is D2 only)
        else if (t == 0) {
            if (PRE3) tile_body(t, P8{}, F_{}, T_{}, F_{}, F_{});          // (D3 went out in the prologue: younger than D1 are D2 and D3, as before)
            else if (ASYNC_TABLE && !(WC_SPLIT_ABL & 8)) tile_body(t, P8{}, T_{}, T_{}, F_{}, F_{});
            else tile_body(t, P4{}, T_{}, F_{}, F_{}, F_{});
        }
        else if (t == 1) tile_body(t, P20{}, T_{}, F_{}, F_{}, F_{});
        else if (dma) tile_body(t, P36{}, T_{}, F_{}, F_{}, F_{});
        else if (t + 2 < n) tile_body(t, P36{}, F_{}, F_{}, F_{}, F_{});
        else if (t + 1 < n) tile_body(t, P32{}, F_{}, F_{}, F_{}, F_{});
        else tile_body(t, N_{}, F_{}, F_{}, F_{}, F_{});
#else
        else if (t == 0) tile_body(t, P8{}, T_{}, T_{}, F_{}, T_{});
        else if (t == 1) tile_body(t, P4{}, T_{}, F_{}, T_{}, T_{});
        else if (t == 2) tile_body(t, P20{}, T_{}, F_{}, T_{}, T_{});
        else if (dma) tile_body(t, P36{}, T_{}, F_{}, T_{}, T_{});
        else if (t + 2 < n) tile_body(t, P36{}, F_{}, F_{}, T_{}, T_{});
        else if (t + 1 < n) tile_body(t, P32{}, F_{}, F_{}, T_{}, T_{});
        else tile_body(t, N_{}, F_{}, F_{}, T_{}, F_{});
#endif
    }
    if (WC_SPLIT_STAMPS && a.dbg && lane == 0) {
        unsigned long long* d = a.dbg + ((int64_t)blockIdx.x * 8 + wave) * 16;
        d[0] = t_wait; d[1] = t_loop; d[2] = t_store; d[3] = __builtin_amdgcn_s_memtime() - k0_;
        d[4] = rt_in; d[5] = rt_loop; d[6] = __builtin_amdgcn_s_memrealtime();      // 100 MHz, chip-wide
#pragma unroll
        for (int i = 0; i < 8; ++i) d[8 + i] = pro_[i];
    }

    // Exact redo of every tile that STRADDLES samples of different slots (HW not a multiple of the tile): the MFMA pass used
    // the table of the tile's first sample for all of its rows.  Same thread, same element, so the second store wins by
    // program order (after a drain: the first pass's stores are asm, invisible to hipcc).
    if (HAS_SLOT && a.mixed) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        for (int t = 0; t < n; ++t) {
            const int64_t r0 = (int64_t)tile_of(t) * TR;
            const int64_t n0 = r0 / a.HW, n1 = (r0 + TR - 1) / a.HW;
            bool mixed = false;
            const int s0 = a.slot[n0];
            for (int64_t q = n0 + 1; q <= n1; ++q) mixed |= (a.slot[q] != s0);
            if (!mixed) continue;
            for (int i = 0; i < 16; ++i) {
                const int row = rbase + 16 * (i >> 3) + 8 * lh + 4 * ((i >> 2) & 1) + (i & 3), ecol = cg * 32 + l31;
                const int slot = a.slot[(r0 + row) / a.HW];
                const float* Bf = a.Bf + (int64_t)slot * a.bf_stride + ecol;
                const _Float16* xh = a.xs + (r0 + row) * C;
                const _Float16* xl = xh + a.plane;
                float accf = 0.f;
                for (int k = 0; k < C; ++k) accf = fmaf(((float)xh[k] + (float)xl[k]) / a.xscale[k], Bf[(int64_t)k * C], accf);
                const float v = accf + a.bias[(int64_t)slot * C + ecol];
                const float o = (MASK || a.relu) ? (!(v <= 0.f) ? v : 0.f) : v;
                if (PL) {
                    const float so = o * os;
                    const _Float16 h = (_Float16)so;
                    a.phi[(r0 + row) * C + ecol] = h;
                    a.plo[(r0 + row) * C + ecol] = (_Float16)(so - (float)h);
                    omax = __builtin_fmaxf(omax, fabsf(so));
                } else a.out[(r0 + row) * C + ecol] = o;
                if (MASK) {       // the redone element's mask bit (rare path: atomics on the word it shares with 31 rows)
                    unsigned* pm = a.maskout + ((r0 + row) >> 5) * C + ecol;
                    const unsigned bit = 1u << ((r0 + row) & 31);
                    if (__builtin_bit_cast(unsigned, o) != 0u) atomicOr(pm, bit); else atomicAnd(pm, ~bit);
                }
            }
        }
    }
    if (PL && a.oamax) {        // this workgroup's max |scaled output| (one partial per workgroup: deterministic, nothing to clear)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) omax = __builtin_fmaxf(omax, __shfl_xor(omax, o));
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __syncthreads();
        if (lane == 0) cnt[8 + wave] = __builtin_bit_cast(int, omax);
        __syncthreads();
        if (tid == 0) {
            float m = 0.f;
            for (int w = 0; w < 8; ++w) m = __builtin_fmaxf(m, __builtin_bit_cast(float, (int)cnt[8 + w]));
            a.oamax[blockIdx.x] = m;
        }
    }
}

template <int C>
hipError_t launch_apply_split(const SplitApplyArgs& a0, hipStream_t st)
{
    constexpr int TR = 8192 / C;
    constexpr size_t lds = 4 * 32768 + 64;
    SplitApplyArgs a = a0;
    a.ntiles = (int)(a.M / TR);
    int nwg = a.ntiles < 256 ? a.ntiles : 256;
    a.tiles_per_wg = (a.ntiles + nwg - 1) / nwg;
    nwg = (a.ntiles + a.tiles_per_wg - 1) / a.tiles_per_wg;
#define WC_LAUNCH_SPLIT(SLOT_, MASK_, PL_)                                                                              \
    do {                                                                                                                \
        static bool attr_set = false;                                                                                   \
        if (!attr_set) {                                                                                                \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(apply_split_kernel<C, SLOT_, MASK_, PL_>), \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                   \
            if (e != hipSuccess) return e;                                                                              \
            attr_set = true;                                                                                            \
        }                                                                                                               \
        hipLaunchKernelGGL((apply_split_kernel<C, SLOT_, MASK_, PL_>), dim3(nwg), dim3(512), lds, st, a);               \
    } while (0)
    const bool mask = a.maskout != nullptr;
    const bool has_slot = a.slot != nullptr;
    if (a.phi != nullptr) {
        // planes form: the pass itself, then the same kernel behind the gate (leaves at once unless the predicted scale overflowed);
        // oscale: the record of wc_launch_out_scale, exactly as affine_ring_kernel<.., PL> uses it (wc_fast.hip)
        if (nwg > kSplitPlaneBounds || a.oscale == nullptr) return hipErrorInvalidValue;
        a.oamax = a.oscale + 2 + kSplitPlaneBounds; a.gate = nullptr; a.ngate = nwg;
        for (int pass = 0; pass < 2; ++pass) {
            if (has_slot) { if (mask) WC_LAUNCH_SPLIT(true, true, true); else WC_LAUNCH_SPLIT(true, false, true); }
            else { if (mask) WC_LAUNCH_SPLIT(false, true, true); else WC_LAUNCH_SPLIT(false, false, true); }
            a.gate = a.oscale + 2 + kSplitPlaneBounds; a.oamax = nullptr;
        }
        return hipGetLastError();
    }
    if (has_slot) { if (mask) WC_LAUNCH_SPLIT(true, true, false); else WC_LAUNCH_SPLIT(true, false, false); }
    else { if (mask) WC_LAUNCH_SPLIT(false, true, false); else WC_LAUNCH_SPLIT(false, false, false); }
#undef WC_LAUNCH_SPLIT
    return hipGetLastError();
}

}  // namespace

static void* g_split_dbg = nullptr;
extern "C" void wc_dev_split_dbg(void* p) { g_split_dbg = p; }      // WC_SPLIT_STAMPS builds: where the stamps go (development only)

bool wc_split_apply_supported(int64_t N, int64_t HW, int C)
{
    if (!(C == 128 || C == 256)) return false;
    const int64_t M = N * HW;
    return M > 0 && (M % (8192 / C)) == 0;
}

hipError_t wc_launch_split_rows(const float* x, const float* center, const float* scale, int64_t M, int C, int relu,
                                void* xs, int* flag, hipStream_t st)
{
    const int64_t n8 = M * C / 8;
    int64_t blocks = (n8 + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    _Float16* hi = static_cast<_Float16*>(xs);
    hipLaunchKernelGGL(split_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, center, scale, n8, C, relu, hi, hi + M * C, flag);
    return hipGetLastError();
}

hipError_t wc_launch_unsplit_rows(const void* xs, const float* center, const float* scale, int64_t M, int C, float* x, hipStream_t st)
{
    const int64_t n8 = M * C / 8;
    int64_t blocks = (n8 + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    const _Float16* hi = static_cast<const _Float16*>(xs);
    hipLaunchKernelGGL(unsplit_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, st, hi, hi + M * C, center, scale, n8, C, x);
    return hipGetLastError();
}

hipError_t wc_launch_split_bias(const float* A, const float* bias, const float* center, const float* mu, int Kc, int C,
                                float* out, hipStream_t st)
{
    hipLaunchKernelGGL(split_bias_kernel, dim3(C / 32, Kc), dim3(256), 0, st, A, bias, center, mu, C, out);
    return hipGetLastError();
}

// plan: the tables of A built for scale = the split tensor's scales (wc_color_f32 with chan_scale = xs_scale, or
// wc_launch_fast_plan_tables); bias2 from wc_launch_split_bias
hipError_t wc_launch_apply_split(const void* xs, const float* xs_scale, const float* A, int Kc, const float* bias2,
                                 const int32_t* slot, int64_t N, int64_t HW, int C, int relu, float* y,
                                 const void* plan_hi, const void* plan_lo, const float* plan_colscale, void* dbg, hipStream_t st,
                                 unsigned* relu_mask, void* planes, float* oscale)
{
    SplitApplyArgs a = {};
    a.xs = static_cast<const _Float16*>(xs); a.plane = N * HW * C;
    a.Bhi = static_cast<const _Float16*>(plan_hi); a.Blo = static_cast<const _Float16*>(plan_lo); a.colscale = plan_colscale;
    a.slot_stride = (int64_t)C * C;
    a.bias = bias2; a.slot = slot; a.M = N * HW; a.HW = HW; a.relu = relu;
    a.mixed = (slot != nullptr && (HW % (8192 / C)) != 0) ? 1 : 0;
    a.Bf = A; a.bf_stride = (int64_t)C * C; a.xscale = xs_scale; a.out = y;
    a.maskout = relu_mask;
    if (relu_mask && (!relu || ((N * HW) % 32) != 0)) return hipErrorInvalidValue;
    if (planes) {       // the output as the next convolution's fp16 planes (hi | lo, N*HW*C halves each)
        if (!oscale) return hipErrorInvalidValue;
        a.phi = static_cast<_Float16*>(planes); a.plo = a.phi + N * HW * C; a.oscale = oscale;
    }
    a.dbg = static_cast<unsigned long long*>(dbg ? dbg : (WC_SPLIT_STAMPS ? g_split_dbg : nullptr));
    (void)Kc;
    switch (C) {
        case 128: return launch_apply_split<128>(a, st);
        case 256: return launch_apply_split<256>(a, st);
    }
    return hipErrorInvalidValue;
}
