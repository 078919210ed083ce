// K1 on a PRE-SPLIT input (wc_split.hip's format): the covariance moments of x given as fp16 hi/lo planes,
//     P[slab] = sum_{m in slab} g[m]^T g[m] / (s_i s_j),   g = hi + lo = (x - center) scale     (+ column sums, + the diagonal)
// i.e. wc_stats_f32's reduction without its conversion.  wc_fast_xty.hip spends 12 vector instructions per MFMA on
// centre / scale / split / transpose of every row, in each of the two workgroups of a slab, and its matrix pipe idles while
// all eight waves convert (22 % MFMA busy, 0.30 of the HBM peak).  Here the planes go HBM -> LDS by
// LDS-DMA and nothing converts them:
//
//  * Both MFMA operands are indexed [channel][row] (the contraction runs over rows) while the planes are [row][channel]:
//    the transpose is the LDS READ.  A DMA piece (1 KiB, one wave-instruction) is 8 rows x 64 channels -- 128 B = one cache
//    line per row from 8 lanes -- and a 32x32x16 operand fragment (8 consecutive rows of one channel per lane) comes out of
//    two ds_read_b64_tr_b16 (a 16-lane group reads 4 rows x 16 channels and receives them channel-major).  The two 64-byte
//    halves of rows 2, 3, 6, 7 of a piece are swapped (in the DMA's per-lane SOURCE address: the LDS side of a DMA is
//    lane-linear), so that the 4 x 64 B a 32-lane half reads cover all 64 banks once.  (The first version fetched
//    16 rows x 64 B per instruction, the layout of wc_conv.hip's weight-gradient kernel, conflict-free as it stands: every
//    line was then requested by two instructions, and the bare wait / barrier / DMA skeleton of the loop -- no MFMA, no
//    statistics -- already took 27 us for two passes over 134 MB.)
//  * Stages of 32 rows (2 k-steps; 32 KiB of hi | lo at C = 256) in four buffers: the DMAs of stage s+3 leave right behind
//    stage s's barrier -- the one workgroup barrier per stage, which also frees buffer (s-1) % 4 -- so 96 KiB per CU are in
//    flight or landed ahead of the MFMAs.  Hand-counted vmcnt: a wave waits for its own pieces of stage s with the pieces
//    of s+1 and s+2 still in flight.
//  * Accuracy as in wc_fast_xty.hip: fp32 MFMA chains of 12 steps (two stages) flushed into float64
//    registers; the block-upper triangle only (36 of 64 blocks at C = 256, 12 per workgroup, three workgroups per slab on one
//    XCD, the later readers served by the L2); the covariance's DIAGONAL and the column sums on the VALU -- the matrix
//    pipe's fp32 sums of all-positive products are biased (section 4.6) -- from the same LDS image: v_fma_mix_f32 forms
//    hi + lo straight from the fp16 halves, 2 vector instructions per element in ONE of the slab's workgroups each.
//  * No range gate: the planes cannot overflow (the producer clamps and flags, wc_split.hip).
// Partials in the layout of wc_fast_xty.hip, finished by the same stats_colsum / stats_xtx kernels (wc_small.hip).
#include "wc_common.h"
#include <stdlib.h>
#include <type_traits>

#ifndef SXT_STAMPS
#define SXT_STAMPS 0
#endif
#ifndef SXT_DMA_FRONT
#define SXT_DMA_FRONT 0     // 1: a stage's DMA pieces leave right behind its barrier (0: behind the block-steps' MFMAs)
#endif
#ifndef SXT_ALT
#define SXT_ALT 0      // 1: every other MFMA chain runs on the negated A fragments and is subtracted at the flush -- the matrix pipe's accumulation bias cancels (wc_fast_xty.hip, XTY_ALT; measured and left off: DESIGN.md section 2)
#endif
#ifndef SXT_ABL
#define SXT_ABL 0      // development ablation bits: 1 no VALU statistics, 2 no fragment reads / MFMAs, 4 no DMAs after the prologue
#endif

namespace {

typedef short s16x4v __attribute__((__vector_size__(4 * sizeof(short))));
typedef short s16x8v __attribute__((__vector_size__(8 * sizeof(short))));

// rows q..q+3 and q+4..q+7 of this lane's channel: two transposing reads 512 B (4 rows of 128 B) apart
__device__ __forceinline__ f16x8 tr_read8(const char* p)
{
    auto a = (__attribute__((address_space(3))) s16x4v*)((__attribute__((address_space(3))) char*)(p));
    auto b = (__attribute__((address_space(3))) s16x4v*)((__attribute__((address_space(3))) char*)(p + 512));
    const s16x4v x = __builtin_amdgcn_ds_read_tr16_b64_v4i16(a);
    const s16x4v y = __builtin_amdgcn_ds_read_tr16_b64_v4i16(b);
    return __builtin_bit_cast(f16x8, __builtin_shufflevector(x, y, 0, 1, 2, 3, 4, 5, 6, 7));
}

struct SplitXtxArgs {
    const _Float16* xs; int64_t plane;       // hi plane at xs, lo plane at xs + plane (elements)
    const float* scale;                      // [C]: the planes' power-of-two scales
    int64_t N, HW;
    int per_sample, nsplit;
    int64_t rows_per_slab;
    int nslab, ntypes;
    double* P;                               // [nslab][C][C], block-upper triangle
    float* colsum;                           // [nslab][C]: sum of g / s
    double* dfix;                            // [nslab][C]: sum of (g / s)^2
    unsigned long long* dbg;
};

template <int C>
__global__ __launch_bounds__(512, 1) void xtx_split_kernel(SplitXtxArgs a)
{
    static_assert(C == 128 || C == 256, "split covariance: C = 128 or 256");
    constexpr int NB = C / 32;                        // channel blocks
    constexpr int NBLK = NB * (NB + 1) / 2;           // block-upper triangle
    constexpr int RS = 32;                            // rows per stage (2 k-steps of 16)
    constexpr int PIECES = 2 * 2 * NB;                // 1-KiB pieces per stage: plane x k-step x channel block
    constexpr int PPW = PIECES / 8;                   // per wave
    constexpr int STAGE = PIECES * 1024;
    constexpr int NBUF = 4;
    constexpr int BW = 2;                             // 32x32 blocks per wave (at most)
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) char*)smem);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;

    // workgroup -> (slab, type): the ntypes workgroups of a slab sit 8 apart in block order (same XCD: the second reader hits L2)
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int type = q % a.ntypes;
    const int64_t z = (int64_t)(q / a.ntypes) * 8 + xcd;
    if (z >= a.nslab) return;
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
    const int nst = (int)((r1 - r0) / RS);            // whole stages (the plan guarantees multiples of 64 rows)

    // this wave's blocks (ib <= jb) of the upper triangle.  C = 256: 36 blocks as 12 per workgroup type (three types per slab),
    // waves 0-3 two blocks, waves 4-7 one: three blocks per SIMD (waves w and w + 4 share one) and at most two per wave, which
    // leaves the registers for a second set of fragments -- the loop below reads block-step n + 1 while block-step n is on
    // the matrix pipe.  (18 per workgroup in two types, three blocks on two of the waves, was the first version: 256 VGPRs, no
    // room to prefetch, every block-step's LDS latency exposed: 57 us, slower than the converting kernel.)  C = 128: 10
    // blocks, waves 0 and 1 two, the others one.
    int ib[BW], jb[BW];
    const int nlive = __builtin_amdgcn_readfirstlane((C == 256) ? (wave < 4 ? 2 : 1) : (wave < 2 ? 2 : 1));
#pragma unroll
    for (int b = 0; b < BW; ++b) {
        int L;
        if (C == 256) L = type * 12 + (wave < 4 ? 2 * wave + b : 8 + (wave - 4));
        else L = (wave < 2 ? wave * 2 + b : 4 + (wave - 2));
        if (b >= nlive) L = 0;
        int i = 0; while (L >= NB - i) { L -= NB - i; ++i; }
        ib[b] = __builtin_amdgcn_readfirstlane(i); jb[b] = __builtin_amdgcn_readfirstlane(i + L);
    }

    // DMA: piece p = PPW wave + i of a stage = (plane pl, k-step ks, row half rr, 64-channel group cg); lane -> row lane / 8 of
    // the piece's 8, 16 bytes (8 channels) j of the row's 128, fetched from j ^ 4 in rows 2, 3, 6, 7 (the bank swizzle)
    constexpr int NG = NB / 2;                        // 64-channel groups
    unsigned voff[PPW];
    const char* pbase[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int p = PPW * wave + i, pl = p / (2 * NB), ks = (p / NB) % 2, rr = (p / NG) % 2, cg = p % NG;
        const int row = lane >> 3, j = (lane & 7) ^ (((row >> 1) & 1) << 2);
        voff[i] = (unsigned)(((16 * ks + 8 * rr + row) * C + cg * 64 + j * 8) * 2);
        pbase[i] = reinterpret_cast<const char*>(a.xs + (int64_t)pl * a.plane + r0 * C);
    }
    auto dma_piece = [&](int s, int i) __attribute__((always_inline)) {
        const unsigned l0 = lds0 + (unsigned)(s & (NBUF - 1)) * STAGE + (unsigned)(PPW * wave) * 1024u;
        {
            const char* g = pbase[i] + (int64_t)s * (RS * C * 2);       // wave-uniform
            const unsigned l = __builtin_amdgcn_readfirstlane(l0 + i * 1024u);
            unsigned keep;
            // (s_nop 4: an SGPR operand restored from a spill lane needs five wait states before a VMEM instruction reads it and
            // hipcc's hazard pass does not look inside asm statements -- wc_split.hip)
            asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(voff[i]), "s"(l), "s"(g) : "memory");
        }
    };
    auto dma_stage = [&](int s) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < PPW; ++i) dma_piece(s, i);
    };

    // transposing-read address of this lane for channel block cb = 2 cg + h of k-step ks: piece (ks, rr = lane / 32, cg); 16-lane
    // group g reads channels 16 (g % 2) + 0..15 of the block, lane 4 q + p of the group supplies row q (and q + 4: + 512 B),
    // channels 4 p .. 4 p + 3; the block's 64-byte half sits at h ^ (q / 2 % 2)
    int tr_off[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int qq = (lane & 15) >> 2;
        tr_off[h] = (lane >> 5) * (NG * 1024) + qq * 128 + ((h ^ (qq >> 1)) & 1) * 64 + ((lane >> 4) & 1) * 32 + (lane & 3) * 8;
    }
    // VALU statistics (waves 4-7): thread -> the 16 bytes at position tid % 64 of piece tid / 64 - 4 (+ 4 per step): the same 8
    // channels in every stage
    const bool want_csum = a.colsum != nullptr && type == 0;
    const bool want_dfix = a.dfix != nullptr && type == (a.ntypes > 1 ? 1 : 0);
    float csum[8], sq[8];
    double lsq[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { csum[e] = 0.f; sq[e] = 0.f; lsq[e] = 0.0; }
    constexpr int VP = (2 * NB * 64) / 256;           // 16-byte pieces per thread, plane and stage (waves 4-7 only): 4 (C = 256), 2 (C = 128)
    auto valu_stats = [&](const char* sb, bool fold, auto MODE_) __attribute__((always_inline)) {
        constexpr int MODE = decltype(MODE_)::value;     // 1: column sums, 2: squares, 3: both
#pragma unroll
        for (int u = 0; u < VP; ++u) {
            const int off = ((tid - 256) + 256 * u) * 16;
            const uint4 h = *reinterpret_cast<const uint4*>(sb + off);
            const uint4 l = *reinterpret_cast<const uint4*>(sb + 2 * NB * 1024 + off);
            const unsigned hw[4] = {h.x, h.y, h.z, h.w}, lw[4] = {l.x, l.y, l.z, l.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float g0, g1;       // g = hi + lo in one mixed-precision FMA per element (exact: both are fp16, the sum fits fp32)
                asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(g0) : "v"(hw[e]), "v"(lw[e]));
                asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(g1) : "v"(hw[e]), "v"(lw[e]));
                if (MODE & 1) { csum[2 * e] += g0; csum[2 * e + 1] += g1; }
                if (MODE & 2) { sq[2 * e] = fmaf(g0, g0, sq[2 * e]); sq[2 * e + 1] = fmaf(g1, g1, sq[2 * e + 1]); }
            }
        }
        if ((MODE & 2) && fold) {       // short fp32 chains (2 VP terms per flush period), folded into float64
#pragma unroll
            for (int e = 0; e < 8; ++e) { lsq[e] += (double)sq[e]; sq[e] = 0.f; }
        }
    };

    // fragment offsets of this wave's blocks inside a stage's plane: 64-channel group (ib / 2) of k-step 0, rows by lane
    int fa_off[BW], fb_off[BW];
#pragma unroll
    for (int b = 0; b < BW; ++b) {
        fa_off[b] = (ib[b] >> 1) * 1024 + ((ib[b] & 1) ? tr_off[1] : tr_off[0]);
        fb_off[b] = (jb[b] >> 1) * 1024 + ((jb[b] & 1) ? tr_off[1] : tr_off[0]);
    }
    double acc64[BW][16];
#pragma unroll
    for (int b = 0; b < BW; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc64[b][r] = 0.0;
    f32x16 acc[BW];
#pragma unroll
    for (int b = 0; b < BW; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;

    unsigned long long t0_ = 0, t_wait = 0, t_valu = 0, t_mfma = 0;
    if (SXT_STAMPS) t0_ = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int s = 0; s < 3; ++s) if (s < nst) dma_stage(s);

    // One stage.  NL_: this wave's blocks; ZERO_: the blocks' accumulators start from zero (the previous stage flushed them);
    // FL_: flush into float64 behind the stage's MFMAs -- a chain is [ZERO_ stage, FL_ stage] = 12 MFMA accumulations.
    // The two waves of a SIMD (w and w + 4) flush in ALTERNATE stages, and the statistics are taken by waves 4-7 only, which
    // own one block where waves 0-3 own two: a stage is one barrier interval for all eight waves, and with every wave in the
    // same phase the vector ALU (two float64 instructions per accumulator element and flush) and the matrix pipe took
    // turns -- the first version ran 12 vector instructions per MFMA, as many as the converting kernel, at 59 us.
    auto stage = [&](int s, auto NL_, auto ZERO_, auto FL_, auto MODE_, bool neg) __attribute__((always_inline)) {
        constexpr int NL = decltype(NL_)::value;
        constexpr bool ZERO = decltype(ZERO_)::value, FL = decltype(FL_)::value;
        const unsigned sgn = (SXT_ALT && neg) ? 0x80008000u : 0u;       // (wave-uniform: the sign of this stage's chain)
        unsigned long long c0_ = 0;
        if (SXT_STAMPS) c0_ = __builtin_amdgcn_s_memtime();
        // my pieces of stage s have landed (younger: my pieces of stages s+1 and s+2, where those exist)
        if (s + 2 < nst) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * PPW) : "memory");
        else if (s + 1 < nst) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // every wave's pieces of stage s have landed, and every wave has finished with stage s-1 (its fragments were consumed by
        // MFMAs, its statistics reads by the VALU): buffer (s+3) % 4 = (s-1) % 4 is free
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        const bool dma_on = s + 3 < nst && !(SXT_ABL & 4);
        if (dma_on && (SXT_DMA_FRONT || (SXT_ABL & 2))) dma_stage(s + 3);
        unsigned long long c1_ = 0;
        if (SXT_STAMPS) { c1_ = __builtin_amdgcn_s_memtime(); t_wait += c1_ - c0_; }
        const char* sb = smem + (s & (NBUF - 1)) * STAGE;
        // block-step u = (k-step u / NL, block u % NL): its four fragments (two transposing reads each) in set u & 1
        f16x8 F[2][4];                                 // [set][A hi, A lo, B hi, B lo]
        auto read_a = [&](int set, int ks, int b) __attribute__((always_inline)) {
            const char* pa = sb + ks * (NB * 1024) + fa_off[b];
            F[set][0] = tr_read8(pa); F[set][1] = tr_read8(pa + 2 * NB * 1024);
            if (SXT_ALT && sgn) {       // (a scalar branch: positive chains skip the eight v_xor)
                typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
                F[set][0] = __builtin_bit_cast(f16x8, __builtin_bit_cast(u32x4v, F[set][0]) ^ sgn);
                F[set][1] = __builtin_bit_cast(f16x8, __builtin_bit_cast(u32x4v, F[set][1]) ^ sgn);
            }
        };
        auto read_b = [&](int set, int ks, int b) __attribute__((always_inline)) {
            const char* pb = sb + ks * (NB * 1024) + fb_off[b];
            F[set][2] = tr_read8(pb); F[set][3] = tr_read8(pb + 2 * NB * 1024);
        };
        if (!(SXT_ABL & 2)) { read_a(0, 0, 0); read_b(0, 0, 0); }      // the first block-step's reads fly under the statistics pass
        if (decltype(MODE_)::value != 0 && !(SXT_ABL & 1)) valu_stats(sb, FL, MODE_);
        unsigned long long c2_ = 0;
        if (SXT_STAMPS) { c2_ = __builtin_amdgcn_s_memtime(); t_valu += c2_ - c1_; }
        if (!(SXT_ABL & 2)) {
            constexpr int NU = 2 * NL;
            const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int b = u % NL, set = u & 1;
                const int un = u + 1, ksn = un / NL, bn = un % NL;
                const bool first = ZERO && u < NL;       // the block's first MFMA of a chain takes a zero C operand: no zeroing pass
                acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F[set][1], F[set][2], first ? zero : acc[b], 0, 0, 0);      // lo * Hi
                if (un < NU) read_a(set ^ 1, ksn, bn);
                __builtin_amdgcn_sched_barrier(0);
                acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F[set][0], F[set][3], acc[b], 0, 0, 0);      // hi * Lo
                if (un < NU) read_b(set ^ 1, ksn, bn);
                __builtin_amdgcn_sched_barrier(0);
                acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F[set][0], F[set][2], acc[b], 0, 0, 0);      // hi * Hi
                // this wave's DMA pieces of stage s+3 behind the block-steps' MFMAs (issued together behind the barrier, all
                // eight waves at once, they cost the stage ~400 cycles: MI355X_MICROARCH.md prices a piece at 60 cycles among
                // MFMAs and 100-185 in a busy phase)
                if (!SXT_DMA_FRONT && dma_on) {
#pragma unroll
                    for (int i = u * PPW / NU; i < (u + 1) * PPW / NU; ++i) dma_piece(s + 3, i);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (FL) {
            const double fsg = sgn ? -1.0 : 1.0;
#pragma unroll
            for (int b = 0; b < NL; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) { if (SXT_ALT) acc64[b][r] = __builtin_fma((double)acc[b][r], fsg, acc64[b][r]); else acc64[b][r] += (double)acc[b][r]; }
        }
        if (SXT_STAMPS) t_mfma += __builtin_amdgcn_s_memtime() - c2_;
    };
    using T_ = std::true_type; using F_ = std::false_type;
    using N1 = std::integral_constant<int, 1>; using N2 = std::integral_constant<int, 2>;
    using M0 = std::integral_constant<int, 0>;
    if (wave >= 4) {        // one block; chains [odd stage, even stage]; stage 0 continues the zero-initialised accumulator
        auto loop_b = [&](auto MODE_) __attribute__((always_inline)) {
            for (int s = 0; s < nst; s += 2) {      // chain k = stages 2k - 1 (zero start) and 2k (flush): stage s ends chain s / 2
                stage(s, N1{}, F_{}, T_{}, MODE_, ((s >> 1) & 1) != 0);
                stage(s + 1, N1{}, T_{}, F_{}, MODE_, (((s >> 1) + 1) & 1) != 0);
            }
        };
        const int mode = (want_csum ? 1 : 0) | (want_dfix ? 2 : 0);
        if (mode == 3) loop_b(std::integral_constant<int, 3>{});
        else if (mode == 2) loop_b(std::integral_constant<int, 2>{});
        else if (mode == 1) loop_b(std::integral_constant<int, 1>{});
        else loop_b(M0{});
        if (SXT_ALT && (nst & 2)) {                                         // the last (odd) stage's half chain: negated when nst = 2 (mod 4)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc64[0][r] -= (double)acc[0][r];
        } else
#pragma unroll
        for (int r = 0; r < 16; ++r) acc64[0][r] += (double)acc[0][r];      // the last (odd) stage's half chain
#pragma unroll
        for (int e = 0; e < 8; ++e) lsq[e] += (double)sq[e];                // ... and its squares
    } else if (nlive == 2) {
        for (int s = 0; s < nst; s += 2) {          // chain k = stages 2k, 2k + 1
            stage(s, N2{}, T_{}, F_{}, M0{}, ((s >> 1) & 1) != 0);
            stage(s + 1, N2{}, F_{}, T_{}, M0{}, ((s >> 1) & 1) != 0);
        }
    } else {
        for (int s = 0; s < nst; s += 2) {
            stage(s, N1{}, T_{}, F_{}, M0{}, ((s >> 1) & 1) != 0);
            stage(s + 1, N1{}, F_{}, T_{}, M0{}, ((s >> 1) & 1) != 0);
        }
    }
    if (SXT_STAMPS && a.dbg && lane == 0) {
        unsigned long long* d = a.dbg + ((int64_t)blockIdx.x * 8 + wave) * 4;
        d[0] = t_wait; d[1] = t_valu; d[2] = t_mfma; d[3] = __builtin_amdgcn_s_memtime() - t0_;
    }

    // partial blocks out, scales undone exactly (powers of two)
    double* P = a.P + z * (int64_t)C * C;
#pragma unroll
    for (int b = 0; b < BW; ++b) {
        if (b >= nlive) continue;
        const int j = jb[b] * 32 + l31;
        const double isj = 1.0 / (double)a.scale[j];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = ib[b] * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            P[(int64_t)i * C + j] = acc64[b][r] * isj / (double)a.scale[i];
        }
    }
    // column sums and squares: the threads of waves 4-7 that hold partial sums of the same 8 channels (8 rows of a piece; at
    // C = 128 two waves share a 64-channel group)
    __syncthreads();
    const int srow = (lane >> 3) & 7, wb = (tid >> 6) - 4;
    const int c0 = (wb % NG) * 64 + (((lane & 7) ^ (((srow >> 1) & 1) << 2)) * 8), rrow = srow + 8 * (wb / NG);
    constexpr int RROWS = 8 * (4 / NG);               // threads per channel: 8 (C = 256), 16 (C = 128)
    if (want_csum && wave >= 4) {
        float* red = reinterpret_cast<float*>(smem);
#pragma unroll
        for (int e = 0; e < 8; ++e) red[rrow * C + c0 + e] = csum[e];
    }
    if (want_dfix && wave >= 4) {
        double* red2 = reinterpret_cast<double*>(smem + RROWS * C * 4);
#pragma unroll
        for (int e = 0; e < 8; ++e) red2[rrow * C + c0 + e] = lsq[e];
    }
    __syncthreads();
    if (want_csum) {
        const float* red = reinterpret_cast<const float*>(smem);
        for (int c = tid; c < C; c += 512) {
            float sacc = 0.f;
            for (int g = 0; g < RROWS; ++g) sacc += red[g * C + c];
            a.colsum[z * C + c] = sacc / a.scale[c];
        }
    }
    if (want_dfix) {
        const double* red2 = reinterpret_cast<const double*>(smem + RROWS * C * 4);
        for (int c = tid; c < C; c += 512) {
            double sacc = 0.0;
            for (int g = 0; g < RROWS; ++g) sacc += red2[g * C + c];
            a.dfix[z * C + c] = sacc / ((double)a.scale[c] * (double)a.scale[c]);
        }
    }
}

template <int C>
hipError_t launch_xtx_split(const SplitXtxArgs& a, hipStream_t st)
{
    constexpr size_t lds = (size_t)4 * (2 * 2 * (C / 32)) * 1024;       // four stages: 128 KiB (C = 256), 64 KiB (C = 128)
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(xtx_split_kernel<C>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const int slab_groups = (a.nslab + 7) / 8;
    hipLaunchKernelGGL((xtx_split_kernel<C>), dim3(slab_groups * a.ntypes * 8), dim3(512), lds, st, a);
    return hipGetLastError();
}

}  // namespace

static void* g_sxt_dbg = nullptr;
extern "C" void wc_dev_split_xtx_dbg(void* p) { g_sxt_dbg = p; }      // SXT_STAMPS builds (development only)

// Plan: slabs of whole 64-row flush periods, ~ one workgroup per CU (wc_fast_xty_plan's rule).  Returns nslab (0 = not eligible).
int wc_split_xtx_plan(int64_t N, int64_t HW, int C, int per_sample, int* nsplit, int64_t* rows_per_slab, int* ntypes)
{
    if (!(C == 128 || C == 256)) return 0;
    const int64_t M = N * HW;
    if (M < wc_fast_xty_min_rows()) return 0;
    const int64_t seg = per_sample ? HW : M;
    if (seg % 64 != 0) return 0;
    *ntypes = (C == 256) ? 3 : 1;
    const int64_t nseg = per_sample ? N : 1;
    const int64_t target = (256 / (8 * *ntypes)) * 8;
    int64_t per_seg = target / nseg;
    if (per_seg < 1) per_seg = 1;
    const int64_t periods = seg / 64;
    if (per_seg > periods) per_seg = periods;
    const int64_t pp_slab = (periods + per_seg - 1) / per_seg;
    *rows_per_slab = pp_slab * 64;
    *nsplit = (int)((seg + *rows_per_slab - 1) / *rows_per_slab);
    return (int)(nseg * (*nsplit));
}

hipError_t wc_launch_split_xtx(const void* xs, const float* scale, int64_t N, int64_t HW, int C, int per_sample, int nsplit,
                               int64_t rows_per_slab, int nslab, int ntypes, double* P, float* colsum, double* dfix, hipStream_t st)
{
    SplitXtxArgs a = {};
    a.xs = static_cast<const _Float16*>(xs); a.plane = N * HW * C; a.scale = scale; a.N = N; a.HW = HW;
    a.per_sample = per_sample; a.nsplit = nsplit; a.rows_per_slab = rows_per_slab; a.nslab = nslab; a.ntypes = ntypes;
    a.P = P; a.colsum = colsum; a.dfix = dfix;
    a.dbg = static_cast<unsigned long long*>(SXT_STAMPS ? g_sxt_dbg : nullptr);
    switch (C) {
        case 128: return launch_xtx_split<128>(a, st);
        case 256: return launch_xtx_split<256>(a, st);
    }
    return hipErrorInvalidValue;
}
