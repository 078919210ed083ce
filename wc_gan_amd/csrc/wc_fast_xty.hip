// Fast path of the two big reductions, K1 (covariance moments, wc_stats_f32) and K4 (R = f^T gbar, wc_bwd_reduce_f32):
//   P[slab] = sum_{m in slab} ((X[m]-cx).*sx)^T ((Y[m]-cy).*sy)  / (sx_i sy_j)          (+ column sums)
// on the split-fp16 scheme of wc_fast.hip: v = hi + lo, three v_mfma_f32_32x32x16_f16 per 32x32 block and
// 16 rows, 3/16 of the f32-MFMA time.  What is specific here:
//
//  * Both MFMA operands are indexed [channel][row] (the contraction runs over rows), so the activation tile is
//    transposed while it is staged: a thread loads 16 B (4 channels) from each of 8 consecutive rows and then
//    holds, per channel, 8 consecutive rows = one 16-byte fp16 fragment chunk -- written with ds_write_b128 into an
//    XOR-swizzled [C][R] image.  No transposed LDS read, no shuffles.
//  * Accuracy of the covariance decides parity on ill-conditioned batches, so the fp32 MFMA
//    accumulators are flushed into FLOAT64 registers after every stage (64 rows: a 12-step fp32 chain).  That
//    costs 48 VGPRs per block, so a wave owns only 3 of the C/32 x C/32 blocks (the LDS images allow one 512-thread
//    workgroup per CU anyway, i.e. 256 VGPRs per thread: 3 blocks fit without spills, 4 do not) and the blocks of one
//    slab are spread over `ntypes` workgroups that stream the same rows (co-scheduled on one XCD: L2 serves the
//    re-reads).  3 instead of 2 blocks: 2 instead of 3 (covariance) / 3 instead of 4 (two operands) workgroups convert
//    every row -- K1 113.6 -> 83.5 us, K4 174.7 -> 146.4 us at 128x32x32x256.
//  * fp16 range: per-channel power-of-two scales from a row subsample, undone exactly in the float64 flush; an
//    out-of-range element raises the same device gate as in wc_fast.hip and the exact kernel redoes the call.
#include "wc_common.h"
#include <stdlib.h>
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
constexpr float kGuard = 60000.0f;
#ifndef XTY_ROLE31
#define XTY_ROLE31 1    // K4 on planes, quadrant form: the waves that stage X (one v_perm per image word) own three blocks of their block row, the waves that
#endif                  // stage and convert gy one -- see the stage loop.  0: two blocks each (rounds 4-5), for A/B
#ifndef XTY_STAMPS
#define XTY_STAMPS 0      // development: s_memtime stamps of wave 0 / workgroup 0 behind the partials (the caller adds 2 KiB to the workspace)
#endif
constexpr int BW_PLAIN = 3;                // 32x32 blocks per wave
#ifndef XTY_BAL
#define XTY_BAL 1
#endif
#ifndef XTY_ALLLIVE
#define XTY_ALLLIVE 1
#endif
#ifndef XTY_ALT
#define XTY_ALT 0       // 1: covariance (K1): every other MFMA chain runs on the NEGATED A fragments and is subtracted at the float64 flush (see the stage loop)
#endif
#ifndef XTY_FLUSH_STAGES
#define XTY_FLUSH_STAGES 1     // covariance (K1): stages per fp32 MFMA chain before the float64 flush.  Development knob, measured in round 3
                               // (tools/seed_sweep.py, worst dx over three seeds at 128x32x32x256, cond 1e6): 1 stage = 12 MFMA accumulations
                               // 1.39e-4; 2 stages 1.72e-4; 4 stages 2.19e-4; 16 stages 1.03e-3 -- and SHORTER chains (a flush every 2 k-steps /
                               // every k-step, built and dropped) 1.92e-4 / 3.23e-4: one stage per chain is the optimum of this scheme
#endif
#ifndef XTY_YPL_ABL
#define XTY_YPL_ABL 0   // development, TIMING ONLY (wrong results): K4 on planes stages the gradient operand from planes as well (it reads x's planes again) --
#endif                  // what K4 would cost if the gradient arrived pre-masked and pre-split (ranked first for round 6)
#ifndef XTY_LOLO
#define XTY_LOLO 0      // 1: covariance (K1): the fourth product lo*lo on every block (round 5, VERDICT r4 item 7: the lever named for the non-gaussian families)
#endif
#ifndef XTY_SUBFLUSH
#define XTY_SUBFLUSH 1  // covariance (K1): float64 flushes per stage -- 2 = a flush every KS/2 k-steps (chains of 6 accumulations instead of 12)
#endif
template <int C, bool TWO> constexpr bool xty_quad() { return TWO && C == 256; }

// The off-diagonal sums of the covariance kernel come out of the matrix pipe LOW by a nearly constant 8-10 e-10 of sqrt(S_ii S_jj)
// (S = the centred sums of squares): v_mfma_f32_*_f16's accumulation is not correctly rounded, and with the
// diagonal taken from exact VALU sums the off-diagonal bias is a rank-one perturbation -kappa s s^T that the whitening amplifies
// by cond(Sigma) -- 1.39e-4 in dx on one seed of five at 128x32x32x256, 3e-5 once the mean bias is taken out (tools/k1_err_structure.py).
// Measured with tools/k1_bias_survey.py (mean over i != j of (Sigma_gpu - Sigma_f64)_ij / sqrt(Sigma_ii Sigma_jj)):
//   cond-1e6 inputs, M = 32768 ... 524288, C = 128 | 256, channel scales over 4 decades   -9.7e-10 ... -1.00e-9
//   independent gaussian -8.3e-10 (C = 128: -6.2e-10), + mean 3: -8.2e-10, heavy tails -9.4e-10, rank-4 + noise -1.12e-9,
//   uniform(-1, 1) -5.9e-10, C = 64 gaussian -5.8e-10
// The compensation adds kappa sqrt(S_ii S_jj) back to every off-diagonal sum (stats_xtx_kernel): with kappa in the middle of that
// range the residual bias is <= 1/3 of the uncompensated one for every input above.  It belongs to THIS kernel's accumulation
// scheme (three products into one fp32 accumulator, 12 accumulations per chain, XTY_FLUSH_STAGES 1): re-measure when that changes.
constexpr double kXtyOffdiagBias = 8.5e-10;


__device__ __forceinline__ f32x4 ldg4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

struct FastXtyArgs {
    const float* X; const float* Y;        // Y == X (and TWO == false) for the covariance
    const float* cx; const float* cy;      // centers [C] or nullptr
    const float* sx; const float* sy;      // power-of-two scales [C]
    int64_t N, HW;
    int per_sample, nsplit;
    int64_t rows_per_slab;
    int nslab, ntypes;
    double* P;                             // [nslab][C][C]
    float* colsum;                         // [nslab][C]: sum of (X-cx) (covariance) or of (Y-cy) (two-operand)
    double* dfix;                          // [nslab][C], covariance only: the diagonal, sum_m g_i[m]^2 on the VALU (see stage_write)
    int* flag;
    const float* Yrelu; float* Yout;       // RELU form (two operands, quadrant scheme): Y := Y where Yrelu > 0 else 0, written to Yout
    const unsigned* Ymask;                 // RELU == 2: the same mask as ONE BIT per element, [M/32][C] words (wc_apply_mask_f32)
    const _Float16* Xhi; const _Float16* Xlo;      // XPL: X as pre-split planes (wc_resadd.hip): sx = their scales, cx = NULL, X itself unused
};

// RELU (K4 behind a site whose ReLU rode in K3's epilogue): the gradient mask gy := gy where y > 0
// is applied to the Y operand as it is staged -- the Y threads load the matching rows of y beside those of gy (32 more
// registers: the quadrant form has them) and the types of quadrant row 0 write the masked rows out for K6 -- instead of a
// separate elementwise pass over three tensors in front of K4.
// RELU: 0 none; 1 the site's output y in fp32 (8 x 16 bytes per Y thread and stage); 2 the bit mask K3 left (ONE 16-byte load of
// the thread's four columns' words per stage: the stage's 64 rows are two 32-row mask blocks, the thread's 8 rows one byte of a
// word -- K4 then reads x, gy and 1/32 of a tensor instead of three tensors)
// XPL (round 4, quadrant form): the X operand (the site's input x) arrives as pre-split planes -- the X threads load 8 bytes of each
// plane per row instead of 16 bytes of fp32, and their whole conversion (centre, scale, range guard, two packs and two mixed FMAs per
// element pair) becomes ONE v_perm_b32 per image word: the 8 rows x 4 channels a thread holds are transposed into 4 channels x 8
// rows by picking the matching half of two rows' words.  The planes hold g = (x - center) scale, the kernel reduces g / scale, and
// the caller adds the rank-one term (center - mu) (sum gy)^T (wc_launch_rank1_add).
template <int C, bool TWO, int RELU = 0, bool XPL = false>
__global__ __launch_bounds__(512, 1) void xty_f16x3_kernel(FastXtyArgs a)
{
    static_assert(!RELU || xty_quad<C, TWO>(), "the masked form exists for the quadrant scheme only");
    static_assert(!XPL || (TWO && (C == 256 || C == 128)), "X from planes: two operands, C = 256 (quadrant scheme) or 128 (round 5: the plain scheme -- same staging: 8 rows x 4 channels per thread, one v_perm per image word)");
    // QUAD (two operands at C = 256): the 8 x 8 blocks are cut into four 4 x 4 quadrants, one workgroup type each.  A
    // quadrant needs only 128 channels of X and 128 of Y, so a workgroup converts HALF of every row (the three types of
    // the plain scheme convert all of it three times), its images hold twice the rows (64 per stage: half the barriers)
    // and a wave owns 2 blocks instead of 3 (no idle block slots: 64 = 4 x 8 x 2).
    constexpr bool QUAD = xty_quad<C, TWO>();
    constexpr bool BAL = XTY_BAL && !TWO && C == 256;     // (two workgroup types, 36 blocks)
    constexpr bool BAL2 = XTY_BAL && TWO && C == 128;
    constexpr int BW = QUAD ? 2 : BW_PLAIN;           // 32x32 blocks per wave
    constexpr int CS = QUAD ? C / 2 : C;              // channels of an operand that this workgroup stages
    constexpr int C4 = CS / 4;
    constexpr int RGRP = 512 / C4;                    // 8-row groups covered by the 512 threads
    constexpr int R = TWO ? RGRP * 4 : RGRP * 8;      // rows per stage and operand (TWO: half the threads per operand)
    constexpr int CPR = R / 8;                        // 16-byte chunks per channel row of an image
    constexpr int KS = R / 16;                        // MFMA k-steps per stage
    constexpr int IMG = CS * R * 2;                   // bytes of one fp16 image
    constexpr int NOP = TWO ? 2 : 1;
    constexpr int NB = C / 32;
    constexpr int NBLK = TWO ? NB * NB : NB * (NB + 1) / 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];      // [2 buffers][operand][hi | lo]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;

    // workgroup -> (slab, type): the ntypes workgroups of a slab sit 8 apart in block order (same XCD, speed only)
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
    const int nst = (int)((r1 - r0) / R);             // whole stages only (the launcher guarantees it)

    // this wave's blocks
    // ib / jb: block coordinates in the C/32 x C/32 grid (outputs, scales); il / jl: channel-block rows of the staged images
    int ib[BW], jb[BW], il[BW], jl[BW]; bool live[BW];
    const int qi = QUAD ? (type >> 1) : 0, qj = QUAD ? (type & 1) : 0;       // quadrant of this workgroup type
#pragma unroll
    for (int b = 0; b < BW; ++b) {
        if (QUAD) {
            const int L = wave * BW + b;                 // 16 blocks of the quadrant on 8 waves x 2
            live[b] = true;
            il[b] = L >> 2; jl[b] = L & 3;
            ib[b] = qi * (NB / 2) + il[b]; jb[b] = qj * (NB / 2) + jl[b];
        } else {
            int L = (type * 8 + wave) * BW + b;
            live[b] = L < NBLK;
            if (BAL) {
                // the covariance at C = 256: 36 blocks as 18 + 18 instead of 24 + 12 -- a slab is as slow as its heavier
                // workgroup.  Waves 0 and 1 (on different SIMDs) take three blocks, the other six take two: at most five
                // blocks per SIMD instead of six, and every wave's blocks are a prefix (NL = 3 or 2: two branch-free loops)
                L = type * (NBLK / 2) + (wave < 2 ? wave * 3 : 6 + (wave - 2) * 2) + b;
                live[b] = b < (wave < 2 ? 3 : 2);
            }
            if (BAL2) {       // two operands at C = 128: 16 blocks as 2 per wave instead of 3, 3, 3, 3, 3, 1, 0, 0 (one workgroup type)
                L = wave * 2 + b;
                live[b] = b < 2;
            }
            if (!live[b]) L = 0;
            if (TWO) { ib[b] = L / NB; jb[b] = L % NB; }
            else { int i = 0; while (L >= NB - i) { L -= NB - i; ++i; } ib[b] = i; jb[b] = i + L; }
            il[b] = ib[b]; jl[b] = jb[b];
        }
    }

    bool all_ = true, any_ = false;
#pragma unroll
    for (int b = 0; b < BW; ++b) { all_ = all_ && live[b]; any_ = any_ || live[b]; }
    const bool all_live = XTY_ALLLIVE && (QUAD || __builtin_amdgcn_readfirstlane(all_ ? 1 : 0) != 0);
    const bool two_live = (BAL || BAL2) && __builtin_amdgcn_readfirstlane((live[0] && live[1] && !live[BW - 1]) ? 1 : 0) != 0;   // blocks 0, 1 only      // wave-uniform by construction (wave index, type)
    const bool any_live = QUAD || __builtin_amdgcn_readfirstlane(any_ ? 1 : 0) != 0;

    // staging: thread -> operand op, float4 column c4, 8-row group rgrp
    const int op = TWO ? (tid >= 256) : 0;
    const int tl = TWO ? (tid & 255) : tid;
    const int c4 = tl % C4, rgrp = tl / C4;
    const float* src = (TWO && op) ? a.Y : a.X;
    const float* cen = (TWO && op) ? a.cy : a.cx;
    const float* scp = (TWO && op) ? a.sy : a.sx;
    const int cbase = QUAD ? (op ? qj : qi) * CS : 0;     // first channel of the operand that this workgroup stages
    const f32x4 scl = ldg4(scp + cbase + 4 * c4);
    f32x4 ncs = {0.f, 0.f, 0.f, 0.f};
    if (cen) ncs = -ldg4(cen + cbase + 4 * c4) * scl;

    // The chunk swizzle has to serve TWO access patterns (MI355X_MICROARCH.md, LDS table): the fragment reads -- ds_read_b128,
    // banks mod 64, lane groups {0-3, 12-15, 20-27} / {4-11, 16-19, 28-31} of 16 consecutive channels, i.e. channel bits
    // c2 ^ c3 ^ c4 constant, c0 and c1 free -- and the transposing stage write -- ds_write_b128, banks mod 32, groups of 8
    // CONTIGUOUS lanes = channels 4 L + j, i.e. c2 c3 c4 free, c0 c1 constant.  Rounds 1-3 swizzled with (c >> 1) & 7 (rows of
    // 8 chunks): conflict-free reads, but the 8 lanes of a write group met in 4 of a row's 8 slots (profiles/r3_k1_xty_pmc.json:
    // SQ_LDS_BANK_CONFLICT 23.5 % of the LDS cycles in K1, 28.5 % in K4).  A GF(2)-linear swizzle whose kernel is a vector with
    // c1 (resp. c0) set AND odd weight on c2 c3 c4 is injective on both kinds of group:
    //   8 chunks per row  (C = 256: K1's plain form, K4's quadrant form):  (c1 ^ c2, c3, c4)            kernel (c1 c2) = 11
    //   16 chunks per row (C = 128 covariance; bit 3 picks the row's 128-byte half: constant per write)  (c0 ^ c2, c3, c4, c1)
    auto swz = [](int c) -> int {
        if (CPR == 8) return (((c >> 1) ^ (c >> 2)) & 1) | (((c >> 3) & 1) << 1) | (((c >> 4) & 1) << 2);
        if (CPR == 16) return ((c ^ (c >> 2)) & 1) | (((c >> 3) & 1) << 1) | (((c >> 4) & 1) << 2) | (((c >> 1) & 1) << 3);
        if (CPR >= 16) return c & 15;
        return (c / (16 / CPR)) & (CPR - 1);
    };
    int st_off[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = 4 * c4 + j;
        st_off[j] = op * 2 * IMG + c * (R * 2) + ((rgrp ^ swz(c)) * 16);
    }

    f32x4 xr0[8];           // (the rows of one stage on their way; ROLE31 keeps a second set: two stages in flight)
    f32x4 yr[RELU == 1 ? 8 : 1];
    uint4 ym0 = {0u, 0u, 0u, 0u};
    const bool y_wave = RELU && __builtin_amdgcn_readfirstlane(op) != 0;       // waves 4-7 stage Y (wave-uniform: a scalar branch)
    const bool x_wave = XPL && (XTY_YPL_ABL || __builtin_amdgcn_readfirstlane(op) == 0);        // waves 0-3 stage X (wave-uniform: a scalar branch)
    // (RL: 0 = the wave's role is a run-time value, as in every form but ROLE31; 1 = an X wave, 2 = a Y wave at compile time -- each role's loop
    //  then holds only its own staging code and registers)
    auto stage_load = [&](int st, auto RL_, f32x4 (&xr)[8], uint4& ym) __attribute__((always_inline)) {
        constexpr int RL = decltype(RL_)::value;
        const int64_t off = (r0 + (int64_t)st * R + rgrp * 8) * C + cbase + 4 * c4;
        if (XPL && (RL == 1 || (RL == 0 && x_wave))) {        // 4 channels of 8 rows from each plane: xr[p] = (hi word 0, hi word 1, lo word 0, lo word 1) of row p
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                const uint2 h = *reinterpret_cast<const uint2*>(a.Xhi + off + p * C), l = *reinterpret_cast<const uint2*>(a.Xlo + off + p * C);
                xr[p] = __builtin_bit_cast(f32x4, make_uint4(h.x, h.y, l.x, l.y));
            }
            return;
        }
        const float* base = src + off;
#pragma unroll
        for (int p = 0; p < 8; ++p) xr[p] = ldg4(base + p * C);
        if (RELU == 1 && (RL == 2 || (RL == 0 && y_wave))) {
#pragma unroll
            for (int p = 0; p < 8; ++p) yr[RELU == 1 ? p : 0] = ldg4(a.Yrelu + off + p * C);
        }
        if (RELU == 2 && (RL == 2 || (RL == 0 && y_wave))) {      // rows row0 .. row0 + 7 (row0 a multiple of 8): byte (row0 % 32) / 8 of the 32-row block's words
            const int64_t row0 = r0 + (int64_t)st * R + rgrp * 8;
            ym = *reinterpret_cast<const uint4*>(a.Ymask + (row0 >> 5) * C + cbase + 4 * c4);
            const int sh = (int)(row0 & 31);
            ym.x >>= sh; ym.y >>= sh; ym.z >>= sh; ym.w >>= sh;
        }
    };
    // The staging is what bounds this kernel: 16 (K1) to 22 (K4) vector instructions per MFMA before this trim (rocprofv3
    // SQ_INSTS_VALU / SQ_INSTS_MFMA, profiles/r2_k1_xty_pmc.json), every row converted by each workgroup of its slab.  So:
    // the range guard is a running max (v_max3_f32 with |.| modifiers, one instruction per two elements), the split
    // remainder is one v_fma_mix_f32 per element (it reads the fp16 half directly), and the column sums / squares are
    // taken only by the workgroup that reports them.
    float gmax = 0.f;
    const bool want_csum = a.colsum != nullptr && (QUAD ? qi == 0 : type == 0) && (!TWO || op == 1);     // QUAD: each half of Y's columns once
    f32x4 csum = {0.f, 0.f, 0.f, 0.f};
    // The DIAGONAL of the covariance does not come from the matrix pipe.  Two systematic errors meet there, both
    // measured on MI355X (tools/probe/mfma_gram_probe.hip, tools/k1_bias.py):
    //  * the product the three MFMAs drop, lo*lo, is zero-mean everywhere except on the diagonal, where it is
    //    sum_m lo_i[m]^2 > 0: -(2..8)e-8 of every variance;
    //  * v_mfma_f32_32x32x16_f16 is NOT a correctly rounded dot product (a third of its outputs differ from
    //    RNE_fp32(C + exact sum)); its error is zero-mean for products of mixed sign, but adding 16 POSITIVE products
    //    to a positive accumulator comes out 0.04 ulp low per instruction: -8e-9 of every variance at this chain length.
    // Either is harmless for the covariance itself, but at cond(Sigma~) ~ 1e6 the whitening amplifies a uniform
    // relative error of the variances by ~3e3: y was off by 0.8-2e-4 and dx by up to 3e-4 at the full-size sites.
    // The scaled values are at hand here, so the lightest workgroup of the slab sums their squares per channel with
    // IEEE fp32 FMAs (round to nearest even: unbiased) and the combine takes the diagonal from there.
    const bool want_dfix = !TWO && a.dfix != nullptr && type == a.ntypes - 1;
    double lsq[4] = {0.0, 0.0, 0.0, 0.0};      // per stage: a fresh 8-term fp32 chain, folded into float64
    auto stage_write = [&](int buf, int st_of_data, auto RL_, f32x4 (&xr)[8], uint4& ym) __attribute__((always_inline)) {
        constexpr int RL = decltype(RL_)::value;
        char* img = smem + buf * (NOP * 2 * IMG);
        if (XPL && (RL == 1 || (RL == 0 && x_wave))) {        // transpose by byte permutes: channel j of rows (2 pp, 2 pp + 1) = half j & 1 of word j >> 1 of the two rows
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned sel = (j & 1) ? 0x07060302u : 0x05040100u;
                unsigned hw[4], lw[4];
#pragma unroll
                for (int pp = 0; pp < 4; ++pp) {
                    const uint4 e = __builtin_bit_cast(uint4, xr[2 * pp]), o = __builtin_bit_cast(uint4, xr[2 * pp + 1]);
                    const unsigned he = (j >> 1) ? e.y : e.x, ho = (j >> 1) ? o.y : o.x, le = (j >> 1) ? e.w : e.z, lo_ = (j >> 1) ? o.w : o.z;
                    hw[pp] = __builtin_amdgcn_perm(ho, he, sel);
                    lw[pp] = __builtin_amdgcn_perm(lo_, le, sel);
                }
                *reinterpret_cast<uint4*>(img + st_off[j]) = make_uint4(hw[0], hw[1], hw[2], hw[3]);
                *reinterpret_cast<uint4*>(img + st_off[j] + IMG) = make_uint4(lw[0], lw[1], lw[2], lw[3]);
            }
            return;
        }
        f32x4 g[8];
        if (RELU && (RL == 2 || (RL == 0 && y_wave))) {
            if (RELU == 2) {
                const unsigned mw[4] = {ym.x, ym.y, ym.z, ym.w};
#pragma unroll
                for (int p = 0; p < 8; ++p)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {       // v_bfe_i32 sign-extends the row's bit to 0 / ~0: two vector instructions per element, as the compare form
                        const unsigned keep = (unsigned)__builtin_amdgcn_sbfe((int)mw[j], p, 1);
                        const float e = xr[p][j];        // (a scalar copy: __builtin_bit_cast applied to the vector ELEMENT took element 0 for every j -- hipcc 7.0)
                        xr[p][j] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, e) & keep);
                    }
            } else {
#pragma unroll
            for (int p = 0; p < 8; ++p)
#pragma unroll
                for (int j = 0; j < 4; ++j) xr[p][j] = !(yr[RELU == 1 ? p : 0][j] <= 0.f) ? xr[p][j] : 0.f;      // (NaN in y: the gradient passes, as in relu_mask_kernel and aten::threshold_backward)
            }
            if (qi == 0 && a.Yout) {        // each half of Y's columns is written by one type (nullable: a K6 that masks for itself)
                float* dst = a.Yout + (r0 + (int64_t)st_of_data * R + rgrp * 8) * C + cbase + 4 * c4;
#pragma unroll
                for (int p = 0; p < 8; ++p) *reinterpret_cast<f32x4*>(dst + p * C) = xr[p];
            }
        }
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            g[p] = xr[p] * scl + ncs;
            gmax = __builtin_fmaxf(__builtin_fmaxf(gmax, fabsf(g[p][0])), fabsf(g[p][1]));
            gmax = __builtin_fmaxf(__builtin_fmaxf(gmax, fabsf(g[p][2])), fabsf(g[p][3]));
        }
        if (want_csum) {
#pragma unroll
            for (int p = 0; p < 8; ++p) csum += g[p];
        }
        if (want_dfix) {
            f32x4 sq = g[0] * g[0];
#pragma unroll
            for (int p = 1; p < 8; ++p) sq += g[p] * g[p];
#pragma unroll
            for (int j = 0; j < 4; ++j) lsq[j] += (double)sq[j];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            unsigned hw[4], lw[4];
#pragma unroll
            for (int pp = 0; pp < 4; ++pp) {
                const float v0 = g[2 * pp][j], v1 = g[2 * pp + 1][j];
                hw[pp] = pk_rne(v0, v1);
                float r0_, r1_;         // remainder = v - float(hi) in one mixed-precision FMA per element
                asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r0_) : "v"(hw[pp]), "v"(v0));
                asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1_) : "v"(hw[pp]), "v"(v1));
                lw[pp] = pk_rne(r0_, r1_);
            }
            *reinterpret_cast<uint4*>(img + st_off[j]) = make_uint4(hw[0], hw[1], hw[2], hw[3]);
            *reinterpret_cast<uint4*>(img + st_off[j] + IMG) = make_uint4(lw[0], lw[1], lw[2], lw[3]);
        }
    };

    double acc64[BW][16];
#pragma unroll
    for (int b = 0; b < BW; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc64[b][r] = 0.0;

    // per-lane fragment addressing: channel row (block*32 + l31), chunk (2*ks + lh) ^ swz(channel).  With 2 ks + lh = 2 ks ^ lh
    // and a row pitch that is a multiple of the row's 16 CPR bytes, the byte offset is  base ^ (ks << 5)  with
    // base = row * pitch + ((lh ^ swz) << 4) kept per block: ONE v_xor per fragment address in the stage loop instead of the
    // xor / shift / add chain (about 70 of a stage's ~370 vector instructions per wave went into these addresses), and the two
    // buffers (64 KiB apart) ride in the same xor.
    static_assert(NOP * 2 * IMG == 65536, "the stage buffers are 64 KiB apart: their bit is xor-ed into the fragment offsets");
    int a_base[BW], b_base[BW];
#pragma unroll
    for (int b = 0; b < BW; ++b) {
        const int ca = il[b] * 32 + l31, cb = jl[b] * 32 + l31;
        a_base[b] = ca * (R * 2) + ((lh ^ swz(ca)) << 4);
        b_base[b] = (TWO ? 2 * IMG : 0) + cb * (R * 2) + ((lh ^ swz(cb)) << 4);
    }

    using RL0 = std::integral_constant<int, 0>;
    constexpr bool ROLE31 = XTY_ROLE31 && QUAD && (XPL || RELU == 2) && !XTY_YPL_ABL && !XTY_STAMPS;      // (tried on the C = 128 planes form -- one 4 x 4 block grid, a quadrant's geometry: 47.1 -> 48.8 / 66.2 -> 68.7 us per stage, not taken)      // (fp32 x with the bit mask: both roles convert -- two blocks
                                                                                                           //  and two sets each: 103 -> 98 us; without a mask the old loop is as fast: 81 against 83)
    if (!ROLE31 && nst > 0) {
        stage_load(0, RL0{}, xr0, ym0);
        stage_write(0, 0, RL0{}, xr0, ym0);
        if (nst > 1) stage_load(1, RL0{}, xr0, ym0);
    }
    if (!ROLE31) __syncthreads();
    const bool stamp_ok = XTY_STAMPS && tid == 0 && blockIdx.x == 0;
    unsigned long long* stamps = reinterpret_cast<unsigned long long*>(a.P + (int64_t)a.nslab * C * C);     // just past P: stamp builds get a larger workspace
    int nstamp = 0;
    (void)stamps; (void)nstamp; (void)stamp_ok;
#define XS() do { if (XTY_STAMPS && stamp_ok && nstamp < 250) stamps[nstamp++] = __builtin_amdgcn_s_memtime(); } while (0)
    // (Tried and dropped, measured: the stage as two half-steps in which waves 0-3 convert while their SIMD partners 4-7 run the
    // MFMAs and vice versa -- both halves slowed down by more than 2x, K1 89 -> 127 us: the ds_write_b128 bursts of the
    // converting waves and the fragment reads of the MFMA waves fight over the LDS.  An L2 prefetch of the stage three steps
    // ahead: 89 -> 100 us.  Reading the A fragments once for a wave's blocks of the same block row (12 -> 9 ds_read_b128 per
    // k-step): no change, +18 spilled registers.)
    f32x16 acc[BW];
    for (int st = 0; !ROLE31 && st < nst; ++st) {
        const int cur = st & 1;
        XS();
        if (st + 1 < nst) stage_write(cur ^ 1, st + 1, RL0{}, xr0, ym0);
        XS();
        if (st + 2 < nst) stage_load(st + 2, RL0{}, xr0, ym0);
        XS();
        // the fp32 accumulators live for one stage only (their first MFMA takes a zero operand: no zeroing pass, and the 16 BW
        // registers are free while the next stage is converted)
        constexpr int FS = TWO ? 1 : XTY_FLUSH_STAGES;
        if (FS == 1 || st % FS == 0) {
#pragma unroll
            for (int b = 0; b < BW; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
        }
        const int kbuf = cur << 16;
        auto frag = [&](int base, int ks, int lo) __attribute__((always_inline)) {
            return *reinterpret_cast<const f16x8*>(smem + (base ^ ((ks << 5) | kbuf)) + lo * IMG);
        };
        // ALL: every block of this wave is live (a scalar, per wave) -- the loop is then ONE basic block; with a `live` test per
        // block and k-step (a per-lane value as far as hipcc can tell: exec-mask branches) every three MFMAs sat in a block of
        // their own (K1 kernel 60 -> 56 us)
        // XTY_ALT (round 4): v_mfma_f32_32x32x16_f16's accumulation is biased -- every chain comes out low by ~1e-9 of sqrt(S_ii S_jj)
        // whatever the sign of its sum (rounds 3-4 added a fitted constant back) -- so every ODD stage runs on the
        // negated A fragments and its chain is SUBTRACTED at the flush: the sums add up as before, the biases of consecutive chains
        // cancel (measured: mean off-diagonal error -9.7e-10 -> +5e-12 of sqrt(S_ii S_jj), the same on uniform / post-ReLU / Laplace
        // inputs).  The f16 MFMAs have no neg modifier: one v_xor per fragment register with a wave-uniform mask (0 in even stages).
        const unsigned sgn = (XTY_ALT && !TWO && FS == 1 && (st & 1)) ? 0x80008000u : 0u;
        constexpr int NSUB = (!TWO && XTY_SUBFLUSH > 1 && KS % XTY_SUBFLUSH == 0) ? XTY_SUBFLUSH : 1;
        int ks_lo = 0, ks_hi = KS;
        auto products = [&](auto ALL_, auto NL_) __attribute__((always_inline)) {
            constexpr bool ALL = decltype(ALL_)::value;
            constexpr int NL = decltype(NL_)::value;          // blocks 0 .. NL-1 (all live when ALL)
#pragma unroll 4
            for (int ks = ks_lo; ks < ks_hi; ++ks) {
#pragma unroll
                for (int b = 0; b < NL; ++b) {
                    if (!ALL && !live[b]) continue;
                    f16x8 ah = frag(a_base[b], ks, 0), al = frag(a_base[b], ks, 1);
                    if (XTY_ALT && !TWO && FS == 1 && sgn) {       // (a scalar branch: even stages skip the eight v_xor)
                        typedef unsigned u32x4v_ __attribute__((ext_vector_type(4)));
                        ah = __builtin_bit_cast(f16x8, __builtin_bit_cast(u32x4v_, ah) ^ sgn);
                        al = __builtin_bit_cast(f16x8, __builtin_bit_cast(u32x4v_, al) ^ sgn);
                    }
                    const f16x8 bh = frag(b_base[b], ks, 0), bl = frag(b_base[b], ks, 1);
                    if (XTY_LOLO && !TWO) acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bl, acc[b], 0, 0, 0);
                    acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[b], 0, 0, 0);
                    acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[b], 0, 0, 0);
                    acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[b], 0, 0, 0);
                }
            }
        };
        // Quadrant form: a wave's two blocks lie in one block row (L = 2 wave + b) and share their A fragments: 6 instead of 8
        // ds_read_b128 per k-step.  (No measurable difference: the stage is the sum of its vector and matrix time, not LDS
        // bound.  Also measured and dropped: reading block-step n + 1 ahead of the MFMAs of block-step n, pinned with
        // sched_barrier: K4 75 -> 77 us.)
        auto products_quad = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const f16x8 ah = frag(a_base[0], ks, 0), al = frag(a_base[0], ks, 1);
#pragma unroll
                for (int b = 0; b < BW; ++b) {
                    const f16x8 bh = frag(b_base[b], ks, 0), bl = frag(b_base[b], ks, 1);
                    acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[b], 0, 0, 0);
                    acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[b], 0, 0, 0);
                    acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[b], 0, 0, 0);
                }
            }
        };
        const double fsg = sgn ? -1.0 : 1.0;             // (wave-uniform: the chain's sign)
#pragma unroll
        for (int sub = 0; sub < NSUB; ++sub) {
        if (NSUB > 1) {
            ks_lo = sub * (KS / NSUB); ks_hi = ks_lo + KS / NSUB;
            if (sub > 0) {
#pragma unroll
                for (int b = 0; b < BW; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
            }
        }
        if (QUAD) products_quad();
        else if (all_live) products(std::true_type{}, std::integral_constant<int, BW>{});
        else if (two_live) products(std::true_type{}, std::integral_constant<int, 2>{});
        else if (any_live) products(std::false_type{}, std::integral_constant<int, BW>{});
        XS();
        // float64 flush: the fp32 rounding chain never exceeds one stage (3*KS MFMA accumulations)
        if (FS > 1 && st % FS != FS - 1 && st + 1 < nst) {}
        else if (two_live) {
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc64[b][r] = __builtin_fma((double)acc[b][r], fsg, acc64[b][r]);
        } else if (any_live) {
#pragma unroll
            for (int b = 0; b < BW; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc64[b][r] = __builtin_fma((double)acc[b][r], fsg, acc64[b][r]);
        }
        }
        XS();
        // LDS hand-off only (__syncthreads() would also drain vmcnt, i.e. wait for the loads of stage st+2 issued a moment ago)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    if (XTY_STAMPS && stamp_ok) { stamps[nstamp++] = __builtin_amdgcn_s_memtime(); stamps[255] = nstamp; }
#undef XS

    // partial blocks out, scales undone exactly (powers of two)
    double* P = a.P + z * (int64_t)C * C;
    if constexpr (ROLE31) {
        // K4 on planes, quadrant form, round 6.  The four waves that stage X have almost nothing to convert (one v_perm_b32 per image word), the four
        // that stage gy mask, scale, range-test and split every element: with two blocks per wave the stage lasted as long as a Y wave's conversion
        // PLUS its two chains, while its SIMD partner -- an X wave -- had long finished (the clock probe and the planes ablation).  So
        // wave w < 4 owns blocks (w, 0), (w, 1), (w, 2) of the quadrant -- one A fragment pair for three chains -- and wave w + 4 owns (w, 3): the
        // matrix work of a SIMD is what it was, the Y wave's critical path is a third shorter.  Each role runs its own copy of the stage loop (the
        // same barriers), so the X role's 144 accumulator registers never meet the Y role's conversion registers.  Block by block the same chains:
        // the partials are bit-identical to the two-blocks-per-wave form.
        auto role = [&](auto NBW_, auto RL_) __attribute__((always_inline)) {
            constexpr int NBW = decltype(NBW_)::value;
            using RLt = decltype(RL_);
            constexpr int jl0 = RLt::value == 1 ? 0 : 4 - NBW;                            // (the Y role's blocks are the last of the row)
            const int ilr = wave & 3;
            double s64[NBW > 0 ? NBW : 1][16];
#pragma unroll
            for (int b = 0; b < NBW; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) s64[b][r] = 0.0;
            const int ca = ilr * 32 + l31;
            const int a_b = ca * (R * 2) + ((lh ^ swz(ca)) << 4);
            int b_b[NBW > 0 ? NBW : 1];
#pragma unroll
            for (int b = 0; b < NBW; ++b) {
                const int cb = (jl0 + b) * 32 + l31;
                b_b[b] = 2 * IMG + cb * (R * 2) + ((lh ^ swz(cb)) << 4);
            }
            // TWO stages of rows in flight (set A = xr0, set B = xr1, alternating): a stage's loads used to have the matrix phase of ONE stage to
            // land (~1 us against a loaded memory latency of two) and every stage began with a wait -- K4 was latency-bound at 2.9 us per stage
            // where its vector + matrix work is 0.8.  Every trip issues the same loads (past the end: the last stage's rows again, an L2 hit nobody
            // reads), so the compiler's vmcnt for "this set has landed" leaves the other set's loads in flight (see resadd_xtx_kernel's history).
            f32x4 xr1[8];
            uint4 ym1 = {0u, 0u, 0u, 0u};
            const int last = nst - 1;
            auto clampst = [&](int st) { return st < last ? st : last; };
            auto stage_body = [&](int st) __attribute__((always_inline)) {
                const int cur = st & 1;
                const int kbuf = cur << 16;
                auto frag = [&](int base, int ks, int lo) __attribute__((always_inline)) {
                    return *reinterpret_cast<const f16x8*>(smem + (base ^ ((ks << 5) | kbuf)) + lo * IMG);
                };
                // (the X role's blocks two at a time, then the third: three chains side by side need 48 accumulator registers that the second set of
                //  rows has taken)
                constexpr int G2 = (NBW == 2 || NBW == 4) ? 2 : (NBW == 3 ? 3 : 1);
                auto chains = [&](auto B0_, auto NB_) __attribute__((always_inline)) {
                    constexpr int b0 = decltype(B0_)::value, nbk = decltype(NB_)::value;
                    f32x16 ac[nbk];
#pragma unroll
                    for (int b = 0; b < nbk; ++b)
#pragma unroll
                        for (int r = 0; r < 16; ++r) ac[b][r] = 0.f;
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        const f16x8 ah = frag(a_b, ks, 0), al = frag(a_b, ks, 1);
#pragma unroll
                        for (int b = 0; b < nbk; ++b) {
                            const f16x8 bh = frag(b_b[b0 + b], ks, 0), bl = frag(b_b[b0 + b], ks, 1);
                            ac[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, ac[b], 0, 0, 0);
                            ac[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, ac[b], 0, 0, 0);
                            ac[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, ac[b], 0, 0, 0);
                        }
                    }
#pragma unroll
                    for (int b = 0; b < nbk; ++b)
#pragma unroll
                        for (int r = 0; r < 16; ++r) s64[b0 + b][r] = __builtin_fma((double)ac[b][r], 1.0, s64[b0 + b][r]);
                };
                if constexpr (NBW > 0) chains(std::integral_constant<int, 0>{}, std::integral_constant<int, G2>{});
                if constexpr (NBW > G2) chains(std::integral_constant<int, G2>{}, std::integral_constant<int, NBW - G2>{});
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            };
            // (the X role with three blocks has no room for a second set: its plane rows keep one stage in flight, the Y role's fp32 rows two)
            constexpr bool TWOSETS = XTY_ROLE31 == 2 || RLt::value == 2 || !XPL;
            if constexpr (TWOSETS) {
                if (nst > 0) {
                    stage_load(0, RLt{}, xr0, ym0);
                    stage_load(clampst(1), RLt{}, xr1, ym1);
                    stage_write(0, 0, RLt{}, xr0, ym0);
                    stage_load(clampst(2), RLt{}, xr0, ym0);
                }
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                for (int st = 0; st < nst; st += 2) {
                    // even stage: set B holds stage st + 1, set A stage st + 2
                    if (st + 1 < nst) stage_write((st & 1) ^ 1, st + 1, RLt{}, xr1, ym1);
                    stage_load(clampst(st + 3), RLt{}, xr1, ym1);
                    stage_body(st);
                    if (st + 1 >= nst) break;
                    // odd stage: set A holds stage st + 2, set B stage st + 3
                    if (st + 2 < nst) stage_write(st & 1, st + 2, RLt{}, xr0, ym0);
                    stage_load(clampst(st + 4), RLt{}, xr0, ym0);
                    stage_body(st + 1);
                }
            } else {
                if (nst > 0) {
                    stage_load(0, RLt{}, xr0, ym0);
                    stage_write(0, 0, RLt{}, xr0, ym0);
                    if (nst > 1) stage_load(1, RLt{}, xr0, ym0);
                }
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                for (int st = 0; st < nst; ++st) {
                    if (st + 1 < nst) stage_write((st & 1) ^ 1, st + 1, RLt{}, xr0, ym0);
                    if (st + 2 < nst) stage_load(st + 2, RLt{}, xr0, ym0);      // (one set: the wait for it is a wait for everything anyway)
                    stage_body(st);
                }
            }
#pragma unroll
            for (int b = 0; b < NBW; ++b) {
                const int j = (qj * (NB / 2) + jl0 + b) * 32 + l31;
                const double isj = 1.0 / (double)a.sy[j];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int i = (qi * (NB / 2) + ilr) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    P[(int64_t)i * C + j] = s64[b][r] * isj / (double)a.sx[i];
                }
            }
        };
        constexpr int NBX = (XPL && XTY_ROLE31 == 3) ? 4 : (XPL && XTY_ROLE31 != 2) ? 3 : 2;          // (XTY_ROLE31 == 2 / 3, development: two / four blocks per X wave on planes)
        if (__builtin_amdgcn_readfirstlane(op) == 0) role(std::integral_constant<int, NBX>{}, std::integral_constant<int, 1>{});
        else role(std::integral_constant<int, 4 - NBX>{}, std::integral_constant<int, 2>{});
    }
#pragma unroll
    for (int b = 0; b < BW; ++b) {
        if (ROLE31 || !live[b]) continue;
        const int j = jb[b] * 32 + l31;
        const double isj = 1.0 / (double)(TWO ? a.sy[j] : a.sx[j]);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = ib[b] * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            P[(int64_t)i * C + j] = acc64[b][r] * isj / (double)a.sx[i];
        }
    }
    // column sums (type-0 workgroups): of the single operand, or of Y when there are two
    if (want_csum) {
        // csum holds sums of SCALED values of channels 4*c4.. over this thread's rows; reduce over the row groups in LDS
        float* red = reinterpret_cast<float*>(smem);            // [row groups][CS], free after the last barrier
#pragma unroll
        for (int j = 0; j < 4; ++j) red[rgrp * CS + 4 * c4 + j] = csum[j];
    }
    if (want_dfix) {
        double* red2 = reinterpret_cast<double*>(smem + RGRP * C * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) red2[rgrp * C + 4 * c4 + j] = lsq[j];
    }
    __syncthreads();
    if ((QUAD ? qi == 0 : type == 0) && a.colsum) {
        constexpr int RG_USED = TWO ? RGRP / 2 : RGRP;
        const float* red = reinterpret_cast<const float*>(smem);
        const int c0 = QUAD ? qj * CS : 0;
        for (int c = tid; c < CS; c += 512) {
            float s = 0.f;
            for (int g = 0; g < RG_USED; ++g) s += red[g * CS + c];
            a.colsum[z * C + c0 + c] = s / (TWO ? a.sy[c0 + c] : a.sx[c0 + c]);
        }
    }
    if (want_dfix) {
        const double* red2 = reinterpret_cast<const double*>(smem + RGRP * C * 4);
        for (int c = tid; c < C; c += 512) {
            double s = 0.0;
            for (int g = 0; g < RGRP; ++g) s += red2[g * C + c];
            a.dfix[z * C + c] = s / ((double)a.sx[c] * (double)a.sx[c]);
        }
    }
    if (!(gmax <= kGuard)) atomicOr(a.flag, 1);
}

template <int C, bool TWO>
constexpr int stage_rows() { return xty_quad<C, TWO>() ? 64 : (TWO ? (512 / (C / 4)) * 4 : (512 / (C / 4)) * 8); }

template <int C, bool TWO, int RELU = 0, bool XPL = false>
hipError_t launch_xty_fast(const FastXtyArgs& a, hipStream_t st)
{
    constexpr int R = stage_rows<C, TWO>();
    constexpr size_t lds = (size_t)2 * (TWO ? 2 : 1) * 2 * (xty_quad<C, TWO>() ? C / 2 : C) * R * 2;       // 128 KiB
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(xty_f16x3_kernel<C, TWO, RELU, XPL>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const int slab_groups = (a.nslab + 7) / 8;
    const int grid = slab_groups * a.ntypes * 8;
    hipLaunchKernelGGL((xty_f16x3_kernel<C, TWO, RELU, XPL>), dim3(grid), dim3(512), lds, st, a);
    return hipGetLastError();
}

}  // namespace

static int xty_stage_rows(int C, bool two)
{
    if (two && C == 256) return 64;           // quadrant scheme (xty_quad)
    const int rg = 512 / (C / 4);
    return two ? rg * 4 : rg * 8;
}

int64_t wc_fast_xty_min_rows()
{
    static const int64_t v = getenv("WC_XTY_MIN_ROWS") ? atoll(getenv("WC_XTY_MIN_ROWS")) : WC_FAST_MIN_ROWS;
    return v;
}

// Plan: slabs of whole stages; total workgroups ~ one per CU.  Returns nslab (0 = shape not eligible).
int wc_fast_xty_plan(int64_t N, int64_t HW, int C, int per_sample, int two, int* nsplit, int64_t* rows_per_slab, int* ntypes)
{
    if (!(C == 32 || C == 64 || C == 128 || C == 256)) return 0;
    const int64_t M = N * HW;
    // K1 (the covariance: the whole error budget of dx at cond 1e6) stays on the exact kernel up to WC_FAST_MIN_ROWS; K4's R
    // (6e-7 of that budget) takes the fast kernel from 16384 rows on as before
    if (M < (two ? (wc_fast_xty_min_rows() < 16384 ? wc_fast_xty_min_rows() : 16384) : wc_fast_xty_min_rows())) return 0;
    const int R = xty_stage_rows(C, two != 0);
    const int64_t seg = per_sample ? HW : M;                  // rows of one segment (slabs never cross segments)
    if (seg % R != 0) return 0;
    const int nb = C / 32;
    const int nblk = two ? nb * nb : nb * (nb + 1) / 2;
    *ntypes = (two && C == 256) ? 4 : (nblk + 8 * BW_PLAIN - 1) / (8 * BW_PLAIN);
    const int64_t nseg = per_sample ? N : 1;
    // 128 KiB of LDS = one workgroup per CU: keep the grid (slab groups of 8 x ntypes) within the 256 CUs when the
    // segment count allows, or the surplus workgroups would run as a second, mostly idle round
    const int64_t target = (256 / (8 * *ntypes)) * 8;
    // rounded DOWN: five statistic groups at 26 slabs each made 130 slabs = 272 workgroups, i.e. a second round of 16 on a
    // chip of 256 CUs and nearly twice the time (measured 280 us at 320x32x32x256 where 2.5 x the headline's 61 is 152)
    int64_t per_seg = target / nseg;
    if (per_seg < 1) per_seg = 1;
    const int64_t stages = seg / R;
    if (per_seg > stages) per_seg = stages;
    int64_t st_per_slab = (stages + per_seg - 1) / per_seg;
    *rows_per_slab = st_per_slab * R;
    *nsplit = (int)((seg + *rows_per_slab - 1) / *rows_per_slab);
    return (int)(nseg * (*nsplit));
}

hipError_t wc_launch_fast_xty(const float* X, const float* Y, const float* cx, const float* cy,
                              const float* sx, const float* sy, int64_t N, int64_t HW, int C,
                              int per_sample, int nsplit, int64_t rows_per_slab, int nslab, int ntypes,
                              double* P, float* colsum, double* dfix, int* gate, hipStream_t st,
                              const float* yrelu, float* yout, const unsigned* ymask, const void* xs)
{
    FastXtyArgs a = {};
    a.Yrelu = yrelu; a.Yout = yout; a.Ymask = ymask;
    if (xs) {       // X as pre-split planes: two operands, C = 256 (quadrant form, with or without the bit mask) or C = 128 (plain form, no mask); sx = the planes' scales, cx = NULL
        if ((C != 256 && C != 128) || yrelu || cx || (C == 128 && ymask)) return hipErrorInvalidValue;
        a.Xhi = static_cast<const _Float16*>(xs); a.Xlo = a.Xhi + N * HW * C;      // (N * HW = all rows in either slab layout)
    }
    a.dfix = (Y == X) ? dfix : nullptr;
    a.X = X; a.Y = Y; a.cx = cx; a.cy = cy; a.sx = sx; a.sy = sy; a.N = N; a.HW = HW;
    a.per_sample = per_sample; a.nsplit = nsplit; a.rows_per_slab = rows_per_slab; a.nslab = nslab; a.ntypes = ntypes;
    a.P = P; a.colsum = colsum; a.flag = gate;
    const bool two = (Y != X) || xs != nullptr;
    switch (C) {
        case 32: return two ? launch_xty_fast<32, true>(a, st) : launch_xty_fast<32, false>(a, st);
        case 64: return two ? launch_xty_fast<64, true>(a, st) : launch_xty_fast<64, false>(a, st);
        case 128:
            if (xs) return launch_xty_fast<128, true, 0, true>(a, st);
            return two ? launch_xty_fast<128, true>(a, st) : launch_xty_fast<128, false>(a, st);
        case 256:
            if (xs) return ymask ? launch_xty_fast<256, true, 2, true>(a, st) : launch_xty_fast<256, true, 0, true>(a, st);
            return two ? (ymask ? launch_xty_fast<256, true, 2>(a, st) : yrelu ? launch_xty_fast<256, true, 1>(a, st) : launch_xty_fast<256, true>(a, st))
                       : launch_xty_fast<256, false>(a, st);
    }
    return hipErrorInvalidValue;
}

double wc_fast_xty_offdiag_bias(void)
{
    static const char* off = getenv("WC_K1_NO_BIAS_COMP");       // development: the uncompensated sums (tools/k1_bias_survey.py)
    return off ? 0.0 : kXtyOffdiagBias;
}
