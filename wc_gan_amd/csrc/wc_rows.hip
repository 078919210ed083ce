// Big-tensor kernels of the WC path, exact-fp32 generic versions on the f32-input MFMA
// (v_mfma_f32_32x32x2_f32: bit-for-bit an fmaf chain, cdna_hip_programming.md section 3).
//
//   rows_gemm_kernel   K3 (wc_apply_f32) and K6 (wc_bwd_apply_f32):
//                      out[m,:] = sum_s (in_s[m,:] - center_s) B_s[slot(m)] + bias[slot(m)] - sub
//                      replaces  W f -> transpose -> Conv2D 1x1 (+ConditionalConv11/FactorizedConv11 + Add)
//                      of generator.py:83-87 and their TF gradients.
//   xty_kernel         K1 (wc_stats_f32) and K4 (wc_bwd_reduce_f32):
//                      P[z] = sum_{m in slab z} (X[m]-cx)^T (Y[m]-cy), fp32 per-slab partials that the
//                      small stage combines in float64; replaces transpose + f f^T GEMM of
//                      DecorelationNormalization.call (generator.py:24) and its gradient reduction.
//
// MFMA 32x32x2 f32 operand maps: a = A[i = lane&31][k = lane>>5], b = B[k = lane>>5][j = lane&31],
// D register r of lane l = D[(r&3) + 8*(r>>2) + 4*(l>>5)][l&31].
#include "wc_common.h"

namespace {

constexpr int BM = 128;      // tile rows (rows of x for rows_gemm; channel i for xty)
constexpr int BN = 128;      // tile cols
constexpr int BK = 32;       // depth per LDS chunk
constexpr int AS_LD = BM + 1;   // transposed x tile [k][row]; +1 makes the 4-way-k scatter write conflict-free
constexpr int BS_LD = BN;
constexpr int64_t WC_EXACT_ROWS = 20479;   // M at or below this: exact float64-MFMA reductions (round 3: was 16384; 128x12x12 = 18432 rows, the STL-10 generator-update site, read dx 7.9e-5 on the split-fp16 kernel and has the least averaging of the sites above the old threshold)

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

__device__ __forceinline__ void mfma_step(const float* __restrict__ As_, int a_ld, const float* __restrict__ Bs_, int b_ld,
                                          int kk, int lane, int arow, int bcol, bool u0, bool u1, f32x16 (&acc)[2][2])
{
    const int kq = kk + (lane >> 5);
    const int l31 = lane & 31;
    const float a0 = As_[kq * a_ld + arow + l31];
    const float a1 = As_[kq * a_ld + arow + 32 + l31];
    const float b0 = Bs_[kq * b_ld + bcol + l31];
    const float b1 = Bs_[kq * b_ld + bcol + 32 + l31];
    if (u0) {
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
    }
    if (u1) {
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
}

// ---------------------------------------------------------------------------------------------
// rows_gemm: 128 x 128 output tile per 256-thread workgroup, 4 waves as 2 x 2, 64 x 64 per wave.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rows_gemm_kernel(WcRowsGemmArgs a, int ncb)
{
    __shared__ float As[BK * AS_LD];
    __shared__ __attribute__((aligned(16))) float Bs[BK * BS_LD];

    if (a.gate && *a.gate == 0) return;      // exact redo of a fast-path call: only when it flagged an overflow
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int C = a.C;
    const int cb = blockIdx.x % ncb;
    const int64_t rb = blockIdx.x / ncb;

    int64_t row0; int rows_valid; int slot = 0;
    if (a.slot) {
        const int64_t bps = (a.HW + BM - 1) / BM;
        const int64_t n = rb / bps, r = rb % bps;
        row0 = n * a.HW + r * BM;
        const int64_t left = a.HW - r * BM;
        rows_valid = left < BM ? (int)left : BM;
        slot = a.slot[n];
    } else {
        const int64_t M = a.N * a.HW;
        row0 = rb * BM;
        const int64_t left = M - row0;
        rows_valid = left < BM ? (int)left : BM;
    }
    const int col0 = cb * BN;

    // staging coordinates (fixed per thread): x tile 128 rows x 32 k as float4 along k; B tile 32 k x 128 cols
    const int xkq = tid & 7;            // float4 index along k
    const int xrow = tid >> 3;          // + 32*p
    const int bnq = tid & 31;           // float4 index along n
    const int bk = tid >> 5;            // + 8*p
    const bool bcol_ok = (col0 + 4 * bnq) < C;

    f32x16 acc[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;

    const bool u0 = (col0 + wc * 64) < C;
    const bool u1 = (col0 + wc * 64 + 32) < C;

    const int cpc = C / BK;                       // chunks per stream
    const int nchunks = a.nstreams * cpc;

    f32x4 xv[4], bv[4], cen;
    auto load_chunk = [&](int c) {
        const int s = c / cpc;
        const int k0 = (c - s * cpc) * BK;
        const float* in = a.in[s];
        const float* Bm = a.B[s] + (int64_t)slot * a.B_slot_stride[s];
        const float* ce = a.center[s];
        if (ce) cen = ld4(ce + k0 + 4 * xkq); else cen = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int row = xrow + 32 * p;
            if (row < rows_valid) xv[p] = ld4(in + (row0 + row) * C + k0 + 4 * xkq) - cen;
            else xv[p] = f32x4{0.f, 0.f, 0.f, 0.f};
            const int k = bk + 8 * p;
            if (bcol_ok) bv[p] = ld4(Bm + (int64_t)(k0 + k) * C + col0 + 4 * bnq);
            else bv[p] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };

    load_chunk(0);
    for (int c = 0; c < nchunks; ++c) {
        __syncthreads();
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int row = xrow + 32 * p;
#pragma unroll
            for (int j = 0; j < 4; ++j) As[(4 * xkq + j) * AS_LD + row] = xv[p][j];
            *reinterpret_cast<f32x4*>(&Bs[(bk + 8 * p) * BS_LD + 4 * bnq]) = bv[p];
        }
        __syncthreads();
        if (c + 1 < nchunks) load_chunk(c + 1);
#pragma unroll 4
        for (int kk = 0; kk < BK; kk += 2)
            mfma_step(As, AS_LD, Bs, BS_LD, kk, lane, wr * 64, wc * 64, u0, u1, acc);
    }

    // epilogue
    const int l31 = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int col = col0 + wc * 64 + u * 32 + l31;
        if (col >= C) continue;
        float add = 0.f;
        if (a.bias) add += a.bias[(int64_t)slot * C + col];
        if (a.sub) add -= a.sub[col];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wr * 64 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (row < rows_valid) {
                    float v = acc[t][u][r] + add;
                    if (a.relu) v = v > 0.f ? v : (v == v ? 0.f : v);        // NaN stays NaN
                    a.out[(row0 + row) * C + col] = v;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// xty: one 128 x 128 tile of P = Xc^T Yc over one slab of rows per workgroup.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void xty_kernel(WcXtyArgs a, int ntiles, int nb)
{
    __shared__ __attribute__((aligned(16))) float Xs[BK * BM];
    __shared__ __attribute__((aligned(16))) float Ys[BK * BN];

    if (a.gate && *a.gate == 0) return;      // exact redo of a fast-path call: only when it flagged an overflow
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int C = a.C;
    const int64_t z = blockIdx.x / ntiles;
    int t = blockIdx.x % ntiles;
    int ib, jb;
    if (a.sym) {            // enumerate the upper triangle row by row
        ib = 0;
        while (t >= nb - ib) { t -= nb - ib; ++ib; }
        jb = ib + t;
    } else {
        ib = t / nb; jb = t % nb;
    }
    const bool diag = a.sym && ib == jb;

    int64_t r0, r1;
    if (a.per_sample) {
        const int64_t n = z / a.nsplit, q = z % a.nsplit;
        r0 = n * a.HW + q * a.rows_per_slab;
        r1 = r0 + a.rows_per_slab;
        const int64_t end = (n + 1) * a.HW;
        if (r1 > end) r1 = end;
    } else {
        const int64_t M = a.N * a.HW;
        r0 = z * a.rows_per_slab;
        r1 = r0 + a.rows_per_slab;
        if (r1 > M) r1 = M;
    }

    const int q4 = tid & 31;          // float4 column group inside the tile
    const int rbase = tid >> 5;       // + 8*p
    const int ci = ib * BM + 4 * q4, cj = jb * BN + 4 * q4;
    const bool vi = ci < C, vj = cj < C;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const f32x4 cx = (a.cx && vi) ? ld4(a.cx + ci) : zero4;
    const f32x4 cy = (a.cy && vj) ? ld4(a.cy + cj) : zero4;

    // fp32 MFMA accumulators are flushed into float64 registers after every 32-row chunk: the fp32
    // rounding chain is 16 steps long instead of rows_per_slab/2, which is what keeps the covariance
    // (and hence the Cholesky factor of an ill-conditioned batch) at ~1e-8 instead of ~1e-6.
    f32x16 acc[2][2];
    double acc64[2][2][16];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[tt][u][r] = 0.f; acc64[tt][u][r] = 0.0; }
    f32x4 csum = zero4;

    const bool u0 = (jb * BN + wc * 64) < C, u1 = (jb * BN + wc * 64 + 32) < C;
    const bool t_ok = (ib * BM + wr * 64) < C;     // whole wave row-range outside C -> nothing to do

    f32x4 xv[4], yv[4];
    auto load_chunk = [&](int64_t m0) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int64_t m = m0 + rbase + 8 * p;
            const bool ok = m < r1;
            if (a.Xhi) {        // X as pre-split planes (the exact redo of a K4 whose x exists as planes only): (hi + lo) / scale
                xv[p] = zero4;
                if (ok && vi) {
                    const f16x4 h = *reinterpret_cast<const f16x4*>(a.Xhi + m * C + ci), l = *reinterpret_cast<const f16x4*>(a.Xlo + m * C + ci);
                    const f32x4 sc = ld4(a.xscale + ci);
#pragma unroll
                    for (int e = 0; e < 4; ++e) xv[p][e] = ((float)h[e] + (float)l[e]) / sc[e];
                }
            } else
            xv[p] = (ok && vi) ? ld4(a.X + m * C + ci) - cx : zero4;
            if (!diag) {
                yv[p] = (ok && vj) ? ld4(a.Y + m * C + cj) : zero4;
                if (a.ymask && ok && vj) {       // the ReLU's bit mask of row m, columns cj .. cj + 3
                    const uint4 w = *reinterpret_cast<const uint4*>(a.ymask + (m >> 5) * C + cj);
                    const int b = (int)(m & 31);
                    yv[p][0] = (w.x >> b) & 1u ? yv[p][0] : 0.f; yv[p][1] = (w.y >> b) & 1u ? yv[p][1] : 0.f;
                    yv[p][2] = (w.z >> b) & 1u ? yv[p][2] : 0.f; yv[p][3] = (w.w >> b) & 1u ? yv[p][3] : 0.f;
                }
                if (ok && vj) yv[p] -= cy;
            }
        }
    };

    if (r0 < r1) load_chunk(r0);
    for (int64_t m0 = r0; m0 < r1; m0 += BK) {
        __syncthreads();
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            *reinterpret_cast<f32x4*>(&Xs[(rbase + 8 * p) * BM + 4 * q4]) = xv[p];
            if (!diag) *reinterpret_cast<f32x4*>(&Ys[(rbase + 8 * p) * BN + 4 * q4]) = yv[p];
            csum += a.sym ? xv[p] : yv[p];
        }
        __syncthreads();
        if (m0 + BK < r1) load_chunk(m0 + BK);
        if (t_ok) {
            const float* Bsrc = diag ? Xs : Ys;
#pragma unroll 4
            for (int kk = 0; kk < BK; kk += 2)
                mfma_step(Xs, BM, Bsrc, BN, kk, lane, wr * 64, wc * 64, u0, u1, acc);
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int r = 0; r < 16; ++r) { acc64[tt][u][r] += (double)acc[tt][u][r]; acc[tt][u][r] = 0.f; }
        }
    }

    // partial tile out
    const int l31 = lane & 31, lh = lane >> 5;
    double* P = a.P + z * (int64_t)C * C;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int j = jb * BN + wc * 64 + u * 32 + l31;
        if (j >= C) continue;
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = ib * BM + wr * 64 + tt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (i < C) P[(int64_t)i * C + j] = acc64[tt][u][r];
            }
    }

    // column sums: X columns on diagonal tiles (sym), Y columns on the ib == 0 tiles (non-sym)
    const bool want = a.sym ? diag : (ib == 0);
    if (want && a.colsum) {
        __syncthreads();
        float* red = Xs;                           // [8][128]
#pragma unroll
        for (int j = 0; j < 4; ++j) red[rbase * 128 + 4 * q4 + j] = csum[j];
        __syncthreads();
        if (tid < 128) {
            float s = 0.f;
#pragma unroll
            for (int r = 0; r < 8; ++r) s += red[r * 128 + tid];
            const int col = (a.sym ? ib : jb) * 128 + tid;
            if (col < C) a.colsum[z * C + col] = s;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// xty, exact variant for small M: products of float32 values are exact in float64, and the float64
// MFMA (v_mfma_f64_16x16x4_f64) accumulates them in float64, so the partials carry no fp32 rounding
// at all.  Used when M <= WC_EXACT_ROWS (the 4x4 .. 16x16 sites of a batch of 64), where the statistics
// of few rows get no help from averaging.
//   a = A[i = lane&15][k = lane>>4], b = B[k = lane>>4][j = lane&15], D reg r = D[(lane>>4) + 4r][lane&15]
// The float64 MFMA takes 64 cycles per 16 x 16 x 4 product, so what decides the time of these small calls is how many
// SIMDs work on them: 64 x 64 tiles (4 waves, 32 x 32 each) give 10 (covariance) / 16 workgroups per slab of 256 rows at
// C = 256 instead of 3 / 4 with the 128 x 128 tiles of xty_kernel (measured in the CIFAR-10 step: 57-70 us per call
// with those, 7 calls per step).  Rows come in chunks of 128 (all of a thread's loads of a chunk in flight at once).
// ---------------------------------------------------------------------------------------------
constexpr int XT = 64;            // tile edge (channels)
constexpr int XK = 128;           // rows per LDS chunk
constexpr int XLD = XT + 16;      // chunk row pitch in floats: the four k rows of one ds_read_b32 land on different banks

__global__ __launch_bounds__(256) void xty_f64_kernel(WcXtyArgs a, int ntiles, int nb)
{
    extern __shared__ __attribute__((aligned(16))) float xty_lds[];      // Xs[XK][XLD] | Ys[XK][XLD]: 80 KB, two workgroups per CU
    float* Xs = xty_lds;
    float* Ys = xty_lds + XK * XLD;

    if (a.gate && *a.gate == 0) return;      // exact redo of a fast-path call: only when it flagged an overflow
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int C = a.C;
    const int64_t z = blockIdx.x / ntiles;
    int t = blockIdx.x % ntiles;
    int ib, jb;
    if (a.sym) {
        ib = 0;
        while (t >= nb - ib) { t -= nb - ib; ++ib; }
        jb = ib + t;
    } else {
        ib = t / nb; jb = t % nb;
    }
    const bool diag = a.sym && ib == jb;

    int64_t r0, r1;
    if (a.per_sample) {
        const int64_t n = z / a.nsplit, q = z % a.nsplit;
        r0 = n * a.HW + q * a.rows_per_slab;
        r1 = r0 + a.rows_per_slab;
        const int64_t end = (n + 1) * a.HW;
        if (r1 > end) r1 = end;
    } else {
        const int64_t M = a.N * a.HW;
        r0 = z * a.rows_per_slab;
        r1 = r0 + a.rows_per_slab;
        if (r1 > M) r1 = M;
    }

    const int q4 = tid & 15;          // float4 column group inside the tile
    const int rbase = tid >> 4;       // + 16*p
    const int ci = ib * XT + 4 * q4, cj = jb * XT + 4 * q4;
    const bool vi = ci < C, vj = cj < C;
    const int cic = vi ? ci : 0, cjc = vj ? cj : 0;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const f32x4 cx = (a.cx && vi) ? ld4(a.cx + ci) : zero4;
    const f32x4 cy = (a.cy && vj) ? ld4(a.cy + cj) : zero4;

    f64x4 acc[2][2];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int u = 0; u < 2; ++u) acc[tt][u] = f64x4{0.0, 0.0, 0.0, 0.0};
    f32x4 csum = zero4;

    const bool t_ok = (ib * XT + wr * 32) < C && (jb * XT + wc * 32) < C;
    const int li = lane & 15, lq = lane >> 4;

    f32x4 xv[8], yv[8];
    auto load_chunk = [&](int64_t m0) {
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int64_t m = m0 + rbase + 16 * p;
            const bool ok = m < r1;
            // clamped addresses, not predicated loads (hipcc branches around those one by one): all in flight together
            const int64_t mc = ok ? m : r0;
            const f32x4 vx = ld4(a.X + mc * C + cic) - cx;
            xv[p] = (ok && vi) ? vx : zero4;
            if (!diag) {
                const f32x4 vy = ld4(a.Y + mc * C + cjc) - cy;
                yv[p] = (ok && vj) ? vy : zero4;
            }
        }
    };

    if (r0 < r1) load_chunk(r0);
    for (int64_t m0 = r0; m0 < r1; m0 += XK) {
        __syncthreads();
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            *reinterpret_cast<f32x4*>(&Xs[(rbase + 16 * p) * XLD + 4 * q4]) = xv[p];
            if (!diag) *reinterpret_cast<f32x4*>(&Ys[(rbase + 16 * p) * XLD + 4 * q4]) = yv[p];
            csum += a.sym ? xv[p] : yv[p];
        }
        __syncthreads();
        if (m0 + XK < r1) load_chunk(m0 + XK);
        if (t_ok) {
            const float* Bsrc = diag ? Xs : Ys;
            const int64_t left = r1 - m0;
            const int kend = left >= XK ? XK : (int)((left + 3) & ~(int64_t)3);      // rows past r1 are zero in the chunk
            // one wave per SIMD: nothing else hides the LDS latency, so the operands of k-step kk + 4 are read before the
            // MFMAs of k-step kk are issued (sched_barrier: hipcc otherwise moves each read back next to its use)
            const float* xa = Xs + lq * XLD + wr * 32 + li;
            const float* xb = Bsrc + lq * XLD + wc * 32 + li;
            float a0 = xa[0], a1 = xa[16], b0 = xb[0], b1 = xb[16];
#pragma unroll 2
            for (int kk = 0; kk < kend; kk += 4) {
                const double av[2] = {(double)a0, (double)a1}, bv[2] = {(double)b0, (double)b1};
                const int kn = (kk + 4 < XK ? kk + 4 : kk) * XLD;          // the last prefetch re-reads its own rows
                a0 = xa[kn]; a1 = xa[kn + 16]; b0 = xb[kn]; b1 = xb[kn + 16];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                    for (int u = 0; u < 2; ++u)
                        acc[tt][u] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[tt], bv[u], acc[tt][u], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }

    double* P = a.P + z * (int64_t)C * C;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int j = jb * XT + wc * 32 + u * 16 + li;
        if (j >= C) continue;
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = ib * XT + wr * 32 + tt * 16 + lq + 4 * r;
                if (i < C) P[(int64_t)i * C + j] = acc[tt][u][r];
            }
    }

    // column sums: X columns on diagonal tiles (sym), Y columns on the ib == 0 tiles (non-sym)
    const bool want = a.sym ? diag : (ib == 0);
    if (want && a.colsum) {
        __syncthreads();
        float* red = Xs;                           // [16][64]
#pragma unroll
        for (int j = 0; j < 4; ++j) red[rbase * XT + 4 * q4 + j] = csum[j];
        __syncthreads();
        if (tid < XT) {
            float sacc = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc += red[r * XT + tid];
            const int col = (a.sym ? ib : jb) * XT + tid;
            if (col < C) a.colsum[z * C + col] = sacc;
        }
    }
}

// shift[c] = mean of a strided sample of <= 256 rows: removes the mean before fp32 products are summed.
// 16 row groups x 64 channels per 1024-thread block, so no thread walks more than 16 dependent loads.
__global__ __launch_bounds__(1024) void subsample_mean_kernel(const float* __restrict__ x, int64_t M, int C,
                                                              float* __restrict__ shift)
{
    __shared__ float red[16][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int part = threadIdx.x >> 6;
    const int64_t nsamp = M < 256 ? M : 256;
    const int64_t stride = M / nsamp;
    float s = 0.f;
    if (c < C)
        for (int64_t r = part; r < nsamp; r += 16) s += x[wc_sample_row(r, stride) * C + c];
    red[part][threadIdx.x & 63] = s;
    __syncthreads();
    if (threadIdx.x < 64 && c < C) {
        float t = 0.f;
#pragma unroll
        for (int p = 0; p < 16; ++p) t += red[p][threadIdx.x];
        shift[c] = t / (float)nsamp;
    }
}

// The same sample once more for the fast path's per-channel power-of-two scale (2^(4 - ceil(log2 max|x - shift|)), as
// wc_fast.hip's channel_scale_kernel computes it) and the clearing of the overflow gate: one launch instead of three.
__global__ __launch_bounds__(1024) void subsample_mean_scale_kernel(const float* __restrict__ x, int64_t M, int C,
                                                                    float* __restrict__ shift, float* __restrict__ scale,
                                                                    int* __restrict__ gate)
{
    __shared__ float red[16][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int part = threadIdx.x >> 6;
    const int64_t nsamp = M < 256 ? M : 256;
    const int64_t stride = M / nsamp;
    if (blockIdx.x == 0 && threadIdx.x < 64) gate[threadIdx.x] = 0;
    float v[16];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int64_t r = part + 16 * i;
        v[i] = (c < C && r < nsamp) ? x[wc_sample_row(r, stride) * C + c] : 0.f;
        s += v[i];
    }
    red[part][threadIdx.x & 63] = s;
    __syncthreads();
    __shared__ float centre2[2][64];             // [0] mean, [1] median of the 16 group means (the outlier-proof centre)
    if (threadIdx.x < 64) {
        float t = 0.f, pm[16];
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            const float ps = red[p][threadIdx.x];
            t += ps;
            const int64_t cnt = (nsamp - p + 15) / 16;                     // rows of group p
            pm[p] = cnt > 0 ? ps / (float)cnt : 0.f;
        }
        const float mean_ = t / (float)nsamp;
        centre2[0][threadIdx.x] = mean_;
        centre2[1][threadIdx.x] = nsamp >= 16 ? wc_median16(pm) : mean_;
    }
    __syncthreads();
    const float mean = centre2[0][threadIdx.x & 63], med = centre2[1][threadIdx.x & 63];
    // group maxima of |v - mean| and of |v - med|: the first decides (and is the only one used on ordinary data), the
    // second replaces it when an outlier sits on a sampled row (it would have moved the mean by 1/256 of itself)
    float mx = 0.f, mx2 = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int64_t r = part + 16 * i;
        if (c < C && r < nsamp) { mx = fmaxf(mx, fabsf(v[i] - mean)); mx2 = fmaxf(mx2, fabsf(v[i] - med)); }
    }
    __shared__ float red2[16][64];
    red[part][threadIdx.x & 63] = mx;
    red2[part][threadIdx.x & 63] = mx2;
    __syncthreads();
    if (threadIdx.x < 64 && c < C) {
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
        shift[c] = centre;
        scale[c] = sc;
    }
}

// gy where y > 0, else 0 (the gradient of the ReLU that rode in K3's epilogue) -- the fallback of wc_bwd_reduce_relu_f32 for
// the shapes whose K4 kernel does not mask while it stages
__global__ __launch_bounds__(256) void relu_mask_kernel(const f32x4* __restrict__ gy, const f32x4* __restrict__ y,
                                                        f32x4* __restrict__ out, int64_t n4)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const f32x4 g = gy[i], v = y[i];
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = !(v[j] <= 0.f) ? g[j] : 0.f;      // NaN in y lets the gradient through, as aten::threshold_backward does
        out[i] = o;
    }
}

// The same gradient mask from the ONE-BIT form K3 leaves behind (wc_apply_mask_f32): mask[(m / 32) * C + c] bit m % 32 = "the
// activation passed".  One thread = 4 channels of a row: one 16-byte load of the four columns' words instead of 16 bytes of y.
__global__ __launch_bounds__(256) void relu_mask_bits_kernel(const f32x4* __restrict__ gy, const unsigned* __restrict__ mask,
                                                             f32x4* __restrict__ out, int64_t n4, int C4)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const int64_t m = i / C4;
        const int c4 = (int)(i - m * C4);
        const uint4 w = *reinterpret_cast<const uint4*>(mask + ((m >> 5) * C4 + c4) * 4);
        const int b = (int)(m & 31);
        const f32x4 g = gy[i];
        f32x4 o;
        o[0] = (w.x >> b) & 1u ? g[0] : 0.f; o[1] = (w.y >> b) & 1u ? g[1] : 0.f;
        o[2] = (w.z >> b) & 1u ? g[2] : 0.f; o[3] = (w.w >> b) & 1u ? g[3] : 0.f;
        out[i] = o;
    }
}

// the bit mask of an activation that exists in fp32 (the K3 paths that do not write it themselves): one thread per
// (32-row block, column); 32 coalesced loads a row apart
__global__ __launch_bounds__(256) void mask_from_y_kernel(const float* __restrict__ y, int64_t blocks, int C, unsigned* __restrict__ mask)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x, n = blocks * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int64_t t = i / C;
        const int c = (int)(i - t * C);
        const float* p = y + (t * 32) * C + c;
        unsigned w = 0;
#pragma unroll 8
        for (int b = 0; b < 32; ++b) w |= (!(p[(int64_t)b * C] <= 0.f) ? 1u : 0u) << b;
        mask[i] = w;
    }
}

__global__ __launch_bounds__(256) void stream_copy_kernel(const f32x4* __restrict__ src, f32x4* __restrict__ dst, int64_t n4)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
#ifndef WC_NT_COPY
#define WC_NT_COPY 1      // the stream-copy yardstick gets the same nontemporal stores as K3's epilogue (40.8 -> 40.5 us)
#endif
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
#if WC_NT_COPY
        __builtin_nontemporal_store(src[i], &dst[i]);
#else
        dst[i] = src[i];
#endif
    }
}

}  // namespace

hipError_t wc_launch_rows_gemm(const WcRowsGemmArgs& a, hipStream_t st)
{
    const int ncb = (a.C + BN - 1) / BN;
    int64_t nrb;
    if (a.slot) nrb = a.N * ((a.HW + BM - 1) / BM);
    else nrb = (a.N * a.HW + BM - 1) / BM;
    const int64_t grid = nrb * ncb;
    hipLaunchKernelGGL(rows_gemm_kernel, dim3((unsigned)grid), dim3(256), 0, st, a, ncb);
    return hipGetLastError();
}

int wc_xty_plan(int64_t N, int64_t HW, int C, int per_sample, int sym, int* nsplit, int64_t* rows_per_slab)
{
    const int nb = (C + BM - 1) / BM;
    const int64_t target = 512;                       // ~2 workgroups per CU (the fp64 flush registers cap it at 2)
    // (round 6: the smallest sites -- 4 x 4 pixels, 2 048 rows, the exact float64 kernel -- as slabs of ONE 128-row chunk: 160 workgroups instead
    //  of 80 on a chip of 256 CUs; above 4 096 rows the slabs' float64 partials would double what the tail reads for nothing)
    const int64_t min_rows = (per_sample ? HW : N * HW) <= 4096 ? 128 : 256;
    const int ntiles = sym ? nb * (nb + 1) / 2 : nb * nb;
    if (per_sample) {
        int64_t want = (target + N * ntiles - 1) / (N * ntiles);         // slabs per sample
        int64_t maxs = (HW + min_rows - 1) / min_rows;
        if (want > maxs) want = maxs;
        if (want < 1) want = 1;
        int64_t rps = (HW + want - 1) / want;
        rps = (rps + BK - 1) / BK * BK;
        *nsplit = (int)((HW + rps - 1) / rps);
        *rows_per_slab = rps;
        return (int)(N * (*nsplit));
    }
    const int64_t M = N * HW;
    int64_t want = (target + ntiles - 1) / ntiles;
    int64_t maxs = (M + min_rows - 1) / min_rows;
    if (want > maxs) want = maxs;
    if (want < 1) want = 1;
    int64_t rps = (M + want - 1) / want;
    rps = (rps + BK - 1) / BK * BK;
    *nsplit = (int)((M + rps - 1) / rps);
    *rows_per_slab = rps;
    return *nsplit;
}

hipError_t wc_launch_xty(const WcXtyArgs& a, int nslab, hipStream_t st)
{
    if (a.N * a.HW <= WC_EXACT_ROWS) {
        const int nb = (a.C + XT - 1) / XT;
        const int ntiles = a.sym ? nb * (nb + 1) / 2 : nb * nb;
        constexpr size_t lds = (size_t)2 * XK * XLD * sizeof(float);
        static bool attr_set = false;
        if (!attr_set) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(xty_f64_kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
            attr_set = true;
        }
        hipLaunchKernelGGL(xty_f64_kernel, dim3((unsigned)((int64_t)nslab * ntiles)), dim3(256), lds, st, a, ntiles, nb);
        return hipGetLastError();
    }
    const int nb = (a.C + BM - 1) / BM;
    const int ntiles = a.sym ? nb * (nb + 1) / 2 : nb * nb;
    const int64_t grid = (int64_t)nslab * ntiles;
    hipLaunchKernelGGL(xty_kernel, dim3((unsigned)grid), dim3(256), 0, st, a, ntiles, nb);
    return hipGetLastError();
}

hipError_t wc_launch_subsample_mean(const float* x, int64_t M, int C, float* shift, hipStream_t st)
{
    hipLaunchKernelGGL(subsample_mean_kernel, dim3((C + 63) / 64), dim3(1024), 0, st, x, M, C, shift);
    return hipGetLastError();
}

hipError_t wc_launch_subsample_mean_scale(const float* x, int64_t M, int C, float* shift, float* scale, int* gate, hipStream_t st)
{
    hipLaunchKernelGGL(subsample_mean_scale_kernel, dim3((C + 63) / 64), dim3(1024), 0, st, x, M, C, shift, scale, gate);
    return hipGetLastError();
}

hipError_t wc_launch_relu_mask(const float* gy, const float* y, float* out, int64_t n, hipStream_t st)
{
    hipLaunchKernelGGL(relu_mask_kernel, dim3(2048), dim3(256), 0, st, reinterpret_cast<const f32x4*>(gy),
                       reinterpret_cast<const f32x4*>(y), reinterpret_cast<f32x4*>(out), n / 4);
    return hipGetLastError();
}

hipError_t wc_launch_relu_mask_bits(const float* gy, const unsigned* mask, float* out, int64_t M, int C, hipStream_t st)
{
    hipLaunchKernelGGL(relu_mask_bits_kernel, dim3(2048), dim3(256), 0, st, reinterpret_cast<const f32x4*>(gy), mask,
                       reinterpret_cast<f32x4*>(out), M * C / 4, C / 4);
    return hipGetLastError();
}

hipError_t wc_launch_mask_from_y(const float* y, int64_t M, int C, unsigned* mask, hipStream_t st)
{
    const int64_t n = (M / 32) * C;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(mask_from_y_kernel, dim3((unsigned)blocks), dim3(256), 0, st, y, M / 32, C, mask);
    return hipGetLastError();
}

hipError_t wc_launch_stream_copy(const float* src, float* dst, int64_t n, hipStream_t st)
{
    const int64_t n4 = n / 4;
    hipLaunchKernelGGL(stream_copy_kernel, dim3(2048), dim3(256), 0, st,
                       reinterpret_cast<const f32x4*>(src), reinterpret_cast<f32x4*>(dst), n4);
    return hipGetLastError();
}
