// Small-matrix (C x C) stage of the WC path, float64 throughout.
//
// Replaces, inside DecorelationNormalization.call (generator.py:24,26; body in the un-vendored gan/
// submodule): the covariance normalisation and shrinkage, tf.cholesky, tf.matrix_triangular_solve
// against I, the moving-statistics add_update ops, and -- for the coloring layers of
// generator.py:49-78 -- the product A_k = W^T Gamma_k that lets K3 run one fused affine.  The backward
// half (wc_bwd_factor_f64) is the closed form of SURVEY.md row a10.
//
// float64 here is what keeps the path within 1e-4 of the float64 oracle on ill-conditioned batches
// (hard part 2); the flops are negligible (O(C^3)) next to the M*C^2 kernels.
#include "wc_common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

__device__ __forceinline__ double readlane64(double v, int l)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}

// ---------------------------------------------------------------------------------------------
// K1 tail
// ---------------------------------------------------------------------------------------------
// 16 slab groups x 64 channels per block; partial sums meet in LDS (fixed order: deterministic)
// dfix / Dp (nullable): the fast path's per-slab VALU diagonal -> Dp[c] = the centred sum of squares of channel c, which the
// off-diagonal elements' bias compensation scales with (stats_xtx_kernel)
__global__ __launch_bounds__(1024) void stats_colsum_kernel(const float* __restrict__ colsum, const float* __restrict__ shift,
                                                            int nslab, int64_t M, int C, double* __restrict__ Sp,
                                                            double* __restrict__ sum, const double* __restrict__ dfix = nullptr,
                                                            double* __restrict__ Dp = nullptr)
{
    __shared__ double red[16][64];
    __shared__ double redd[16][64];
    // statistic group = blockIdx.y: its nslab partial slabs, its Sp / sum rows (shift is common to all groups)
    colsum += (int64_t)blockIdx.y * nslab * C; Sp += (int64_t)blockIdx.y * C; sum += (int64_t)blockIdx.y * C;
    const bool diag = dfix && Dp;
    if (diag) { dfix += (int64_t)blockIdx.y * nslab * C; Dp += (int64_t)blockIdx.y * C; }
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int part = threadIdx.x >> 6;
    double s = 0.0, d = 0.0;
    if (c < C) {
        for (int z = part; z < nslab; z += 16) s += (double)colsum[(int64_t)z * C + c];
        if (diag) for (int z = part; z < nslab; z += 16) d += dfix[(int64_t)z * C + c];      // (both loops' loads are in flight together)
    }
    red[part][threadIdx.x & 63] = s;
    if (diag) redd[part][threadIdx.x & 63] = d;
    __syncthreads();
    if (threadIdx.x < 64 && c < C) {
        double t = 0.0;
#pragma unroll
        for (int p = 0; p < 16; ++p) t += red[p][threadIdx.x];
        Sp[c] = t;
        sum[c] = t + (double)M * (double)shift[c];
    } else if (diag && threadIdx.x >= 64 && threadIdx.x < 128 && c < C) {
        double t = 0.0;
#pragma unroll
        for (int p = 0; p < 16; ++p) t += redd[p][threadIdx.x - 64];
        Dp[c] = t;
    }
}

// Sum of the strided terms p[z * stride], z = z0, z0 + step, ... (< n), with up to 16 loads in flight at once: the slab
// reductions below are latency-bound otherwise (4 loads in flight per thread: 15 us for 34 MB at C = 256).  The index is
// clamped instead of predicated -- hipcc puts an s_waitcnt vmcnt(0) between exec-masked loads -- so the batch shrinks
// with n (the grouped critic-phase sites have 26 slabs per group: no 12 idle loads per thread).  Fixed order: deterministic.
template <int UB, typename T>
__device__ __forceinline__ double strided_batch(const T* __restrict__ p, int64_t stride, int z0, int step, int n)
{
    double v[UB];
#pragma unroll
    for (int u = 0; u < UB; ++u) {
        const int z = z0 + u * step;
        v[u] = (double)p[(int64_t)(z < n ? z : n - 1) * stride];
    }
    double s = 0.0;
#pragma unroll
    for (int u = 0; u < UB; ++u) s += (z0 + u * step < n) ? v[u] : 0.0;
    return s;
}
template <typename T>
__device__ __forceinline__ double strided_sum(const T* __restrict__ p, int64_t stride, int part, int step, int n)
{
    double s = 0.0;
    if (n <= 4 * step) { if (part < n) s = strided_batch<4>(p, stride, part, step, n); }
    else if (n <= 8 * step) s = strided_batch<8>(p, stride, part, step, n);
    else
        for (int z0 = part; z0 < n; z0 += 16 * step) s += strided_batch<16>(p, stride, z0, step, n);
    return s;
}

// xtx = G' + s Sp^T + Sp s^T + M s s^T with G' = sum_z P[z]; only block-upper tiles of P were written.
// block = 64 columns x 8 slab groups, one row i per blockIdx.y
constexpr int SX_PARTS = 8;
__global__ __launch_bounds__(64 * SX_PARTS) void stats_xtx_kernel(const double* __restrict__ P, const float* __restrict__ shift,
                                                                  const double* __restrict__ Sp, int nslab, int64_t M, int C,
                                                                  double* __restrict__ xtx, const double* __restrict__ dfix,
                                                                  const int* __restrict__ gate, double kappa = 0.0,
                                                                  const double* __restrict__ Dp = nullptr)
{
    __shared__ double red[SX_PARTS][64];
    const int j = blockIdx.x * 64 + (threadIdx.x & 63);
    const int part = threadIdx.x >> 6;
    const int i = blockIdx.y;
    if (blockIdx.x * 64 + 63 < i) return;          // whole block below the diagonal
    const int64_t CC = (int64_t)C * C;
    P += (int64_t)blockIdx.z * nslab * CC; Sp += (int64_t)blockIdx.z * C; xtx += (int64_t)blockIdx.z * CC;   // group
    if (Dp) Dp += (int64_t)blockIdx.z * C;
    double g = 0.0;
    if (j < C && j >= i) {
        // the fast reduction's diagonal comes from its VALU sums of squares, not from the matrix pipe (wc_fast_xty.hip:
        // the MFMA's rounding is biased for all-positive products) -- unless the exact redo has replaced the partials
        const bool diag = j == i && dfix && !(gate && *gate != 0);
        const double* p = diag ? dfix + (int64_t)blockIdx.z * nslab * C + i : P + (int64_t)i * C + j;
        const int64_t stride = diag ? C : CC;
        g = strided_sum(p, stride, part, SX_PARTS, nslab);
    }
    red[part][threadIdx.x & 63] = g;
    __syncthreads();
    if (threadIdx.x < 64 && j < C && j >= i) {
        g = 0.0;
#pragma unroll
        for (int q = 0; q < SX_PARTS; ++q) g += red[q][threadIdx.x];
        // the matrix pipe's accumulation bias on the off-diagonal sums (wc_fast_xty.hip, kXtyOffdiagBias), taken out again
        if (j != i && Dp && kappa != 0.0 && dfix && !(gate && *gate != 0)) g += kappa * sqrt(Dp[i] * Dp[j]);
        const double si = shift[i], sj = shift[j];
        const double v = g + si * Sp[j] + Sp[i] * sj + (double)M * si * sj;
        xtx[(int64_t)i * C + j] = v;
        if (j != i) xtx[(int64_t)j * C + i] = v;   // mirrored by the same thread: exact symmetry
    }
}

// K1 tail + K2 head in ONE launch, for the caller that wants the factorisation and not the moments (wc_whiten_f32: training mode,
// per-replica statistics -- the reference's behaviour): stats_xtx_kernel's slab reduction and
// factor_prepare_kernel's bookkeeping by the same thread, element by element the SAME float64 expressions in the same order
// (tests compare the two routes for equality).  The statistic groups are walked by every workgroup in turn (their
// moving-statistics updates are applied one after the other, as `groups` separate calls would); xtx and sum are never stored.
__global__ __launch_bounds__(64 * SX_PARTS) void stats_xtx_prepare_kernel(const double* __restrict__ P, const float* __restrict__ shift,
                                                                          const double* __restrict__ Sp, int nslab, int64_t M, int C,
                                                                          const double* __restrict__ dfix, const int* __restrict__ gate,
                                                                          double kappa, const double* __restrict__ Dp,
                                                                          int groups, double eps, double momentum, int ddof,
                                                                          float* __restrict__ moving_mean, float* __restrict__ moving_cov,
                                                                          float* __restrict__ mu, float* __restrict__ chan_scale,
                                                                          double* __restrict__ T, int lower_only,
                                                                          unsigned* __restrict__ rows, int nrows)
{
    __shared__ double red[SX_PARTS][64];
    const int j = blockIdx.x * 64 + (threadIdx.x & 63);
    const int part = threadIdx.x >> 6;
    const int i = blockIdx.y;
    if (i == 0 && part == 0 && j < nrows) rows[j] = 0u;       // the row-block counters of the fused factor launch
    if (blockIdx.x * 64 + 63 < i) return;                     // whole block below the diagonal (its elements are mirrored from above)
    const int64_t CC = (int64_t)C * C;
    const bool live = j < C && j >= i;
    const bool exact_diag = dfix && !(gate && *gate != 0);
    const int64_t e = (int64_t)i * C + j, et = (int64_t)j * C + i;
    const double invM = 1.0 / (double)M;
    double tmax = 0.0;
    for (int g = 0; g < groups; ++g) {
        double acc = 0.0;
        if (live) {
            const bool diag = j == i && exact_diag;
            const double* p = diag ? dfix + (int64_t)g * nslab * C + i : P + (int64_t)g * nslab * CC + e;
            acc = strided_sum(p, diag ? (int64_t)C : CC, part, SX_PARTS, nslab);
        }
        red[part][threadIdx.x & 63] = acc;
        __syncthreads();
        if (threadIdx.x < 64 && live) {
            acc = 0.0;
#pragma unroll
            for (int q = 0; q < SX_PARTS; ++q) acc += red[q][threadIdx.x];
            if (j != i && Dp && kappa != 0.0 && exact_diag) acc += kappa * sqrt(Dp[(int64_t)g * C + i] * Dp[(int64_t)g * C + j]);
            const double* spg = Sp + (int64_t)g * C;
            const double si = shift[i], sj = shift[j];
            const double v = acc + si * spg[j] + spg[i] * sj + (double)M * si * sj;              // xtx[i][j] (= xtx[j][i])
            const double sum_i = spg[i] + (double)M * si, sum_j = spg[j] + (double)M * sj;       // sum[i], sum[j]
            const double sig = (0.5 * (v + v) - sum_i * sum_j * invM) / (double)(M - ddof);
            if (moving_cov) {
                moving_cov[e] = (float)(momentum * (double)moving_cov[e] + (1.0 - momentum) * sig);
                if (j != i) moving_cov[et] = (float)(momentum * (double)moving_cov[et] + (1.0 - momentum) * sig);
            }
            if (i == 0) {
                const double m = sum_j * invM;
                mu[(int64_t)g * C + j] = (float)m;
                if (moving_mean) moving_mean[j] = (float)(momentum * (double)moving_mean[j] + (1.0 - momentum) * m);
            }
            const double t = (1.0 - eps) * sig + (i == j ? eps : 0.0);
            T[g * CC + e] = (lower_only && (j >> 4) > (i >> 4)) ? 0.0 : t;
            if (j != i) T[g * CC + et] = t;
            tmax = t > tmax ? t : tmax;
        }
        __syncthreads();
    }
    if (chan_scale && threadIdx.x < 64 && live && i == j) {
        int ex;
        frexp(sqrt(tmax), &ex);
        chan_scale[j] = (float)ldexp(1.0, 3 - ex);
    }
}

// ---------------------------------------------------------------------------------------------
// K4 tail: per-slab partials -> per-slot R, gsum.  A block of 512 threads is `parts` slab groups x 512/parts elements: 8 groups
// when one long run of slabs is summed, fewer when every sample has only one or two slabs of its own (class slots) -- with
// 8 groups and one slab per sample seven threads of eight idled and the grid was 33 000 workgroups at Kc = N = 128 (the
// per-sample tables of K = 200 classes): 524 us.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64 * SX_PARTS) void bwd_combine_kernel(const double* __restrict__ P, const float* __restrict__ colsum,
                                                                    const int32_t* __restrict__ slot, int64_t N, int nsplit,
                                                                    int per_sample, int C, double* __restrict__ R,
                                                                    double* __restrict__ gsum, int parts)
{
    __shared__ double red[64 * SX_PARTS];
    const int64_t CC = (int64_t)C * C;
    const int epw = (64 * SX_PARTS) / parts;                             // elements per workgroup
    const int el = threadIdx.x % epw, part = threadIdx.x / epw;
    const int64_t e = (int64_t)blockIdx.x * epw + el;                    // element of C*C (+ C for gsum)
    const int k = blockIdx.y;
    const bool live = e < CC + C;
    const bool is_sum = e >= CC;
    double acc = 0.0;
    auto terms = [&](int64_t zbase, int n) {           // sum over z = zbase + part, + parts, ... < zbase + n
        const double t = is_sum ? strided_sum(colsum + zbase * C + (e - CC), (int64_t)C, part, parts, n)
                                : strided_sum(P + zbase * CC + e, CC, part, parts, n);
        return t;
    };
    if (per_sample) {
        // the samples of slot k; the slot vector goes through LDS once per workgroup
        __shared__ int sl[1024];
        for (int64_t n0 = 0; n0 < N; n0 += 1024) {
            const int cnt = (int)(N - n0 < 1024 ? N - n0 : 1024);
            __syncthreads();
            for (int i = threadIdx.x; i < cnt; i += 64 * SX_PARTS) sl[i] = slot[n0 + i];
            __syncthreads();
            if (live)
                for (int i = 0; i < cnt; ++i)
                    if (sl[i] == k) acc += terms((n0 + i) * nsplit, nsplit);
        }
    } else if (live) {
        acc = terms(0, nsplit);
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    if (part == 0 && live) {
        acc = 0.0;
        for (int q = 0; q < parts; ++q) acc += red[q * epw + el];
        if (is_sum) gsum[(int64_t)k * C + (e - CC)] = acc;
        else R[k * CC + e] = acc;
    }
}

// ---------------------------------------------------------------------------------------------
// K2 head: moments -> mu, Sigma, moving statistics, T = (1-eps) Sigma + eps I
// ---------------------------------------------------------------------------------------------
__global__ void factor_prepare_kernel(const double* __restrict__ sum, const double* __restrict__ xtx, int64_t M, int C,
                                      double eps, double momentum, int ddof, int training, int groups,
                                      float* __restrict__ moving_mean, float* __restrict__ moving_cov,
                                      float* __restrict__ mu, float* __restrict__ chan_scale, double* __restrict__ T, int lower_only,
                                      unsigned* __restrict__ rows, int nrows)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (i == 0 && j < nrows) rows[j] = 0u;       // the row-block counters of the fused factor launch (tri_inverse_role)
    if (j >= C) return;
    const int64_t e = (int64_t)i * C + j, CC = (int64_t)C * C;
    double tmax = 0.0;
    // statistic groups are independent batches; their moving-statistics updates are applied one after the other,
    // exactly as `groups` separate calls would
    for (int g = 0; g < groups; ++g) {
        double sig;
        if (training) {
            const double* sg = sum + (int64_t)g * C;
            const double* xg = xtx + g * CC;
            const double invM = 1.0 / (double)M;
            sig = (0.5 * (xg[e] + xg[(int64_t)j * C + i]) - sg[i] * sg[j] * invM) / (double)(M - ddof);
            if (moving_cov) moving_cov[e] = (float)(momentum * (double)moving_cov[e] + (1.0 - momentum) * sig);
            if (i == 0) {
                const double m = sg[j] * invM;
                mu[(int64_t)g * C + j] = (float)m;
                if (moving_mean) moving_mean[j] = (float)(momentum * (double)moving_mean[j] + (1.0 - momentum) * m);
            }
        } else {
            sig = 0.5 * ((double)moving_cov[e] + (double)moving_cov[(int64_t)j * C + i]);
            if (i == 0) mu[(int64_t)g * C + j] = moving_mean[j];
        }
        const double t = (1.0 - eps) * sig + (i == j ? eps : 0.0);
        // lower_only: the fused Cholesky reads block rows at and below the diagonal only and leaves the rest as it finds
        // it -- the zeros of L's upper triangle are written here, off its critical path
        T[g * CC + e] = (lower_only && (j >> 4) > (i >> 4)) ? 0.0 : t;
        tmax = t > tmax ? t : tmax;
    }
    if (chan_scale && i == j) {
        // power-of-two scale for the fp16 fast path (common to all groups: the largest variance decides):
        // (x - mu) * s has a standard deviation in [4, 8), 7500 sigma below the fp16 guard, exact to undo
        int ex;
        frexp(sqrt(tmax), &ex);
        chan_scale[j] = (float)ldexp(1.0, 3 - ex);
    }
}

// ---------------------------------------------------------------------------------------------
// Cholesky, one 1024-thread workgroup, right-looking with 16-wide panels.
//   per panel: wave 0 factors the 16x16 diagonal block in registers (lane = row, readlane broadcasts),
//   every thread solves one row of the panel against it, then 4x4 register micro-tiles apply the
//   rank-16 update to the trailing lower triangle with the panel held in LDS as [c][row].
// ---------------------------------------------------------------------------------------------
constexpr int CH_NB = 16;

__global__ __launch_bounds__(1024) void cholesky_kernel(double* __restrict__ T, int C, int ldp)
{
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double* D = sm;                    // [16][17]
    double* rdiag = D + 16 * 17;       // [16]
    double* Pn = rdiag + 16;           // [16][ldp]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    T += (int64_t)blockIdx.x * C * C;  // one matrix (statistic group) per workgroup

    for (int j0 = 0; j0 < C; j0 += CH_NB) {
        const int rows = C - j0 - CH_NB;
        const int g0 = j0 + CH_NB;
        // (0) prefetch this thread's (at most two) 4x4 micro-tiles of the trailing matrix: their global loads fly
        //     while wave 0 runs the serial 16x16 factorisation and the panel rows are solved
        const int nt = rows > 0 ? rows >> 2 : 0;
        const int count = nt * (nt + 1) / 2;
        int ti_[2], tk_[2];
        double tile[2][4][4];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int e = tid + 1024 * q;
            ti_[q] = -1; tk_[q] = 0;
            if (e < count) {
                int ti = (int)((sqrt(8.0 * (double)e + 1.0) - 1.0) * 0.5);
                while ((ti + 1) * (ti + 2) / 2 <= e) ++ti;
                while (ti * (ti + 1) / 2 > e) --ti;
                ti_[q] = ti; tk_[q] = e - ti * (ti + 1) / 2;
#pragma unroll
                for (int x = 0; x < 4; ++x) {
                    const double* tr = T + (int64_t)(g0 + 4 * ti + x) * C + g0 + 4 * tk_[q];
#pragma unroll
                    for (int y = 0; y < 4; ++y) tile[q][x][y] = tr[y];
                }
            }
        }
        // (1) wave 0 factors the 16x16 diagonal block in registers: lane = row, readlane broadcasts the pivot column;
        //     the pivot's reciprocal square root comes from v_rsq_f64 + two Newton steps (no fp64 sqrt/divide chain)
        if (wave == 0) {
            double a[CH_NB];
            const int li = lane & 15;
            double myrd = 1.0;
#pragma unroll
            for (int c = 0; c < CH_NB; ++c) a[c] = T[(int64_t)(j0 + li) * C + j0 + c];
#pragma unroll
            for (int j = 0; j < CH_NB; ++j) {
                const double p = readlane64(a[j], j);
                double rd = __builtin_amdgcn_rsq(p);
                rd = rd * (1.5 - 0.5 * p * rd * rd);
                rd = rd * (1.5 - 0.5 * p * rd * rd);
                const double d = p * rd;
                if (li == j) myrd = rd;
                a[j] = (li == j) ? d : a[j] * rd;
#pragma unroll
                for (int k = j + 1; k < CH_NB; ++k) {
                    const double lkj = readlane64(a[j], k);
                    a[k] -= a[j] * lkj;
                }
            }
            if (lane < 16) {
#pragma unroll
                for (int c = 0; c < CH_NB; ++c) {
                    const double v = (c <= lane) ? a[c] : 0.0;
                    D[lane * 17 + c] = v;
                    T[(int64_t)(j0 + lane) * C + j0 + c] = v;
                }
                rdiag[lane] = myrd;
            }
        }
        __syncthreads();
        if (rows <= 0) break;
        // (2) one thread per panel row: forward substitution against the diagonal block
        if (tid < rows) {
            double x[CH_NB];
            double* trow = T + (int64_t)(g0 + tid) * C + j0;
#pragma unroll
            for (int c = 0; c < CH_NB; ++c) x[c] = trow[c];
#pragma unroll
            for (int c = 0; c < CH_NB; ++c) {
                double s = x[c];
#pragma unroll
                for (int k = 0; k < c; ++k) s -= x[k] * D[c * 17 + k];
                x[c] = s * rdiag[c];
            }
#pragma unroll
            for (int c = 0; c < CH_NB; ++c) { trow[c] = x[c]; Pn[c * ldp + tid] = x[c]; }
        }
        __syncthreads();
        // (3) rank-16 update of the prefetched micro-tiles (any beyond two per thread: loaded here)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            if (ti_[q] < 0) continue;
            const int ti = ti_[q], tk = tk_[q];
#pragma unroll 4
            for (int c = 0; c < CH_NB; ++c) {
                const double* pi = Pn + c * ldp + 4 * ti;
                const double* pk = Pn + c * ldp + 4 * tk;
                const double ai[4] = {pi[0], pi[1], pi[2], pi[3]};
                const double ak[4] = {pk[0], pk[1], pk[2], pk[3]};
#pragma unroll
                for (int x = 0; x < 4; ++x)
#pragma unroll
                    for (int y = 0; y < 4; ++y) tile[q][x][y] -= ai[x] * ak[y];
            }
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                double* tr = T + (int64_t)(g0 + 4 * ti + x) * C + g0 + 4 * tk;
#pragma unroll
                for (int y = 0; y < 4; ++y) tr[y] = tile[q][x][y];
            }
        }
        for (int e = tid + 2048; e < count; e += 1024) {        // C > 272 only
            int ti = (int)((sqrt(8.0 * (double)e + 1.0) - 1.0) * 0.5);
            while ((ti + 1) * (ti + 2) / 2 <= e) ++ti;
            while (ti * (ti + 1) / 2 > e) --ti;
            const int tk = e - ti * (ti + 1) / 2;
            double acc[4][4];
#pragma unroll
            for (int x = 0; x < 4; ++x)
#pragma unroll
                for (int y = 0; y < 4; ++y) acc[x][y] = 0.0;
#pragma unroll 4
            for (int c = 0; c < CH_NB; ++c) {
                const double* pi = Pn + c * ldp + 4 * ti;
                const double* pk = Pn + c * ldp + 4 * tk;
                const double ai[4] = {pi[0], pi[1], pi[2], pi[3]};
                const double ak[4] = {pk[0], pk[1], pk[2], pk[3]};
#pragma unroll
                for (int x = 0; x < 4; ++x)
#pragma unroll
                    for (int y = 0; y < 4; ++y) acc[x][y] += ai[x] * ak[y];
            }
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                double* tr = T + (int64_t)(g0 + 4 * ti + x) * C + g0 + 4 * tk;
#pragma unroll
                for (int y = 0; y < 4; ++y) tr[y] -= acc[x][y];
            }
        }
        __syncthreads();
    }
    // strict upper triangle := 0
    for (int64_t e = tid; e < (int64_t)C * C; e += 1024) {
        const int i = (int)(e / C), j = (int)(e % C);
        if (j > i) T[e] = 0.0;
    }
}

// Register-resident form for C <= 256: the lower triangle lives in the 1024 threads' registers as 4x4 tiles (two per
// thread, tile e of the row-major triangle on thread e % 1024) for the whole factorisation.  Per 16-wide
// panel: its tiles go to LDS, wave 0 factors AND inverts the 16x16 diagonal block, the panel solve is a product with that inverse on the MFMA, every
// thread applies the rank-16 update to the tiles it owns -- four barriers and no global round trip inside the loop
// (the form above pays three L2 round trips per panel: measured 190 us at C = 256, of which the flops are ~25).
#ifndef CHOL_SKIP
#define CHOL_SKIP 0      // development: 1 no diagonal factor, 2 no panel solve, 4 no update, 8 no extraction (wrong results)
#endif
typedef double f64x4 __attribute__((ext_vector_type(4)));
// Register-resident form for C <= 256: the trailing matrix lives in the 16 waves' registers as 16x16 blocks in the
// f64-MFMA accumulator layout (register r of lane l = block[(l>>4) + 4r][l&15]) for the whole factorisation (8 waves).  Per
// 16-wide panel: its blocks go to LDS, wave 0 factors AND inverts the 16x16 diagonal block, the panel solve is a product with that inverse on the MFMA,
// every wave applies the rank-16 update to its blocks with four v_mfma_f64_16x16x4_f64 each (operands: two doubles
// per lane and MFMA from the solved panel in LDS) -- four barriers and no global round trip inside the loop.  (The
// form above pays three L2 round trips per panel and runs the update on 4x4 register micro-tiles fed from LDS.)
// Blocks are ranked by block column, LAST column first, and dealt to the waves round-robin: the blocks still active
// at any panel step are a prefix of the ranking, so the waves stay evenly loaded as the matrix shrinks.
__global__ __launch_bounds__(512) void cholesky_reg_kernel(double* __restrict__ T, int C, int ldp)
{
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double* Praw = sm;                  // [C][17]   the panel as its owners hold it (row-major)
    double* D = Praw + C * 17;          // [16][17]  INVERSE of the factored diagonal block
    double* rdiag = D + 16 * 17;        // [16]  (unused)
    double* Pn = rdiag + 16;            // [16][ldp] solved panel, column-major: the update's operands
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // scalar: block ownership stays in SGPRs
    const int li = lane & 15, lq = lane >> 4;
    T += (int64_t)blockIdx.x * C * C;   // one matrix (statistic group) per workgroup
    const int nb = C >> 4;              // block rows; block column 0 (the first panel) goes from global straight to LDS
    const int nblk = (nb - 1) * nb / 2;

    constexpr int SLOTS = 15;           // 120 blocks at C = 256 over 8 waves (512 threads: 256 VGPRs each, no spills)
    int bi_[SLOTS], bj_[SLOTS];
    f64x4 blk[SLOTS];
#pragma unroll
    for (int q = 0; q < SLOTS; ++q) {
        const int r = wave + 8 * q;
        bi_[q] = -1; bj_[q] = 0;
        if (r < nblk) {
            int m = (int)((sqrt(8.0 * (double)r + 1.0) - 1.0) * 0.5);
            while ((m + 1) * (m + 2) / 2 <= r) ++m;
            while (m * (m + 1) / 2 > r) --m;                    // column nb-1-m holds m+1 blocks
            bj_[q] = nb - 1 - m; bi_[q] = bj_[q] + (r - m * (m + 1) / 2);
#pragma unroll
            for (int e = 0; e < 4; ++e) blk[q][e] = T[(int64_t)(16 * bi_[q] + lq + 4 * e) * C + 16 * bj_[q] + li];
        }
    }
    for (int e = tid; e < C * 16; e += 512) Praw[(e >> 4) * 17 + (e & 15)] = T[(int64_t)(e >> 4) * C + (e & 15)];

    for (int j = 0; j < nb; ++j) {
        const int j0 = 16 * j;
        const int rows = C - j0 - CH_NB;
        const int g0 = j0 + CH_NB;
        // (1) the panel's blocks (block column j) leave the registers
        if (j > 0) {
#pragma unroll
            for (int q = 0; q < SLOTS; ++q) {
                if (bi_[q] >= 0 && bj_[q] == j && !(CHOL_SKIP & 8)) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) Praw[(16 * bi_[q] + lq + 4 * e) * 17 + li] = blk[q][e];
                }
            }
        }
        __syncthreads();
        // (2) wave 0 factors the 16x16 diagonal block in registers: lane = row, readlane broadcasts the pivot column;
        //     the pivot's reciprocal square root comes from v_rsq_f64 + two Newton steps
        if (wave == 0 && !(CHOL_SKIP & 1)) {
            double a[CH_NB];
            double myrd = 1.0;
#pragma unroll
            for (int c = 0; c < CH_NB; ++c) a[c] = Praw[(j0 + li) * 17 + c];
#pragma unroll
            for (int jj = 0; jj < CH_NB; ++jj) {
                const double p = readlane64(a[jj], jj);
                double rd = __builtin_amdgcn_rsq(p);
                rd = rd * (1.5 - 0.5 * p * rd * rd);
                rd = rd * (1.5 - 0.5 * p * rd * rd);
                const double d = p * rd;
                if (li == jj) myrd = rd;
                a[jj] = (li == jj) ? d : a[jj] * rd;
#pragma unroll
                for (int k = jj + 1; k < CH_NB; ++k) {
                    const double lkj = readlane64(a[jj], k);
                    a[k] -= a[jj] * lkj;
                }
            }
            // ... and inverts it: lane = column c of the inverse, forward substitution with the rows of L read by
            // readlane (the panel solve below is then a product on the MFMA instead of 136 dependent steps per row)
            double w[CH_NB];
#pragma unroll
            for (int i = 0; i < CH_NB; ++i) w[i] = (i == li) ? myrd : 0.0;
#pragma unroll
            for (int i = 1; i < CH_NB; ++i) {
                double acc = 0.0;
#pragma unroll
                for (int kk = 0; kk < i; ++kk) acc += readlane64(a[kk], i) * w[kk];
                const double wi = -acc * readlane64(myrd, i);
                w[i] = (i > li) ? wi : w[i];
            }
            if (lane < 16) {
#pragma unroll
                for (int c = 0; c < CH_NB; ++c) {
                    const double v = (c <= lane) ? a[c] : 0.0;
                    T[(int64_t)(j0 + lane) * C + j0 + c] = v;
                }
#pragma unroll
                for (int i = 0; i < CH_NB; ++i) D[i * 17 + lane] = w[i];       // D[i][c] = (L^-1)[i][c]
            }
        }
        __syncthreads();
        if (rows <= 0) break;
        // (3) panel solve X = P L^-T as a product with the inverted diagonal block: X[r][c] = sum_k P[r][k] Linv[c][k],
        //     one 16-row block per wave and pass, four f64 MFMAs each
        if (!(CHOL_SKIP & 2)) {
            for (int rb = wave; 16 * rb < rows; rb += 8) {
                const int row0 = g0 + 16 * rb;
                f64x4 x = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const double av = Praw[(row0 + li) * 17 + 4 * kk + lq];
                    const double bv = D[li * 17 + 4 * kk + lq];
                    x = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, x, 0, 0, 0);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = row0 + lq + 4 * e;
                    T[(int64_t)r * C + j0 + li] = x[e];
                    Pn[li * ldp + r] = x[e];
                }
            }
        }
        __syncthreads();
        // (4) rank-16 update of the blocks still in registers: block(bi, bj) -= P[bi] P[bj]^T.  The active blocks are a
        //     prefix of this wave's slots; they are taken three at a time -- all 24 operand reads first, then the three
        //     independent MFMA chains interleaved -- and a dead block in the last group is simply updated too.
        int nact = 0;
#pragma unroll
        for (int q = 0; q < SLOTS; ++q) nact += (bi_[q] >= 0 && bj_[q] > j) ? 1 : 0;
        if (CHOL_SKIP & 4) nact = 0;
#pragma unroll
        for (int g3 = 0; g3 < SLOTS / 3; ++g3) {
            if (3 * g3 >= nact) break;
            double av[3][4], bv[3][4];
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int q = 3 * g3 + u;
                const int bi = bi_[q] >= 0 ? bi_[q] : 0, bj = bi_[q] >= 0 ? bj_[q] : 0;
                const double* pa = Pn + lq * ldp + 16 * bi + li;
                const double* pb = Pn + lq * ldp + 16 * bj + li;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) { av[u][kk] = -pa[4 * kk * ldp]; bv[u][kk] = pb[4 * kk * ldp]; }
            }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int u = 0; u < 3; ++u)
                    blk[3 * g3 + u] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u][kk], bv[u][kk], blk[3 * g3 + u], 0, 0, 0);
        }
        __syncthreads();
    }
    // strict upper triangle := 0
    for (int64_t e = tid; e < (int64_t)C * C; e += 512) {
        const int i = (int)(e / C), jx = (int)(e % C);
        if (jx > i) T[e] = 0.0;
    }
}

// ---------------------------------------------------------------------------------------------
// K2 for C <= 256, second form (round 2): ONE launch factors T = L L^T and inverts the 16 x 16 diagonal blocks of
// L; a second launch builds W = L^-1 column block by column block.  What changed against cholesky_reg_kernel + the
// seven block-doubling launches below:
//  * The serial part -- factoring AND inverting the 16 x 16 diagonal block -- runs on DPP row broadcasts
//    (gfx90a+: v_fmac_f64_dpp / v_mov_b64_dpp with row_newbcast:k read lane k of the 16-lane row, which IS "row k
//    of the block" when lane = row): one instruction where the v_readlane form needed three, no SGPR round trip, and
//    the inverse's rows are taken as the factor's pivots complete (their fmacs fill the pivots' dependency stalls).
//  * Look-ahead: wave 0 owns no block of the trailing matrix and runs at raised priority.  Once the next panel's
//    blocks (block column j+1) have taken the rank-16 update of step j, wave 0 factors the next diagonal block WHILE
//    waves 1-15 update the rest of the trailing matrix: three LDS-only barriers per step, the serial part off the
//    other waves' critical path.  The roles run different code between the same barriers, so the register allocator
//    never sees wave 0's working set next to the trailing blocks (together they spilled).
//  * 16 waves of 128 registers instead of 8 of 256: 8 blocks per owner wave, three to four waves per SIMD to cover
//    the LDS and f64-MFMA latencies of the update.
// Operand maps of v_mfma_f64_16x16x4_f64 as in gemm_f64_kernel: a = A[i = lane&15][k = lane>>4],
// b = B[k = lane>>4][j = lane&15], D register r of lane l = D[(l>>4) + 4r][l&15].
// ---------------------------------------------------------------------------------------------
template <int K> __device__ __forceinline__ double row_bcast(double v)
{
    double r;      // (s_nop: a VALU write of the source needs two wait states before a DPP read; inline asm gets no hazard pass)
    asm("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v), "n"(K));
    return r;
}
// acc += (lane K of this lane's 16-lane row).src * mul.  NO wait states inside: the caller guarantees that `src` was not
// written by the two instructions in front (dpp_settle below, or a value that has been final for a while)
template <int K> __device__ __forceinline__ void fmac_bcast(double& acc, double src, double mul)
{
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(mul), "n"(K));
}
// acc -= (lane K of this lane's row).src * mul  (the negation rides on the source modifier)
template <int K> __device__ __forceinline__ void fmac_nbcast(double& acc, double src, double mul)
{
    asm volatile("v_fmac_f64_dpp %0, %1, -%2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(mul), "n"(K));
}
// The head of a pivot step of the leaf (round 6, tools/probe/leaf_probe.hip V5): the three nearest columns take the update of column
// `col`, then column K's pivot is broadcast -- p += bcast_K(a1) * 1.0, hp += bcast_K(a1) * 0.5 (p, hp zero on entry) -- through
// v_fmac_f64_dpp: the two updates in between are the DPP read's wait states, and no v_mov_b64_dpp (29 cycles of the dependent chain) is
// needed.  (v_rsq_f64_dpp would read the pivot directly and assembles, but the hardware returns garbage: probe's dpp_check.)
template <int K, int K2, int K3>
__device__ __forceinline__ void leaf_head3(double& a1, double& a2, double& a3, double col, double& p, double& hp, double one, double half)
{
    asm volatile("v_fmac_f64_dpp %0, %5, -%5 row_newbcast:%8 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %1, %5, -%5 row_newbcast:%9 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %2, %5, -%5 row_newbcast:%10 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %3, %0, %6 row_newbcast:%8 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %4, %0, %7 row_newbcast:%8 row_mask:0xf bank_mask:0xf"
                 : "+v"(a1), "+v"(a2), "+v"(a3), "+v"(p), "+v"(hp) : "v"(col), "v"(one), "v"(half), "n"(K), "n"(K2), "n"(K3));
}
template <int K>
__device__ __forceinline__ void leaf_head1(double& a1, double col, double& p, double& hp, double one, double half)
{
    asm volatile("v_fmac_f64_dpp %0, %3, -%3 row_newbcast:%6 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_fmac_f64_dpp %1, %0, %4 row_newbcast:%6 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %2, %0, %5 row_newbcast:%6 row_mask:0xf bank_mask:0xf"
                 : "+v"(a1), "+v"(p), "+v"(hp) : "v"(col), "v"(one), "v"(half), "n"(K));
}
// two wait states between the write of v and every DPP read that follows ("modifies" v, so its readers are ordered behind it)
__device__ __forceinline__ void dpp_settle(double& v) { asm("s_nop 1" : "+v"(v)); }
template <int B, int E, typename F> __device__ __forceinline__ void static_for(F&& f)
{
    if constexpr (B < E) { f(std::integral_constant<int, B>{}); static_for<B + 1, E>(f); }
}

// ---------------------------------------------------------------------------------------------
// W = L^-1 BEHIND the factorisation, in the same launch (round 2).  Row block i of W needs row block i of L and the
// inverse of L_ii -- both final once the factorisation has passed step i -- so the inverse does not have to wait for
// the whole factor: TI_WG more workgroups of the Cholesky launch take the role below and follow the factorisation one
// row block behind it, told by a counter in global memory how many row blocks are complete.  28 us of a second launch
// become a tail of a few microseconds.
//   * X_j = Linv_jj;  X_i = -Linv_ii sum_{k=j}^{i-1} L_ik X_k  (i > j),  W[i][j] = X_i, as in tri_inverse_cols_kernel --
//     but the launch is 1024 threads wide (the factorisation's shape), i.e. 128 VGPRs per thread, and a whole column
//     of X blocks (up to 16 x 8 registers) no longer fits one wave.  So FOUR waves share a column: wave p keeps the
//     blocks X_{j+e} with e = p (mod 4), adds up its own products of a step, the three that do not own the new block leave
//     their partial sums in LDS and the owner (p = (i-j) mod 4) finishes the block.  The chain of dependent MFMAs of a
//     step is a quarter as long (column 0, last step: 16 instead of 60), which also makes the tail behind the last panel short.
//   * A workgroup takes the columns {w, 7-w, 8+w, 15-w} (17 blocks per pair): four workgroups per matrix.
//   * The XCDs' L2s are not coherent with each other: the factorisation stores what this role reads (the solved panels and
//     the inverted diagonal blocks) write-through and this role loads it past its L2 (relaxed agent-scope atomics = sc1
//     accesses, as in wc_sn.hip); the counter is stored after an s_waitcnt vmcnt(0) of every storing wave and a barrier.
//   * Row block i+1 is fetched while step i computes if the counter already allows it; if not, the step waits for it (the
//     common case: the factorisation is the slower of the two except in its last steps).
// ---------------------------------------------------------------------------------------------
constexpr int TI_WG = 4;                         // inverse workgroups per matrix
constexpr int TI_COLS = 4;                       // block columns per workgroup (four waves each)
__device__ __forceinline__ double ld_sc1(const double* p)
{ return __builtin_bit_cast(double, __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)); }
__device__ __forceinline__ void st_sc1(double* p, double v)
{ __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), __builtin_bit_cast(unsigned long long, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__host__ __device__ constexpr size_t ti_role_lds_doubles(int C) { return (size_t)2 * 16 * (C + 2) + 2 * 16 * 17 + 2 * TI_COLS * 4 * 256 + 2; }

// flag == nullptr: L is complete (a launch of its own behind the factorisation); otherwise *flag counts its complete row blocks
__device__ __forceinline__ void tri_inverse_role(const double* __restrict__ L, const double* __restrict__ Linv, double* __restrict__ W,
                                                 int C, int wg, const unsigned* flag, double* sm)
{
    const int ldr = C + 2;
    double* Lrow = sm;                           // [2][16][ldr]  row block i of L, even / odd i
    double* Dv = Lrow + 2 * 16 * ldr;            // [2][16][17]   inverse of L_ii, even / odd i
    double* Part = Dv + 2 * 16 * 17;             // [2][TI_COLS][4][4 x 64] partial sums of a step, even / odd i
    unsigned* avail = reinterpret_cast<unsigned*>(Part + 2 * TI_COLS * 4 * 256);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lq = lane >> 4;
    const int nb = C >> 4;
    const int p = wave & 3, cl = wave >> 2;      // a column's four waves sit on the four SIMDs
    int j = (cl == 0) ? wg : (cl == 1) ? 7 - wg : (cl == 2) ? 8 + wg : 15 - wg;
    const bool has_col = j < nb;
    if (!has_col) j = nb;
    if (wg >= nb) return;                        // (C = 32: two columns)

    // staging of row block i: 16 x C doubles, 8 bytes per thread and load (the sc1 form), a surplus thread repeats the last one
    int off[4], doff[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        int e = tid + 1024 * q; e = e < 16 * C ? e : 16 * C - 1;
        off[q] = e; doff[q] = (e / C) * ldr + (e % C);
    }
    double st[4], dst = 0.0;
    auto fetch = [&](int i) {
        const double* src = L + (int64_t)16 * i * C;
#pragma unroll
        for (int q = 0; q < 4; ++q) st[q] = flag ? ld_sc1(src + off[q]) : src[off[q]];
        if (tid < 256) dst = flag ? ld_sc1(Linv + i * 256 + tid) : Linv[i * 256 + tid];
    };
    auto stash = [&](int i) {
        double* d = Lrow + (i & 1) * 16 * ldr;
#pragma unroll
        for (int q = 0; q < 4; ++q) d[doff[q]] = st[q];
        if (tid < 256) Dv[(i & 1) * (16 * 17) + (tid >> 4) * 17 + (tid & 15)] = dst;
    };
    auto wait_rows = [&](unsigned target) {
        if (tid == 0) {
            unsigned spins = 0;          // (bounded: a lost producer must not hang the device -- W is poisoned below instead)
            while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && ++spins < (1u << 22)) __builtin_amdgcn_s_sleep(2);
            if (spins >= (1u << 22)) avail[1] = 1u;
        }
        __syncthreads();
    };
    auto store = [&](int i, const f64x4& v) {
#pragma unroll
        for (int r = 0; r < 4; ++r) W[(int64_t)(16 * i + lq + 4 * r) * C + 16 * j + li] = v[r];
    };
    if (has_col && p == 0)
        for (int i = 0; i < j; ++i) store(i, f64x4{0.0, 0.0, 0.0, 0.0});      // blocks above the diagonal block: zero

    f64x4 X[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) X[s] = f64x4{0.0, 0.0, 0.0, 0.0};
    int have = -1;
    if (tid == 0) avail[1] = 0u;                  // set if a wait ran out (ordered before its first reader by the barriers below)
#pragma unroll 1
    for (int i = 0; i < nb; ++i) {
        if (have != i) {
            if (flag) wait_rows((unsigned)i + 1);
            fetch(i);
        }
        stash(i);
        if (flag && tid == 0) *avail = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        have = -1;
        if (i + 1 < nb && (!flag || *avail >= (unsigned)i + 2)) { fetch(i + 1); have = i + 1; }
        const int d = i - j;                      // wave-uniform
        const int own = d & 3;
        double* part = Part + (((i & 1) * TI_COLS + cl) * 4) * 256;
        f64x4 S = {0.0, 0.0, 0.0, 0.0}, S1 = {0.0, 0.0, 0.0, 0.0};
        if (has_col && d >= 1) {
            const double* lr = Lrow + (i & 1) * 16 * ldr + li * ldr + lq + 16 * j;
            static_for<0, 4>([&](auto Sx) {
                constexpr int s = decltype(Sx)::value;
                const int e = 4 * s + p;
                if (e < d) {
                    double an[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) an[r] = lr[16 * e + 4 * r];
                    S = __builtin_amdgcn_mfma_f64_16x16x4f64(an[0], X[s][0], S, 0, 0, 0);
                    S1 = __builtin_amdgcn_mfma_f64_16x16x4f64(an[1], X[s][1], S1, 0, 0, 0);
                    S = __builtin_amdgcn_mfma_f64_16x16x4f64(an[2], X[s][2], S, 0, 0, 0);
                    S1 = __builtin_amdgcn_mfma_f64_16x16x4f64(an[3], X[s][3], S1, 0, 0, 0);
                }
            });
            S += S1;
            if (p != own && p < d) {
#pragma unroll
                for (int r = 0; r < 4; ++r) part[p * 256 + r * 64 + lane] = S[r];
            }
        }
        __syncthreads();
        if (has_col && d >= 0 && p == own) {
            const double* dv = Dv + (i & 1) * (16 * 17);
            f64x4 Xi;
            if (d == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) Xi[r] = dv[(lq + 4 * r) * 17 + li];
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (q != p && q < d) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) S[r] += part[q * 256 + r * 64 + lane];
                    }
                f64x4 Xa = {0.0, 0.0, 0.0, 0.0}, Xb = {0.0, 0.0, 0.0, 0.0};
                const double* pi = dv + li * 17 + lq;
                Xa = __builtin_amdgcn_mfma_f64_16x16x4f64(-pi[0], S[0], Xa, 0, 0, 0);
                Xb = __builtin_amdgcn_mfma_f64_16x16x4f64(-pi[4], S[1], Xb, 0, 0, 0);
                Xa = __builtin_amdgcn_mfma_f64_16x16x4f64(-pi[8], S[2], Xa, 0, 0, 0);
                Xb = __builtin_amdgcn_mfma_f64_16x16x4f64(-pi[12], S[3], Xb, 0, 0, 0);
                Xi = Xa + Xb;
            }
            static_for<0, 4>([&](auto Sx) {
                constexpr int s = decltype(Sx)::value;
                if ((d >> 2) == s) X[s] = Xi;
            });
            store(i, Xi);
        }
    }
    // a wait that ran out means the factorising workgroup never got there (it cannot happen while the launch is resident as a
    // whole); fail loudly: NaN on this workgroup's first diagonal entry poisons every table built from W
    __syncthreads();
    if (flag && avail[1] != 0u && tid == 0) {
        W[(int64_t)(16 * wg) * C + 16 * wg] = __builtin_nan("");
        // ... and a sticky error word the host can read: the matrix's row-block counter is rows[16 g], the word behind
        // it is free and zeroed by the same prepare launch -- wc_factor_error_offset / wc_whiten_error_offset say where it is
        atomicOr(const_cast<unsigned*>(flag) + 1, 1u);
    }
}

__global__ __launch_bounds__(1024) void tri_inverse_split_kernel(const double* __restrict__ L, const double* __restrict__ Linv,
                                                                 double* __restrict__ W, int C)
{
    extern __shared__ __attribute__((aligned(16))) double sm[];
    tri_inverse_role(L + (int64_t)blockIdx.y * C * C, Linv + (int64_t)blockIdx.y * C * 16, W + (int64_t)blockIdx.y * C * C, C,
                     (int)blockIdx.x, nullptr, sm);
}

#ifndef CF_OWN
#define CF_OWN 15                   // waves that own trailing blocks and solve the panel.  (12 = all but 4, 8, 12, which share wave 0's
#endif                              // SIMD: measured 74 against 69 us -- the early steps are bound by the CU's f64-MFMA rate and lose a quarter of it)
constexpr int CF_SLOTS = (120 + CF_OWN - 1) / CF_OWN;
// (CF_SPLIT_B counts one s_barrier per step for every wave: the idle-wave form of CF_OWN = 12 keeps two)
#if CF_OWN != 15 && !defined(CF_SPLIT_B)
#define CF_SPLIT_B 0
#endif      // 120 blocks at C = 256 (16 waves x 128 VGPRs: 10 blocks = 80 of them)

#ifndef CF_NEWTON
#define CF_NEWTON 1        // Newton steps on v_rsq_f64 per pivot (measured: L to 1e-13 with one, 1e-7 with none)
#endif
#ifndef CF_SPLIT_B
#define CF_SPLIT_B 1     // round 4: barrier (B) is a counter among the owner waves only; wave 0 solves the next diagonal block's rows itself and never waits for the panel
#endif
#ifndef CF_STAMPS
#define CF_STAMPS 0      // development: s_memtime stamps around every barrier of waves 0, 1 and 5 into the workspace behind Linv
#endif
// LDS hand-off only: every wave's LDS traffic has completed, then the barrier.  (__syncthreads() would also wait for the
// global stores of L in flight, half a microsecond per step for nothing: nobody reads them inside this launch.)
#define CF_BARRIER() do { if (CF_STAMPS && stamp_ok) stamps[nstamp++] = __builtin_amdgcn_s_memtime();              \
                          asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                          \
                          if (CF_STAMPS && stamp_ok) stamps[nstamp++] = __builtin_amdgcn_s_memtime(); } while (0)

// rows != nullptr: the launch carries TI_WG inverse workgroups per matrix behind the `groups` factorising ones (tri_inverse_role);
// rows[16 g] counts the complete row blocks of matrix g (zeroed by factor_prepare_kernel)
__global__ __launch_bounds__(1024) void cholesky_fused_kernel(double* __restrict__ T, double* __restrict__ Linv, int C, int ldp,
                                                              double* __restrict__ Winv, unsigned* __restrict__ rows, int groups)
{
    extern __shared__ __attribute__((aligned(16))) double sm[];
    if ((int)blockIdx.x >= groups) {
        const int idx = (int)blockIdx.x - groups, g = idx / TI_WG;
        tri_inverse_role(T + (int64_t)g * C * C, Linv + (int64_t)g * C * 16, Winv + (int64_t)g * C * C, C, idx % TI_WG, rows + 16 * g, sm);
        return;
    }
    unsigned* rowflag = rows ? rows + 16 * blockIdx.x : nullptr;
    double* Praw = sm;                          // [2][C][17]  the current / next panel as its owners hold it (row-major)
    double* Pn = Praw + 2 * C * 17;             // [C][17]     solved panel, row-major like Praw: the update's MFMA operands.  (Round 6: it was
                                                // [16][C + 2] column-major -- lanes (li, lq) and (li + 2, lq - 1) of an operand read met in one bank,
                                                // every read ran twice: 95 cycles per f64 MFMA against the instruction's own 64,
                                                // tools/probe/mfma_f64_rate.hip)
    double* Dinv = Pn + C * 17;                 // [2][16][17] INVERSE of the factored diagonal block (even / odd steps)
    double* Dpre = Dinv + 2 * 16 * 17;          // [2][16][17] diagonal block (b, b) with every update but the last one, b even / odd
    volatile int* const cntB = reinterpret_cast<volatile int*>(Dpre + 2 * 16 * 17);      // CF_SPLIT_B: owner waves through the panel solve (running count)
    const unsigned cntB_lds = (unsigned)(size_t)((__attribute__((address_space(3))) char*)(Dpre + 2 * 16 * 17));
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lq = lane >> 4;
    T += (int64_t)blockIdx.x * C * C;           // one matrix (statistic group) per workgroup
    Linv += (int64_t)blockIdx.x * C * 16;
    const int nb = C >> 4;
    const int nblk = (nb - 1) * nb / 2;         // blocks (bi >= bj >= 1); block column 0 goes straight to LDS

    if (tid == 0) cntB[0] = 0;
    for (int e = tid; e < C * 16; e += 1024) Praw[(e >> 4) * 17 + (e & 15)] = T[(int64_t)(e >> 4) * C + (e & 15)];
    if (nb > 1)
        for (int e = tid; e < 256; e += 1024) Dpre[16 * 17 + (e >> 4) * 17 + (e & 15)] = T[(int64_t)(16 + (e >> 4)) * C + 16 + (e & 15)];    // block (1,1) as it stands
    __syncthreads();
    const bool stamp_ok = CF_STAMPS && lane == 0 && blockIdx.x == 0;
    unsigned long long* stamps = reinterpret_cast<unsigned long long*>(Linv + 8192) + wave * 128;
    int nstamp = 0;
    (void)stamps; (void)nstamp; (void)stamp_ok;

    // Two barriers per step.  (A): the inverse of L_jj is in LDS and panel j is complete.  (B): panel j is solved.
    // Between (B) and the next (A) wave 0 gives the next diagonal block its last update itself and factors it, while the
    // other waves update the trailing matrix and publish the next panel.
    if (wave == 0) {
        // wave 0: factor a 16 x 16 diagonal block (src: [16][17]) and invert the factor, lane = row / column (all four 16-lane
        // rows of the wave do the same work).  Pivot: v_rsq_f64 + one Newton step.  Row jj of the inverse (lane = its
        // column c): w[jj] = -rd_jj sum_{k<jj} l[jj][k] w[k], taken as soon as row jj of L is final.
        __builtin_amdgcn_s_setprio(3);
        auto factor = [&](int j, const double* src) __attribute__((always_inline)) {
            double a[16], w[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) a[c] = src[li * 17 + c];
            static_for<0, 16>([&](auto J) {
                constexpr int jj = decltype(J)::value;
                const double p = row_bcast<jj>(a[jj]);
                double rd = __builtin_amdgcn_rsq(p);
#if CF_NEWTON >= 1
                rd = rd * (1.5 - 0.5 * p * rd * rd);
#endif
#if CF_NEWTON >= 2
                rd = rd * (1.5 - 0.5 * p * rd * rd);
#endif
                a[jj] = (li == jj) ? p * rd : a[jj] * rd;
                dpp_settle(a[jj]);
                const double nj = -a[jj];
                static_for<jj + 1, 16>([&](auto K) {
                    constexpr int k = decltype(K)::value;
                    fmac_bcast<k>(a[k], a[jj], nj);                    // a[li][k] -= l[k][jj] * l[li][jj]
                });
                double acc = 0.0;
                static_for<0, jj>([&](auto K) {
                    constexpr int k = decltype(K)::value;
                    fmac_bcast<jj>(acc, a[k], w[k]);                   // acc += l[jj][k] * w[k][c]
                });
                w[jj] = (li == jj) ? rd : (li < jj ? -acc * rd : 0.0);
            });
            double* dv = Dinv + (j & 1) * (16 * 17);
            if (rowflag) {
                // row blocks 0..j-1 are complete: the owners' panel stores landed before barrier (A) of step j-1, this wave's
                // own (the inverse of block j-1) were issued a whole step ago
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0) __hip_atomic_store(rowflag, (unsigned)j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (lane < 16) {
#pragma unroll
                for (int c = 0; c < 16; ++c) T[(int64_t)(16 * j + lane) * C + 16 * j + c] = (c <= lane) ? a[c] : 0.0;
#pragma unroll
                for (int i = 0; i < 16; ++i) { dv[i * 17 + lane] = w[i]; st_sc1(Linv + j * 256 + i * 16 + lane, w[i]); }   // [i][c], zero above the diagonal
            }
        };
        factor(0, Praw);
#pragma unroll 1
        for (int j = 0; j < nb; ++j) {
            CF_BARRIER();                                          // (A)
            if (C - 16 * (j + 1) <= 0) {
                if (rowflag) {                                     // the last block's inverse has landed: L is complete
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if (lane == 0) __hip_atomic_store(rowflag, (unsigned)nb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                break;
            }
            // the last update of block (j+1, j+1): D -= X X^T with X = rows 16(j+1).. of the solved panel; through LDS into
            // the lane = row layout of the factorisation
            double* dp = Dpre + ((j + 1) & 1) * (16 * 17);
            f64x4 d4, d5 = {0.0, 0.0, 0.0, 0.0};
#if CF_SPLIT_B
            // Those 16 rows of X are solved HERE as well (the owner of row block 0 publishes them for everybody else): X^T = Linv P^T with
            // the operand roles of the owners' product swapped, same pairs of k-groups in the same order, so register kk of the result is
            // X[li][lq + 4 kk] -- exactly the A (and B) operand of the update below.  Wave 0 then needs nothing the other waves produce
            // between (A) and the next (A): it does not take part in (B), which is a counter among the owners.  (Steps 5-15 are bound by
            // this wave's chain, stamps of round 4: 1 300 - 2 000 cycles of waiting for the slowest solver per step, 10 % of the kernel.)
            {
                const double* pr = Praw + (j & 1) * (C * 17) + (16 * (j + 1) + li) * 17 + lq;
                const double* dv0 = Dinv + (j & 1) * (16 * 17) + li * 17 + lq;
                f64x4 xa = {0.0, 0.0, 0.0, 0.0}, xb = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int r = 0; r < 4; ++r) d4[r] = dp[(lq + 4 * r) * 17 + li];
                xa = __builtin_amdgcn_mfma_f64_16x16x4f64(dv0[0], pr[0], xa, 0, 0, 0);
                xb = __builtin_amdgcn_mfma_f64_16x16x4f64(dv0[4], pr[4], xb, 0, 0, 0);
                xa = __builtin_amdgcn_mfma_f64_16x16x4f64(dv0[8], pr[8], xa, 0, 0, 0);
                xb = __builtin_amdgcn_mfma_f64_16x16x4f64(dv0[12], pr[12], xb, 0, 0, 0);
                xa += xb;
                d4 = __builtin_amdgcn_mfma_f64_16x16x4f64(-xa[0], xa[0], d4, 0, 0, 0);
                d5 = __builtin_amdgcn_mfma_f64_16x16x4f64(-xa[1], xa[1], d5, 0, 0, 0);
                d4 = __builtin_amdgcn_mfma_f64_16x16x4f64(-xa[2], xa[2], d4, 0, 0, 0);
                d5 = __builtin_amdgcn_mfma_f64_16x16x4f64(-xa[3], xa[3], d5, 0, 0, 0);
            }
#else
            CF_BARRIER();                                          // (B) the others have solved panel j
            const double* px = Pn + (16 * (j + 1) + li) * 17 + lq;
#pragma unroll
            for (int r = 0; r < 4; ++r) d4[r] = dp[(lq + 4 * r) * 17 + li];
            d4 = __builtin_amdgcn_mfma_f64_16x16x4f64(-px[0], px[0], d4, 0, 0, 0);
            d5 = __builtin_amdgcn_mfma_f64_16x16x4f64(-px[4], px[4], d5, 0, 0, 0);
            d4 = __builtin_amdgcn_mfma_f64_16x16x4f64(-px[8], px[8], d4, 0, 0, 0);
            d5 = __builtin_amdgcn_mfma_f64_16x16x4f64(-px[12], px[12], d5, 0, 0, 0);
#endif
            d4 += d5;
#pragma unroll
            for (int r = 0; r < 4; ++r) dp[(lq + 4 * r) * 17 + li] = d4[r];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // own LDS writes before own reads (one wave: no barrier needed)
            factor(j + 1, dp);
        }
    } else if (CF_OWN == 15 || (wave & 3) != 0) {
        const int ow = CF_OWN == 15 ? wave - 1 : wave - 1 - (wave >> 2);        // owner index 0..CF_OWN-1
        // the owner waves hold the trailing blocks: ranked by block column, LAST column first, the diagonal block first within a
        // column, dealt round-robin -- the blocks still active at a step are a prefix of every wave's slots.
        // One packed SGPR per slot (bi << 8 | bj), integer arithmetic only.
        int bc_[CF_SLOTS];
        f64x4 blk[CF_SLOTS];
#pragma unroll
        for (int q = 0; q < CF_SLOTS; ++q) {
            const int r = ow + CF_OWN * q;
            bc_[q] = 0;
            blk[q] = f64x4{0.0, 0.0, 0.0, 0.0};
            if (r < nblk) {
                int m = 0;
                while ((m + 1) * (m + 2) / 2 <= r) ++m;             // column nb-1-m holds m+1 blocks
                const int bj = nb - 1 - m, bi = bj + (r - m * (m + 1) / 2);
                bc_[q] = (bi << 8) | bj;
#pragma unroll
                for (int e = 0; e < 4; ++e) blk[q][e] = T[(int64_t)(16 * bi + lq + 4 * e) * C + 16 * bj + li];
            }
        }
        int li_t = li, lq_t = lq;       // per-step opaque copies of the lane constants (see the loop)
        // the rank-16 update of one block: block(bi, bj) -= P[bi] P[bj]^T, operands from the solved panel in LDS
        auto update = [&](auto Q) __attribute__((always_inline)) {
            constexpr int q = decltype(Q)::value;
            const int bc = bc_[q];
            const double* pa = Pn + (16 * (bc >> 8) + li_t) * 17 + lq_t;
            const double* pb = Pn + (16 * (bc & 255) + li_t) * 17 + lq_t;
            double av[4], bv[4];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) { av[kk] = -pa[4 * kk]; bv[kk] = pb[4 * kk]; }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) blk[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[kk], bv[kk], blk[q], 0, 0, 0);
        };
        int cur = 0;
#pragma unroll 1
        for (int j = 0; j < nb; ++j) {
            // (the panel stores of step j-1, a trailing update ago, have landed: what wave 0 counts as complete row blocks)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            CF_BARRIER();                                          // (A) the inverse of L_jj and panel j are in LDS
            const int j0 = 16 * j, g0 = j0 + 16, rows = C - g0;
            if (rows <= 0) break;
            // every LDS address below is loop-invariant per slot; left alone, hipcc hoists all of them out of the step loop
            // (~40 VGPRs) and spills the blocks instead
            asm volatile("" : "+v"(li_t), "+v"(lq_t));
            const double* pan = Praw + cur * (C * 17);
            double* nxt = Praw + (cur ^ 1) * (C * 17);
            // (S2) panel solve X = P L_jj^-T as a product with the inverted diagonal block: X[r][c] = sum_k P[r][k] Linv[c][k],
            //      one 16-row block per wave, four f64 MFMAs.  (Row block 0 = the rows of the next diagonal block: from
            //      panel j's own rows, which hold block (j+1, j).)
            {
                const double* dv = Dinv + (j & 1) * (16 * 17);
                for (int rb = ow; 16 * rb < rows; rb += CF_OWN) {
                    const int row0 = g0 + 16 * rb;
                    f64x4 x = {0.0, 0.0, 0.0, 0.0}, x1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int kk = 0; kk < 4; kk += 2) {
                        x = __builtin_amdgcn_mfma_f64_16x16x4f64(pan[(row0 + li_t) * 17 + 4 * kk + lq_t], dv[li_t * 17 + 4 * kk + lq_t], x, 0, 0, 0);
                        x1 = __builtin_amdgcn_mfma_f64_16x16x4f64(pan[(row0 + li_t) * 17 + 4 * kk + 4 + lq_t], dv[li_t * 17 + 4 * kk + 4 + lq_t], x1, 0, 0, 0);
                    }
                    x += x1;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int r = row0 + lq_t + 4 * e;
                        st_sc1(T + (int64_t)r * C + j0 + li_t, x[e]);      // write-through: the inverse role reads it in this launch
                        Pn[r * 17 + li_t] = x[e];
                    }
                }
            }
#if CF_SPLIT_B
            {   // (B) among the owners: every owner's rows of the solved panel are in LDS
                if (CF_STAMPS && stamp_ok) stamps[nstamp++] = __builtin_amdgcn_s_memtime();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane == 0) { const unsigned one = 1u; asm volatile("ds_add_u32 %0, %1" :: "v"(cntB_lds), "v"(one) : "memory"); }
                const int target = CF_OWN * (j + 1);
                for (;;) {
                    int v;
                    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(cntB_lds) : "memory");
                    if (__builtin_amdgcn_readfirstlane(v) >= target) break;
                }
                if (CF_STAMPS && stamp_ok) stamps[nstamp++] = __builtin_amdgcn_s_memtime();
            }
#else
            CF_BARRIER();                                          // (B) panel j is solved
#endif
            // (S4) this wave's active slots are [0, n34): ranks below R4 lie right of the next panel, the next nb-j-1 ranks ARE
            // the next panel (column j+1; rank = owner index + CF_OWN * slot).  The next panel's blocks go first and leave for the other
            // panel buffer -- except the diagonal one, which wave 0 updates and factors itself.  The diagonal block after
            // that, (j+2, j+2), leaves a copy behind once it has this step's update: wave 0's input at the next step.
            const int R4 = (nb - j - 2) * (nb - j - 1) / 2, w1 = ow;
            const int n4 = R4 > w1 ? (R4 - w1 + CF_OWN - 1) / CF_OWN : 0;
            const int n34 = R4 + nb - j - 1 > w1 ? (R4 + nb - j - 1 - w1 + CF_OWN - 1) / CF_OWN : 0;
            const int R5 = (nb - j - 3) * (nb - j - 2) / 2;        // rank of block (j+2, j+2), the first of its column
            const int qd = (j + 2 < nb && R5 % CF_OWN == w1) ? R5 / CF_OWN : -1;
            static_for<0, CF_SLOTS>([&](auto Q) {                  // next panel first (the highest active slots)
                constexpr int q = CF_SLOTS - 1 - decltype(Q)::value;
                if (q >= n4 && q < n34 && (bc_[q] >> 8) != (bc_[q] & 255)) {
                    update(std::integral_constant<int, q>{});
#pragma unroll
                    for (int e = 0; e < 4; ++e) nxt[(16 * (bc_[q] >> 8) + lq_t + 4 * e) * 17 + li_t] = blk[q][e];
                }
            });
            static_for<0, CF_SLOTS>([&](auto Q) {
                constexpr int q = decltype(Q)::value;
                if (q < n4) {
                    update(std::integral_constant<int, q>{});
                    if (q == qd) {
                        double* dp = Dpre + (j & 1) * (16 * 17);
#pragma unroll
                        for (int e = 0; e < 4; ++e) dp[(lq_t + 4 * e) * 17 + li_t] = blk[q][e];
                    }
                }
            });
            cur ^= 1;
        }
    }
    else {
        // waves 4, 8, 12: on wave 0's SIMD; they only keep the barrier count
#pragma unroll 1
        for (int j = 0; j < nb; ++j) {
            CF_BARRIER();
            if (C - 16 * (j + 1) <= 0) break;
            CF_BARRIER();
        }
    }
    if (CF_STAMPS && stamp_ok) { stamps[nstamp++] = __builtin_amdgcn_s_memtime(); stamps[127] = nstamp; }
    // L's upper triangle: the blocks above the diagonal were zero on entry (factor_prepare_kernel), the diagonal blocks
    // were written whole
}

// =================================================================================================================================
// K2 as a RELAY of two factorising workgroups (round 6; VERDICT r5 item 1: "take the small-matrix stage off one CU").
//
// What the stamps of cholesky_fused_kernel said at C = 256 (profiles/r6_k2_timeline.txt): the kernel is the sum over its 16 steps of
// max(wave 0's chain, the owners' panel solve + trailing update).  Wave 0 needs 6 300 - 7 700 cycles per step (3 100 - 5 700 of them the
// 16 x 16 leaf, 1 200 - 2 000 the 48 stores behind it); the owners need 14 900 / 13 700 / 13 100 / 12 000 / 10 100 ... cycles in steps
// 0 .. 4 for work whose f64-MFMA floor on one CU is 8 600 / 7 600 / 6 700 / 5 800 / 4 900: the first eight steps are bound by ONE CU's
// matrix pipe, the last eight by wave 0.  Spreading every step's update over several CUs puts a cross-CU hand-off (~1 us each way) into
// every step's critical path -- longer than the step; that is what killed the forms of DESIGN 4.7 / 8.  The split here is by COLUMNS
// and by TIME instead:
//   F1 owns block columns 0 .. split-1 only (C = 256: split = 6, 65 trailing blocks instead of 120) and runs the look-ahead algorithm on
//      them unchanged, steps 0 .. split-1 -- every row of every panel is solved by F1 (the panel is its own column) and leaves for global
//      memory write-through, as it always did for the inverse role.  Then it is done.
//   F2 holds the trailing blocks of the columns >= split (55 blocks) from the start.  While F1 factors, F2 is PASSIVE: it follows a
//      counter of complete panels, fetches rows >= 16 split of panel j (20 KB) into LDS and applies it to its blocks -- a step or so
//      behind F1, which never waits for it.  After panel split-1 it has the trailing matrix of step split in its registers, the next
//      panel in LDS and the next diagonal block factored: it IS the factoriser from there on, steps split .. nb-1.
// One hand-off per launch, and it moves 20 KB (the last panel), not the trailing matrix.  Three forms that moved the trailing blocks
// instead were built and measured first (four helper workgroups, F taking the blocks back at a takeover step; C = 256, K2 per call):
// helpers through LDS with a boundary wait 67.6 us; helpers fed by 8-byte sc1 loads 70.7 (one fabric transaction per LANE: their waves
// ran 12 000 - 15 000 cycles apart); 16-byte sc1 accesses + the last two panels applied by F itself 64.5 -- there the takeover alone
// was 22 000 cycles, 12 000 of them F fetching 110 KB that write-through stores had dropped from every L2 (profiles/r6_k2_timeline.txt).
// Each factoriser's owner waves hold 5 blocks instead of 8 (40 VGPRs: the register file is no longer at its limit, which is what let
// the publisher below in without spills).
// Also new on the factorisers' side: (a) wave 0 leaves the factored block and its inverse in LDS only, in 128-bit accesses; an owner wave
// that solves nothing copies them to global memory behind barrier (A) (publish_diag) and counts the complete row blocks for the inverse
// role; (b) the panel solve's row blocks go to the twelve owners that do not share wave 0's SIMD; (c) no workgroup barrier in front of
// the first step: wave 0 takes its block straight from global memory while the others stage the panel.
// Hand-off protocol (MI355X_MICROARCH.md, "Valid forms": all payload bytes sc1 stores / sc1 loads, every storing wave's vmcnt(0), a
// workgroup barrier, ONE lane's sc1 flag store; the consumer's polling wave joins a workgroup barrier, the others load behind it).
// Every workgroup of the launch must be resident at once, as for tri_inverse_role: groups * (2 + TI_WG) <= 48.
// sync words of matrix g (zeroed by the prepare launch): sync[16 g] complete row blocks, +1 error bits, +2 complete panels.
// =================================================================================================================================
typedef double f64x2 __attribute__((ext_vector_type(2)));
constexpr int CP_SLOTS = 5;                 // trailing blocks per owner wave of F
constexpr int CP_MAXB = 15 * CP_SLOTS;      // ... so a phase holds at most 75 blocks
__host__ __device__ constexpr int cp_colblocks(int c, int c1, int nb)          // blocks (bi >= bj) of the block columns c .. c1-1
{ return c1 > c ? (c1 - c) * (2 * nb - c - c1 + 1) / 2 : 0; }
// the block column where the helpers' part begins (nb: no helpers, one phase)
__host__ __device__ constexpr size_t cp_factor_lds_doubles(int C) { return (size_t)3 * C * 17 + 6 * 16 * 18 + 2; }       // (+ two counter words)

__device__ __forceinline__ bool cp_wait_word(const unsigned* w, unsigned target)
{
    unsigned spins = 0;              // bounded: a lost producer must not hang the device
    while (__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && ++spins < (1u << 22)) __builtin_amdgcn_s_sleep(1);
    return spins < (1u << 22);
}

// ---- 16-byte write-through stores / L1-bypassing loads -------------------------------------------------------------------------
// The 8-byte agent-scope atomics ld_sc1 / st_sc1 are ONE FABRIC TRANSACTION PER LANE (MI355X_MICROARCH.md: "scalar sc1 stores are one
// fabric write each: dwordx2 2.7x ... the dwordx4 time per byte"): a 16 x 16 block costs a wave 256 of them, a helper workgroup's panel
// round 8 192 -- the stamps of the first helper version showed its sixteen waves 12 000 - 15 000 cycles apart behind that queue, and the
// takeover waiting for the last of them.  A dwordx4 access with the sc1 bit coalesces like a plain one; as raw-buffer builtins (not inline
// asm) the compiler keeps the vmcnt bookkeeping -- an asm load's destination is a register the allocator may copy or spill before it lands.
typedef unsigned int cp_u32x4 __attribute__((ext_vector_type(4)));
// a raw buffer over one matrix (element offsets below are in doubles); aux 16 = the sc1 bit on gfx940+
__device__ __forceinline__ __amdgpu_buffer_rsrc_t cp_rsrc(const double* p)
{ return __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(p), 0, 0x7fffffff, 0x00020000); }
__device__ __forceinline__ void st_sc1_x2(__amdgpu_buffer_rsrc_t r, int elem, f64x2 v)
{ __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(cp_u32x4, v), r, elem * 8, 0, 16); }
__device__ __forceinline__ f64x2 ld_sc1_x2(__amdgpu_buffer_rsrc_t r, int elem)
{ return __builtin_bit_cast(f64x2, __builtin_amdgcn_raw_buffer_load_b128(r, elem * 8, 0, 16)); }
// the value of lane ^ 1
__device__ __forceinline__ double cp_swap1(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0xB1, 0xf, 0xf, false);      // quad_perm [1, 0, 3, 2]
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0xB1, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
// A 16 x 16 block in the f64 MFMA accumulator layout (register e of lane (li, lq) = element [lq + 4 e][li]) to / from the matrix behind
// `r` at element offset `b` (leading dimension ld) in 16-byte pieces: neighbouring lanes trade halves, an even lane moves rows lq, lq + 4
// (two columns each), an odd lane rows lq + 8, lq + 12.
__device__ __forceinline__ void cp_store_block_sc1(__amdgpu_buffer_rsrc_t r, int b, int ld, const f64x4& x, int li, int lq)
{
    const bool odd = li & 1;
    const double r0 = cp_swap1(odd ? x[0] : x[2]), r1 = cp_swap1(odd ? x[1] : x[3]);
    const int p = b + (lq + (odd ? 8 : 0)) * ld + (li & ~1);
    st_sc1_x2(r, p, odd ? f64x2{r0, x[2]} : f64x2{x[0], r0});
    st_sc1_x2(r, p + 4 * ld, odd ? f64x2{r1, x[3]} : f64x2{x[1], r1});
}
__device__ __forceinline__ f64x4 cp_load_block_sc1(__amdgpu_buffer_rsrc_t r, int b, int ld, int li, int lq)
{
    const bool odd = li & 1;
    const int p = b + (lq + (odd ? 8 : 0)) * ld + (li & ~1);
    const f64x2 v0 = ld_sc1_x2(r, p), v1 = ld_sc1_x2(r, p + 4 * ld);
    const double r0 = cp_swap1(odd ? v0[0] : v0[1]), r1 = cp_swap1(odd ? v1[0] : v1[1]);
    return odd ? f64x4{r0, r1, v0[1], v1[1]} : f64x4{v0[0], v1[0], r0, r1};
}

#ifndef CP_SKIP
#define CP_SKIP 0        // development ablations (WRONG results): 1 no update MFMAs, 2 no trailing update at all, 4 no panel-solve MFMAs, 8 no pivots in the leaf, 16 no panel stores to global memory
#endif
#define CP_BARRIER() do { if (CF_STAMPS && stamp_ok && nstamp < 100) stamps[nstamp++] = __builtin_amdgcn_s_memtime();              \
                          asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                          \
                          if (CF_STAMPS && stamp_ok && nstamp < 100) stamps[nstamp++] = __builtin_amdgcn_s_memtime(); } while (0)

// One factoriser of the relay: active in the steps [jbeg, jend), where it owns the trailing blocks of the block columns [max(jbeg, 1), jend)
// (rows from the diagonal down to the matrix' last).  Before that it is PASSIVE: it applies the panels 0 .. jbeg-1 to those blocks as the
// factorisers before it publish them (fetch rows >= 16 jbeg of the solved panel, update).  jbeg = 0: the first one; jend = nb: the last.
__device__ __forceinline__ void cp_factor_role(double* __restrict__ T, double* __restrict__ Linv, int C, unsigned* sync, int jbeg, int jend,
                                               double* sm, unsigned long long* stamp_base)
{
    double* Praw = sm;                          // [2][C][17]  the current / next panel as its owners hold it (row-major)
    double* Pn = Praw + 2 * C * 17;             // [C][17]     solved panel: the update's MFMA operands
    // the three 16 x 16 hand-off blocks have rows of 18 doubles: 16-byte aligned, so wave 0 moves them with 128-bit LDS accesses
    // (16 writes + 8 reads per step instead of 32 + 16: ~600 cycles of its chain), conflict-free for the MFMA operand reads
    double* DinvT = Pn + C * 17;                // [2][16][18] INVERSE of the factored diagonal block, TRANSPOSED: [c][i] = Linv[i][c] (even / odd steps)
    double* Dpre = DinvT + 2 * 16 * 18;         // [2][16][18] diagonal block (b, b) with every update but the last one, b even / odd
    double* Ldg = Dpre + 2 * 16 * 18;           // [2][16][18] the factored diagonal block on its way to global memory (even / odd steps)
    const unsigned cntB_lds = (unsigned)(size_t)((__attribute__((address_space(3))) char*)(Ldg + 2 * 16 * 18));
    volatile int* const cntB = reinterpret_cast<volatile int*>(Ldg + 2 * 16 * 18);      // owner waves through the panel solve (running count)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lq = lane >> 4;
    const int nb = C >> 4;
    const __amdgpu_buffer_rsrc_t rT = cp_rsrc(T), rLinv = cp_rsrc(Linv);
    const bool stamp_ok = CF_STAMPS && lane == 0 && stamp_base != nullptr;
    unsigned long long* stamps = stamp_base + wave * 128;
    int nstamp = 0;
    (void)stamps; (void)nstamp; (void)stamp_ok;
    if (tid == 0) cntB[0] = 0;
    // a passive step's fetch: rows >= 16 jbeg of the solved panel j -> Pn, 16 bytes per thread and load (all 16 waves)
    auto stage_panel = [&](int j) __attribute__((always_inline)) {
        const int row0 = 16 * jbeg, n16 = (nb - jbeg) * 128;
        const int e0 = tid, e1 = tid + 1024;
        f64x2 v0 = {0.0, 0.0}, v1 = {0.0, 0.0};
        if (e0 < n16) v0 = ld_sc1_x2(rT, (row0 + (e0 >> 3)) * C + 16 * j + 2 * (e0 & 7));
        if (e1 < n16) v1 = ld_sc1_x2(rT, (row0 + (e1 >> 3)) * C + 16 * j + 2 * (e1 & 7));
        if (e0 < n16) { double* d = Pn + (row0 + (e0 >> 3)) * 17 + 2 * (e0 & 7); d[0] = v0[0]; d[1] = v0[1]; }
        if (e1 < n16) { double* d = Pn + (row0 + (e1 >> 3)) * 17 + 2 * (e1 & 7); d[0] = v1[0]; d[1] = v1[1]; }
    };

    if (wave == 0) {
        // ---- wave 0: the diagonal blocks.  Factor + invert block (j, j) [lane = row / column, all four 16-lane rows do the same work],
        //      results into LDS; the look-ahead between two barriers (A) as in cholesky_fused_kernel.
        __builtin_amdgcn_s_setprio(3);
        auto factor = [&](int j, const double* src) __attribute__((always_inline)) {
            double a[16], w[16];
            if (src) {
#pragma unroll
                for (int c = 0; c < 16; c += 2) {
                    const f64x2 v = *reinterpret_cast<const f64x2*>(src + li * 18 + c);
                    a[c] = v[0]; a[c + 1] = v[1];
                }
            } else {                            // block (0, 0): straight from global memory (the launch before wrote it), nothing is in LDS yet
#pragma unroll
                for (int c = 0; c < 16; ++c) a[c] = T[(int64_t)(16 * j + li) * C + 16 * j + c];
            }
            if (CF_STAMPS && stamp_ok) { if (src) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                                         stamps[40 + 5 * (j - jbeg) + 1] = __builtin_amdgcn_s_memtime(); }
            // Round 6 (probe V5: 3 720 -> 3 070 cycles alone on its SIMD): the pivot column is scaled by the raw rsq first and by the Newton
            // factor after (a * rn runs beside the Newton step, not behind it); lane jj's a[jj] IS the pivot, no select; the negation rides
            // on the FMA's source modifier; the next pivot's broadcast is issued behind the first three updates of this one (leaf_head3).
            double half = 0.5, one = 1.0;
            asm volatile("" : "+v"(half), "+v"(one));
            double p = 0.0, hp = 0.0;
            asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %2, %3 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %1, %2, %4 row_newbcast:0 row_mask:0xf bank_mask:0xf" : "+v"(p), "+v"(hp) : "v"(a[0]), "v"(one), "v"(half));
            static_for<0, 16>([&](auto J) {
                constexpr int jj = decltype(J)::value;
                if (CP_SKIP & 8) { w[jj] = a[jj]; return; }
                const double rn = __builtin_amdgcn_rsq(p);
                const double t = hp * rn;
                const double ua = a[jj] * rn;
                const double c = __builtin_fma(-rn, t, 1.5);          // one Newton step: rd = rn (1.5 - 0.5 p rn^2)
                a[jj] = ua * c;
                const double rd = rn * c;
                dpp_settle(a[jj]);
                p = 0.0; hp = 0.0;
                if constexpr (jj + 3 < 16) leaf_head3<jj + 1, jj + 2, jj + 3>(a[jj + 1], a[jj + 2], a[jj + 3], a[jj], p, hp, one, half);
                else if constexpr (jj + 1 < 16) {
                    leaf_head1<jj + 1>(a[jj + 1], a[jj], p, hp, one, half);
                    if constexpr (jj + 2 < 16) fmac_nbcast<jj + 2>(a[jj + 2], a[jj], a[jj]);
                }
                static_for<jj + 4, 16>([&](auto K) {
                    constexpr int k = decltype(K)::value;
                    fmac_nbcast<k>(a[k], a[jj], a[jj]);                // a[li][k] -= l[k][jj] * l[li][jj]
                });
                double acc = (li == jj) ? -1.0 : 0.0;                  // lane jj: w = rd; lanes above jj: the sum is zero by itself
                static_for<0, jj>([&](auto K) {
                    constexpr int k = decltype(K)::value;
                    fmac_bcast<jj>(acc, a[k], w[k]);                   // acc += l[jj][k] * w[k][c]
                });
                w[jj] = -acc * rd;
            });
            if (CF_STAMPS && stamp_ok) { asm volatile("" :: "v"(a[15]), "v"(w[15])); stamps[40 + 5 * (j - jbeg) + 2] = __builtin_amdgcn_s_memtime(); }
            if (lane < 16) {
                double* lg = Ldg + (j & 1) * (16 * 18);
                double* dv = DinvT + (j & 1) * (16 * 18);
#pragma unroll
                for (int c = 0; c < 16; c += 2) *reinterpret_cast<f64x2*>(lg + lane * 18 + c) = f64x2{a[c], a[c + 1]};     // (publish_diag zeroes what lies above the diagonal)
#pragma unroll
                for (int i = 0; i < 16; i += 2) *reinterpret_cast<f64x2*>(dv + lane * 18 + i) = f64x2{w[i], w[i + 1]};     // [c][i], zero where i < c
            }
            if (CF_STAMPS && stamp_ok) stamps[40 + 5 * (j - jbeg) + 3] = __builtin_amdgcn_s_memtime();
        };
        if (jbeg == 0) factor(0, nullptr);
        else {
#pragma unroll 1
            for (int j = 0; j < jbeg; ++j) {
                CP_BARRIER();                                          // (P1) panel j is complete in global memory
                stage_panel(j);
                CP_BARRIER();                                          // (P2) ... and its rows >= 16 split in Pn
                if (j + 1 == jbeg) {
                    // the look-ahead into the first active step: block (jbeg, jbeg) with every update but panel j's waits in Dpre (the
                    // passive step before left it there), its 16 rows of panel j are SOLVED already (the work of the factoriser before)
                    double* dp = Dpre + (jbeg & 1) * (16 * 18);
                    const double* px = Pn + (16 * jbeg + li) * 17 + lq;
                    f64x4 d4, d5 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int r = 0; r < 4; ++r) d4[r] = dp[(lq + 4 * r) * 18 + li];
                    d4 = __builtin_amdgcn_mfma_f64_16x16x4f64(-px[0], px[0], d4, 0, 0, 0);
                    d5 = __builtin_amdgcn_mfma_f64_16x16x4f64(-px[4], px[4], d5, 0, 0, 0);
                    d4 = __builtin_amdgcn_mfma_f64_16x16x4f64(-px[8], px[8], d4, 0, 0, 0);
                    d5 = __builtin_amdgcn_mfma_f64_16x16x4f64(-px[12], px[12], d5, 0, 0, 0);
                    d4 += d5;
#pragma unroll
                    for (int r = 0; r < 4; ++r) dp[(lq + 4 * r) * 18 + li] = d4[r];
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    factor(jbeg, dp);
                    // (Measured and dropped: the owner waves on this SIMD holding their share of this step's update back until the leaf is
                    // done -- f64 MFMAs beside it stretch the leaf from 2 600 to ~6 000 cycles here -- made K2 2 us SLOWER: their blocks then
                    // finish last.)
                }
            }
        }
#pragma unroll 1
        for (int j = jbeg; j < jend; ++j) {
            CP_BARRIER();                                          // (A)
            if (j + 1 >= jend) break;                              // this factoriser's last panel: nothing to look ahead to
            // the next diagonal block: its 16 rows of panel j solved here, its last update, through LDS into the lane = row layout
            double* dp = Dpre + ((j + 1) & 1) * (16 * 18);
            f64x4 d4, d5 = {0.0, 0.0, 0.0, 0.0};
            {
                const double* pr = Praw + (j & 1) * (C * 17) + (16 * (j + 1) + li) * 17 + lq;
                const double* dv0 = DinvT + (j & 1) * (16 * 18) + lq * 18 + li;        // Linv[li][lq + 4 kk] = DinvT[lq + 4 kk][li]
                f64x4 xa = {0.0, 0.0, 0.0, 0.0}, xb = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int r = 0; r < 4; ++r) d4[r] = dp[(lq + 4 * r) * 18 + li];
                xa = __builtin_amdgcn_mfma_f64_16x16x4f64(dv0[0], pr[0], xa, 0, 0, 0);
                xb = __builtin_amdgcn_mfma_f64_16x16x4f64(dv0[4 * 18], pr[4], xb, 0, 0, 0);
                xa = __builtin_amdgcn_mfma_f64_16x16x4f64(dv0[8 * 18], pr[8], xa, 0, 0, 0);
                xb = __builtin_amdgcn_mfma_f64_16x16x4f64(dv0[12 * 18], pr[12], xb, 0, 0, 0);
                xa += xb;
                d4 = __builtin_amdgcn_mfma_f64_16x16x4f64(-xa[0], xa[0], d4, 0, 0, 0);
                d5 = __builtin_amdgcn_mfma_f64_16x16x4f64(-xa[1], xa[1], d5, 0, 0, 0);
                d4 = __builtin_amdgcn_mfma_f64_16x16x4f64(-xa[2], xa[2], d4, 0, 0, 0);
                d5 = __builtin_amdgcn_mfma_f64_16x16x4f64(-xa[3], xa[3], d5, 0, 0, 0);
            }
            d4 += d5;
#pragma unroll
            for (int r = 0; r < 4; ++r) dp[(lq + 4 * r) * 18 + li] = d4[r];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // own LDS writes before own reads (one wave: no barrier needed)
            if (CF_STAMPS && stamp_ok) stamps[40 + 5 * (j + 1 - jbeg)] = __builtin_amdgcn_s_memtime();
            factor(j + 1, dp);
        }
    } else {
        // ---- waves 1 .. 15: the panel solve and the trailing update of this factoriser's block columns
        const int ow = wave - 1;
        const bool publisher = wave == 4;                 // (on wave 0's SIMD: it has no row block to solve)
        const bool solver = (wave & 3) != 0;               // not on wave 0's SIMD
        const int sv = ow - (wave >> 2);                   // 0 .. 11 among the solvers
        const int cb = jbeg ? jbeg : 1, ce = jend;         // this factoriser's trailing block columns [cb, ce)
        int bc_[CP_SLOTS];
        f64x4 blk[CP_SLOTS];
        int li_t = li, lq_t = lq;       // per-step opaque copies of the lane constants (nothing built from them may be hoisted out of the step loop)
        // (S4) the trailing update of step j: block (bi, bj) -= P[bi] P[bj]^T, operands from the solved panel in LDS.  This wave's active
        // slots are [0, n34): ranks below R4 lie right of the next panel, the next ranks ARE the next panel (column j+1).  The slots run from
        // the highest down, so the next panel's blocks go first; they leave for the other panel buffer -- except the diagonal one, which
        // wave 0 updates and factors itself.  The diagonal block after that, (j+2, j+2), leaves a copy behind once it has this step's
        // update: wave 0's input at the next step.  (Columns left of cb do not exist here: a passive step's ranks.)
        // Round 6: software-pipelined.  A block used to be eight LDS reads, a wait, four dependent MFMAs, and (next panel) four LDS writes
        // that wait for the last MFMA -- with every wave of a SIMD in the same phase of that at the same time its matrix pipe was idle half
        // of the update (stamps: the youngest wave of a SIMD finished 2 700 cycles behind the oldest).  Now the next slot's operands are on
        // their way while this slot's MFMAs run, and a slot's LDS writes follow the NEXT slot's MFMAs.
        auto trailing = [&](int j) __attribute__((always_inline)) {
            const int c3 = j + 1 > cb ? j + 1 : cb, c4 = j + 2 > cb ? j + 2 : cb, c5 = j + 3 > cb ? j + 3 : cb;
            const int R4 = cp_colblocks(c4, ce, nb), R3 = cp_colblocks(c3, ce, nb);
            const int n4 = R4 > ow ? (R4 - ow + 14) / 15 : 0;
            const int n34 = R3 > ow ? (R3 - ow + 14) / 15 : 0;
            const int R5 = cp_colblocks(c5, ce, nb);               // rank of block (j+2, j+2), the first of its column
            const int qd = (j + 2 >= cb && j + 2 < ce && R5 % 15 == ow) ? R5 / 15 : -1;
            double* nxt = Praw + ((j + 1) & 1) * (C * 17);
            double* dpq = Dpre + (j & 1) * (16 * 18);
            double av[2][4], bv[2][4];
            auto fetch = [&](auto Q) __attribute__((always_inline)) {
                constexpr int q = decltype(Q)::value;
                const int bc = bc_[q];
                const double* pa = Pn + (16 * (bc >> 8) + li_t) * 17 + lq_t;
                const double* pb = Pn + (16 * (bc & 255) + li_t) * 17 + lq_t;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) { av[q & 1][kk] = -pa[4 * kk]; bv[q & 1][kk] = pb[4 * kk]; }
            };
            auto leave = [&](auto Q) __attribute__((always_inline)) {      // what a finished slot leaves in LDS
                constexpr int q = decltype(Q)::value;
                if (q >= n4 && (bc_[q] >> 8) != (bc_[q] & 255)) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) nxt[(16 * (bc_[q] >> 8) + lq_t + 4 * e) * 17 + li_t] = blk[q][e];
                }
                if (q == qd) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) dpq[(lq_t + 4 * e) * 18 + li_t] = blk[q][e];
                }
            };
            static_for<0, CP_SLOTS>([&](auto Q) {
                constexpr int q = CP_SLOTS - 1 - decltype(Q)::value;
                if (q < n34) {
                    if (CF_STAMPS && stamp_ok && j == jbeg) stamps[100 + q] = __builtin_amdgcn_s_memtime();
                    if (q == n34 - 1) fetch(std::integral_constant<int, q>{});
                    if constexpr (q > 0) fetch(std::integral_constant<int, q - 1>{});
                    // (a next-panel slot that holds the diagonal block takes no update here: wave 0 gives it its last one)
                    if (!(q >= n4 && (bc_[q] >> 8) == (bc_[q] & 255))) {
                        if (CP_SKIP & 1) { blk[q][0] += av[q & 1][0] * bv[q & 1][1] + av[q & 1][2] * bv[q & 1][3] + av[q & 1][1] * bv[q & 1][0] + av[q & 1][3] * bv[q & 1][2]; }
                        else {
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk) blk[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[q & 1][kk], bv[q & 1][kk], blk[q], 0, 0, 0);
                        }
                    }
                    if constexpr (q + 1 < CP_SLOTS) { if (q + 1 < n34) leave(std::integral_constant<int, q + 1>{}); }
                }
            });
            if (n34 > 0) leave(std::integral_constant<int, 0>{});
            if (CF_STAMPS && stamp_ok && j == jbeg) { stamps[106] = __builtin_amdgcn_s_memtime(); asm volatile("s_nop 0" :: "v"(blk[0])); stamps[107] = __builtin_amdgcn_s_memtime(); }
        };
        // the diagonal block of step j and its inverse, from wave 0's LDS copies to global memory (off wave 0's chain)
        auto publish_diag = [&](int j) __attribute__((always_inline)) {
            const double* lg = Ldg + (j & 1) * (16 * 18);
            const double* dv = DinvT + (j & 1) * (16 * 18);
#pragma unroll
            for (int t = 0; t < 2; ++t) {       // 128 pairs of neighbouring columns: 16 bytes per lane and store
                const int r = 2 * lq_t + (li_t >> 3) + 8 * t, c = 2 * (li_t & 7);
                const f64x2 l2 = *reinterpret_cast<const f64x2*>(lg + r * 18 + c);
                *reinterpret_cast<f64x2*>(T + (int64_t)(16 * j + r) * C + 16 * j + c) = f64x2{c <= r ? l2[0] : 0.0, c + 1 <= r ? l2[1] : 0.0};
                st_sc1_x2(rLinv, j * 256 + 16 * r + c, f64x2{dv[c * 18 + r], dv[(c + 1) * 18 + r]});
            }
        };
        // this wave's trailing blocks: ranked by block column, LAST column first, the diagonal block first within a column, dealt
        // round-robin -- the blocks still active at a step are a prefix of every wave's slots.  (As the launch before left them.)
        {
            const int nblk = cp_colblocks(cb, ce, nb);
#pragma unroll
            for (int q = 0; q < CP_SLOTS; ++q) {
                const int r = ow + 15 * q;
                bc_[q] = 0;
                blk[q] = f64x4{0.0, 0.0, 0.0, 0.0};
                if (r < nblk) {
                    int bj = ce - 1;
                    while (cp_colblocks(bj, ce, nb) <= r) --bj;       // column bj holds the ranks [blocks right of it, blocks from it on)
                    const int bi = bj + (r - cp_colblocks(bj + 1, ce, nb));
                    bc_[q] = (bi << 8) | bj;
#pragma unroll
                    for (int e = 0; e < 4; ++e) blk[q][e] = T[(int64_t)(16 * bi + lq + 4 * e) * C + 16 * bj + li];
                }
            }
        }
        if (jbeg == 0) {
            // panel 0 (block column 0 below its diagonal block) and block (1, 1) as they stand -> LDS
            for (int e = tid - 64; e < (C - 16) * 16; e += 960)
                Praw[(16 + (e >> 4)) * 17 + (e & 15)] = T[(int64_t)(16 + (e >> 4)) * C + (e & 15)];
            if (nb > 1)
                for (int e = tid - 64; e < 256; e += 960)
                    Dpre[16 * 18 + (e >> 4) * 18 + (e & 15)] = T[(int64_t)(16 + (e >> 4)) * C + 16 + (e & 15)];
        } else {
#pragma unroll 1
            for (int j = 0; j < jbeg; ++j) {
                if (wave == 1) {
                    bool good = true;
                    if (lane == 0) good = cp_wait_word(sync + 2, (unsigned)j + 1);
                    if (!__builtin_amdgcn_readfirstlane((int)good) && lane == 0) atomicOr(sync + 1, 2u);     // a wait ran out: the host reads the error word
                }
                CP_BARRIER();                                      // (P1) panel j is complete in global memory
                asm volatile("" : "+v"(li_t), "+v"(lq_t));
                stage_panel(j);
                CP_BARRIER();                                      // (P2) ... and its rows >= 16 split in Pn
                trailing(j);
            }
        }
        int nB = 0;                     // meetings (B) so far
#pragma unroll 1
        for (int j = jbeg; j < jend; ++j) {
            // every store of the step before has landed (the solved panel's rows; the publisher's copy of diagonal block j-1), and
            // every owner passed this wait before barrier (A) of step j-1 with its rows of the panels before: row blocks 0..j-1 and
            // panels 0..j-1 are complete in global memory once the barrier below is behind us
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            CP_BARRIER();                                          // (A) the inverse of L_jj and panel j are in LDS
            asm volatile("" : "+v"(li_t), "+v"(lq_t));
            if (publisher && lane == 0 && j > jbeg) {
                __hip_atomic_store(sync, (unsigned)j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (jend < nb) __hip_atomic_store(sync + 2, (unsigned)j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (a factoriser behind this one follows the panels)
            }
            if (publisher) publish_diag(j);
            const int j0 = 16 * j, g0 = j0 + 16, rows = C - g0;
            if (rows <= 0) {
                if (publisher) {                                   // the last block and its inverse have landed: L is complete
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if (lane == 0) __hip_atomic_store(sync, (unsigned)nb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                break;
            }
            const double* pan = Praw + (j & 1) * (C * 17);
            // (S2) panel solve X = P L_jj^-T as a product with the inverted diagonal block, one 16-row block per solver wave and turn
            if (solver) {
                const double* dv = DinvT + (j & 1) * (16 * 18);        // B operand Linv[c = li][k = lq + 4 kk] = DinvT[k][li]
                for (int rb = sv; 16 * rb < rows; rb += 12) {
                    const int row0 = g0 + 16 * rb;
                    f64x4 x = {0.0, 0.0, 0.0, 0.0}, x1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int kk = 0; kk < 4; kk += 2) {
                        if (CP_SKIP & 4) { x[kk] += pan[(row0 + li_t) * 17 + 4 * kk + lq_t] * dv[(4 * kk + lq_t) * 18 + li_t]; x1[kk] += pan[(row0 + li_t) * 17 + 4 * kk + 4 + lq_t] * dv[(4 * kk + 4 + lq_t) * 18 + li_t]; continue; }
                        x = __builtin_amdgcn_mfma_f64_16x16x4f64(pan[(row0 + li_t) * 17 + 4 * kk + lq_t], dv[(4 * kk + lq_t) * 18 + li_t], x, 0, 0, 0);
                        x1 = __builtin_amdgcn_mfma_f64_16x16x4f64(pan[(row0 + li_t) * 17 + 4 * kk + 4 + lq_t], dv[(4 * kk + 4 + lq_t) * 18 + li_t], x1, 0, 0, 0);
                    }
                    x += x1;
                    if (!(CP_SKIP & 16)) cp_store_block_sc1(rT, row0 * C + j0, C, x, li_t, lq_t);      // write-through: the inverse role and the second factoriser read it in this launch
#pragma unroll
                    for (int e = 0; e < 4; ++e) Pn[(row0 + lq_t + 4 * e) * 17 + li_t] = x[e];
                }
            }
            const bool last = j + 1 >= jend;                        // this factoriser's last panel (rows > 0: not the matrix' last)
            if (last) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the meeting below then says: panel j and diagonal block j are complete in memory
            {   // (B) among the owners: every owner's rows of the solved panel are in LDS
                if (CF_STAMPS && stamp_ok && nstamp < 100) stamps[nstamp++] = __builtin_amdgcn_s_memtime();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane == 0) { const unsigned one = 1u; asm volatile("ds_add_u32 %0, %1" :: "v"(cntB_lds), "v"(one) : "memory"); }
                const int target = 15 * (++nB);
                for (;;) {
                    int v;
                    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(cntB_lds) : "memory");
                    if (__builtin_amdgcn_readfirstlane(v) >= target) break;
                }
                if (CF_STAMPS && stamp_ok && nstamp < 100) stamps[nstamp++] = __builtin_amdgcn_s_memtime();
            }
            if (last) {
                // hand over: the next factoriser's last passive step waits for this panel; the inverse role for row block j
                if (publisher && lane == 0) {
                    __hip_atomic_store(sync + 2, (unsigned)j + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(sync, (unsigned)j + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                break;
            }
            if (!(CP_SKIP & 2)) trailing(j);
        }
    }
    if (CF_STAMPS && stamp_ok) { if (nstamp < 100) stamps[nstamp++] = __builtin_amdgcn_s_memtime(); stamps[127] = nstamp; }
}

// grid: [0, groups) first factorisers, [groups, groups (1 + TI_WG)) inverse, then the factorisers 2 .. nparts of every matrix.
// bounds: the first steps of the factorisers 2, 3, 4, one byte each.
__global__ __launch_bounds__(1024) void cholesky_phased_kernel(double* __restrict__ T, double* __restrict__ Linv, int C, double* __restrict__ Winv,
                                                               unsigned* __restrict__ sync, int groups, int nparts, int bounds)
{
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int b = (int)blockIdx.x, nb = C >> 4;
    if (b >= groups && b < groups * (1 + TI_WG)) {
        const int idx = b - groups, g = idx / TI_WG;
        tri_inverse_role(T + (int64_t)g * C * C, Linv + (int64_t)g * C * 16, Winv + (int64_t)g * C * C, C, idx % TI_WG, sync + 16 * g, sm);
        return;
    }
    const int part = b < groups ? 0 : 1 + (b - groups * (1 + TI_WG)) / groups;
    const int g = b < groups ? b : (b - groups * (1 + TI_WG)) % groups;
    const int jbeg = part ? (bounds >> (8 * (part - 1))) & 255 : 0;
    const int jend = part + 1 < nparts ? (bounds >> (8 * part)) & 255 : nb;
    // (development stamps: 16 x 128 words per factoriser of matrix 0)
    cp_factor_role(T + (int64_t)g * C * C, Linv + (int64_t)g * C * 16, C, sync + 16 * g, jbeg, jend, sm,
                   (CF_STAMPS && g == 0) ? reinterpret_cast<unsigned long long*>(Linv + 8192) + part * 16 * 128 : nullptr);
}

// W = L^-1 from L and the inverses of its 16 x 16 diagonal blocks, one WAVE per block column j:
//   X_j = Linv_jj;   X_i = -Linv_ii sum_{k=j}^{i-1} L_ik X_k   (i > j),   W[i][j] = X_i.
// The X blocks of a column stay in its wave's registers in the MFMA accumulator layout, which is at the same time a valid
// B-operand layout (the four MFMAs of a 16-deep product take k' = (lane>>4) + 4r for r = 0..3; the order in which a
// contraction index is visited is free).  All columns walk the row blocks i in lock-step: row block i of L (16 x C, 32 KB) is
// staged through LDS once per workgroup, one step ahead of its use (fed straight from L2 the MFMAs waited ~240 cycles each
// for their operands: 54 us instead of 18).  Column 0 is the longest chain (540 MFMAs); columns j and nb-1-j share a SIMD.
template <int PER>        // 16-byte pieces of a row block per thread: ceil(8 C / 512)
__global__ __launch_bounds__(512) void tri_inverse_cols_kernel(const double* __restrict__ L, const double* __restrict__ Linv,
                                                               double* __restrict__ W, int C)
{
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int ldr = C + 2;
    double* Lrow = sm;                          // [2][16][ldr]  row block i of L, even / odd i
    double* Dv = Lrow + 2 * 16 * ldr;           // [nb][16][17]  the inverted diagonal blocks
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lq = lane >> 4;
    const int nb = C >> 4;
    L += (int64_t)blockIdx.y * C * C; W += (int64_t)blockIdx.y * C * C; Linv += (int64_t)blockIdx.y * C * 16;
    const int idx = 4 * blockIdx.x + (wave & 3);
    const bool has_col = idx < (nb >> 1);
    const int j = has_col ? ((wave < 4) ? idx : nb - 1 - idx) : nb;       // idle waves: a column beyond the matrix

    // cooperative staging of a row block: 16 x C doubles = C/2 float4-sized pieces per row, 8 * C / 512 = C/64 per thread
    typedef double f64x2 __attribute__((ext_vector_type(2)));
    // (no predication: a surplus thread repeats the last piece.  Exec-masked loads made hipcc wait for each one before it
    // issued the next -- four serial L2 round trips, 2100 cycles per step)
    const int pieces = 8 * C;                   // 16-byte pieces of a row block
    f64x2 st[PER];
    int soff[PER], doff[PER];                   // the piece's place in a row block of L / in its LDS image: computed ONCE (two
#pragma unroll                                  // integer divisions per piece; left inside fetch / stash they ran every step)
    for (int p = 0; p < PER; ++p) {
        int e = tid + 512 * p; e = e < pieces ? e : pieces - 1;
        const int row = e / (C >> 1), c2 = e % (C >> 1);
        soff[p] = row * C + 2 * c2; doff[p] = row * ldr + 2 * c2;
    }
    auto fetch = [&](int i) {
        const double* src = L + (int64_t)16 * i * C;
#pragma unroll
        for (int p = 0; p < PER; ++p) st[p] = *reinterpret_cast<const f64x2*>(src + soff[p]);
    };
    auto stash = [&](int i) {
        double* dst = Lrow + (i & 1) * 16 * ldr;
#pragma unroll
        for (int p = 0; p < PER; ++p) *reinterpret_cast<f64x2*>(dst + doff[p]) = st[p];
    };
    for (int e = tid; e < nb * 256; e += 512) Dv[(e >> 8) * (16 * 17) + ((e >> 4) & 15) * 17 + (e & 15)] = Linv[e];
    if (nb > 1) { fetch(1); stash(1); }
    if (nb > 2) fetch(2);                       // (head(1) stashes it)
    __syncthreads();

    f64x4 X[16];
    auto store = [&](int i, const f64x4& v) {
#pragma unroll
        for (int r = 0; r < 4; ++r) W[(int64_t)(16 * i + lq + 4 * r) * C + 16 * j + li] = v[r];
    };
    if (has_col) {
        for (int i = 0; i < j; ++i) store(i, f64x4{0.0, 0.0, 0.0, 0.0});      // blocks above the diagonal block: zero
#pragma unroll
        for (int r = 0; r < 4; ++r) X[0][r] = Dv[j * (16 * 17) + (lq + 4 * r) * 17 + li];
        store(j, X[0]);
    }
    // global step i, head: row block i+1 (fetched one step ago) into the other buffer -- its last readers passed the barrier that
    // ended step i-1 -- and row block i+2 on its way.  Tail: the barrier.
    auto head = [&](int i) {
        if (i + 1 < nb) stash(i + 1);
        if (i + 2 < nb) fetch(i + 2);
    };
    const bool stamp_ok = CF_STAMPS && lane == 0 && blockIdx.x == 0 && blockIdx.y == 0 && (wave == 0 || wave == 4);
    unsigned long long* stamps = reinterpret_cast<unsigned long long*>(W + (int64_t)C * C) + (wave == 0 ? 0 : 128);   // (the workspace behind W: development builds only)
    int nstamp = 0;
    (void)stamps; (void)nstamp; (void)stamp_ok;
#define TI_STAMP() do { if (CF_STAMPS && stamp_ok) stamps[nstamp++] = __builtin_amdgcn_s_memtime(); } while (0)
    auto tail = [&]() { TI_STAMP(); asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); TI_STAMP(); };
    for (int i = 1; i <= j && i < nb; ++i) { head(i); tail(); }                 // steps before this column starts
    static_for<1, 16>([&](auto D) {
        constexpr int d = decltype(D)::value;
        const int i = j + d;
        if (i < nb) {
            head(i);
            TI_STAMP();
            const double* lr = Lrow + (i & 1) * 16 * ldr + li * ldr + lq + 16 * j;
            // the A operands of product e+1 are read from LDS before the MFMAs of product e issue
            f64x4 S = {0.0, 0.0, 0.0, 0.0}, S1 = {0.0, 0.0, 0.0, 0.0};
            double an[4], ac[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) an[r] = lr[4 * r];
            static_for<0, d>([&](auto E) {
                constexpr int e = decltype(E)::value;
#pragma unroll
                for (int r = 0; r < 4; ++r) ac[r] = an[r];
                if constexpr (e + 1 < d) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) an[r] = lr[16 * (e + 1) + 4 * r];
                }
                __builtin_amdgcn_sched_barrier(0);
                S = __builtin_amdgcn_mfma_f64_16x16x4f64(ac[0], X[e][0], S, 0, 0, 0);
                S1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ac[1], X[e][1], S1, 0, 0, 0);
                S = __builtin_amdgcn_mfma_f64_16x16x4f64(ac[2], X[e][2], S, 0, 0, 0);
                S1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ac[3], X[e][3], S1, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            });
            S += S1;
            f64x4 Xi = {0.0, 0.0, 0.0, 0.0}, Xi1 = {0.0, 0.0, 0.0, 0.0};
            const double* pi = Dv + i * (16 * 17) + li * 17 + lq;
#pragma unroll
            for (int r = 0; r < 4; r += 2) {
                Xi = __builtin_amdgcn_mfma_f64_16x16x4f64(-pi[4 * r], S[r], Xi, 0, 0, 0);
                Xi1 = __builtin_amdgcn_mfma_f64_16x16x4f64(-pi[4 * r + 4], S[r + 1], Xi1, 0, 0, 0);
            }
            Xi += Xi1;
            X[d] = Xi;
            TI_STAMP();
            tail();
        }
    });
    // the column leaves the registers at the end, in one burst: a store inside the loop sits in the same in-order memory
    // counter as the row-block fetches, and every wait for fetched data then also waited ~2000 cycles for the store before it
    static_for<1, 16>([&](auto D) {
        constexpr int d = decltype(D)::value;
        if (j + d < nb) store(j + d, X[d]);
    });
    if (CF_STAMPS && stamp_ok) stamps[127] = nstamp;
#undef TI_STAMP
}

// inverse of each 32 x 32 diagonal block of L (lower): one wave per block, lane = column of the inverse
__global__ __launch_bounds__(256) void tri_inv_diag_kernel(const double* __restrict__ L, double* __restrict__ W, int C)
{
    __shared__ double Lb[32 * 33];
    const int b0 = blockIdx.x * 32;
    const int tid = threadIdx.x, lane = tid & 63;
    L += (int64_t)blockIdx.y * C * C; W += (int64_t)blockIdx.y * C * C;      // group
    // this block row of W outside its diagonal block starts as zero (the upper triangle stays so; the block doubling
    // fills the lower part): no separate memset launch.  All 256 threads, 16 B each.
    typedef double f64x2 __attribute__((ext_vector_type(2)));
    for (int e = tid; e < 16 * C; e += 256) {       // pairs of doubles: 32 rows x C/2 pairs
        const int i = e / (C >> 1), k = 2 * (e - i * (C >> 1));
        if (k < b0 || k >= b0 + 32) *reinterpret_cast<f64x2*>(W + (int64_t)(b0 + i) * C + k) = f64x2{0.0, 0.0};
    }
    for (int e = tid; e < 32 * 32; e += 256) {
        const int i = e >> 5, k = e & 31;
        Lb[i * 33 + k] = L[(int64_t)(b0 + i) * C + b0 + k];
    }
    __syncthreads();
    if (tid >= 64) return;                          // one wave inverts
    const int c = lane & 31;
    double w[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        double s = (i == c) ? 1.0 : 0.0;
#pragma unroll
        for (int k = 0; k < i; ++k) s -= Lb[i * 33 + k] * w[k];
        w[i] = s / Lb[i * 33 + i];
    }
    if (lane < 32) {
#pragma unroll
        for (int i = 0; i < 32; ++i) W[(int64_t)(b0 + i) * C + b0 + c] = w[i];
    }
}

// ---------------------------------------------------------------------------------------------
// generic small GEMM on v_mfma_f64_16x16x4_f64.
//   operand maps: a = A[i = lane&15][k = lane>>4], b = B[k = lane>>4][j = lane&15],
//   D register r of lane l = D[(l>>4) + 4*r][l&15]   (NOT the f32 row formula).
// One 32 x 32 output block per 256-thread workgroup; the four waves split K and meet in LDS.
// ---------------------------------------------------------------------------------------------
template <typename TA, typename TB>
__device__ __forceinline__ void gemm_f64_body(const WcGemm& g, int bz, double (&red)[4][32 * 32])
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lq = lane >> 4;
    const int bm = blockIdx.x, bn = blockIdx.y;
    const int b = bz % g.batch, b2 = bz / g.batch;       // two batch levels (e.g. class x group)
    const int kper = g.k >> 2;

    f64x4 acc[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u) acc[t][u] = f64x4{0.0, 0.0, 0.0, 0.0};

    for (int r = 0; r < g.nred; ++r) {
        if (g.red_total > 0 && b * g.nred + r >= g.red_total) break;
        const TA* A = reinterpret_cast<const TA*>(g.A) + (int64_t)b * g.a_bs + (int64_t)b2 * g.a_b2s + (int64_t)r * g.a_red;
        const TB* B = reinterpret_cast<const TB*>(g.B) + (int64_t)b * g.b_bs + (int64_t)b2 * g.b_b2s + (int64_t)r * g.b_red;
        const TA* A0 = A + (int64_t)(bm * 32 + li) * g.a_rs;
        const TA* A1 = A0 + 16 * g.a_rs;
        const TB* B0 = B + (int64_t)(bn * 32 + li) * g.b_cs;
        const TB* B1 = B0 + 16 * g.b_cs;
        // the loop is load latency, not flops: 16 k at a time, every load of the group ahead of its MFMAs and the NEXT group's
        // loads issued before this group's MFMAs (one exposed L2 latency per call instead of one per group: the chains of
        // six to nine of these launches in K5 and the coloring are what their 7-9 us each add up to)
        // (the trip count is a run-time value, so the unrolling is spelled out: hipcc declines it otherwise)
        const int kend = (wave + 1) * kper;
        int kk = wave * kper;
        double a0[4], a1[4], b0[4], b1[4];
        auto load16 = [&](int k0, double (&x0)[4], double (&x1)[4], double (&y0)[4], double (&y1)[4]) __attribute__((always_inline)) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t ka = k0 + 4 * u + lq;
                x0[u] = (double)A0[ka * g.a_cs]; x1[u] = (double)A1[ka * g.a_cs];
                y0[u] = (double)B0[ka * g.b_rs]; y1[u] = (double)B1[ka * g.b_rs];
            }
        };
        // 64 k at once when the wave's share allows (C = 256: all of it): every load of the call in flight together
        for (; kk + 64 <= kend; kk += 64) {
            double wa0[16], wa1[16], wb0[16], wb1[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int64_t ka = kk + 4 * u + lq;
                wa0[u] = (double)A0[ka * g.a_cs]; wa1[u] = (double)A1[ka * g.a_cs];
                wb0[u] = (double)B0[ka * g.b_rs]; wb1[u] = (double)B1[ka * g.b_rs];
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(wa0[u], wb0[u], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(wa0[u], wb1[u], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(wa1[u], wb0[u], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(wa1[u], wb1[u], acc[1][1], 0, 0, 0);
            }
        }
        if (kk + 16 <= kend) load16(kk, a0, a1, b0, b1);
        for (; kk + 16 <= kend; kk += 16) {
            double na0[4], na1[4], nb0[4], nb1[4];
            const bool more = kk + 32 <= kend;
            load16(more ? kk + 16 : kk, na0, na1, nb0, nb1);         // (the last group re-reads itself: no predicated loads)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[u], b0[u], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[u], b1[u], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[u], b0[u], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[u], b1[u], acc[1][1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 4; ++u) { a0[u] = na0[u]; a1[u] = na1[u]; b0[u] = nb0[u]; b1[u] = nb1[u]; }
        }
        for (; kk < kend; kk += 4) {
            const int64_t ka = kk + lq;
            const double a0 = (double)A0[ka * g.a_cs], a1 = (double)A1[ka * g.a_cs];
            const double b0 = (double)B0[ka * g.b_rs], b1 = (double)B1[ka * g.b_rs];
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
        }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wave][(t * 16 + lq + 4 * r) * 32 + u * 16 + li] = acc[t][u][r];
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int e = tid + 256 * p;
        const int row = e >> 5, col = e & 31;
        double v = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
        v *= g.alpha;
        const int gi = bm * 32 + row, gj = bn * 32 + col;
        if (g.epi != WC_EPI_NONE) {
            if (gj > gi) v = 0.0;
            else if (gj == gi && g.epi == WC_EPI_PHI) v *= 0.5;
        }
        const int64_t off = (int64_t)b * g.c_bs + (int64_t)b2 * g.c_b2s + (int64_t)gi * g.c_rs + (int64_t)gj * g.c_cs;
        if (g.c_is_f32) reinterpret_cast<float*>(g.Cm)[off] = (float)v;
        else reinterpret_cast<double*>(g.Cm)[off] = v;
        if (g.Cm2)
            reinterpret_cast<float*>(g.Cm2)[(int64_t)b * g.c2_bs + (int64_t)b2 * g.c_b2s + (int64_t)gi * g.c2_rs + (int64_t)gj * g.c2_cs] = (float)v;
    }
}

template <typename TA, typename TB>
__global__ __launch_bounds__(256) void gemm_f64_kernel(WcGemm g)
{
    __shared__ double red[4][32 * 32];
    gemm_f64_body<TA, TB>(g, blockIdx.z, red);
}

// two independent products of the same output tiling in ONE launch (blockIdx.z < nz0: the first): K5 starts with
// dgamma = W R[k] and Wbar = sum_k Gamma_k R_k^T, neither of which waits for the other
template <typename TA0, typename TB0, typename TA1, typename TB1>
__global__ __launch_bounds__(256) void gemm_f64_pair_kernel(WcGemm g0, WcGemm g1, int nz0)
{
    __shared__ double red[4][32 * 32];
    if ((int)blockIdx.z < nz0) gemm_f64_body<TA0, TB0>(g0, blockIdx.z, red);
    else gemm_f64_body<TA1, TB1>(g1, blockIdx.z - nz0, red);
}

// ---------------------------------------------------------------------------------------------
// element-wise helpers
// ---------------------------------------------------------------------------------------------
__global__ void transpose_to_f32_kernel(const double* __restrict__ W, int C, float* __restrict__ A, float* __restrict__ At)
{
    __shared__ double tile[32][33];
    W += (int64_t)blockIdx.z * C * C; A += (int64_t)blockIdx.z * C * C;          // one matrix (statistic group) per blockIdx.z
    if (At) At += (int64_t)blockIdx.z * C * C;
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;     // 256 threads: ty in 0..7
    for (int r = ty; r < 32; r += 8) {
        const double v = W[(int64_t)(by + r) * C + bx + tx];
        tile[r][tx] = v;
        if (At) At[(int64_t)(by + r) * C + bx + tx] = (float)v;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) A[(int64_t)(bx + r) * C + by + tx] = (float)tile[tx][r];
}

// K5's tail in one launch: S = scale * sym(Q) (rows y < C), gmean[c] = (1/M) sum_k sum_j gsum[k][j] A[k][c][j] (rows y >= C,
// first column block) and dbeta = float(gsum) (same rows) -- three launches of ~5 us each on the backward's critical path before
__global__ __launch_bounds__(128) void bwd_tail_kernel(const double* __restrict__ Q, int C, double scale, float* __restrict__ S,
                                                       const double* __restrict__ gsum, const float* __restrict__ A, int Kc,
                                                       int64_t M, float* __restrict__ gmean, float* __restrict__ dbeta)
{
    __shared__ double red[128];
    const int y = blockIdx.y;
    if (y < C) {
        const int j = blockIdx.x * 128 + threadIdx.x;
        if (j < C) S[(int64_t)y * C + j] = (float)(scale * 0.5 * (Q[(int64_t)y * C + j] + Q[(int64_t)j * C + y]));
        return;
    }
    if (blockIdx.x != 0) return;
    const int c = y - C;
    if (dbeta)
        for (int64_t e = (int64_t)c * 128 + threadIdx.x; e < (int64_t)Kc * C; e += (int64_t)C * 128) dbeta[e] = (float)gsum[e];
    double s = 0.0;
    for (int64_t e = threadIdx.x; e < (int64_t)Kc * C; e += 128) {
        const int64_t k = e / C, j = e % C;
        s += gsum[e] * (double)A[(k * C + c) * C + j];
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 64; w > 0; w >>= 1) {
        if (threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) gmean[c] = (float)(red[0] / (double)M);
}

// grouped forward: center[c] = mean_g mu[g][c];  bias[s][n] = beta[s % Kc][n] - sum_c (mu[g][c] - center[c]) A[s][c][n],
// s = g*Kc + k.  One block per slot s.
__global__ __launch_bounds__(1024) void group_bias_kernel(const float* __restrict__ mu, const float* __restrict__ A,
                                                          const float* __restrict__ beta, int G, int Kc, int C, int per_group,
                                                          float* __restrict__ center, float* __restrict__ bias,
                                                          const float* __restrict__ center_in)
{
    // (round 6: 256 threads walked the C rows of A one dependent load after the other, 12 us for a C x C matrix-vector product; now four
    //  quarters of the rows side by side, 16 loads in flight per thread, the quarters added in a fixed order)
    __shared__ double dm[1024];
    __shared__ double part[4][256];
    const int s_ = blockIdx.x, g = s_ / Kc, k = s_ % Kc;
    for (int c = threadIdx.x; c < C; c += 1024) {
        // center_in (round 4): the common centre is GIVEN -- the centre of a pre-split input's planes -- so that the biases are the
        // planes route's additive terms beta - (mu_g - center) A directly (no wc_split_bias_f32 launch behind this one)
        double m = 0.0;
        if (center_in) m = (double)center_in[c];
        else {
            for (int gg = 0; gg < G; ++gg) m += (double)mu[(int64_t)gg * C + c];
            m /= (double)G;
            if (s_ == 0) center[c] = (float)m;
        }
        dm[c] = (double)mu[(int64_t)g * C + c] - m;
    }
    __syncthreads();
    const float* As = A + (int64_t)s_ * C * C;
    const int q = threadIdx.x >> 8, t = threadIdx.x & 255;
    const int c0 = (C * q) / 4, c1 = (C * (q + 1)) / 4;
    for (int n0 = 0; n0 < C; n0 += 256) {
        const int n = n0 + t;
        double acc = 0.0;
        if (n < C) {
            int c = c0;
            for (; c + 16 <= c1; c += 16) {
                float v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) v[u] = As[(int64_t)(c + u) * C + n];
#pragma unroll
                for (int u = 0; u < 16; ++u) acc += dm[c + u] * (double)v[u];
            }
            for (; c < c1; ++c) acc += dm[c] * (double)As[(int64_t)c * C + n];
        }
        part[q][t] = acc;
        __syncthreads();
        if (q == 0 && n < C) {
            const double b0 = beta ? (double)beta[(int64_t)(per_group ? s_ : k) * C + n] : 0.0;
            bias[(int64_t)s_ * C + n] = (float)(b0 - ((part[0][t] + part[1][t]) + (part[2][t] + part[3][t])));
        }
        __syncthreads();
    }
}

__global__ void f64_to_f32_kernel(const double* __restrict__ src, float* __restrict__ dst, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = (float)src[i];
}

}  // namespace

hipError_t wc_launch_stats_finalize(const double* P, const float* colsum, const float* shift, int nslab,
                                    int64_t M, int C, int groups, double* Sp, double* sum, double* xtx,
                                    const double* dfix, const int* gate, hipStream_t st, double kappa)
{
    // nslab and M are PER GROUP; group g owns slabs [g*nslab, (g+1)*nslab).  Sp holds 2*groups*C doubles: the column sums, then
    // (kappa != 0) the diagonal's slab sums Dp
    double* Dp = (dfix && kappa != 0.0) ? Sp + (size_t)groups * C : nullptr;
    hipLaunchKernelGGL(stats_colsum_kernel, dim3((C + 63) / 64, groups), dim3(1024), 0, st, colsum, shift, nslab, M, C, Sp, sum, dfix, Dp);
    hipLaunchKernelGGL(stats_xtx_kernel, dim3((C + 63) / 64, C, groups), dim3(64 * SX_PARTS), 0, st, P, shift, (const double*)Sp, nslab, M, C, xtx, dfix, gate,
                       kappa, (const double*)Dp);
    return hipGetLastError();
}

hipError_t wc_launch_bwd_combine(const double* P, const float* colsum, const int32_t* slot, int64_t N, int nsplit,
                                 int per_sample, int C, int Kc, double* R, double* gsum, hipStream_t st)
{
    const int64_t total = (int64_t)C * C + C;
    int parts = SX_PARTS;                      // slab groups per block: no more than the slabs of one run, a power of two
    while (parts > 1 && parts > nsplit) parts >>= 1;
    const int epw = (64 * SX_PARTS) / parts;
    hipLaunchKernelGGL(bwd_combine_kernel, dim3((unsigned)((total + epw - 1) / epw), Kc), dim3(64 * SX_PARTS), 0, st,
                       P, colsum, slot, N, nsplit, per_sample, C, R, gsum, parts);
    return hipGetLastError();
}

// C <= 256: Cholesky with look-ahead + the inverses of the diagonal blocks in one launch, W in a second one
static bool use_fused_factor(int C)
{
    static const bool off = getenv("WC_CHOL_OLD") != nullptr;             // development: the round-1 kernels
    return C <= 256 && !off;
}
// factor and inverse in ONE launch (tri_inverse_role): every workgroup of the launch has to be resident at once (the inverse
// workgroups spin on the factorising ones), so only for a handful of matrices.  Measured on MI355X (tools/k2_pipe_check.py), K2
// per call, two launches -> one: C = 256: 94.4 -> 73.1 us (5 groups 96.1 -> 76.5, 8 groups 99.0 -> 77.7), C = 224 (3 groups):
// 76.7 -> 59.6, C = 128: 35.6 -> 29.5; below that the hand-off per row block (counter, sc1 fetch: ~3 us) is longer than the
// factorisation's step and two launches are faster (C = 64: 18.9 against 24-30 us).
static bool factor_one_launch(int C, int groups)
{
    static const bool off = getenv("WC_K2_TWO_LAUNCH") != nullptr || getenv("WC_K2_SPLIT") != nullptr;
    return use_fused_factor(C) && C >= 128 && !off && groups * (1 + TI_WG) <= 40;
}

hipError_t wc_launch_factor_prepare(const double* sum, const double* xtx, int64_t M, int C, double eps, double momentum,
                                    int ddof, int training, int groups, float* moving_mean, float* moving_cov, float* mu,
                                    float* chan_scale, double* T, hipStream_t st, double* tmp)
{
    // the row-block counters of the one-launch factor + inverse sit behind the inverted diagonal blocks in `tmp`
    unsigned* rows = (tmp && factor_one_launch(C, groups)) ? reinterpret_cast<unsigned*>(tmp + (size_t)groups * C * 16) : nullptr;
    if (rows && 16 * groups > 128) return hipErrorInvalidValue;
    hipLaunchKernelGGL(factor_prepare_kernel, dim3((C + 127) / 128, C), dim3(128), 0, st,
                       sum, xtx, M, C, eps, momentum, ddof, training, groups, moving_mean, moving_cov, mu, chan_scale, T,
                       use_fused_factor(C) ? 1 : 0, rows, rows ? 16 * groups : 0);
    return hipGetLastError();
}

// K1 tail and K2 head together (wc_whiten_f32): the column sums, then ONE launch for slab reduction + moments -> T, mu,
// moving statistics, chan_scale.  nslab and M are PER GROUP; `tmp` as in wc_launch_factor_prepare; sum_scratch [groups*C].
hipError_t wc_launch_stats_prepare(const double* P, const float* colsum, const float* shift, int nslab, int64_t M, int C, int groups,
                                   double* Sp, double* sum_scratch, const double* dfix, const int* gate, double eps, double momentum,
                                   int ddof, float* moving_mean, float* moving_cov, float* mu, float* chan_scale, double* T,
                                   hipStream_t st, double* tmp, double kappa)
{
    unsigned* rows = (tmp && factor_one_launch(C, groups)) ? reinterpret_cast<unsigned*>(tmp + (size_t)groups * C * 16) : nullptr;
    if (rows && 16 * groups > 128) return hipErrorInvalidValue;
    double* Dp = (dfix && kappa != 0.0) ? Sp + (size_t)groups * C : nullptr;       // Sp holds 2*groups*C doubles
    hipLaunchKernelGGL(stats_colsum_kernel, dim3((C + 63) / 64, groups), dim3(1024), 0, st, colsum, shift, nslab, M, C, Sp, sum_scratch, dfix, Dp);
    hipLaunchKernelGGL(stats_xtx_prepare_kernel, dim3((C + 63) / 64, C), dim3(64 * SX_PARTS), 0, st, P, shift, (const double*)Sp, nslab, M, C,
                       dfix, gate, kappa, (const double*)Dp, groups, eps, momentum, ddof, moving_mean, moving_cov, mu, chan_scale, T,
                       use_fused_factor(C) ? 1 : 0, rows, rows ? 16 * groups : 0);
    return hipGetLastError();
}

hipError_t wc_launch_factor_fused(double* T, double* W, double* tmp, int C, int groups, hipStream_t st)
{
    const int nb = C >> 4, ldp = 17;
    size_t lds = (size_t)(2 * C * 17 + C * 17 + 4 * 16 * 17 + 2) * sizeof(double);      // (+ the owners' panel counter)
    const bool one = factor_one_launch(C, groups);
    static const bool split = getenv("WC_K2_SPLIT") != nullptr;         // development: the four-waves-per-column inverse as a launch of its own
    const size_t lds_role = ti_role_lds_doubles(C) * sizeof(double);
    if (one && lds_role > lds) lds = lds_role;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(cholesky_fused_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    if (one) {
        unsigned* rows = reinterpret_cast<unsigned*>(tmp + (size_t)groups * C * 16);     // zeroed by factor_prepare_kernel
        static const bool old_one = getenv("WC_K2_FUSED_R5") != nullptr;     // development A/B: round 5's one-workgroup factorisation
        if (!old_one) {
            // the relay's stages: WC_K2_BOUNDS="4,9" (development) or the rule below; a stage that does not fit its owners' slots or the
            // passive fetch -> one stage (or round 5's kernel when even that does not fit)
            static const char* benv = getenv("WC_K2_BOUNDS");
            int bnd[4] = {0, 0, 0, 0}, np = 1;
            if (benv && *benv) {
                const char* q = benv;
                while (*q && np < 4) { const int v = atoi(q); if (v > bnd[np - 1] && v < nb) bnd[np++] = v; while (*q && *q != ',') ++q; if (*q == ',') ++q; }
            } else if (nb >= 14) { bnd[1] = nb / 4; bnd[2] = nb * 9 / 16; np = 3; }
            bool fits = true;
            for (int k = 0; k < np; ++k) {
                const int jb = bnd[k], je = k + 1 < np ? bnd[k + 1] : nb;
                if (cp_colblocks(jb ? jb : 1, je, nb) > CP_MAXB || (jb && (nb - jb) * 128 > 2048)) fits = false;
            }
            if (!fits) { np = 1; fits = cp_colblocks(1, nb, nb) <= CP_MAXB; }
            if (fits) {
                size_t l2 = cp_factor_lds_doubles(C) * sizeof(double);
                if (lds_role > l2) l2 = lds_role;
                static size_t l2_set = 0;        // (once per size: the attribute call is not free, and not needed again)
                if (l2 > l2_set) {
                    e = hipFuncSetAttribute(reinterpret_cast<const void*>(cholesky_phased_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)l2);
                    if (e != hipSuccess) return e;
                    l2_set = l2;
                }
                const int nwg = groups * (np + TI_WG);
                hipLaunchKernelGGL(cholesky_phased_kernel, dim3(nwg), dim3(1024), l2, st, T, tmp, C, W, rows, groups, np, bnd[1] | (bnd[2] << 8) | (bnd[3] << 16));
                return hipGetLastError();
            }
        }
        hipLaunchKernelGGL(cholesky_fused_kernel, dim3(groups * (1 + TI_WG)), dim3(1024), lds, st, T, tmp, C, ldp, W, rows, groups);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(cholesky_fused_kernel, dim3(groups), dim3(1024), lds, st, T, tmp, C, ldp, W, (unsigned*)nullptr, groups);
    if (split) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(tri_inverse_split_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_role);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(tri_inverse_split_kernel, dim3(TI_WG, groups), dim3(1024), lds_role, st, (const double*)T, (const double*)tmp, W, C);
        return hipGetLastError();
    }
    const int nwg = ((nb >> 1) + 3) / 4;
    const size_t lds2 = (size_t)(2 * 16 * (C + 2) + nb * 16 * 17) * sizeof(double);
#define WC_TI_LAUNCH(PER_)                                                                                             \
    do {                                                                                                                \
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(tri_inverse_cols_kernel<PER_>),                           \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);                                 \
        if (e != hipSuccess) return e;                                                                                  \
        hipLaunchKernelGGL((tri_inverse_cols_kernel<PER_>), dim3(nwg, groups), dim3(512), lds2, st, (const double*)T,   \
                           (const double*)tmp, W, C);                                                                   \
    } while (0)
    switch ((C + 63) / 64) {
        case 1: WC_TI_LAUNCH(1); break;
        case 2: WC_TI_LAUNCH(2); break;
        case 3: WC_TI_LAUNCH(3); break;
        default: WC_TI_LAUNCH(4); break;
    }
#undef WC_TI_LAUNCH
    return hipGetLastError();
}

bool wc_factor_is_fused(int C) { return use_fused_factor(C); }

hipError_t wc_launch_cholesky(double* T, int C, int groups, hipStream_t st)
{
    static const bool no_reg = getenv("WC_CHOL_GLOBAL") != nullptr;      // development: the global-memory form
    if (C <= 256 && !no_reg) {
        const int ldr = C + 2;
        const size_t ldsr = (size_t)(C * 17 + 16 * 17 + 16 + 16 * ldr) * sizeof(double);
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(cholesky_reg_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsr);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(cholesky_reg_kernel, dim3(groups), dim3(512), ldsr, st, T, C, ldr);
        return hipGetLastError();
    }
    int ldp = C - CH_NB;
    if (ldp < 4) ldp = 4;
    ldp = (ldp + 3) / 4 * 4;
    const size_t lds = (size_t)(16 * 17 + 16 + 16 * ldp) * sizeof(double);
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(cholesky_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(cholesky_kernel, dim3(groups), dim3(1024), lds, st, T, C, ldp);
    return hipGetLastError();
}

hipError_t wc_launch_gemm(const WcGemm& g, hipStream_t st)
{
    const dim3 grid(g.m / 32, g.n / 32, g.batch * (g.batch2 > 0 ? g.batch2 : 1));
    if (g.a_is_f32 && g.b_is_f32) hipLaunchKernelGGL((gemm_f64_kernel<float, float>), grid, dim3(256), 0, st, g);
    else if (g.a_is_f32) hipLaunchKernelGGL((gemm_f64_kernel<float, double>), grid, dim3(256), 0, st, g);
    else if (g.b_is_f32) hipLaunchKernelGGL((gemm_f64_kernel<double, float>), grid, dim3(256), 0, st, g);
    else hipLaunchKernelGGL((gemm_f64_kernel<double, double>), grid, dim3(256), 0, st, g);
    return hipGetLastError();
}

__global__ void sum_partials_kernel(const double* __restrict__ part, int nparts, int64_t n, double* __restrict__ out)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    double s = 0.0;
    for (int p = 0; p < nparts; ++p) s += part[(int64_t)p * n + e];
    out[e] = s;
}
hipError_t wc_launch_sum_partials(const double* part, int nparts, int64_t n, double* out, hipStream_t st)
{
    hipLaunchKernelGGL(sum_partials_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, part, nparts, n, out);
    return hipGetLastError();
}

// K5's head: g0 = (double A, double B), g1 = (float A, double B), same m x n tiling
hipError_t wc_launch_gemm_pair_dd_fd(const WcGemm& g0, const WcGemm& g1, hipStream_t st)
{
    const int nz0 = g0.batch * (g0.batch2 > 0 ? g0.batch2 : 1), nz1 = g1.batch * (g1.batch2 > 0 ? g1.batch2 : 1);
    const dim3 grid(g0.m / 32, g0.n / 32, nz0 + nz1);
    hipLaunchKernelGGL((gemm_f64_pair_kernel<double, double, float, double>), grid, dim3(256), 0, st, g0, g1, nz0);
    return hipGetLastError();
}

// W = L^-1: invert the 32-wide diagonal blocks, then double the block size level by level:
//   [L11 0; L21 L22]^-1 = [W11 0; -W22 L21 W11, W22]
hipError_t wc_launch_tri_inverse(const double* L, double* W, double* tmp, int C, int groups, hipStream_t st)
{
    const int64_t CCg = (int64_t)C * C;
    hipError_t e = hipSuccess;
    hipLaunchKernelGGL(tri_inv_diag_kernel, dim3(C / 32, groups), dim3(256), 0, st, L, W, C);
    for (int b = 32; b < C; b *= 2) {
        const int nfull = C / (2 * b);
        const int rem = C - 2 * b * nfull;
        for (int pass = 0; pass < 2; ++pass) {
            // pass 0: the nfull complete pairs (batched); pass 1: a trailing partial pair, if any
            int start, mrows, batch;
            if (pass == 0) { if (nfull == 0) continue; start = 0; mrows = b; batch = nfull; }
            else { if (rem <= b) continue; start = 2 * b * nfull; mrows = rem - b; batch = 1; }
            const int64_t pair = (int64_t)2 * b * (C + 1);
            WcGemm g1 = {};
            g1.A = L + (int64_t)(start + b) * C + start; g1.a_rs = C; g1.a_cs = 1; g1.a_bs = pair;
            g1.B = W + (int64_t)start * C + start;       g1.b_rs = C; g1.b_cs = 1; g1.b_bs = pair;
            g1.Cm = tmp; g1.c_rs = b; g1.c_cs = 1; g1.c_bs = (int64_t)b * b;
            g1.m = mrows; g1.n = b; g1.k = b; g1.batch = batch; g1.nred = 1; g1.alpha = 1.0; g1.epi = WC_EPI_NONE;
            g1.batch2 = groups; g1.a_b2s = CCg; g1.b_b2s = CCg; g1.c_b2s = CCg;      // tmp is [group][C*C]
            e = wc_launch_gemm(g1, st);
            if (e != hipSuccess) return e;
            WcGemm g2 = {};
            g2.A = W + (int64_t)(start + b) * C + (start + b); g2.a_rs = C; g2.a_cs = 1; g2.a_bs = pair;
            g2.B = tmp; g2.b_rs = b; g2.b_cs = 1; g2.b_bs = (int64_t)b * b;
            g2.Cm = W + (int64_t)(start + b) * C + start; g2.c_rs = C; g2.c_cs = 1; g2.c_bs = pair;
            g2.m = mrows; g2.n = b; g2.k = mrows; g2.batch = batch; g2.nred = 1; g2.alpha = -1.0; g2.epi = WC_EPI_NONE;
            g2.batch2 = groups; g2.a_b2s = CCg; g2.b_b2s = CCg; g2.c_b2s = CCg;
            e = wc_launch_gemm(g2, st);
            if (e != hipSuccess) return e;
        }
    }
    return hipGetLastError();
}

hipError_t wc_launch_transpose_to_f32(const double* W, int C, int groups, float* A, float* At, hipStream_t st)
{
    hipLaunchKernelGGL(transpose_to_f32_kernel, dim3(C / 32, C / 32, groups), dim3(256), 0, st, W, C, A, At);
    return hipGetLastError();
}

hipError_t wc_launch_bwd_tail(const double* Q, int C, double scale, float* S, const double* gsum, const float* A, int Kc,
                              int64_t M, float* gmean, float* dbeta, hipStream_t st)
{
    hipLaunchKernelGGL(bwd_tail_kernel, dim3((C + 127) / 128, 2 * C), dim3(128), 0, st, Q, C, scale, S, gsum, A, Kc, M, gmean, dbeta);
    return hipGetLastError();
}

hipError_t wc_launch_group_bias(const float* mu, const float* A, const float* beta, int G, int Kc, int C, int per_group,
                                float* center, float* bias, hipStream_t st, const float* center_in)
{
    hipLaunchKernelGGL(group_bias_kernel, dim3(G * Kc), dim3(1024), 0, st, mu, A, beta, G, Kc, C, per_group, center, bias, center_in);
    return hipGetLastError();
}

hipError_t wc_launch_f64_to_f32(const double* src, float* dst, int64_t n, hipStream_t st)
{
    hipLaunchKernelGGL(f64_to_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, dst, n);
    return hipGetLastError();
}


// K4 on a pre-split x (wc_fast_xty.hip, XPL): the kernel reduced g / scale = x - center; R needs f = x - mu:
//     R[k] += (center - mu) gsum[k]^T          (float64, one thread per element)
__global__ __launch_bounds__(256) void rank1_add_kernel(double* __restrict__ R, const double* __restrict__ gsum,
                                                        const float* __restrict__ u, const float* __restrict__ v, int C)
{
    const int k = blockIdx.y;
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (int64_t)C * C) return;
    const int i = (int)(e / C), j = (int)(e % C);
    R[(int64_t)k * C * C + e] += ((double)u[i] - (double)v[i]) * gsum[(int64_t)k * C + j];
}

hipError_t wc_launch_rank1_add(double* R, const double* gsum, const float* u, const float* v, int C, int Kc, hipStream_t st)
{
    hipLaunchKernelGGL(rank1_add_kernel, dim3((C * C + 255) / 256, Kc), dim3(256), 0, st, R, gsum, u, v, C);
    return hipGetLastError();
}
