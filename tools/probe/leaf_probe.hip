// Development probe (round 6): the 16 x 16 float64 leaf of K2 (factor a diagonal block + invert the factor on ONE wave with DPP row
// broadcasts) in isolation -- cycles per leaf for schedule variants, and the issue cost of the instructions it is made of.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/probe/bin/leaf_probe tools/probe/leaf_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <type_traits>
#include <vector>
typedef double f64x4 __attribute__((ext_vector_type(4)));

template <int K> __device__ __forceinline__ double row_bcast(double v)
{
    double r;
    asm("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v), "n"(K));
    return r;
}
template <int K> __device__ __forceinline__ void fmac_bcast(double& acc, double src, double mul)
{
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(mul), "n"(K));
}
__device__ __forceinline__ void dpp_settle(double& v) { asm("s_nop 1" : "+v"(v)); }
template <int B, int E, typename F> __device__ __forceinline__ void static_for(F&& f)
{
    if constexpr (B < E) { f(std::integral_constant<int, B>{}); static_for<B + 1, E>(f); }
}
#define SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
__device__ __forceinline__ void fmac_plain(double& acc, double s, double m) { asm("v_fmac_f64 %0, %1, %2" : "+v"(acc) : "v"(s), "v"(m)); }
__device__ __forceinline__ void mul_dep(double& a, double m) { asm("v_mul_f64 %0, %1, %2" : "=v"(a) : "v"(a), "v"(m)); }
__device__ __forceinline__ void mul_to(double& d, double a, double m) { asm("v_mul_f64 %0, %1, %2" : "=v"(d) : "v"(a), "v"(m)); }
__device__ __forceinline__ void rsq_dep(double& a) { asm("v_rsq_f64 %0, %1" : "=v"(a) : "v"(a)); }

// ---- V0: the leaf as cholesky_fused_kernel runs it (round 2-5) ----------------------------------------------------------
__device__ __forceinline__ void leaf_v0(const double* src, double* Lout, double* Wout, int li)
{
    double a[16], w[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) a[c] = src[li * 17 + c];
    static_for<0, 16>([&](auto J) {
        constexpr int jj = decltype(J)::value;
        const double p = row_bcast<jj>(a[jj]);
        double rd = __builtin_amdgcn_rsq(p);
        rd = rd * (1.5 - 0.5 * p * rd * rd);
        a[jj] = (li == jj) ? p * rd : a[jj] * rd;
        dpp_settle(a[jj]);
        const double nj = -a[jj];
        static_for<jj + 1, 16>([&](auto K) { constexpr int k = decltype(K)::value; fmac_bcast<k>(a[k], a[jj], nj); });
        double acc = 0.0;
        static_for<0, jj>([&](auto K) { constexpr int k = decltype(K)::value; fmac_bcast<jj>(acc, a[k], w[k]); });
        w[jj] = (li == jj) ? rd : (li < jj ? -acc * rd : 0.0);
    });
#pragma unroll
    for (int c = 0; c < 16; ++c) { Lout[li * 17 + c] = (c <= li) ? a[c] : 0.0; Wout[c * 17 + li] = w[c]; }
}

// ---- V1: the next pivot's rsq + Newton chain starts as soon as ITS column has the current pivot's update; the other trailing
//      updates and the inverse row fill its latency (program order pinned by scheduling fences) -------------------------------------
template <bool INV>
__device__ __forceinline__ void leaf_v1(const double* src, double* Lout, double* Wout, int li)
{
    double a[16], w[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) a[c] = src[li * 17 + c];
    double p = row_bcast<0>(a[0]);
    double rd = __builtin_amdgcn_rsq(p);
    rd = rd * (1.5 - 0.5 * p * rd * rd);
    static_for<0, 16>([&](auto J) {
        constexpr int jj = decltype(J)::value;
        a[jj] = (li == jj) ? p * rd : a[jj] * rd;
        dpp_settle(a[jj]);
        const double nj = -a[jj];
        const double rd_now = rd;
        double hp = 0.0, rn = 0.0;
        if constexpr (jj < 15) {
            fmac_bcast<jj + 1>(a[jj + 1], a[jj], nj);
            p = row_bcast<jj + 1>(a[jj + 1]);
            rn = __builtin_amdgcn_rsq(p);
            hp = 0.5 * p;
        }
        SCHED_FENCE();
        // fill 1: half of the remaining trailing updates
        static_for<jj + 2, (jj + 2 + 16) / 2>([&](auto K) { constexpr int k = decltype(K)::value; fmac_bcast<k>(a[k], a[jj], nj); });
        SCHED_FENCE();
        double t2 = 0.0;
        if constexpr (jj < 15) t2 = hp * rn;
        SCHED_FENCE();
        static_for<(jj + 2 + 16) / 2, 16>([&](auto K) { constexpr int k = decltype(K)::value; fmac_bcast<k>(a[k], a[jj], nj); });
        SCHED_FENCE();
        double t3 = 0.0;
        if constexpr (jj < 15) t3 = __builtin_fma(-rn, t2, 1.5);
        SCHED_FENCE();
        double acc = 0.0;
        if constexpr (INV) {
            static_for<0, jj>([&](auto K) { constexpr int k = decltype(K)::value; fmac_bcast<jj>(acc, a[k], w[k]); });
        }
        SCHED_FENCE();
        if constexpr (jj < 15) rd = rn * t3;
        if constexpr (INV) w[jj] = (li == jj) ? rd_now : (li < jj ? -acc * rd_now : 0.0);
        else w[jj] = rd_now;
        SCHED_FENCE();
    });
#pragma unroll
    for (int c = 0; c < 16; ++c) { Lout[li * 17 + c] = (c <= li) ? a[c] : 0.0; Wout[c * 17 + li] = w[c]; }
}

// ---- V2: factor + TWO row-specific payloads solved in the same instruction stream.  Every 16-lane row keeps its own copy of the
//      diagonal block (a[]) and a payload block b[] with lane = payload row: b <- b L^-T column by column (the panel solve's
//      recurrence).  Payload = identity gives L^-T, i.e. register k of lane r holds Linv[k][r]: the inverse without its own FMAs'
//      dependence chain (all of a pivot's FMAs have the same multiplicand a[jj]); payload = 16 rows of the raw panel gives those
//      rows SOLVED (what wave 0 computes with four MFMAs + an LDS round trip today). ------------------------------------------------
__device__ __forceinline__ void leaf_v2(const double* src, const double* pay, double* Lout, double* Wout, double* Xout, int li, int lq)
{
    double a[16], b[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) a[c] = src[li * 17 + c];
    // rows 0, 2: identity (-> inverse), rows 1, 3: payload rows
#pragma unroll
    for (int c = 0; c < 16; ++c) b[c] = (lq & 1) ? pay[li * 17 + c] : (c == li ? 1.0 : 0.0);
    double p = row_bcast<0>(a[0]);
    double rd = __builtin_amdgcn_rsq(p);
    rd = rd * (1.5 - 0.5 * p * rd * rd);
    static_for<0, 16>([&](auto J) {
        constexpr int jj = decltype(J)::value;
        a[jj] = (li == jj) ? p * rd : a[jj] * rd;
        dpp_settle(a[jj]);
        const double nj = -a[jj];
        b[jj] = b[jj] * rd;
        const double nb = -b[jj];
        double hp = 0.0, rn = 0.0;
        if constexpr (jj < 15) {
            fmac_bcast<jj + 1>(a[jj + 1], a[jj], nj);
            p = row_bcast<jj + 1>(a[jj + 1]);
            rn = __builtin_amdgcn_rsq(p);
            hp = 0.5 * p;
        }
        SCHED_FENCE();
        static_for<jj + 2, 16>([&](auto K) { constexpr int k = decltype(K)::value; fmac_bcast<k>(a[k], a[jj], nj); });
        SCHED_FENCE();
        double t2 = 0.0;
        if constexpr (jj < 15) t2 = hp * rn;
        SCHED_FENCE();
        static_for<jj + 1, 16>([&](auto K) { constexpr int k = decltype(K)::value; fmac_bcast<k>(b[k], a[jj], nb); });
        SCHED_FENCE();
        if constexpr (jj < 15) { const double t3 = __builtin_fma(-rn, t2, 1.5); rd = rn * t3; }
        SCHED_FENCE();
    });
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        if (lq == 0) { Lout[li * 17 + c] = (c <= li) ? a[c] : 0.0; Wout[c * 17 + li] = b[c]; }
        if (lq == 1) Xout[li * 17 + c] = b[c];
    }
}


// ---- V3 / V4 (round 6): the pivot chain cut to  fmac -> rsq_dpp -> mul -> fma -> mul.  (a) v_rsq_f64_dpp reads the pivot through the row
//      broadcast itself (no v_mov_b64_dpp: 29 cycles of the chain); (b) the column is scaled by rn first and by the Newton factor c after
//      (a * rn runs beside the Newton step, not behind it); (c) lane jj's a[jj] IS the pivot: no select; (d) the negation rides on the
//      FMA's source modifier; (e) two independent FMAs stand where the DPP read hazard needs wait states.
//      V3: inverse rows as today (w[jj] = -acc * rd, acc seeded with -1 in lane jj: no selects).  V4: the inverse as a payload block.
template <int K> __device__ __forceinline__ void fmac_nbcast(double& acc, double src, double mul)      // acc -= bcast_K(src) * mul
{
    asm volatile("v_fmac_f64_dpp %0, %1, -%2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(mul), "n"(K));
}
// a1 -= bcast_K(col) * col ; [two more independent updates] ; rn = rsq(bcast_K(a1)) ; hp = 0.5 * bcast_K(a1)
template <int K, int K2, int K3>
__device__ __forceinline__ void head3(double& a1, double& a2, double& a3, double col, double& rn, double& hp, double half)
{
    asm volatile("v_fmac_f64_dpp %0, %5, -%5 row_newbcast:%7 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %1, %5, -%5 row_newbcast:%8 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %2, %5, -%5 row_newbcast:%9 row_mask:0xf bank_mask:0xf\n\t"
                 "v_rsq_f64_dpp %3, %0 row_newbcast:%7 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %4, %0, %6 row_newbcast:%7 row_mask:0xf bank_mask:0xf"
                 : "+v"(a1), "+v"(a2), "+v"(a3), "=&v"(rn), "+v"(hp) : "v"(col), "v"(half), "n"(K), "n"(K2), "n"(K3));
}
template <int K>
__device__ __forceinline__ void head1(double& a1, double col, double& rn, double& hp, double half)
{
    asm volatile("v_fmac_f64_dpp %0, %3, -%3 row_newbcast:%5 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_rsq_f64_dpp %1, %0 row_newbcast:%5 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %2, %0, %4 row_newbcast:%5 row_mask:0xf bank_mask:0xf"
                 : "+v"(a1), "=&v"(rn), "+v"(hp) : "v"(col), "v"(half), "n"(K));
}
template <bool PAYLOAD>
__device__ __forceinline__ void leaf_v3(const double* src, double* Lout, double* Wout, int li)
{
    double a[16], w[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) a[c] = src[li * 17 + c];
    if constexpr (PAYLOAD) {
#pragma unroll
        for (int c = 0; c < 16; ++c) w[c] = (c == li) ? 1.0 : 0.0;
    }
    double half = 0.5;
    asm volatile("" : "+v"(half));
    double rn, hp = 0.0;
    asm volatile("v_rsq_f64_dpp %0, %2 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %1, %2, %3 row_newbcast:0 row_mask:0xf bank_mask:0xf" : "=&v"(rn), "+v"(hp) : "v"(a[0]), "v"(half));
    static_for<0, 16>([&](auto J) {
        constexpr int jj = decltype(J)::value;
        const double t = hp * rn;
        const double ua = a[jj] * rn;
        double ub = 0.0;
        if constexpr (PAYLOAD) ub = w[jj] * rn;
        const double c = __builtin_fma(-rn, t, 1.5);
        a[jj] = ua * c;
        if constexpr (PAYLOAD) w[jj] = ub * c;
        const double rd = rn * c;
        dpp_settle(a[jj]);
        hp = 0.0;
        if constexpr (jj + 3 < 16) head3<jj + 1, jj + 2, jj + 3>(a[jj + 1], a[jj + 2], a[jj + 3], a[jj], rn, hp, half);
        else if constexpr (jj + 1 < 16) {
            head1<jj + 1>(a[jj + 1], a[jj], rn, hp, half);
            if constexpr (jj + 2 < 16) fmac_nbcast<jj + 2>(a[jj + 2], a[jj], a[jj]);
        }
        static_for<jj + 4, 16>([&](auto K) { constexpr int k = decltype(K)::value; fmac_nbcast<k>(a[k], a[jj], a[jj]); });
        if constexpr (PAYLOAD) {
            static_for<jj + 1, 16>([&](auto K) { constexpr int k = decltype(K)::value; fmac_nbcast<k>(w[k], a[jj], w[jj]); });
        } else {
            double acc = (li == jj) ? -1.0 : 0.0;
            static_for<0, jj>([&](auto K) { constexpr int k = decltype(K)::value; fmac_bcast<jj>(acc, a[k], w[k]); });
            w[jj] = -acc * rd;
        }
    });
#pragma unroll
    for (int c = 0; c < 16; ++c) { Lout[li * 17 + c] = (c <= li) ? a[c] : 0.0; Wout[c * 17 + li] = w[c]; }
}


// ---- V5 (round 6): V3 with a LEGAL pivot broadcast -- v_rsq_f64_dpp assembles but the hardware returns garbage (dpp_check below); the
//      pivot travels through v_fmac_f64_dpp instead (p = 0 + bcast(a) * 1.0), half of it the same way, both beside the column's first update.
template <int K, int K2, int K3>
__device__ __forceinline__ void head3b(double& a1, double& a2, double& a3, double col, double& p, double& hp, double one, double half)
{
    asm volatile("v_fmac_f64_dpp %0, %5, -%5 row_newbcast:%8 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %1, %5, -%5 row_newbcast:%9 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %2, %5, -%5 row_newbcast:%10 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %3, %0, %6 row_newbcast:%8 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %4, %0, %7 row_newbcast:%8 row_mask:0xf bank_mask:0xf"
                 : "+v"(a1), "+v"(a2), "+v"(a3), "+v"(p), "+v"(hp) : "v"(col), "v"(one), "v"(half), "n"(K), "n"(K2), "n"(K3));
}
template <int K>
__device__ __forceinline__ void head1b(double& a1, double col, double& p, double& hp, double one, double half)
{
    asm volatile("v_fmac_f64_dpp %0, %3, -%3 row_newbcast:%6 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_fmac_f64_dpp %1, %0, %4 row_newbcast:%6 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %2, %0, %5 row_newbcast:%6 row_mask:0xf bank_mask:0xf"
                 : "+v"(a1), "+v"(p), "+v"(hp) : "v"(col), "v"(one), "v"(half), "n"(K));
}
__device__ __forceinline__ void leaf_v5(const double* src, double* Lout, double* Wout, int li)
{
    double a[16], w[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) a[c] = src[li * 17 + c];
    double half = 0.5, one = 1.0;
    asm volatile("" : "+v"(half), "+v"(one));
    double p = 0.0, hp = 0.0;
    asm volatile("v_fmac_f64_dpp %0, %2, %3 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %1, %2, %4 row_newbcast:0 row_mask:0xf bank_mask:0xf" : "+v"(p), "+v"(hp) : "v"(a[0]), "v"(one), "v"(half));
    static_for<0, 16>([&](auto J) {
        constexpr int jj = decltype(J)::value;
        const double rn = __builtin_amdgcn_rsq(p);
        const double t = hp * rn;
        const double ua = a[jj] * rn;
        const double c = __builtin_fma(-rn, t, 1.5);
        a[jj] = ua * c;
        const double rd = rn * c;
        dpp_settle(a[jj]);
        p = 0.0; hp = 0.0;
        if constexpr (jj + 3 < 16) head3b<jj + 1, jj + 2, jj + 3>(a[jj + 1], a[jj + 2], a[jj + 3], a[jj], p, hp, one, half);
        else if constexpr (jj + 1 < 16) {
            head1b<jj + 1>(a[jj + 1], a[jj], p, hp, one, half);
            if constexpr (jj + 2 < 16) fmac_nbcast<jj + 2>(a[jj + 2], a[jj], a[jj]);
        }
        static_for<jj + 4, 16>([&](auto K) { constexpr int k = decltype(K)::value; fmac_nbcast<k>(a[k], a[jj], a[jj]); });
        double acc = (li == jj) ? -1.0 : 0.0;
        static_for<0, jj>([&](auto K) { constexpr int k = decltype(K)::value; fmac_bcast<jj>(acc, a[k], w[k]); });
        w[jj] = -acc * rd;
    });
#pragma unroll
    for (int c = 0; c < 16; ++c) { Lout[li * 17 + c] = (c <= li) ? a[c] : 0.0; Wout[c * 17 + li] = w[c]; }
}

// ---------------------------------------------------------------------------------------------------------------------------------------
template <int V>
__global__ __launch_bounds__(1024) void leaf_kernel(const double* A, const double* P, double* L, double* W, double* X, unsigned long long* cyc,
                                                    int reps, int busy_waves)
{
    __shared__ double src[16 * 17], pay[16 * 17], lo[16 * 17], wo[16 * 17], xo[16 * 17];
    __shared__ double junk[16 * 272];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lq = lane >> 4;
    for (int e = tid; e < 256; e += blockDim.x) { src[(e >> 4) * 17 + (e & 15)] = A[e]; pay[(e >> 4) * 17 + (e & 15)] = P[e]; }
    for (int e = tid; e < 16 * 272; e += blockDim.x) junk[e] = 1e-3 * (e % 13);
    __syncthreads();
    if (wave == 0) {
        __builtin_amdgcn_s_setprio(3);
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        for (int r = 0; r < reps; ++r) {
            if constexpr (V == 0) leaf_v0(src, lo, wo, li);
            if constexpr (V == 1) leaf_v1<true>(src, lo, wo, li);
            if constexpr (V == 2) leaf_v2(src, pay, lo, wo, xo, li, lq);
            if constexpr (V == 3) leaf_v1<false>(src, lo, wo, li);
            if constexpr (V == 4) leaf_v3<false>(src, lo, wo, li);
            if constexpr (V == 5) leaf_v3<true>(src, lo, wo, li);
            if constexpr (V == 6) leaf_v5(src, lo, wo, li);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        if (lane == 0) cyc[0] = (t1 - t0) / reps;
        if (lane < 16)
            for (int c = 0; c < 16; ++c) { L[li * 16 + c] = lo[li * 17 + c]; W[li * 16 + c] = wo[li * 17 + c]; X[li * 16 + c] = xo[li * 17 + c]; }
    } else if ((wave & 3) == 0 && (wave >> 2) <= busy_waves) {
        // waves 4, 8, 12 share wave 0's SIMD: f64 MFMAs with LDS operands, as the trailing update's owners
        f64x4 acc = {0.0, 0.0, 0.0, 0.0};
        const double* pa = junk + li * 17 + lq;
        for (int r = 0; r < reps * 8; ++r) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[4 * kk + 17 * (r & 15)], pa[4 * kk + 64], acc, 0, 0, 0);
        }
        if (acc[0] == 12345.0) L[300] = acc[1];
    }
}

// issue cost of the ingredients: N back-to-back instructions, independent or one dependent chain
template <int WHAT>
__global__ void instr_kernel(double* out, unsigned long long* cyc)
{
    double a[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) a[c] = 1.0 + 1e-3 * (threadIdx.x + c);
    double s = 0.5 + 1e-4 * threadIdx.x, m = 1.0 - 1e-5 * threadIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < 64; ++r) {
        if constexpr (WHAT == 0) static_for<0, 16>([&](auto K) { constexpr int k = decltype(K)::value; fmac_bcast<k>(a[k], s, m); });              // 16 independent DPP FMAs
        if constexpr (WHAT == 1) static_for<0, 16>([&](auto K) { constexpr int k = decltype(K)::value; fmac_plain(a[k], s, m); });
        if constexpr (WHAT == 2) static_for<0, 16>([&](auto K) { fmac_plain(a[0], s, m); });                     // dependent chain, plain
        if constexpr (WHAT == 3) static_for<0, 16>([&](auto K) { fmac_bcast<3>(a[0], s, m); });                                                       // dependent chain through the accumulator, DPP
        if constexpr (WHAT == 4) static_for<0, 16>([&](auto K) { mul_dep(a[0], m); });                     // dependent multiplies
        if constexpr (WHAT == 5) static_for<0, 16>([&](auto K) { rsq_dep(a[0]); });                                 // dependent rsq
        if constexpr (WHAT == 6) static_for<0, 16>([&](auto K) { constexpr int k = decltype(K)::value; rsq_dep(a[k]); });   // independent rsq
        if constexpr (WHAT == 7) static_for<0, 16>([&](auto K) { dpp_settle(a[0]); a[0] = row_bcast<5>(a[0]); });                                     // dependent broadcast (with its wait states)
        if constexpr (WHAT == 8) static_for<0, 16>([&](auto K) { dpp_settle(a[0]); fmac_bcast<3>(a[1], a[0], m); mul_to(a[0], a[1], m); });  // mul -> settle -> DPP read
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double t = 0.0;
#pragma unroll
    for (int c = 0; c < 16; ++c) t += a[c];
    out[threadIdx.x] = t;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}


// does the hardware do what the assembler accepts?  per lane: [0] v_rsq_f64_dpp row_newbcast:5, [1] v_fmac_f64_dpp with -src1, [2] src0 == src1 register
__global__ void dpp_check_kernel(double* out)
{
    const int lane = threadIdx.x;
    double x = 1.0 + 0.25 * lane, r = 0.0, acc = 10.0, acc2 = 10.0, m = 2.0 + lane;
    asm volatile("s_nop 1\n\tv_rsq_f64_dpp %0, %1 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "=&v"(r) : "v"(x));
    asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, -%2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x), "v"(m));
    asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, -%1 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(acc2) : "v"(x));
    out[3 * lane] = r; out[3 * lane + 1] = acc; out[3 * lane + 2] = acc2;
}

int main()
{
    {
        double* d; hipMalloc(&d, 64 * 3 * 8);
        hipLaunchKernelGGL(dpp_check_kernel, dim3(1), dim3(64), 0, 0, d);
        std::vector<double> h(192); hipMemcpy(h.data(), d, 192 * 8, hipMemcpyDeviceToHost);
        double e0 = 0, e1 = 0, e2 = 0;
        for (int l = 0; l < 64; ++l) {
            const double xb = 1.0 + 0.25 * ((l & ~15) + 5), x = 1.0 + 0.25 * l, m = 2.0 + l;
            e0 = fmax(e0, fabs(h[3 * l] - 1.0 / sqrt(xb)) * sqrt(xb)); e1 = fmax(e1, fabs(h[3 * l + 1] - (10.0 - xb * m))); e2 = fmax(e2, fabs(h[3 * l + 2] - (10.0 - xb * x)));
        }
        printf("v_rsq_f64_dpp rel err %.2e (lane 7 got %.6f, want %.6f); fmac_dpp -src1 err %.2e; src0 == src1 err %.2e\n", e0, h[21], 1.0 / sqrt(1.0 + 0.25 * 5), e1, e2);
    }

    // an SPD 16 x 16 block and a payload
    std::vector<double> A(256), P(256), Lr(256, 0.0), Wr(256, 0.0), Xr(256);
    srand(3);
    std::vector<double> B(256);
    for (auto& v : B) v = (rand() / (double)RAND_MAX) - 0.5;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = (i == j) ? 0.5 : 0.0; for (int k = 0; k < 16; ++k) s += B[i * 16 + k] * B[j * 16 + k]; A[i * 16 + j] = s; }
    for (auto& v : P) v = (rand() / (double)RAND_MAX) - 0.5;
    for (int j = 0; j < 16; ++j) {
        double d = A[j * 16 + j]; for (int k = 0; k < j; ++k) d -= Lr[j * 16 + k] * Lr[j * 16 + k];
        Lr[j * 16 + j] = sqrt(d);
        for (int i = j + 1; i < 16; ++i) { double s = A[i * 16 + j]; for (int k = 0; k < j; ++k) s -= Lr[i * 16 + k] * Lr[j * 16 + k]; Lr[i * 16 + j] = s / Lr[j * 16 + j]; }
    }
    for (int c = 0; c < 16; ++c)
        for (int i = 0; i < 16; ++i) { double s = (i == c) ? 1.0 : 0.0; for (int k = 0; k < i; ++k) s -= Lr[i * 16 + k] * Wr[k * 16 + c]; Wr[i * 16 + c] = s / Lr[i * 16 + i]; }
    for (int r = 0; r < 16; ++r)      // X = P L^-T
        for (int k = 0; k < 16; ++k) { double s = P[r * 16 + k]; for (int m = 0; m < k; ++m) s -= Xr[r * 16 + m] * Lr[k * 16 + m]; Xr[r * 16 + k] = s / Lr[k * 16 + k]; }
    double *dA, *dP, *dL, *dW, *dX; unsigned long long* dc;
    hipMalloc(&dA, 2048); hipMalloc(&dP, 2048); hipMalloc(&dL, 4096); hipMalloc(&dW, 2048); hipMalloc(&dX, 2048); hipMalloc(&dc, 64);
    hipMemcpy(dA, A.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(dP, P.data(), 2048, hipMemcpyHostToDevice);
    auto check = [&](const char* name, int busy, bool has_x, bool has_w) {
        std::vector<double> L(256), W(256), X(256); unsigned long long c;
        hipDeviceSynchronize();
        hipMemcpy(L.data(), dL, 2048, hipMemcpyDeviceToHost); hipMemcpy(W.data(), dW, 2048, hipMemcpyDeviceToHost);
        hipMemcpy(X.data(), dX, 2048, hipMemcpyDeviceToHost); hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
        double eL = 0, eW = 0, eX = 0;
        for (int i = 0; i < 256; ++i) { eL = fmax(eL, fabs(L[i] - Lr[i])); eW = fmax(eW, fabs(W[i] - Wr[i])); eX = fmax(eX, fabs(X[i] - Xr[i])); }
        printf("%-34s busy MFMA waves on the SIMD %d: %6llu cycles per leaf   max err L %.1e  W %.1e%s  X %.1e%s\n", name, busy, c, eL, eW, has_w ? "" : " (n/a)",
               eX, has_x ? "" : " (n/a)");
    };
    for (int busy : {0, 3}) {
        hipLaunchKernelGGL(leaf_kernel<0>, dim3(1), dim3(1024), 0, 0, dA, dP, dL, dW, dX, dc, 200, busy); check("V0 current", busy, false, true);
        hipLaunchKernelGGL(leaf_kernel<1>, dim3(1), dim3(1024), 0, 0, dA, dP, dL, dW, dX, dc, 200, busy); check("V1 pipelined pivots", busy, false, true);
        hipLaunchKernelGGL(leaf_kernel<3>, dim3(1), dim3(1024), 0, 0, dA, dP, dL, dW, dX, dc, 200, busy); check("V1 without the inverse", busy, false, false);
        hipLaunchKernelGGL(leaf_kernel<2>, dim3(1), dim3(1024), 0, 0, dA, dP, dL, dW, dX, dc, 200, busy); check("V2 inverse + panel rows as payloads", busy, true, true);
        hipLaunchKernelGGL(leaf_kernel<4>, dim3(1), dim3(1024), 0, 0, dA, dP, dL, dW, dX, dc, 200, busy); check("V3 short chain, inverse rows", busy, false, true);
        hipLaunchKernelGGL(leaf_kernel<5>, dim3(1), dim3(1024), 0, 0, dA, dP, dL, dW, dX, dc, 200, busy); check("V4 short chain, inverse as payload", busy, false, true);
        hipLaunchKernelGGL(leaf_kernel<6>, dim3(1), dim3(1024), 0, 0, dA, dP, dL, dW, dX, dc, 200, busy); check("V5 short chain, pivot through fmac_dpp", busy, false, true);
    }
    const char* names[] = {"16 independent v_fmac_f64_dpp", "16 independent v_fmac_f64", "16 dependent v_fmac_f64", "16 dependent v_fmac_f64_dpp (acc)",
                           "16 dependent v_mul_f64", "16 dependent v_rsq_f64", "16 independent v_rsq_f64", "16 dependent settle+v_mov_b64_dpp",
                           "16 x (settle, dpp fmac, mul) chain"};
    double* dout; hipMalloc(&dout, 64 * 8);
    auto instr = [&](int what, unsigned long long& c) {
        switch (what) {
        case 0: hipLaunchKernelGGL(instr_kernel<0>, dim3(1), dim3(64), 0, 0, dout, dc); break;
        case 1: hipLaunchKernelGGL(instr_kernel<1>, dim3(1), dim3(64), 0, 0, dout, dc); break;
        case 2: hipLaunchKernelGGL(instr_kernel<2>, dim3(1), dim3(64), 0, 0, dout, dc); break;
        case 3: hipLaunchKernelGGL(instr_kernel<3>, dim3(1), dim3(64), 0, 0, dout, dc); break;
        case 4: hipLaunchKernelGGL(instr_kernel<4>, dim3(1), dim3(64), 0, 0, dout, dc); break;
        case 5: hipLaunchKernelGGL(instr_kernel<5>, dim3(1), dim3(64), 0, 0, dout, dc); break;
        case 6: hipLaunchKernelGGL(instr_kernel<6>, dim3(1), dim3(64), 0, 0, dout, dc); break;
        case 7: hipLaunchKernelGGL(instr_kernel<7>, dim3(1), dim3(64), 0, 0, dout, dc); break;
        case 8: hipLaunchKernelGGL(instr_kernel<8>, dim3(1), dim3(64), 0, 0, dout, dc); break;
        }
        hipDeviceSynchronize(); hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
    };
    for (int w = 0; w < 9; ++w) { unsigned long long c; instr(w, c); instr(w, c); printf("%-40s %.1f cycles per instruction (group)\n", names[w], c / (64.0 * 16)); }
    return 0;
}
