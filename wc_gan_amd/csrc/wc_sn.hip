// Spectral normalisation of a weight matrix as ONE kernel per direction.
//   reference call sites: discriminator.py:26-33, generator.py:104-113 (SNConv2D / SNDense / SNEmbeding of the
//   missing gan.spectral_normalized_layers; knobs spectral_iterations, fully_diff_spectral, run.py:265-270)
//   forward : `iterations` power-iteration steps  v <- normalize(W^T u),  u <- normalize(W v)   (training only),
//             sigma = u^T W v,   w_sn = W / sigma
//   backward: dW = (g - fully_diff * <g, w_sn> u v^T) / sigma         (u, v are constants of the step)
// W is (R, K) row-major: a convolution kernel in the memory order it is stored in (Cout rows; the singular values do
// not depend on the order of the columns, only v is permuted with them).  A critic step evaluates this for every
// layer before every forward pass -- as separate launches it is ~15 small kernels per layer (GEMVs, norms, divisions);
// here up to SN_MAXWG co-resident workgroups split the matrix and meet twice per power-iteration step.
#include "wc_common.h"

namespace {

constexpr int SN_THREADS = 256;
#ifndef SN_MAXWG_
#define SN_MAXWG_ 128
#endif
constexpr int SN_MAXWG = SN_MAXWG_;      // workgroups per weight (32 until round 2: a 1024 x 9216 critic weight of the Tiny-ImageNet recipe then streamed through 32 CUs)

struct SnArgs {
    const float* W; int R, K;
    float* u; float* v;          // [R], [K]: read, and written back when iterations > 0
    int iterations; float eps;
    float* w_sn; float* sigma;   // [R*K], [1]
    float* u_used; float* v_used;   // nullable: copies of u, v as used for sigma (what the backward needs)
    float* t; float* s;          // scratch [K], [R]: the two matrix-vector products, assembled from the workgroups' slices
    unsigned* sync;              // [0] barrier arrivals, [1] workgroups done: both 0 between launches
    int nwg;
    float* amax;                 // [SN_MAXWG]: max |w_sn| of each workgroup's rows (entries >= nwg stay as the caller zeroed them)
};

__device__ __forceinline__ float block_sum(float x, float* red)       // SN_THREADS threads; every thread gets the sum
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();                       // red may still be read from the previous call
    if (lane == 0) red[wave] = x;
    __syncthreads();
    float r = 0.f;
#pragma unroll
    for (int w = 0; w < SN_THREADS / 64; ++w) r += red[w];
    return r;
}

// What the workgroups hand each other (slices of the two matrix-vector products, a few hundred bytes) is stored
// write-through and loaded past the L1 (relaxed agent-scope atomics = sc1 accesses): the XCDs' L2s are not coherent
// and an agent-scope release would first write back everything the kernels before this one left dirty in the L2.
__device__ __forceinline__ void put(float* p, float x)
{ __hip_atomic_store(reinterpret_cast<unsigned*>(p), __float_as_uint(x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float get(const float* p)
{ return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)); }

// all workgroups of the launch (co-resident: at most SN_MAXWG of them) meet here for the `phase`-th time.
// The wait is BOUNDED: the peers of a weight are dispatched in block order, so a resident workgroup only ever
// waits for peers that the dispatcher starts as soon as earlier blocks leave -- but if CUs are held by something that does
// not leave (another process, a kernel on a second stream that spins itself) the meeting would otherwise hang the device.
// After ~2^21 polls (> 0.1 s) the waiter gives up, sets the weight's sticky error word (the last word of its scratch:
// wc_spectral_norm_error_offset) and goes on: the results of this weight are then garbage, the flag says so, nothing hangs.
constexpr unsigned SN_MEET_POLLS = 1u << 21;
__device__ __forceinline__ void grid_meet(unsigned* sync, int nwg, int phase, unsigned* err)
{
    if (nwg == 1) { __syncthreads(); return; }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every storing wave: my write-through stores have landed
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned target = (unsigned)(phase + 1) * (unsigned)nwg;
        unsigned polls = 0;
        while (__hip_atomic_load(sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (++polls > SN_MEET_POLLS) { __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            __builtin_amdgcn_s_sleep(2);
        }
    }
    __syncthreads();
}

// One launch, nwg <= SN_MAXWG workgroups.  Every workgroup owns a COLUMN slice of W for v = N(W^T u) and a ROW slice for
// u = N(W v) and for the final scaling; the two products are assembled in global scratch between grid meetings and
// each workgroup normalises them for itself (K + R floats: nothing), so all sums run in a fixed order.
__device__ __forceinline__ void sn_forward_body(const SnArgs& a, const int b)
{
    extern __shared__ float sm[];
    float* us = sm;                 // [R]
    float* vs = us + a.R;           // [K]
    float* red = vs + a.K;          // [SN_THREADS / 64]
    float* part = red + SN_THREADS / 64;      // [SN_THREADS]: partial column sums
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int R = a.R, K = a.K, nwg = a.nwg;
    const int j0 = (int)((int64_t)K * b / nwg), j1 = (int)((int64_t)K * (b + 1) / nwg);      // my columns
    const int r0 = (int)((int64_t)R * b / nwg), r1 = (int)((int64_t)R * (b + 1) / nwg);      // my rows
    for (int r = tid; r < R; r += SN_THREADS) us[r] = a.u[r];
    for (int j = tid; j < K; j += SN_THREADS) vs[j] = a.v[j];
    __syncthreads();
    int phase = 0;

    auto rows_times_v = [&]() {     // a.s[r0..r1) = W[r0..r1) vs: one wave per row, lanes along the row
        for (int r = r0 + wave; r < r1; r += SN_THREADS / 64) {
            const float* wr = a.W + (int64_t)r * K;
            float p = 0.f;
#pragma unroll 8                    // the loads of a row go out together: this loop is latency, not bandwidth
            for (int j = lane; j < K; j += 64) p = fmaf(wr[j], vs[j], p);
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) p += __shfl_xor(p, off);
            if (lane == 0) put(a.s + r, p);
        }
    };

    float sig_norm = 0.f;
    for (int it = 0; it < a.iterations; ++it) {
        // t[j0..j1) = sum_r W[r][j] u[r]: the slice's columns across the threads, the rows split over thread groups
        const int nc = j1 - j0;
        if (nc > 0) {
            const int groups = SN_THREADS / nc > 0 ? (SN_THREADS / nc < R ? SN_THREADS / nc : R) : 1;
            for (int c0 = 0; c0 < nc; c0 += SN_THREADS) {       // (nc > SN_THREADS only for very wide, few-row matrices)
                const int c = c0 + tid % (nc < SN_THREADS ? nc : SN_THREADS), gidx = tid / (nc < SN_THREADS ? nc : SN_THREADS);
                float tsum = 0.f;
                if (c < nc && gidx < groups) {
#pragma unroll 8
                    for (int r = gidx; r < R; r += groups) tsum = fmaf(a.W[(int64_t)r * K + j0 + c], us[r], tsum);
                }
                part[tid] = (c < nc && gidx < groups) ? tsum : 0.f;
                __syncthreads();
                if (gidx == 0 && c < nc) {
                    float tot = 0.f;
                    const int stride = nc < SN_THREADS ? nc : SN_THREADS;
                    for (int q = 0; q < groups; ++q) tot += part[q * stride + (c - c0)];
                    put(a.t + j0 + c, tot);
                }
                __syncthreads();
            }
        }
        grid_meet(a.sync, nwg, phase++, a.sync - 4);
        float nrm = 0.f;
        for (int j = tid; j < K; j += SN_THREADS) { const float tv = get(a.t + j); vs[j] = tv; nrm = fmaf(tv, tv, nrm); }
        nrm = block_sum(nrm, red);
        const float inv_v = 1.0f / fmaxf(sqrtf(nrm), a.eps);
        for (int j = tid; j < K; j += SN_THREADS) vs[j] *= inv_v;
        __syncthreads();
        rows_times_v();
        grid_meet(a.sync, nwg, phase++, a.sync - 4);
        float n2 = 0.f;
        for (int r = tid; r < R; r += SN_THREADS) { const float sv = get(a.s + r); us[r] = sv; n2 = fmaf(sv, sv, n2); }
        n2 = block_sum(n2, red);
        const float ns = sqrtf(n2), inv_u = 1.0f / fmaxf(ns, a.eps);
        for (int r = tid; r < R; r += SN_THREADS) us[r] *= inv_u;
        __syncthreads();
        sig_norm = n2 * inv_u;                       // u^T (W v) with u = (W v) / max(|W v|, eps)
        if (it + 1 < a.iterations) grid_meet(a.sync, nwg, phase++, a.sync - 4);      // a.t / a.s are rewritten by the next round
    }
    float sigma = sig_norm;
    if (a.iterations == 0) {
        rows_times_v();
        grid_meet(a.sync, nwg, phase++, a.sync - 4);
        float d = 0.f;
        for (int r = tid; r < R; r += SN_THREADS) d = fmaf(us[r], get(a.s + r), d);
        sigma = block_sum(d, red);
    }
    if (b == 0) {
        if (a.iterations > 0) {
            for (int r = tid; r < R; r += SN_THREADS) a.u[r] = us[r];
            for (int j = tid; j < K; j += SN_THREADS) a.v[j] = vs[j];
        }
        if (tid == 0) a.sigma[0] = sigma;
        if (a.u_used) for (int r = tid; r < R; r += SN_THREADS) a.u_used[r] = us[r];
        if (a.v_used) for (int j = tid; j < K; j += SN_THREADS) a.v_used[j] = vs[j];
    }
    const float inv = 1.0f / sigma;
    const int64_t e0 = (int64_t)r0 * K, e1 = (int64_t)r1 * K;
    float wmax = 0.f;
#pragma unroll 8
    for (int64_t e = e0 + tid; e < e1; e += SN_THREADS) { const float w = a.W[e] * inv; a.w_sn[e] = w; wmax = fmaxf(wmax, fabsf(w)); }
    // max |w_sn| of this slice, for the consumer that splits w_sn into fp16 pairs (wc_conv_weights_f32): one sweep less there
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) wmax = fmaxf(wmax, __shfl_xor(wmax, off));
    __syncthreads();
    if (lane == 0) red[wave] = wmax;
    __syncthreads();
    if (tid == 0) {
        float m = red[0];
#pragma unroll
        for (int w = 1; w < SN_THREADS / 64; ++w) m = fmaxf(m, red[w]);
        a.amax[b] = m;
    }
    // the last workgroup out re-arms the meeting counter for the next launch
    if (nwg > 1 && tid == 0) {
        const unsigned old = __hip_atomic_fetch_add(a.sync + 1, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (old == (unsigned)nwg - 1) {
            __hip_atomic_store(a.sync, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(a.sync + 1, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

struct SnBwdArgs {
    const float* g; const float* w_sn; const float* u; const float* v; const float* sigma;
    int R, K, fully_diff; float* dW;
    float* partial; unsigned* sync; int nwg;          // fully_diff: per-workgroup partial sums of <g, w_sn>
};

__device__ __forceinline__ void sn_backward_body(const SnBwdArgs& a, const int b)
{
    __shared__ float red[SN_THREADS / 64];
    const int tid = threadIdx.x, nwg = a.nwg;
    const int64_t n = (int64_t)a.R * a.K;
    const int64_t e0 = n * b / nwg, e1 = n * (b + 1) / nwg;
    float c = 0.f;
    if (a.fully_diff) {
        float p = 0.f;
        for (int64_t e = e0 + tid; e < e1; e += SN_THREADS) p = fmaf(a.g[e], a.w_sn[e], p);
        p = block_sum(p, red);
        if (tid == 0) put(a.partial + b, p);
        grid_meet(a.sync, nwg, 0, a.sync - 6);      // (the backward's words are the forward's + 2)
        for (int q = 0; q < nwg; ++q) c += get(a.partial + q);   // same order in every workgroup
    }
    const float inv = 1.0f / a.sigma[0];
#pragma unroll 4
    for (int64_t e = e0 + tid; e < e1; e += SN_THREADS) {
        const int r = (int)(e / a.K), j = (int)(e % a.K);
        a.dW[e] = (a.g[e] - c * a.u[r] * a.v[j]) * inv;
    }
    if (a.fully_diff && nwg > 1 && tid == 0) {
        const unsigned old = __hip_atomic_fetch_add(a.sync + 1, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (old == (unsigned)nwg - 1) {
            __hip_atomic_store(a.sync, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(a.sync + 1, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

__global__ __launch_bounds__(SN_THREADS) void sn_forward_kernel(SnArgs a) { sn_forward_body(a, blockIdx.x); }
__global__ __launch_bounds__(SN_THREADS) void sn_backward_kernel(SnBwdArgs a) { sn_backward_body(a, blockIdx.x); }

// Every spectrally normalised layer of a network in ONE launch: the layers are independent of each other and of the
// activations (they depend on the weights only), but as separate launches of <= SN_MAXWG workgroups each they run one
// after the other on a mostly idle chip.  Items ride in the kernel arguments; first[i] is the first block of item i.
constexpr int SN_MAXITEMS = 16;
struct SnBatch { SnArgs item[SN_MAXITEMS]; int first[SN_MAXITEMS + 1]; int count; };
struct SnBwdBatch { SnBwdArgs item[SN_MAXITEMS]; int first[SN_MAXITEMS + 1]; int count; };

__global__ __launch_bounds__(SN_THREADS) void sn_forward_batched_kernel(SnBatch q)
{
    int i = 0;
    while (i + 1 < q.count && (int)blockIdx.x >= q.first[i + 1]) ++i;
    sn_forward_body(q.item[i], (int)blockIdx.x - q.first[i]);
}
__global__ __launch_bounds__(SN_THREADS) void sn_backward_batched_kernel(SnBwdBatch q)
{
    int i = 0;
    while (i + 1 < q.count && (int)blockIdx.x >= q.first[i + 1]) ++i;
    sn_backward_body(q.item[i], (int)blockIdx.x - q.first[i]);
}

int sn_workgroups(int R, int K)
{
    int nwg = (int)(((int64_t)R * K + 4095) / 4096);          // ~4096 elements per workgroup and pass, up to 32 workgroups ...
    if (nwg > 32) {                                           // ... and beyond that ~16384 elements each, up to SN_MAXWG (the
        nwg = (int)(((int64_t)R * K + 16383) / 16384);        // 1024-wide critic of the Tiny-ImageNet recipe: 9.4 M elements
        if (nwg < 32) nwg = 32;                               // per weight; spectral norm 7.5 -> 4 ms of its 145 ms step)
    }
    if (nwg > SN_MAXWG) nwg = SN_MAXWG;
    if (nwg > R) nwg = R;                                     // at least one row each
    return nwg < 1 ? 1 : nwg;
}

}  // namespace

size_t wc_sn_lds_bytes(int R, int K) { return (size_t)(R + K + SN_THREADS / 64 + SN_THREADS) * sizeof(float); }
// scratch: t[K] | s[R] | partial[SN_MAXWG] | amax[SN_MAXWG] | sync[4] (the sync words must be zero before the first launch; every launch leaves them zero)
// ... | 16 bytes whose first word is the sticky error word of grid_meet | 16 bytes of meeting words (forward [0..1], backward [2..3])
size_t wc_sn_workspace_bytes(int R, int K) { return ((size_t)(R + K + 2 * SN_MAXWG) * sizeof(float) + 15) / 16 * 16 + 32; }
size_t wc_sn_error_offset(int R, int K) { return wc_sn_workspace_bytes(R, K) - 32; }
size_t wc_sn_amax_offset(int R, int K) { return (size_t)(R + K + SN_MAXWG) * sizeof(float); }

hipError_t wc_launch_spectral_norm(const float* W, int R, int K, float* u, float* v, int iterations, float eps,
                                   float* w_sn, float* sigma, float* u_used, float* v_used, void* ws, hipStream_t st)
{
    const size_t lds = wc_sn_lds_bytes(R, K);
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(sn_forward_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    float* t = static_cast<float*>(ws);
    unsigned* sync = reinterpret_cast<unsigned*>(static_cast<char*>(ws) + wc_sn_workspace_bytes(R, K) - 16);
    SnArgs a{W, R, K, u, v, iterations, eps, w_sn, sigma, u_used, v_used, t, t + K, sync, sn_workgroups(R, K), t + K + R + SN_MAXWG};
    hipLaunchKernelGGL(sn_forward_kernel, dim3(a.nwg), dim3(SN_THREADS), lds, st, a);
    return hipGetLastError();
}

hipError_t wc_launch_spectral_norm_bwd(const float* g, const float* w_sn, const float* u, const float* v, const float* sigma,
                                       int R, int K, int fully_diff, float* dW, void* ws, hipStream_t st)
{
    float* t = static_cast<float*>(ws);
    unsigned* sync = reinterpret_cast<unsigned*>(static_cast<char*>(ws) + wc_sn_workspace_bytes(R, K) - 16);
    SnBwdArgs a{g, w_sn, u, v, sigma, R, K, fully_diff, dW, t + K + R, sync + 2, sn_workgroups(R, K)};
    hipLaunchKernelGGL(sn_backward_kernel, dim3(a.nwg), dim3(SN_THREADS), 0, st, a);
    return hipGetLastError();
}

// How many workgroups of `kernel` (SN_THREADS threads, `lds` bytes) the device holds at once: a batched launch never asks for
// more (16 items x 128 workgroups of 42 KB could not all be resident, and forward progress then rested on
// in-order dispatch alone).  0 = unknown: no limit is applied.
template <typename K>
static int sn_resident_capacity(K kernel, size_t lds)
{
    int per_cu = 0, dev = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, SN_THREADS, lds) != hipSuccess) return 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    return per_cu * cus;
}

hipError_t wc_launch_spectral_norm_batched(const WcSnItem* items, int count, int iterations, float eps, hipStream_t st)
{
    size_t lds_all = 0;
    for (int i = 0; i < count; ++i) { const size_t l = wc_sn_lds_bytes(items[i].rows, items[i].cols); if (l > lds_all) lds_all = l; }
    static size_t cap_lds = ~(size_t)0; static int cap = 0;
    if (cap_lds != lds_all) {
        if (lds_all > 48 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(sn_forward_batched_kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_all);
            if (e != hipSuccess) return e;
        }
        cap = sn_resident_capacity(sn_forward_batched_kernel, lds_all); cap_lds = lds_all;
    }
    for (int c0 = 0; c0 < count;) {
        SnBatch q = {};
        int take = 0, total = 0;
        while (c0 + take < count && take < SN_MAXITEMS) {        // as many items as are resident together (at least one)
            const int nw = sn_workgroups(items[c0 + take].rows, items[c0 + take].cols);
            if (take > 0 && cap > 0 && total + nw > cap) break;
            total += nw; ++take;
        }
        q.count = take;
        size_t lds = 0;
        int blocks = 0;
        for (int i = 0; i < q.count; ++i) {
            const WcSnItem& it = items[c0 + i];
            float* t = static_cast<float*>(it.ws);
            unsigned* sync = reinterpret_cast<unsigned*>(static_cast<char*>(it.ws) + wc_sn_workspace_bytes(it.rows, it.cols) - 16);
            q.item[i] = SnArgs{it.W, it.rows, it.cols, it.u, it.v, iterations, eps, it.w_sn, it.sigma, it.u_used, it.v_used,
                               t, t + it.cols, sync, sn_workgroups(it.rows, it.cols), t + it.cols + it.rows + SN_MAXWG};
            q.first[i] = blocks;
            blocks += q.item[i].nwg;
            const size_t l = wc_sn_lds_bytes(it.rows, it.cols);
            if (l > lds) lds = l;
        }
        q.first[q.count] = blocks;
        if (lds > 48 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(sn_forward_batched_kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(sn_forward_batched_kernel, dim3(blocks), dim3(SN_THREADS), lds, st, q);
        c0 += take;
    }
    return hipGetLastError();
}

hipError_t wc_launch_spectral_norm_bwd_batched(const WcSnBwdItem* items, int count, int fully_diff, hipStream_t st)
{
    static int cap = -1;
    if (cap < 0) cap = sn_resident_capacity(sn_backward_batched_kernel, 0);
    for (int c0 = 0; c0 < count;) {
        SnBwdBatch q = {};
        int take = 0, total = 0;
        while (c0 + take < count && take < SN_MAXITEMS) {
            const int nw = sn_workgroups(items[c0 + take].rows, items[c0 + take].cols);
            if (take > 0 && cap > 0 && total + nw > cap) break;
            total += nw; ++take;
        }
        q.count = take;
        int blocks = 0;
        for (int i = 0; i < q.count; ++i) {
            const WcSnBwdItem& it = items[c0 + i];
            float* t = static_cast<float*>(it.ws);
            unsigned* sync = reinterpret_cast<unsigned*>(static_cast<char*>(it.ws) + wc_sn_workspace_bytes(it.rows, it.cols) - 16);
            q.item[i] = SnBwdArgs{it.g, it.w_sn, it.u, it.v, it.sigma, it.rows, it.cols, fully_diff, it.dW,
                                  t + it.cols + it.rows, sync + 2, sn_workgroups(it.rows, it.cols)};
            q.first[i] = blocks;
            blocks += q.item[i].nwg;
        }
        q.first[q.count] = blocks;
        hipLaunchKernelGGL(sn_backward_batched_kernel, dim3(blocks), dim3(SN_THREADS), 0, st, q);
        c0 += take;
    }
    return hipGetLastError();
}
