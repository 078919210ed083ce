// Development probe (round 6): issue rate of v_mfma_f64_16x16x4_f64 on gfx950 -- cycles per MFMA on one SIMD with 1 / 2 / 4 waves,
// independent accumulators vs one dependent chain, operands from registers vs re-read from LDS before every group of four.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_f64_rate tools/probe/mfma_f64_rate.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double f64x4 __attribute__((ext_vector_type(4)));

template <int NACC, bool LDS>
__global__ __launch_bounds__(1024) void rate_kernel(double* out, unsigned long long* cyc, int iters, int active_waves)
{
    __shared__ double sm[16 * 260];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int e = threadIdx.x; e < 16 * 260; e += blockDim.x) sm[e] = 1e-3 * (e % 97);
    __syncthreads();
    if (wave >= active_waves) return;
    f64x4 acc[NACC];
#pragma unroll
    for (int q = 0; q < NACC; ++q) acc[q] = f64x4{0.0, 0.0, 0.0, 0.0};
    double a = 1.0 + lane * 1e-3, b = 0.5 - lane * 1e-3;
    const double* pa = sm + (lane >> 4) * 260 + (lane & 15);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < NACC; ++q) {
            double av[4], bv[4];
            if (LDS) {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) { av[kk] = -pa[4 * kk * 260 + 16 * q]; bv[kk] = pa[4 * kk * 260 + 16 * q + 64]; }
            } else {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) { av[kk] = a; bv[kk] = b; }
            }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[kk], bv[kk], acc[q], 0, 0, 0);
        }
        asm volatile("" : "+v"(a), "+v"(b));
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0.0;
#pragma unroll
    for (int q = 0; q < NACC; ++q) s += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
    out[blockIdx.x * 1024 + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 16 + wave] = t1 - t0;
}

// interleaved: the four MFMAs of a group go to four DIFFERENT accumulators (no back-to-back dependence inside a wave)
template <int NACC>
__global__ __launch_bounds__(1024) void rate_interleaved_kernel(double* out, unsigned long long* cyc, int iters, int active_waves)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave >= active_waves) return;
    f64x4 acc[NACC];
#pragma unroll
    for (int q = 0; q < NACC; ++q) acc[q] = f64x4{0.0, 0.0, 0.0, 0.0};
    double a = 1.0 + lane * 1e-3, b = 0.5 - lane * 1e-3;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int q = 0; q < NACC; ++q) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[q], 0, 0, 0);
        asm volatile("" : "+v"(a), "+v"(b));
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0.0;
#pragma unroll
    for (int q = 0; q < NACC; ++q) s += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
    out[blockIdx.x * 1024 + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 16 + wave] = t1 - t0;
}

template <typename K>
static void run(const char* name, K kern, int nacc, int threads, int active)
{
    double* out; unsigned long long* cyc;
    hipMalloc(&out, 1024 * 8); hipMalloc(&cyc, 16 * 8);
    hipMemset(cyc, 0, 16 * 8);
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(1), dim3(threads), 0, 0, out, cyc, iters, active);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(1), dim3(threads), 0, 0, out, cyc, iters, active);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[16]; hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    unsigned long long mx = 0; for (int w = 0; w < 16; ++w) mx = h[w] > mx ? h[w] : mx;
    const double per_simd = (double)iters * nacc * 4 * ((active + 3) / 4);       // MFMAs issued on the busiest SIMD
    printf("%-44s waves %2d: %9llu memtime ticks, %7.1f us, ticks per MFMA per SIMD %.1f, ns per MFMA per SIMD %.2f\n", name, active, mx, ms * 1e3,
           mx / per_simd, ms * 1e6 / per_simd);
    hipFree(out); hipFree(cyc);
}

int main()
{
    for (int active : {1, 4, 8, 16}) {
        run("chain-of-4, 1 acc, regs", rate_kernel<1, false>, 1, 1024, active);
        run("chain-of-4, 4 acc, regs", rate_kernel<4, false>, 4, 1024, active);
        run("chain-of-4, 8 acc, LDS operands", rate_kernel<8, true>, 8, 1024, active);
        run("interleaved, 4 acc, regs", rate_interleaved_kernel<4>, 4, 1024, active);
        run("interleaved, 8 acc, regs", rate_interleaved_kernel<8>, 8, 1024, active);
    }
    return 0;
}
