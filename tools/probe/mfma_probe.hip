// Development probe: how much non-MFMA issue hides under v_mfma_f32_32x32x16_f16 on gfx950, one or two waves per SIMD.
// build+run on the GPU box: hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_probe tools/probe/mfma_probe.hip && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// same flops per iteration as `probe` (48 x 32x32x16), issued as 96 x 16x16x32 on four 16x16 accumulators
template <int MODE>
__global__ __launch_bounds__(512, 2) void probe16(const _Float16* __restrict__ in, float* __restrict__ out,
                                                  unsigned long long* __restrict__ cyc, int iters)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    f16x8 b[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) b[i] = *reinterpret_cast<const f16x8*>(in + ((i * 64 + lane) * 8));
    f16x8 a0 = *reinterpret_cast<const f16x8*>(in + ((40 * 64 + lane) * 8));
    f16x8 a1 = *reinterpret_cast<const f16x8*>(in + ((41 * 64 + lane) * 8));
    f32x4 acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    float v0 = lane, v1 = lane * 2.f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 96; ++g) {
            acc[g & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16((g & 4) ? a1 : a0, b[(g >> 1) % 32], acc[g & 3], 0, 0, 0);
            if (MODE == 1) { v0 = __builtin_fmaf(v0, 1.0001f, 0.5f); v1 = __builtin_fmaf(v1, 0.9999f, 0.25f); }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = v0 + v1;
#pragma unroll
    for (int q = 0; q < 4; ++q) r += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
    out[blockIdx.x * blockDim.x + tid] = r;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}
template <int MODE>
void run16(const char* name, int threads, const _Float16* in, float* out, unsigned long long* cyc)
{
    const int iters = 64, grid = 256;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((probe16<MODE>), dim3(grid), dim3(threads), 0, 0, in, out, cyc, iters);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((probe16<MODE>), dim3(grid), dim3(threads), 0, 0, in, out, cyc, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s threads=%3d  %8.1f us  (16x16x32: %.1f ns per 32x32x16-equivalent per SIMD)\n", name, threads, ms * 1e3,
           ms * 1e6 / (iters * 48.0 * (threads / 256.0)));
}

template <int MODE, int NB>
__global__ __launch_bounds__(512, 2) void probe(const _Float16* __restrict__ in, float* __restrict__ out,
                                                unsigned long long* __restrict__ cyc, int iters)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    f16x8 b[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) b[i] = *reinterpret_cast<const f16x8*>(in + ((i * 64 + lane) * 8));
    f16x8 a0 = *reinterpret_cast<const f16x8*>(in + ((40 * 64 + lane) * 8));
    f16x8 a1 = *reinterpret_cast<const f16x8*>(in + ((41 * 64 + lane) * 8));
    for (int i = tid; i < 16384; i += blockDim.x) reinterpret_cast<float*>(smem)[i] = (float)i;
    __syncthreads();
    f32x16 acc, acc2;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acc2[r] = 0.f; }
    float v0 = lane, v1 = lane * 2.f, v2 = 3.f, v3 = wave;
    f32x4 lv = {0.f, 0.f, 0.f, 0.f};
    const char* lp = smem + lane * 16 + wave * 1024;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 48; ++g) {
            const int s = g / 3, m = g % 3;
            if (MODE == 6) {   // two independent accumulators
                if (g & 1) acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(m ? a1 : a0, b[(2 * s + (m == 1)) % NB], acc2, 0, 0, 0);
                else acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(m ? a1 : a0, b[(2 * s + (m == 1)) % NB], acc, 0, 0, 0);
            } else
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(m ? a1 : a0, b[(2 * s + (m == 1)) % NB], acc, 0, 0, 0);
            if (MODE == 1 || MODE == 4) {   // 4 independent fp32 FMAs
                v0 = __builtin_fmaf(v0, 1.0001f, 0.5f); v1 = __builtin_fmaf(v1, 0.9999f, 0.25f);
                v2 = __builtin_fmaf(v2, 1.0002f, 0.125f); v3 = __builtin_fmaf(v3, 0.9998f, 0.0625f);
            }
            if (MODE == 5) {   // 2 FMAs
                v0 = __builtin_fmaf(v0, 1.0001f, 0.5f); v1 = __builtin_fmaf(v1, 0.9999f, 0.25f);
            }
            if (MODE == 2 || MODE == 4) {   // one LDS read per gap, consumed a k-step later
                if (m == 0) { v0 += lv[0]; }
                lv = *reinterpret_cast<const f32x4*>(lp + ((g * 64) & 8191));
            }
            if (MODE == 3) {   // conversion-like: cvt + sub
                const _Float16 h = (_Float16)v0; v1 = v0 - (float)h; v0 = v1 * 1.5f + v2;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = v0 + v1 + v2 + v3 + lv[1];
#pragma unroll
    for (int i = 0; i < 16; ++i) r += acc[i] + acc2[i];
    out[blockIdx.x * blockDim.x + tid] = r;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int MODE, int NB>
void run(const char* name, int threads, const _Float16* in, float* out, unsigned long long* cyc)
{
    const int iters = 64, grid = 256;
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe<MODE, NB>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((probe<MODE, NB>), dim3(grid), dim3(threads), 128 * 1024, 0, in, out, cyc, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<MODE, NB>), dim3(grid), dim3(threads), 128 * 1024, 0, in, out, cyc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(grid * 8);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double avg = 0; int nw = threads / 64;
    for (int b = 0; b < grid; ++b) for (int w = 0; w < nw; ++w) avg += (double)h[b * 8 + w];
    avg /= grid * nw;
    const double per_wave = avg / (iters * 48.0);                 // cycles per MFMA as seen by one wave
    const double per_simd = per_wave / (threads / 256.0);         // pipe cycles per MFMA on a SIMD
    printf("%-44s threads=%3d  %8.1f us  (%.1f ns per MFMA per SIMD)  cycles/MFMA/wave %6.1f  /SIMD %6.1f  clock %.2f GHz\n", name, threads, ms * 1e3,
           ms * 1e6 / (iters * 48.0 * (threads / 256.0)), per_wave, per_simd, avg / (ms * 1e6));
}

int main()
{
    _Float16* in; float* out; unsigned long long* cyc;
    hipMalloc(&in, 64 * 64 * 8 * 2); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
    std::vector<_Float16> h(64 * 64 * 8);
    for (auto& v : h) v = (_Float16)((rand() % 2001 - 1000) / 1000.0f);
    hipMemcpy(in, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    for (int threads : {256, 512}) {
        run16<0>("16x16x32 bare", threads, in, out, cyc);
        run16<1>("16x16x32 +2 v_fma per MFMA", threads, in, out, cyc);
        run<0, 32>("bare MFMA chain, 32 B regs", threads, in, out, cyc);
        run<0, 2>("bare MFMA chain, 2 B regs", threads, in, out, cyc);
        run<6, 32>("two accumulators", threads, in, out, cyc);
        run<5, 32>("+2 v_fma per gap", threads, in, out, cyc);
        run<1, 32>("+4 v_fma per gap", threads, in, out, cyc);
        run<2, 32>("+1 ds_read_b128 per gap", threads, in, out, cyc);
        run<4, 32>("+4 v_fma +1 ds_read_b128 per gap", threads, in, out, cyc);
        run<3, 32>("+cvt/sub/fma chain per gap", threads, in, out, cyc);
    }
    return 0;
}
