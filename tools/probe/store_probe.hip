// Development probe: cost of the three candidate store patterns for a wave's 32 x 32 fp32 output block (row pitch 1 KiB),
// eight waves per workgroup writing a 32-row x 256-column tile, 16 tiles per workgroup, 256 workgroups.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(512, 2) void probe(float* __restrict__ out, unsigned long long* __restrict__ cyc, int tiles)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
    float v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = lane + r;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int t = 0; t < tiles; ++t) {
        float* tile = out + ((size_t)(blockIdx.x + t * gridDim.x) * 32) * 256 + wave * 32;
        if (MODE == 0) {          // 16 dword stores: reg r -> row (r&3)+8(r>>2)+4lh, col l31
#pragma unroll
            for (int r = 0; r < 16; ++r) tile[((r & 3) + 8 * (r >> 2) + 4 * lh) * 256 + l31] = v[r];
        } else if (MODE == 1) {   // 4 dwordx4 stores: lane -> row l31, cols 8q+4lh..+3 (transposed accumulator)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 w = {v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
                *reinterpret_cast<f32x4*>(tile + l31 * 256 + 8 * q + 4 * lh) = w;
            }
        } else {                  // 4 dwordx4 stores: lane -> row 8q + lane/8, cols 4(lane%8)..+3 (8 full 128-B lines)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 w = {v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
                *reinterpret_cast<f32x4*>(tile + (8 * q + (lane >> 3)) * 256 + 4 * (lane & 7)) = w;
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] += 1.f;
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)");
    unsigned long long t2 = __builtin_amdgcn_s_memtime();
    if (lane == 0) { cyc[(blockIdx.x * 8 + wave) * 2] = t1 - t0; cyc[(blockIdx.x * 8 + wave) * 2 + 1] = t2 - t0; }
}
template <int MODE> void run(const char* name, float* out, unsigned long long* cyc)
{
    const int tiles = 16, grid = 256;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((probe<MODE>), dim3(grid), dim3(512), 0, 0, out, cyc, tiles);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((probe<MODE>), dim3(grid), dim3(512), 0, 0, out, cyc, tiles);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(grid * 16);
    (void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double issue = 0, done = 0;
    for (int i = 0; i < grid * 8; ++i) { issue += h[2 * i]; done += h[2 * i + 1]; }
    issue /= grid * 8; done /= grid * 8;
    printf("%-52s %7.1f us  %6.0f GB/s   issue ticks/tile/wave %6.0f   until drained %6.0f\n", name, ms * 1e3,
           (double)grid * tiles * 32768 / (ms * 1e-3) / 1e9, issue / tiles, done / tiles);
}
int main()
{
    float* out; unsigned long long* cyc;
    (void)hipMalloc(&out, (size_t)256 * 16 * 32768); (void)hipMalloc(&cyc, 256 * 16 * 8);
    run<0>("16 x dword, 2 rows x 128 B per instr", out, cyc);
    run<1>("4 x dwordx4, 32 rows x 32 B per instr (transposed)", out, cyc);
    run<2>("4 x dwordx4, 8 rows x 128 B per instr", out, cyc);
    return 0;
}
