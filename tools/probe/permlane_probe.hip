// v_permlane16_swap_b32 semantics on gfx950: prints which (operand, 16-lane row) each result row holds.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out)
{
    const unsigned a = 0x100 + threadIdx.x / 16, b = 0x200 + threadIdx.x / 16;     // operand tag | row
    const auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    out[threadIdx.x] = r[0]; out[64 + threadIdx.x] = r[1];
}
int main()
{
    unsigned* d; hipMalloc(&d, 128 * 4);
    k<<<1, 64>>>(d);
    unsigned h[128]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    for (int w = 0; w < 2; ++w) { printf("r[%d] rows:", w); for (int q = 0; q < 4; ++q) printf(" %03x", h[64 * w + 16 * q]); printf("\n"); }
    return 0;
}
