// Development probe (round 6): the shader clock while other kernels run.  One wave samples (s_memtime, s_memrealtime) pairs -- shader cycles and the
// chip-wide 100 MHz counter -- `n` times, `sleep` x 64 cycles apart, on a side stream; stamp_kernel leaves one s_memrealtime on the stream under test
// in front of and behind the launch whose clock is wanted.  tools/clock_under_kernels.py loads this library into the process that launches the kernels.
// build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools/probe/bin/libclock_probe.so tools/probe/clock_probe.hip
#include <hip/hip_runtime.h>
__global__ void clock_probe_kernel(unsigned long long* out, int n, int sleep)
{
    if (threadIdx.x != 0) return;
    for (int i = 0; i < n; ++i) {
        const unsigned long long t = __builtin_amdgcn_s_memtime(), r = __builtin_amdgcn_s_memrealtime();
        out[2 * i] = t; out[2 * i + 1] = r;
        for (int k = 0; k < sleep; ++k) __builtin_amdgcn_s_sleep(1);
    }
}
__global__ void stamp_kernel(unsigned long long* out) { if (threadIdx.x == 0) *out = __builtin_amdgcn_s_memrealtime(); }
extern "C" int clock_probe_launch(unsigned long long* out, int n, int sleep, void* stream)
{
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out, n, sleep);
    return (int)hipGetLastError();
}
extern "C" int clock_stamp_launch(unsigned long long* out, void* stream)
{
    hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out);
    return (int)hipGetLastError();
}
