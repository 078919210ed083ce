// Development probe: does v_mfma_f32_32x32x16_f16 lose low bits of SMALL products when the accumulator is LARGE?
// D = C0 + sum_k a[i][k] b[k][j] with C0 a constant; the host compares with the exact float64 sum.
// build+run on the GPU box: hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_bias tools/probe/mfma_bias_probe.hip && /tmp/mfma_bias
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(64) void probe(const _Float16* __restrict__ A, const _Float16* __restrict__ B, float c0, float* __restrict__ D)
{
    const int t = blockIdx.x, lane = threadIdx.x, l31 = lane & 31, lh = lane >> 5;
    const f16x8 a = *reinterpret_cast<const f16x8*>(A + ((size_t)t * 32 + l31) * 16 + 8 * lh);      // A[t][i][k]
    f16x8 b;
    for (int q = 0; q < 8; ++q) b[q] = B[((size_t)t * 16 + 8 * lh + q) * 32 + l31];                  // B[t][k][j]
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = c0;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) D[((size_t)t * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * 32 + l31] = acc[r];
}

static double urand() { return (rand() + 0.5) / (RAND_MAX + 1.0); }
static double nrand() { return sqrt(-2.0 * log(urand())) * cos(6.283185307179586 * urand()); }

int main()
{
    const int T = 8192;
    std::vector<_Float16> hA((size_t)T * 32 * 16), hB((size_t)T * 16 * 32);
    std::vector<float> hD((size_t)T * 32 * 32);
    _Float16 *dA, *dB; float* dD;
    (void)hipMalloc(&dA, hA.size() * 2); (void)hipMalloc(&dB, hB.size() * 2); (void)hipMalloc(&dD, hD.size() * 4);
    const double amag[3] = {6.0, 6.0 / 2048.0, 6.0 / 2048.0};     // "hi-like" operand, "lo-like" operand, lo-like
    const int apos[3] = {0, 0, 1};                                 // 1: A and B are the SAME sign pattern (all products >= 0)
    const float c0s[4] = {0.f, 16.f, 2048.f, -2048.f};
    for (int av = 0; av < 3; ++av) {
        srand(1234 + av);
        for (size_t i = 0; i < hB.size(); ++i) hB[i] = (_Float16)(6.0 * nrand());
        for (size_t i = 0; i < hA.size(); ++i) hA[i] = (_Float16)(amag[av] * nrand());
        if (apos[av]) {            // force a*b >= 0: a[i][k] takes the sign of b[k][j=i] -- only the diagonal outputs are then all-positive
            for (int t = 0; t < T; ++t) for (int i = 0; i < 32; ++i) for (int k = 0; k < 16; ++k) {
                _Float16& a = hA[((size_t)t * 32 + i) * 16 + k];
                const float bv = (float)hB[((size_t)t * 16 + k) * 32 + i];
                if (((float)a < 0) != (bv < 0)) a = -a;
            }
        }
        (void)hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice);
        (void)hipMemcpy(dB, hB.data(), hB.size() * 2, hipMemcpyHostToDevice);
        for (int ci = 0; ci < 4; ++ci) {
            hipLaunchKernelGGL(probe, dim3(T), dim3(64), 0, 0, dA, dB, c0s[ci], dD);
            (void)hipMemcpy(hD.data(), dD, hD.size() * 4, hipMemcpyDeviceToHost);
            double bias = 0, bias_diag = 0, rms = 0; size_t n = 0, nd = 0;
            for (int t = 0; t < T; ++t) for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
                double s = 0;
                for (int k = 0; k < 16; ++k) s += (double)(float)hA[((size_t)t * 32 + i) * 16 + k] * (double)(float)hB[((size_t)t * 16 + k) * 32 + j];
                const double e = ((double)hD[((size_t)t * 32 + i) * 32 + j] - (double)c0s[ci]) - s;
                bias += e; rms += e * e; ++n;
                if (i == j) { bias_diag += e; ++nd; }
            }
            printf("A~%.4f%s  C0=%7.0f : mean err %+.3e (diag %+.3e)  rms %.3e   [fp32 ulp(C0)/2 = %.3e]\n", amag[av], apos[av] ? " same-sign" : "", c0s[ci],
                   bias / n, bias_diag / nd, sqrt(rms / n), c0s[ci] != 0 ? ldexp(1.0, (int)floor(log2(fabs(c0s[ci]))) - 24) : 0.0);
        }
    }
    return 0;
}
