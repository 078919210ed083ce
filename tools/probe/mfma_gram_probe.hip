// Development probe: v_mfma_f32_32x32x16_f16 on GRAM data (B = A^T, so the diagonal outputs are sums of squares), as a chain
// of 4 accumulations from zero -- one K1 stage.  Is D bitwise RNE_fp32(C + exact sum of 16 products)?  If not, how does it differ?
// build+run on the GPU box: hipcc --offload-arch=gfx950 -O3 -o /tmp/gram tools/probe/mfma_gram_probe.hip && /tmp/gram
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int CH = 4;     // chain length

// A[t][c][i][k]: 32 channels x 16 rows per link; both operands are the same fragment (Gram block on the diagonal)
__global__ __launch_bounds__(64) void probe(const _Float16* __restrict__ A, float* __restrict__ D)
{
    const int t = blockIdx.x, lane = threadIdx.x, l31 = lane & 31, lh = lane >> 5;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int c = 0; c < CH; ++c) {
        const f16x8 a = *reinterpret_cast<const f16x8*>(A + (((size_t)t * CH + c) * 32 + l31) * 16 + 8 * lh);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, a, acc, 0, 0, 0);
        for (int r = 0; r < 16; ++r) D[(((size_t)t * CH + c) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * 32 + l31] = acc[r];
    }
}
static double urand() { return (rand() + 0.5) / (RAND_MAX + 1.0); }
static double nrand() { return sqrt(-2.0 * log(urand())) * cos(6.283185307179586 * urand()); }
int main()
{
    const int T = 4096;
    std::vector<_Float16> hA((size_t)T * CH * 32 * 16);
    std::vector<float> hD((size_t)T * CH * 32 * 32);
    _Float16* dA; float* dD;
    (void)hipMalloc(&dA, hA.size() * 2); (void)hipMalloc(&dD, hD.size() * 4);
    for (double sigma : {6.0, 4.2, 7.9}) {
        srand(7);
        for (auto& v : hA) v = (_Float16)(sigma * nrand());
        (void)hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe, dim3(T), dim3(64), 0, 0, dA, dD);
        (void)hipMemcpy(hD.data(), dD, hD.size() * 4, hipMemcpyDeviceToHost);
        // emulate: acc_c = RNE_f32(acc_{c-1} + exact) with the GPU's own acc_{c-1} as input (so one link at a time)
        size_t n = 0, mism = 0, nd = 0, mism_d = 0; double be_d = 0, be_o = 0, ulps_d = 0;
        size_t ties = 0, tie_down = 0, tie_up = 0, tie_even_ok = 0;
        for (int t = 0; t < T; ++t) for (int c = 0; c < CH; ++c) for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
            double s = 0;
            for (int k = 0; k < 16; ++k) s += (double)(float)hA[(((size_t)t * CH + c) * 32 + i) * 16 + k] * (double)(float)hA[(((size_t)t * CH + c) * 32 + j) * 16 + k];
            const double prev = c ? (double)hD[(((size_t)t * CH + c - 1) * 32 + i) * 32 + j] : 0.0;
            const double exact = prev + s;                 // exact in double (products 22 bits, |sum| < 2^13, lsb >= 2^-24)
            const float rne = (float)exact;                // host: round to nearest even
            const float got = hD[(((size_t)t * CH + c) * 32 + i) * 32 + j];
            ++n; if (got != rne) ++mism;
            const double e = (double)got - exact;
            if (i == j) { ++nd; be_d += e; if (got != rne) ++mism_d; ulps_d += e / (double)(std::nextafter(fabsf(rne), INFINITY) - fabsf(rne)); }
            else be_o += e;
            // is the exact value a tie between two floats?
            const float lo = (rne > exact) ? std::nextafter(rne, -INFINITY) : rne, hi = (rne < exact) ? std::nextafter(rne, INFINITY) : rne;
            if (lo != hi && (exact - (double)lo) == ((double)hi - exact)) { ++ties; if (got == lo) ++tie_down; else if (got == hi) ++tie_up; if (got == rne) ++tie_even_ok; }
        }
        printf("sigma %.1f: %zu outputs, %zu differ from RNE (%.4f%%); diagonal: %zu of %zu differ, mean err %+.3e (%+.4f ulp), off-diagonal mean err %+.3e\n",
               sigma, n, mism, 100.0 * mism / n, mism_d, nd, be_d / nd, ulps_d / nd, be_o / (n - nd));
        printf("           exact ties: %zu (%.3f%%): GPU took the lower %zu, the upper %zu; agrees with ties-to-even on %zu\n", ties, 100.0 * ties / n, tie_down, tie_up, tie_even_ok);
    }
    return 0;
}
