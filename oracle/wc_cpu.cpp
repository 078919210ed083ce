// TEST INFRASTRUCTURE -- not part of the product path.
//
// CPU restatement of the WC hot path behind the SAME C ABI as libwc_hip.so (include/wc_hip.h), every entry point with a
// `_cpu` suffix, HOST pointers, and the workspace / plan / stream arguments accepted and ignored (SURVEY.md section 8b:
// "The C++ CPU restatement exports the same symbols with a `_cpu` suffix and a null stream, and is what the conformance
// tests link against"; section 8d: it is also timed beside the GPU path).  Plain OpenMP loops, float64 accumulation,
// no blocking tricks: it restates WHAT each stage computes --
//   wc_stats_f32        DecorelationNormalization.call (class imported generator.py:9, instantiated generator.py:24,26):
//                       the sums behind mean and f f^T/(M-1)
//   wc_factor_f64       (1-eps) Sigma + eps I, tf.cholesky, tf.matrix_triangular_solve against I, moving-statistics updates
//   wc_color_f32        Conv2D 1x1 / ConditionalConv11 / FactorizedConv11 kernels folded into W (generator.py:50-78)
//   wc_apply(_act)_f32  W f, transpose back, 1x1 coloring conv + bias + Add (generator.py:83-87), optional ReLU
//   wc_bwd_*            the closed-form gradients
// -- with the argument meaning, layouts and error codes of the header.  Parity of this file with upstream is as unpinned as
// the numpy oracle's: it is checked against oracle/wc_oracle.py (tests/test_cpu_port.py), and the
// HIP library is checked against both.  Only tests/, __graft_entry__ and bench.py's cpu legs may load it.
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <vector>
#include <omp.h>
#include "../include/wc_hip.h"

namespace {

bool bad_channels(int C) { return C < 32 || C > 1024 || (C % 32) != 0; }

// out (C x C) = alpha * op(A) op(B), row-major; ta / tb: use the transpose of A / B
void mm(int C, const double* A, bool ta, const double* B, bool tb, double alpha, double* out)
{
#pragma omp parallel for schedule(static)
    for (int i = 0; i < C; ++i) {
        std::vector<double> row(C, 0.0);
        for (int k = 0; k < C; ++k) {
            const double a = ta ? A[(size_t)k * C + i] : A[(size_t)i * C + k];
            if (a == 0.0) continue;
            if (!tb) { const double* b = B + (size_t)k * C; for (int j = 0; j < C; ++j) row[j] += a * b[j]; }
            else for (int j = 0; j < C; ++j) row[j] += a * B[(size_t)j * C + k];
        }
        for (int j = 0; j < C; ++j) out[(size_t)i * C + j] = alpha * row[j];
    }
}

}  // namespace

extern "C" {

int wc_cpu_threads(void) { return omp_get_max_threads(); }

int wc_stats_f32_cpu(const float* x, int64_t M, int C, int groups, double* sum, double* xtx,
                     void*, size_t, wc_stream_t)
{
    if (!x || !sum || !xtx) return WC_ERR_NULL;
    if (M <= 0 || groups <= 0 || (M % groups) != 0) return WC_ERR_SHAPE;
    if (bad_channels(C)) return WC_ERR_CHANNELS;
    const int64_t Mg = M / groups;
    const size_t CC = (size_t)C * C;
    for (int g = 0; g < groups; ++g) {
        const float* xg = x + (size_t)g * Mg * C;
        double* sg = sum + (size_t)g * C;
        double* tg = xtx + (size_t)g * CC;
        std::memset(sg, 0, sizeof(double) * C);
        std::memset(tg, 0, sizeof(double) * CC);
        // row slabs in a fixed number and order (so the result does not depend on the thread count), upper triangle
        const int nslab = (int)((Mg + 255) / 256);
        const int lanes = nslab < 64 ? nslab : 64;
        std::vector<double> part((size_t)lanes * (CC + C), 0.0);
#pragma omp parallel for schedule(static)
        for (int l = 0; l < lanes; ++l) {
            double* ps = part.data() + (size_t)l * (CC + C);
            double* pt = ps + C;
            for (int z = l; z < nslab; z += lanes) {
                const int64_t m0 = (int64_t)z * 256, m1 = m0 + 256 < Mg ? m0 + 256 : Mg;
                for (int64_t m = m0; m < m1; ++m) {
                    const float* r = xg + (size_t)m * C;
                    for (int i = 0; i < C; ++i) {
                        const double a = r[i];
                        ps[i] += a;
                        double* t = pt + (size_t)i * C;
                        for (int j = i; j < C; ++j) t[j] += a * (double)r[j];
                    }
                }
            }
        }
        for (int l = 0; l < lanes; ++l) {
            const double* ps = part.data() + (size_t)l * (CC + C);
            const double* pt = ps + C;
            for (int i = 0; i < C; ++i) sg[i] += ps[i];
            for (size_t e = 0; e < CC; ++e) tg[e] += pt[e];
        }
        for (int i = 0; i < C; ++i)
            for (int j = 0; j < i; ++j) tg[(size_t)i * C + j] = tg[(size_t)j * C + i];
    }
    return WC_OK;
}

int wc_factor_f64_cpu(const double* sum, const double* xtx, int64_t M, int C, int groups, double eps, double momentum, int ddof,
                      int training, float* moving_mean, float* moving_cov, float* mu, float* chan_scale, double* L, double* W,
                      void*, size_t, wc_stream_t)
{
    if (!mu || !L || !W) return WC_ERR_NULL;
    if (training && (!sum || !xtx)) return WC_ERR_NULL;
    if (!training && (!moving_mean || !moving_cov)) return WC_ERR_NULL;
    if (bad_channels(C)) return WC_ERR_CHANNELS;
    if (groups <= 0 || (training && (M <= ddof || M <= 0))) return WC_ERR_SHAPE;
    if (!(eps > 0.0) || eps >= 1.0 || momentum < 0.0 || momentum > 1.0 || ddof < 0 || ddof > 1) return WC_ERR_ARG;
    const size_t CC = (size_t)C * C;
    std::vector<double> tmax(C, 0.0);
    for (int g = 0; g < groups; ++g) {
        double* T = L + (size_t)g * CC;
        double* Wg = W + (size_t)g * CC;
        for (int i = 0; i < C; ++i)
            for (int j = 0; j < C; ++j) {
                const size_t e = (size_t)i * C + j;
                double sig;
                if (training) {
                    const double* sg = sum + (size_t)g * C;
                    const double* xg = xtx + (size_t)g * CC;
                    sig = (0.5 * (xg[e] + xg[(size_t)j * C + i]) - sg[i] * sg[j] / (double)M) / (double)(M - ddof);
                    if (moving_cov) moving_cov[e] = (float)(momentum * (double)moving_cov[e] + (1.0 - momentum) * sig);
                    if (i == 0) {
                        const double m = sg[j] / (double)M;
                        mu[(size_t)g * C + j] = (float)m;
                        if (moving_mean) moving_mean[j] = (float)(momentum * (double)moving_mean[j] + (1.0 - momentum) * m);
                    }
                } else {
                    sig = 0.5 * ((double)moving_cov[e] + (double)moving_cov[(size_t)j * C + i]);
                    if (i == 0) mu[(size_t)g * C + j] = moving_mean[j];
                }
                const double t = (1.0 - eps) * sig + (i == j ? eps : 0.0);
                T[e] = t;
                if (i == j && t > tmax[j]) tmax[j] = t;
            }
        // Cholesky, lower, in place; zeros above the diagonal
        for (int j = 0; j < C; ++j) {
            double d = T[(size_t)j * C + j];
            for (int k = 0; k < j; ++k) d -= T[(size_t)j * C + k] * T[(size_t)j * C + k];
            d = std::sqrt(d);
            T[(size_t)j * C + j] = d;
#pragma omp parallel for schedule(static)
            for (int i = j + 1; i < C; ++i) {
                double v = T[(size_t)i * C + j];
                for (int k = 0; k < j; ++k) v -= T[(size_t)i * C + k] * T[(size_t)j * C + k];
                T[(size_t)i * C + j] = v / d;
            }
        }
        for (int i = 0; i < C; ++i)
            for (int j = i + 1; j < C; ++j) T[(size_t)i * C + j] = 0.0;
        // W = L^-1: forward substitution against I, one column per task
#pragma omp parallel for schedule(dynamic, 4)
        for (int c = 0; c < C; ++c) {
            for (int i = 0; i < C; ++i) {
                if (i < c) { Wg[(size_t)i * C + c] = 0.0; continue; }
                double v = (i == c) ? 1.0 : 0.0;
                for (int k = c; k < i; ++k) v -= T[(size_t)i * C + k] * Wg[(size_t)k * C + c];
                Wg[(size_t)i * C + c] = v / T[(size_t)i * C + i];
            }
        }
    }
    if (chan_scale)
        for (int j = 0; j < C; ++j) {
            int ex;
            std::frexp(std::sqrt(tmax[j]), &ex);
            chan_scale[j] = (float)std::ldexp(1.0, 3 - ex);
        }
    return WC_OK;
}

int wc_color_f32_cpu(const double* W, const float* gamma, int Kc, int C, int groups, int per_group, float* A, float* At,
                     const float*, void*, void*, size_t, wc_stream_t)
{
    if (!W || !A) return WC_ERR_NULL;
    if (bad_channels(C)) return WC_ERR_CHANNELS;
    if (Kc <= 0 || groups <= 0 || (!gamma && Kc != 1)) return WC_ERR_SHAPE;
    const size_t CC = (size_t)C * C;
#pragma omp parallel for collapse(2) schedule(static)
    for (int s = 0; s < groups * Kc; ++s)
        for (int i = 0; i < C; ++i) {
            const int g = s / Kc, k = s % Kc;
            const double* Wg = W + (size_t)g * CC;
            const float* G = gamma ? gamma + (size_t)(per_group ? s : k) * CC : nullptr;
            std::vector<double> row(C, 0.0);
            if (G) {
                for (int c = 0; c < C; ++c) {            // A[i][j] = sum_c W[c][i] Gamma[c][j]
                    const double w = Wg[(size_t)c * C + i];
                    if (w == 0.0) continue;
                    const float* gr = G + (size_t)c * C;
                    for (int j = 0; j < C; ++j) row[j] += w * (double)gr[j];
                }
            } else {
                for (int j = 0; j < C; ++j) row[j] = Wg[(size_t)j * C + i];
            }
            for (int j = 0; j < C; ++j) {
                A[(size_t)s * CC + (size_t)i * C + j] = (float)row[j];
                if (At) At[(size_t)s * CC + (size_t)j * C + i] = (float)row[j];
            }
        }
    return WC_OK;
}

int wc_group_bias_f32_cpu(const float* mu, const float* A, const float* beta, int groups, int Kc, int C, int per_group,
                          float* center, float* bias, wc_stream_t)
{
    if (!mu || !A || !center || !bias) return WC_ERR_NULL;
    if (groups <= 0 || Kc <= 0) return WC_ERR_SHAPE;
    if (bad_channels(C)) return WC_ERR_CHANNELS;
    const size_t CC = (size_t)C * C;
    for (int c = 0; c < C; ++c) {
        double s = 0.0;
        for (int g = 0; g < groups; ++g) s += mu[(size_t)g * C + c];
        center[c] = (float)(s / groups);
    }
    for (int s = 0; s < groups * Kc; ++s) {
        const int g = s / Kc, k = s % Kc;
        for (int n = 0; n < C; ++n) {
            double v = beta ? (double)beta[(size_t)(per_group ? s : k) * C + n] : 0.0;
            for (int c = 0; c < C; ++c)
                v -= ((double)mu[(size_t)g * C + c] - (double)center[c]) * (double)A[(size_t)s * CC + (size_t)c * C + n];
            bias[(size_t)s * C + n] = (float)v;
        }
    }
    return WC_OK;
}

// out[m] = sum_s (in_s[m] - center_s) B_s[slot] + bias[slot] - sub   (the shape of K3 and K6)
static void rows_affine(const float* in0, const float* c0, const float* B0, int64_t b0_stride,
                        const float* in1, const float* c1, const float* B1,
                        const float* bias, const float* sub, const int32_t* slot, int64_t N, int64_t HW, int C, int relu, float* out)
{
    const int64_t M = N * HW;
#pragma omp parallel for schedule(static)
    for (int64_t m = 0; m < M; ++m) {
        const int s = slot ? slot[m / HW] : 0;
        std::vector<double> acc(C, 0.0);
        const float* B = B0 + (size_t)s * b0_stride;
        const float* r = in0 + (size_t)m * C;
        for (int k = 0; k < C; ++k) {
            const double a = (double)r[k] - (c0 ? (double)c0[k] : 0.0);
            const float* b = B + (size_t)k * C;
            for (int j = 0; j < C; ++j) acc[j] += a * (double)b[j];
        }
        if (in1) {
            const float* r1 = in1 + (size_t)m * C;
            for (int k = 0; k < C; ++k) {
                const double a = (double)r1[k] - (c1 ? (double)c1[k] : 0.0);
                const float* b = B1 + (size_t)k * C;
                for (int j = 0; j < C; ++j) acc[j] += a * (double)b[j];
            }
        }
        float* o = out + (size_t)m * C;
        for (int j = 0; j < C; ++j) {
            double v = acc[j] + (bias ? (double)bias[(size_t)s * C + j] : 0.0) - (sub ? (double)sub[j] : 0.0);
            if (relu && !(v > 0.0) && v == v) v = 0.0;
            o[j] = (float)v;
        }
    }
}

int wc_apply_act_f32_cpu(const float* x, const float* mu, const float* A, const float* bias, const int32_t* slot,
                         int64_t N, int64_t HW, int C, int Kc, int relu, float* y, const void*, void*, size_t, wc_stream_t)
{
    if (!x || !A || !y) return WC_ERR_NULL;
    if (relu != 0 && relu != 1) return WC_ERR_ARG;
    if (N <= 0 || HW <= 0 || Kc <= 0) return WC_ERR_SHAPE;
    if (bad_channels(C)) return WC_ERR_CHANNELS;
    rows_affine(x, mu, A, (int64_t)C * C, nullptr, nullptr, nullptr, bias, nullptr, slot, N, HW, C, relu, y);
    return WC_OK;
}

int wc_apply_f32_cpu(const float* x, const float* mu, const float* A, const float* bias, const int32_t* slot,
                     int64_t N, int64_t HW, int C, int Kc, float* y, const void* plan, void* ws, size_t wsb, wc_stream_t st)
{
    return wc_apply_act_f32_cpu(x, mu, A, bias, slot, N, HW, C, Kc, 0, y, plan, ws, wsb, st);
}

int wc_bwd_reduce_f32_cpu(const float* x, const float* mu, const float* gy, const int32_t* slot, int64_t N, int64_t HW,
                          int C, int Kc, double* R, double* gsum, void*, size_t, wc_stream_t)
{
    if (!x || !gy || !R || !gsum) return WC_ERR_NULL;
    if (N <= 0 || HW <= 0 || Kc <= 0 || (!slot && Kc != 1)) return WC_ERR_SHAPE;
    if (bad_channels(C)) return WC_ERR_CHANNELS;
    const size_t CC = (size_t)C * C;
    std::memset(R, 0, sizeof(double) * CC * Kc);
    std::memset(gsum, 0, sizeof(double) * (size_t)C * Kc);
    // one task per (slot, block of 8 rows of R): walks every row of the slot's samples -- simple, deterministic (fixed
    // summation order whatever the thread count), parallel over the blocks
    const int nib = C / 8;
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = 0; k < Kc; ++k)
        for (int ib = 0; ib < nib; ++ib) {
            double* Rb = R + (size_t)k * CC + (size_t)ib * 8 * C;
            double ci[8];
            for (int u = 0; u < 8; ++u) ci[u] = mu ? (double)mu[ib * 8 + u] : 0.0;
            for (int64_t n = 0; n < N; ++n) {
                if (slot && slot[n] != k) continue;
                for (int64_t p = 0; p < HW; ++p) {
                    const size_t m = (size_t)(n * HW + p);
                    const float* g = gy + m * C;
                    const float* xr = x + m * C + ib * 8;
                    for (int u = 0; u < 8; ++u) {
                        const double f = (double)xr[u] - ci[u];
                        double* Ri = Rb + (size_t)u * C;
                        for (int j = 0; j < C; ++j) Ri[j] += f * (double)g[j];
                    }
                    if (ib == 0) { double* gs = gsum + (size_t)k * C; for (int j = 0; j < C; ++j) gs[j] += (double)g[j]; }
                }
            }
        }
    return WC_OK;
}

int wc_bwd_factor_f64_cpu(const double* R, const double* gsum, const double* W, const double* L, const float* gamma,
                          const float* A, int Kc, int C, int64_t M, double eps, int ddof, int training,
                          float* dgamma, float* dbeta, float* S, float* gmean, void*, size_t, wc_stream_t)
{
    if (!R || !gsum || !W) return WC_ERR_NULL;
    if (training && (!L || !A || !S || !gmean)) return WC_ERR_NULL;
    if (bad_channels(C)) return WC_ERR_CHANNELS;
    if (Kc <= 0 || (!gamma && Kc != 1) || (training && M <= ddof)) return WC_ERR_SHAPE;
    const size_t CC = (size_t)C * C;
    std::vector<double> t0(CC), t1(CC), t2(CC);
    if (dgamma && gamma)
        for (int k = 0; k < Kc; ++k) {                  // dgamma[k] = W R[k]
            mm(C, W, false, R + (size_t)k * CC, false, 1.0, t0.data());
            for (size_t e = 0; e < CC; ++e) dgamma[(size_t)k * CC + e] = (float)t0[e];
        }
    if (dbeta) for (size_t e = 0; e < (size_t)Kc * C; ++e) dbeta[e] = (float)gsum[e];
    if (!training) return WC_OK;
    std::vector<double> Wbar(CC, 0.0), G(CC);
    if (gamma) {                                         // Wbar = sum_k Gamma_k R_k^T
        for (int k = 0; k < Kc; ++k) {
            for (size_t e = 0; e < CC; ++e) G[e] = (double)gamma[(size_t)k * CC + e];
            mm(C, G.data(), false, R + (size_t)k * CC, true, 1.0, t0.data());
            for (size_t e = 0; e < CC; ++e) Wbar[e] += t0[e];
        }
    } else {
        for (int i = 0; i < C; ++i) for (int j = 0; j < C; ++j) Wbar[(size_t)i * C + j] = R[(size_t)j * C + i];
    }
    mm(C, W, true, Wbar.data(), false, 1.0, t0.data());                 // U1 = W^T Wbar
    mm(C, t0.data(), false, W, true, -1.0, t1.data());                  // -U1 W^T
    for (int i = 0; i < C; ++i) for (int j = i + 1; j < C; ++j) t1[(size_t)i * C + j] = 0.0;      // Lbar = tril
    mm(C, L, true, t1.data(), false, 1.0, t2.data());                   // L^T Lbar
    for (int i = 0; i < C; ++i) {                                       // P = Phi(.)
        for (int j = i + 1; j < C; ++j) t2[(size_t)i * C + j] = 0.0;
        t2[(size_t)i * C + i] *= 0.5;
    }
    mm(C, W, true, t2.data(), false, 1.0, t0.data());                   // Q1 = W^T P
    mm(C, t0.data(), false, W, false, 1.0, t1.data());                  // Q2 = Q1 W
    const double scale = 2.0 * (1.0 - eps) / (double)(M - ddof);
    for (int i = 0; i < C; ++i)
        for (int j = 0; j < C; ++j) S[(size_t)i * C + j] = (float)(scale * 0.5 * (t1[(size_t)i * C + j] + t1[(size_t)j * C + i]));
    for (int c = 0; c < C; ++c) {
        double s = 0.0;
        for (int k = 0; k < Kc; ++k)
            for (int j = 0; j < C; ++j) s += gsum[(size_t)k * C + j] * (double)A[(size_t)k * CC + (size_t)c * C + j];
        gmean[c] = (float)(s / (double)M);
    }
    return WC_OK;
}

int wc_bwd_apply_f32_cpu(const float* gy, const float* x, const float* mu, const float* At, const float* S, const float* gmean,
                         const int32_t* slot, int64_t N, int64_t HW, int C, int Kc, float* dx, void*, size_t, wc_stream_t)
{
    if (!gy || !At || !dx) return WC_ERR_NULL;
    if (S && !x) return WC_ERR_NULL;
    if (N <= 0 || HW <= 0 || Kc <= 0) return WC_ERR_SHAPE;
    if (bad_channels(C)) return WC_ERR_CHANNELS;
    rows_affine(gy, nullptr, At, (int64_t)C * C, S ? x : nullptr, mu, S, nullptr, gmean, slot, N, HW, C, 0, dx);
    return WC_OK;
}

}  // extern "C"
