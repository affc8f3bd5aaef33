// Host-side entry points of libiif_amd.so (no device code).
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include <atomic>

#include <hip/hip_runtime.h>

#include "../../include/iif_amd.h"

extern "C" const char* iif_version(void) { return "iif_amd 0.1.0 gfx950"; }

// ---- compute-unit budget of the persistent grids (see common.h: iif_persistent_cus) -------------------------------------
namespace {
std::atomic<int> g_cu_budget{0};          // 0: the whole device
int device_cus() {
    static const int n = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) v = 0;
        return v > 0 ? v : 256;
    }();
    return n;
}
}  // namespace

int iif_persistent_cus() {
    const int n = device_cus(), b = g_cu_budget.load(std::memory_order_relaxed);
    return (b > 0 && b < n) ? b : n;
}

// Blocks of a persistent grid launched in whole units (8 * S blocks: the S slices of eight tile sequences, one per XCD).  Under a
// CU budget the grid is the largest whole-unit count within the budget - unless that rounding would give up more CUs than the
// reservation itself asked for (budget 240 and units of 64 or 128: 192 / 128 blocks, a 25-50 % smaller grid for a 6 % reservation):
// such launches keep the device's whole-unit grid (round-5 advice).
int iif_persistent_grid(int unit) {
    const int n = device_cus(), b = iif_persistent_cus();
    if (unit <= 0) return b;
    const int g = b / unit * unit, full = n / unit * unit;
    return (full - g > n - b) ? full : g;
}

extern "C" int iif_set_cu_budget(int cus) {
    if (cus < 0) return IIF_EINVAL;
    // whole groups of 8 (one block per XCD and group), at least 64: below that the persistent kernels refuse small layers
    if (cus != 0) { cus = cus / 8 * 8; if (cus < 64) cus = 64; }
    g_cu_budget.store(cus, std::memory_order_relaxed);
    return IIF_OK;
}

extern "C" int iif_get_cu_budget(void) { return iif_persistent_cus(); }

namespace {

// Inverse of the standard normal CDF.  scipy.special.ndtri (used at
// classification/custom.py:20) is Cephes' rational approximation; here the
// root of Phi(x) - p is refined by Newton steps on libm's erfc, which agrees
// with it to a few ulp of double — far below float32 resolution.
double ndtri_newton(double p) {
    if (p <= 0.0) return -INFINITY;
    if (p >= 1.0) return INFINITY;
    const bool upper = p > 0.5;
    const double q = upper ? 1.0 - p : p;  // tail probability, <= 0.5
    double t = sqrt(-2.0 * log(q));
    // Abramowitz-Stegun 26.2.23 start value for the lower-tail quantile (negative)
    double x = -(t - (2.515517 + 0.802853 * t + 0.010328 * t * t) /
                         (1.0 + 1.432788 * t + 0.189269 * t * t + 0.001308 * t * t * t));
    const double inv_sqrt2 = 0.70710678118654752440, inv_sqrt2pi = 0.39894228040143267794;
    for (int it = 0; it < 60; ++it) {
        const double cdf = 0.5 * erfc(-x * inv_sqrt2);       // accurate in the lower tail
        const double pdf = inv_sqrt2pi * exp(-0.5 * x * x);
        const double f = cdf - q;
        // Halley step
        const double dx = f / (pdf - 0.5 * f * (-x));
        x -= dx;
        if (fabs(dx) <= 1e-16 * fabs(x) || dx == 0.0) break;
    }
    return upper ? -x : x;
}

}  // namespace

extern "C" int iif_build_table(const int64_t* counts, int C, int variant, int norm_p, float* out) {
    if (!counts || !out || C <= 0 || norm_p < 0) return IIF_EINVAL;
    if (variant < IIF_RAW || variant > IIF_BASE10) return IIF_EINVAL;
    int64_t total = 0;
    for (int i = 0; i < C; ++i) {
        if (counts[i] < 0) return IIF_EINVAL;
        total += counts[i];
    }
    const double S = (double)total;
    for (int i = 0; i < C; ++i) {
        const double f = (double)counts[i];
        double v;
        switch (variant) {
            case IIF_RAW: v = log(S / f); break;
            case IIF_SMOOTH: v = log((S + 1.0) / (f + 1.0)) + 1.0; break;
            case IIF_REL: v = log((S - f) / f); break;
            case IIF_NORMIT: v = -ndtri_newton(f / S); break;
            case IIF_GOMBIT: v = -log(-log(1.0 - (f / S))); break;
            case IIF_BASE2: v = log2(S / f); break;
            default: v = log10(S / f); break;
        }
        out[i] = (float)v;
    }
    if (norm_p > 0) {
        // torch.norm(v, p) on the float32 vector (custom.py:25-26); accumulate in double
        double acc = 0.0;
        for (int i = 0; i < C; ++i) acc += pow(fabs((double)out[i]), (double)norm_p);
        const float nrm = (float)pow(acc, 1.0 / (double)norm_p);
        for (int i = 0; i < C; ++i) out[i] = out[i] / nrm;
    }
    return IIF_OK;
}
