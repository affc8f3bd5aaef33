// Gradient of a 3x3 / stride-2 / pad-1 max pool seen from an INPUT pixel: the sum of the pooled gradients of the
// (at most four) windows whose arg-max code points at this pixel.  Row h lies in window (h+1)>>1 and, when h is
// odd, also in the one before it; same for columns.  One window row at a time (both columns' loads in flight,
// clamped addresses, predicated use); windows are added in (row, column) order, the order of the general loop.
#pragma once
#include "vec16.h"

namespace {
template <typename T>
__device__ __forceinline__ void pool321_gather(const T* gy, const unsigned char* idx, int n, int h, int w, int c, int C,
                                               int Ho, int Wo, float (&acc)[VT<T>::V]) {
    constexpr int V = VT<T>::V;
    const int th = h + 1, tw = w + 1;
    const int hoA = th >> 1, woA = tw >> 1;
    const int rA = th & 1, sA = tw & 1;                 // code row / column inside window A (0 or 1)
    const bool vhA = hoA < Ho, vwA = woA < Wo;
    const bool vhB = rA == 0, vwB = sA == 0;            // the earlier window (code row / column 2) exists for odd h / w
    const int hA = vhA ? hoA : Ho - 1, wA = vwA ? woA : Wo - 1;
    const int hB = vhB ? hoA - 1 : hA, wB = vwB ? woA - 1 : wA;
    const int64_t base = (int64_t)n * Ho;
    const int k1 = 2 * 3 + sA, k3 = rA * 3 + sA;
#pragma unroll
    for (int q = 0; q < V; ++q) acc[q] = 0.f;
    // one window row at a time; the row predicates are wave-uniform when a wave covers one image row
    auto add_row = [&](int hh, int kB, int kA) {
        const int64_t oB = ((base + hh) * Wo + wB) * C + c, oA = ((base + hh) * Wo + wA) * C + c;
        float gB[V], gA[V];
        VT<T>::load(gy + oB, gB); VT<T>::load(gy + oA, gA);
        unsigned cB[2], cA[2];
        if constexpr (V == 8) {
            const uint2 tb = *reinterpret_cast<const uint2*>(idx + oB), ta = *reinterpret_cast<const uint2*>(idx + oA);
            cB[0] = tb.x; cB[1] = tb.y; cA[0] = ta.x; cA[1] = ta.y;
        } else {
            cB[0] = *reinterpret_cast<const unsigned int*>(idx + oB); cA[0] = *reinterpret_cast<const unsigned int*>(idx + oA);
            cB[1] = 0; cA[1] = 0;
        }
#pragma unroll
        for (int q = 0; q < V; ++q) {
            if (vwB && (int)((cB[q >> 2] >> (8 * (q & 3))) & 0xffu) == kB) acc[q] += gB[q];
            if (vwA && (int)((cA[q >> 2] >> (8 * (q & 3))) & 0xffu) == kA) acc[q] += gA[q];
        }
    };
    if (vhB) add_row(hB, 2 * 3 + 2, k1);
    if (vhA) add_row(hA, rA * 3 + 2, k3);
}
}  // namespace
