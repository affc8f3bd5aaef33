/* Plain-C (double precision) restatement of the IIF softmax cross-entropy.
 * TEST INFRASTRUCTURE — only tests/, smoke() and bench.py's cpu_baseline may use
 * it.  Follows classification/custom.py:28-36 (z = pred*iif; CrossEntropyLoss(
 * reduction='none', weight); mean|sum) and, for the row weights / ignore_index,
 * mmdet/models/losses/iif_loss.py:184-202 + losses/utils.py:42-55.
 * Pinned against tests/golden/g4_loss.npz (taken from the reference itself) by
 * tests/test_oracle_golden.py::test_c_oracle_matches_golden.                  */
#include <math.h>
#include <stdint.h>
#include <stddef.h>

/* returns scale * sum_i r_i; rows[i] = r_i; dpred (nullable) = gradient */
double iif_oracle_ce(const float* pred, const float* table, const int64_t* ta, const int64_t* tb,
                     double lam, const float* row_w, const float* cls_w, int64_t ignore,
                     double scale, int B, int C, double* rows, double* dpred) {
    double total = 0.0;
    for (int i = 0; i < B; ++i) {
        const float* x = pred + (size_t)i * C;
        double m = -INFINITY;
        for (int c = 0; c < C; ++c) {
            double z = (double)(float)(x[c] * table[c]);      /* the fp32 product the reference forms */
            if (z > m) m = z;
        }
        double s = 0.0;
        for (int c = 0; c < C; ++c) s += exp((double)(float)(x[c] * table[c]) - m);
        const double lse = m + log(s);
        const double rw = row_w ? row_w[i] : 1.0;
        int64_t a = ta[i], b = tb ? tb[i] : ignore;
        double la = tb ? lam : 1.0, lb = tb ? 1.0 - lam : 0.0;
        double wa = 0.0, wb = 0.0, r = 0.0;
        if (a != ignore && a >= 0 && a < C) { wa = la * (cls_w ? cls_w[a] : 1.0); r += wa * (lse - (double)(float)(x[a] * table[a])); }
        if (tb && b != ignore && b >= 0 && b < C) { wb = lb * (cls_w ? cls_w[b] : 1.0); r += wb * (lse - (double)(float)(x[b] * table[b])); }
        rows[i] = rw * r;
        total += rows[i];
        if (dpred) {
            for (int c = 0; c < C; ++c) {
                double p = exp((double)(float)(x[c] * table[c]) - m) / s;
                double g = (wa + wb) * p - (c == a ? wa : 0.0) - ((tb && c == b) ? wb : 0.0);
                dpred[(size_t)i * C + c] = scale * rw * g * (double)table[c];
            }
        }
    }
    return scale * total;
}
