/* experiment hook (tools/exp/iter_cycle.py): for every 9-state step of the host emulation that runs the IEKF to its
 * cap, at which iteration does the iterate first repeat an earlier one bit for bit, and with which period? */
#pragma once
#include <cstring>
#include <cstdio>
#include <cmath>
struct KfposIterTrace {
    int n, first, period;
    double ve[21][6];
    double c[21];
    long capped, hist_first[21], hist_period[21], never;
    double min_rel_gap[21];
};
extern "C" { extern KfposIterTrace kfpos_iter_trace_cur; }
static inline void kfpos_iter_trace_flush() {
    KfposIterTrace &t = kfpos_iter_trace_cur;
    if (t.n == 20) {
        t.capped++;
        if (t.capped % 2000 == 1) {
            for (int i = 0; i < 20; ++i) {
                double d1 = 0, d2 = 0;
                for (int k = 0; k < 3; ++k) { if (i >= 1) d1 = std::fmax(d1, std::fabs(t.ve[i][k] - t.ve[i-1][k])); if (i >= 2) d2 = std::fmax(d2, std::fabs(t.ve[i][k] - t.ve[i-2][k])); }
                std::fprintf(stderr, "  it %2d c %.10g  |dx1| %.3e |dx2| %.3e\n", i, t.c[i], d1, d2);
            }
            std::fprintf(stderr, "\n");
        }
        if (t.first >= 0) { t.hist_first[t.first]++; t.hist_period[t.period]++; }
        else t.never++;
    }
    t.n = 0;
}
static inline void kfpos_emu_iter_trace(int iter, const double *ve, double c) {
    KfposIterTrace &t = kfpos_iter_trace_cur;
    if (iter == 0) { kfpos_iter_trace_flush(); t.first = -1; t.period = 0; }
    std::memcpy(t.ve[iter], ve, 48);
    t.c[iter] = c;
    t.n = iter + 1;
    if (t.first < 0)
        for (int j = iter - 1; j >= 0; --j)
            if (std::memcmp(t.ve[j], ve, 48) == 0) { t.first = iter; t.period = iter - j; break; }
}
