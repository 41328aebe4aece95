// tests/host_emul/host_ex.h -- TEST INFRASTRUCTURE: the CPU executor of csrc/lap_exact.h (a workgroup's threads one after the
// other, ascending or descending) shared by the two harnesses.
#pragma once
#include <stdint.h>
struct HostEx {
    int n, order;
    int nth() const { return n; }
    template <class F> void phase(F f) {
        if (order == 0) for (int t = 0; t < n; ++t) f(t);
        else for (int t = n - 1; t >= 0; --t) f(t);
    }
    int atomic_add(int32_t *p, int v) { const int o = *p; *p = o + v; return o; }
    int atomic_max(int32_t *p, int v) { const int o = *p; if (v > o) *p = v; return o; }
};
