// hostprof.hpp -- development aid: named host wall-clock accumulators (no HIP dependency).
// XRHIP_HOSTPROF=1 makes xrhip_klt_destroy / XRSLAMDestroy dump them to stderr.
#pragma once
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>

namespace xrhip {

struct HostProf {
    static constexpr int N = 32;
    double sec[N] = {0};
    long calls[N] = {0};
    const char *name[N] = {nullptr};
};
inline HostProf g_hostprof;
// Process-wide accumulators: written only when XRHIP_HOSTPROF is set (several instances / the backend thread of the pipelined
// mode then add to them without synchronisation -- a development aid, figures are approximate in that case; without the
// variable nothing is written, so instances share no mutable state through this header).
inline bool hostprof_enabled() {
    static const bool on = std::getenv("XRHIP_HOSTPROF") != nullptr;
    return on;
}
struct HostProfScope {
    int slot;
    bool on = hostprof_enabled();
    std::chrono::steady_clock::time_point t0;
    HostProfScope(int s, const char *name) : slot(s) {
        if (on) {
            g_hostprof.name[s] = name;
            t0 = std::chrono::steady_clock::now();
        }
    }
    ~HostProfScope() {
        if (!on) return;
        g_hostprof.sec[slot] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        g_hostprof.calls[slot] += 1;
    }
};
inline void hostprof_dump() {
    if (!std::getenv("XRHIP_HOSTPROF")) return;
    for (int i = 0; i < HostProf::N; ++i)
        if (g_hostprof.name[i])
            std::fprintf(stderr, "[hostprof] %-28s calls %7ld  total %9.3f ms  avg %8.2f us\n", g_hostprof.name[i],
                         g_hostprof.calls[i], 1e3 * g_hostprof.sec[i], 1e6 * g_hostprof.sec[i] / std::max(1L, g_hostprof.calls[i]));
}

}   // namespace xrhip
