// group_api.hip -- the instance group's submission thread and its C ABI (group.hip.h says what it is for).
#include "group.hip.h"

#include "common.hip.h"

#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

using namespace xrhip;

namespace {

GroupLaunchFn g_launch[GK_COUNT] = {nullptr};

inline void relax() {
#if defined(__x86_64__)
    _mm_pause();
#else
    std::this_thread::yield();
#endif
}

constexpr int MAX_BATCH = 32;   // requests one batch takes (the launch functions cut them into chunks of XB entries)

}   // namespace

struct xrhip_group {
    int device = 0;
    hipStream_t stream[GQ_COUNT] = {nullptr};
    std::mutex m;   // the queues, `sleeping`
    std::condition_variable cv;
    std::deque<GroupRequest *> q[GQ_COUNT];
    bool sleeping = false;
    std::atomic<long> submitted{0};
    std::atomic<bool> quit{false};
    std::atomic<int> members{0};
    std::atomic<bool> profiling{false};
    std::atomic<int> busy_elsewhere{0};   // members inside a window solve (their own stream): they will not submit for a while
    std::atomic<int> sequences{0};        // front-end contexts joined == sequences in the group
    int linger_us = 0;                    // hold a batch back this long for the members that have not submitted yet (XRHIP_GROUP_LINGER_US)
    int linger_queues = 7;                // bit k: queue k lingers (XRHIP_GROUP_LINGER_QUEUES)
    std::chrono::steady_clock::time_point first_seen[GQ_COUNT];
    bool lingering[GQ_COUNT] = {false};
    std::thread th;
    // submission thread only
    bool inflight[GQ_COUNT] = {false};
    int inflight_kind[GQ_COUNT] = {0};
    hipEvent_t ev0[GQ_COUNT] = {nullptr}, ev1[GQ_COUNT] = {nullptr};
    bool timed[GQ_COUNT] = {false};
    std::mutex stats_m;
    xrhip_group_stats stats;

    void finish_timing(int k) {
        if (!timed[k]) return;
        timed[k] = false;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, ev0[k], ev1[k]) == hipSuccess) {
            std::lock_guard<std::mutex> lk(stats_m);
            stats.ms[inflight_kind[k]] += ms;
            stats.timed[inflight_kind[k]] += 1;
        } else {
            (void)hipGetLastError();
        }
    }

    // the head request and every later one of its kind whose owner has nothing older still waiting in this queue
    // -> false: requests are pending but held back (linger): the caller comes back
    bool take(int k, std::vector<GroupRequest *> &batch) {
        batch.clear();
        std::lock_guard<std::mutex> lk(m);
        std::deque<GroupRequest *> &dq = q[k];
        if (dq.empty()) {
            lingering[k] = false;
            return true;
        }
        const int kind = dq.front()->kind;
        if (linger_us > 0 && kind != GK_CALL && ((linger_queues >> k) & 1)) {
            // Members run the same frame loop: when some have submitted this kind and the others are about to, a short wait turns
            // several small launches into one -- and members that travelled in one batch come back together.  Never longer than
            // linger_us past the first pending request, and not at all for members that are busy with a window solve.
            int same = 0;
            for (GroupRequest *r : dq) same += r->kind == kind ? 1 : 0;
            const int expected = std::max(1, sequences.load(std::memory_order_relaxed) - busy_elsewhere.load(std::memory_order_relaxed));
            if (same < expected) {
                const auto now = std::chrono::steady_clock::now();
                if (!lingering[k]) {
                    lingering[k] = true;
                    first_seen[k] = now;
                }
                if (now - first_seen[k] < std::chrono::microseconds(linger_us)) return false;
            }
            lingering[k] = false;
        }
        if (kind == GK_CALL) {
            batch.push_back(dq.front());
            dq.pop_front();
            return true;
        }
        void *blocked[64];
        int nb = 0;
        for (auto it = dq.begin(); it != dq.end() && (int)batch.size() < MAX_BATCH;) {
            GroupRequest *r = *it;
            bool held = false;
            for (int i = 0; i < nb; ++i) held = held || blocked[i] == r->owner;
            if (!held && r->kind == kind) {
                batch.push_back(r);
                it = dq.erase(it);
            } else {
                if (!held && nb < 64) blocked[nb++] = r->owner;
                else if (!held) break;   // (more distinct waiting owners than the scan tracks: stop here, order stays safe)
                ++it;
            }
        }
        return true;
    }

    void launch(int k, std::vector<GroupRequest *> &batch) {
        const int kind = batch[0]->kind, n = (int)batch.size();
        const bool prof = profiling.load(std::memory_order_relaxed);
        if (prof) {
            if (!ev0[k]) {
                hipEventCreate(&ev0[k]);
                hipEventCreate(&ev1[k]);
            }
            hipEventRecord(ev0[k], stream[k]);
        }
        int rc;
        if (kind == GK_CALL) rc = batch[0]->call ? batch[0]->call(stream[k]) : XRHIP_OK;
        else if (g_launch[kind]) rc = g_launch[kind](batch.data(), n, stream[k]);
        else rc = xr_fail(XRHIP_ESTATE, "instance group: no launcher registered for this request kind");
        if (prof) {
            hipEventRecord(ev1[k], stream[k]);
            timed[k] = true;
        }
        {
            std::lock_guard<std::mutex> lk(stats_m);
            stats.batches[kind] += 1;
            stats.entries[kind] += n;
        }
        inflight_kind[k] = kind;
        const char *text = rc ? xr_err_buf() : "";
        for (GroupRequest *r : batch) {
            r->rc = rc;
            if (rc) std::snprintf(r->err, sizeof r->err, "%s", text);
            r->state.store(2, std::memory_order_release);   // the owner may reuse or free the request from here on
        }
    }

    void run() {
        hipSetDevice(device);
        std::vector<GroupRequest *> batch;
        batch.reserve(MAX_BATCH);
        auto idle_since = std::chrono::steady_clock::now();
        for (;;) {
            bool active = false;
            for (int k = 0; k < GQ_COUNT; ++k) {
                if (inflight[k]) {
                    const hipError_t qr = hipStreamQuery(stream[k]);
                    if (qr == hipErrorNotReady) {
                        active = true;
                        continue;
                    }
                    if (qr != hipSuccess) (void)hipGetLastError();   // a fault surfaces at the owners' mailbox waits
                    inflight[k] = false;
                    finish_timing(k);
                }
                if (!take(k, batch)) {   // held back for a few microseconds: stay awake
                    active = true;
                    continue;
                }
                if (batch.empty()) continue;
                launch(k, batch);
                inflight[k] = true;
                active = true;
            }
            if (active) {
                idle_since = std::chrono::steady_clock::now();
                continue;
            }
            if (quit.load(std::memory_order_acquire)) {
                std::lock_guard<std::mutex> lk(m);
                bool empty = true;
                for (int k = 0; k < GQ_COUNT; ++k) empty = empty && q[k].empty();
                if (empty) return;
                continue;
            }
            // nothing queued, nothing in flight: spin on the wake word for a while (a frame is a fraction of a millisecond), then sleep
            const long seen = submitted.load(std::memory_order_acquire);
            bool woke = false;
            for (int spin = 0; spin < 4096 && !woke; ++spin) {
                relax();
                woke = submitted.load(std::memory_order_acquire) != seen || quit.load(std::memory_order_relaxed);
            }
            if (woke) continue;
            if (std::chrono::steady_clock::now() - idle_since < std::chrono::milliseconds(3)) continue;
            std::unique_lock<std::mutex> lk(m);
            bool empty = true;
            for (int k = 0; k < GQ_COUNT; ++k) empty = empty && q[k].empty();
            if (!empty || quit.load(std::memory_order_relaxed)) continue;
            sleeping = true;
            cv.wait_for(lk, std::chrono::milliseconds(20));
            sleeping = false;
        }
    }
};

namespace xrhip {

void group_register(int kind, GroupLaunchFn fn) {
    if (kind >= 0 && kind < GK_COUNT) g_launch[kind] = fn;
}

int group_submit(xrhip_group *g, int queue, GroupRequest *r) {
    while (r->state.load(std::memory_order_acquire) == 1) relax();   // its previous use is still queued
    r->rc = 0;
    r->state.store(1, std::memory_order_release);
    bool wake;
    {
        std::lock_guard<std::mutex> lk(g->m);
        g->q[queue].push_back(r);
        wake = g->sleeping;
    }
    g->submitted.fetch_add(1, std::memory_order_release);
    if (wake) g->cv.notify_one();
    return XRHIP_OK;
}

int group_wait_launched(GroupRequest *r) {
    while (r->state.load(std::memory_order_acquire) == 1) relax();
    if (r->rc) {
        std::snprintf(xr_err_buf(), 512, "%s", r->err);
        return r->rc;
    }
    return XRHIP_OK;
}

int group_call(xrhip_group *g, int queue, void *owner, std::function<int(hipStream_t)> fn) {
    GroupRequest r;
    r.kind = GK_CALL;
    r.owner = owner;
    r.call = std::move(fn);
    group_submit(g, queue, &r);
    return group_wait_launched(&r);
}

int group_drain(xrhip_group *g, int queue, void *owner) {
    int rc = group_call(g, queue, owner, [](hipStream_t) { return XRHIP_OK; });
    if (rc) return rc;
    XR_HIP(hipStreamSynchronize(g->stream[queue]));
    return XRHIP_OK;
}

hipStream_t group_stream(xrhip_group *g, int queue) { return g->stream[queue]; }
void group_member_add(xrhip_group *g, bool front_end) {
    g->members.fetch_add(1);
    if (front_end) g->sequences.fetch_add(1);
}
void group_member_remove(xrhip_group *g, bool front_end) {
    g->members.fetch_sub(1);
    if (front_end) g->sequences.fetch_sub(1);
}
void group_busy_elsewhere(xrhip_group *g, int delta) {
    if (g) g->busy_elsewhere.fetch_add(delta, std::memory_order_relaxed);
}

int wait_flag(volatile int *flag, int seq, hipStream_t s, GroupRequest *req, const char *what) {
    if (req) {
        const int rc = group_wait_launched(req);
        if (rc) return rc;
    }
    for (unsigned long spin = 1;; ++spin) {
        if (*flag == seq) return XRHIP_OK;
        if ((spin & 0x3FFF) == 0) {
            const hipError_t q = hipStreamQuery(s);
            if (q == hipSuccess) {
                if (*flag == seq) return XRHIP_OK;
                std::snprintf(xr_err_buf(), 512, "%s: kernel retired without publishing its result", what);
                return XRHIP_ESTATE;
            }
            if (q != hipErrorNotReady) {
                (void)hipGetLastError();
                std::snprintf(xr_err_buf(), 512, "%s: stream error while waiting for the kernel", what);
                return XRHIP_EHIP;
            }
        }
    }
}

}   // namespace xrhip

extern "C" {

int xrhip_group_create(xrhip_group **out) {
    if (!out) return xr_fail(XRHIP_EINVAL, "xrhip_group_create: null argument");
    int rc = xr_require_device();
    if (rc) return rc;
    xrhip_group *g = new xrhip_group();
    std::memset(&g->stats, 0, sizeof(g->stats));
    hipGetDevice(&g->device);
    if (const char *e = std::getenv("XRHIP_GROUP_LINGER_US")) g->linger_us = std::max(0, std::atoi(e));
    if (const char *e = std::getenv("XRHIP_GROUP_LINGER_QUEUES")) g->linger_queues = std::atoi(e);
    // The group's streams carry every member's per-frame critical path; the members' own streams carry window solves and
    // marginalisations (long, one frame in five).  The runtime multiplexes a process's streams over a few hardware queues PER
    // PRIORITY LEVEL, in order within a queue: at the default priority a batch would queue behind whichever member's 100 us
    // factorisation shares its hardware queue.  Highest priority gives the three group streams hardware queues of their own.
    int prio_low = 0, prio_high = 0;
    const bool use_prio = std::getenv("XRHIP_GROUP_NO_PRIORITY") == nullptr &&
                          hipDeviceGetStreamPriorityRange(&prio_low, &prio_high) == hipSuccess && prio_high != prio_low;
    for (int k = 0; k < GQ_COUNT; ++k) {
        if (use_prio) XR_HIP(hipStreamCreateWithPriority(&g->stream[k], hipStreamNonBlocking, prio_high));
        else XR_HIP(hipStreamCreateWithFlags(&g->stream[k], hipStreamNonBlocking));
    }
    g->th = std::thread([g] { g->run(); });
    *out = g;
    return XRHIP_OK;
}

int xrhip_group_destroy(xrhip_group *g) {
    if (!g) return XRHIP_OK;
    if (g->members.load() != 0) return xr_fail(XRHIP_ESTATE, "xrhip_group_destroy: contexts are still joined to this group");
    g->quit.store(true, std::memory_order_release);
    g->submitted.fetch_add(1, std::memory_order_release);
    {
        std::lock_guard<std::mutex> lk(g->m);
    }
    g->cv.notify_all();
    g->th.join();
    for (int k = 0; k < GQ_COUNT; ++k) {
        hipStreamSynchronize(g->stream[k]);
        hipStreamDestroy(g->stream[k]);
        if (g->ev0[k]) hipEventDestroy(g->ev0[k]);
        if (g->ev1[k]) hipEventDestroy(g->ev1[k]);
    }
    delete g;
    return XRHIP_OK;
}

int xrhip_group_set_profiling(xrhip_group *g, int enable) {
    if (!g) return xr_fail(XRHIP_EINVAL, "xrhip_group_set_profiling: null group");
    g->profiling.store(enable != 0);
    return XRHIP_OK;
}

int xrhip_group_get_stats(xrhip_group *g, xrhip_group_stats *out, int reset) {
    if (!g || !out) return xr_fail(XRHIP_EINVAL, "xrhip_group_get_stats: null argument");
    std::lock_guard<std::mutex> lk(g->stats_m);
    *out = g->stats;
    if (reset) std::memset(&g->stats, 0, sizeof(g->stats));
    return XRHIP_OK;
}

}   // extern "C"
