// group_api.hip -- the instance group's submission threads and its C ABI (group.hip.h says what it is for).
#include "group.hip.h"

#include "common.hip.h"

#include <cstdio>
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

// One queue: its requests, its stream (and a side stream for work that need not hold up the next batch), its submission thread.
struct GroupQueueState {
    int index = 0;
    hipStream_t stream = nullptr, side = nullptr;
    std::mutex m;   // the request list, `sleeping`
    std::condition_variable cv;
    std::deque<GroupRequest *> q;
    bool sleeping = false;
    std::atomic<long> submitted{0};
    std::thread th;
    // submission thread only
    struct Inflight {
        int kind;
        hipEvent_t begin, end;
    };
    std::deque<Inflight> inflight;   // batches on the stream, oldest first (an in-order stream retires them in that order)
    std::vector<hipEvent_t> free_events;
    std::chrono::steady_clock::time_point first_seen;
    bool lingering = false;
};

struct xrhip_group {
    int device = 0;
    int queue_split = 0;                  // 1: two hardware queues carry the batches, two the members' window solves (xrhip_group_queue_split)
    GroupQueueState qs[GQ_COUNT];
    std::atomic<bool> quit{false};
    std::atomic<int> members{0};
    std::atomic<int> sequences{0};        // front-end contexts joined == sequences in the group
    std::atomic<bool> profiling{false};
    std::atomic<int> busy_elsewhere{0};   // members inside a window solve (their own stream): they will not submit for a while
    std::atomic<int> linger_us{0};        // hold a batch back this long for the members that have not submitted yet (XRHIP_GROUP_LINGER_US)
    int linger_queues = 7;                // bit k: queue k lingers (XRHIP_GROUP_LINGER_QUEUES)
    int preint_shares = -1;               // >= 0: the pre-integration queue launches into that queue's stream
    bool per_kind = true;                 // a batch waits for the previous batch OF ITS KIND only (XRHIP_GROUP_PER_KIND=0: for any batch)
    std::mutex stats_m;
    xrhip_group_stats stats;

    // frame gate (group.hip.h)
    struct GateMember {
        void *owner = nullptr;
        bool busy = false, absent = false;
        unsigned arrived_gen = ~0u;   // the generation this member is waiting in
    };
    std::mutex gate_m;                      // gate_members, and every transition of the words below
    std::vector<GateMember> gate_members;
    std::atomic<unsigned> gate_gen{0};      // bumped when the gate opens
    std::atomic<int> gate_active{0};        // members that are neither busy nor absent
    std::atomic<int> gate_waiting{0};       // ... of which at the gate right now
    std::atomic<bool> gate_used{false};     // somebody has been through the gate: the linger below counts with it
    std::atomic<bool> gate_on{false};       // XRHIP_GROUP_GATE=1 switches it on (measured: bigger batches, no more frames per second -- group.hip.h);
                                            // too many timeouts in a row switch it off again: members are not driven concurrently
    int gate_timeout_us = 2500;             // XRHIP_GROUP_GATE_TIMEOUT_US: longer than a frame -- members that are out of phase (at the start, after a
                                            // straggler) meet at the gate within one frame; a member that is really gone costs the others this once
    int gate_timeouts_in_a_row = 0;
    long long gate_opens = 0, gate_full = 0, gate_timeouts = 0;

    void gate_recount_locked() {
        int a = 0, w = 0;
        const unsigned gen = gate_gen.load(std::memory_order_relaxed);
        for (const GateMember &m : gate_members) {
            if (m.busy || m.absent) continue;
            ++a;
            if (m.arrived_gen == gen) ++w;
        }
        gate_active.store(a, std::memory_order_relaxed);
        gate_waiting.store(w, std::memory_order_relaxed);
    }
    // every expected member is here (and at least one is)
    bool gate_ready_locked() const {
        const unsigned gen = gate_gen.load(std::memory_order_relaxed);
        bool any = false;
        for (const GateMember &m : gate_members) {
            if (m.arrived_gen == gen) {
                any = true;
                continue;
            }
            if (!m.busy && !m.absent) return false;
        }
        return any;
    }
    void gate_open_locked() {
        gate_opens++;
        gate_gen.fetch_add(1, std::memory_order_release);
        gate_recount_locked();
    }

    hipEvent_t take_event(GroupQueueState &Q) {
        if (!Q.free_events.empty()) {
            hipEvent_t e = Q.free_events.back();
            Q.free_events.pop_back();
            return e;
        }
        hipEvent_t e = nullptr;
        hipEventCreate(&e);
        return e;
    }

    // batches whose last kernel has finished leave the in-flight list (in order); their duration goes into the statistics
    void retire(GroupQueueState &Q) {
        while (!Q.inflight.empty()) {
            GroupQueueState::Inflight &f = Q.inflight.front();
            const hipError_t qr = hipEventQuery(f.end);
            if (qr == hipErrorNotReady) return;
            if (qr != hipSuccess) (void)hipGetLastError();   // a fault surfaces at the owners' mailbox waits
            if (f.begin) {
                float ms = 0.f;
                if (qr == hipSuccess && hipEventElapsedTime(&ms, f.begin, f.end) == hipSuccess) {
                    std::lock_guard<std::mutex> lk(stats_m);
                    stats.ms[f.kind] += ms;
                    stats.timed[f.kind] += 1;
                } else {
                    (void)hipGetLastError();
                }
                Q.free_events.push_back(f.begin);
            }
            Q.free_events.push_back(f.end);
            Q.inflight.pop_front();
        }
    }

    // The head request and every later one of its kind whose owner has nothing older still waiting in this queue -- once no batch
    // of that kind is in flight any more: while one runs, requests of its kind accumulate and leave together.  Batches of
    // DIFFERENT kinds follow each other on the stream without a wait in between (frame -> pyramid -> tracking of the same members
    // is one pipeline on an in-order stream: nothing is gained by holding its stages apart).
    // -> false: requests are pending but held back: the caller comes back
    bool take(GroupQueueState &Q, std::vector<GroupRequest *> &batch) {
        batch.clear();
        std::lock_guard<std::mutex> lk(Q.m);
        std::deque<GroupRequest *> &dq = Q.q;
        if (dq.empty()) {
            Q.lingering = false;
            return true;
        }
        const int kind = dq.front()->kind;
        for (const GroupQueueState::Inflight &f : Q.inflight)
            if (!per_kind || f.kind == kind || kind == GK_CALL || f.kind == GK_CALL) return false;
        const int linger_now = linger_us.load(std::memory_order_relaxed);
        if (linger_now > 0 && kind != GK_CALL && ((linger_queues >> Q.index) & 1)) {
            // Members run the same frame loop: when some have submitted this kind and the others are about to, a short wait turns
            // several small launches into one -- and members that travelled in one batch come back together.  Never longer than
            // linger_us past the first pending request, and not at all for members that are busy with a window solve.
            int same = 0;
            for (GroupRequest *r : dq) same += r->kind == kind ? 1 : 0;
            // with the frame gate in use: the members that started this frame together (neither busy with a keyframe, nor absent, nor
            // waiting at the gate for the next frame); without it: everybody who is not inside a window solve
            const int expected = gate_used.load(std::memory_order_relaxed)
                                     ? std::max(1, gate_active.load(std::memory_order_relaxed) - gate_waiting.load(std::memory_order_relaxed))
                                     : std::max(1, sequences.load(std::memory_order_relaxed) - busy_elsewhere.load(std::memory_order_relaxed));
            if (same < expected) {
                const auto now = std::chrono::steady_clock::now();
                if (!Q.lingering) {
                    Q.lingering = true;
                    Q.first_seen = now;
                }
                if (now - Q.first_seen < std::chrono::microseconds(linger_now)) return false;
            }
            Q.lingering = false;
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

    void launch(GroupQueueState &Q, std::vector<GroupRequest *> &batch) {
        const int kind = batch[0]->kind, n = (int)batch.size();
        GroupQueueState::Inflight f{kind, nullptr, nullptr};
        if (profiling.load(std::memory_order_relaxed)) {
            f.begin = take_event(Q);
            hipEventRecord(f.begin, Q.stream);
        }
        int rc;
        if (kind == GK_CALL) rc = batch[0]->call ? batch[0]->call(Q.stream) : XRHIP_OK;
        else if (g_launch[kind]) rc = g_launch[kind](batch.data(), n, Q.stream, Q.side);
        else rc = xr_fail(XRHIP_ESTATE, "instance group: no launcher registered for this request kind");
        f.end = take_event(Q);
        hipEventRecord(f.end, Q.stream);
        Q.inflight.push_back(f);
        {
            std::lock_guard<std::mutex> lk(stats_m);
            stats.batches[kind] += 1;
            stats.entries[kind] += n;
        }
        const char *text = rc ? xr_err_buf() : "";
        for (GroupRequest *r : batch) {
            r->rc = rc;
            if (rc) std::snprintf(r->err, sizeof r->err, "%s", text);
            r->state.store(2, std::memory_order_release);   // the owner may reuse or free the request from here on
        }
    }

    void run(GroupQueueState &Q) {
        hipSetDevice(device);
        std::vector<GroupRequest *> batch;
        batch.reserve(MAX_BATCH);
        auto idle_since = std::chrono::steady_clock::now();
        for (;;) {
            retire(Q);
            bool active = !Q.inflight.empty();
            if (!take(Q, batch)) {
                active = true;   // held back (a batch of the kind in flight, or lingering): stay awake
            } else if (!batch.empty()) {
                launch(Q, batch);
                active = true;
            }
            if (active) {
                idle_since = std::chrono::steady_clock::now();
                relax();   // a batch in flight or a held-back request: poll, but leave the core's issue slots to the members' threads
                continue;
            }
            if (quit.load(std::memory_order_acquire)) {
                std::lock_guard<std::mutex> lk(Q.m);
                if (Q.q.empty()) return;
                continue;
            }
            // nothing queued, nothing in flight: spin on the wake word for a while (a frame is a fraction of a millisecond), then sleep
            const long seen = Q.submitted.load(std::memory_order_acquire);
            bool woke = false;
            for (int spin = 0; spin < 4096 && !woke; ++spin) {
                relax();
                woke = Q.submitted.load(std::memory_order_acquire) != seen || quit.load(std::memory_order_relaxed);
            }
            if (woke) continue;
            if (std::chrono::steady_clock::now() - idle_since < std::chrono::milliseconds(3)) continue;
            std::unique_lock<std::mutex> lk(Q.m);
            if (!Q.q.empty() || quit.load(std::memory_order_relaxed)) continue;
            Q.sleeping = true;
            Q.cv.wait_for(lk, std::chrono::milliseconds(20));
            Q.sleeping = false;
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
    GroupQueueState &Q = g->qs[queue];
    bool wake;
    {
        std::lock_guard<std::mutex> lk(Q.m);
        Q.q.push_back(r);
        wake = Q.sleeping;
    }
    Q.submitted.fetch_add(1, std::memory_order_release);
    if (wake) Q.cv.notify_one();
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
    XR_HIP(hipStreamSynchronize(g->qs[queue].stream));
    if (g->qs[queue].side) XR_HIP(hipStreamSynchronize(g->qs[queue].side));
    return XRHIP_OK;
}

hipStream_t group_stream(xrhip_group *g, int queue) { return g->qs[queue].stream; }
int group_device(xrhip_group *g) { return g->device; }
hipStream_t group_side_stream(xrhip_group *g, int queue) { return g->qs[queue].side; }
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

void group_gate_register(xrhip_group *g, void *owner) {
    std::lock_guard<std::mutex> lk(g->gate_m);
    for (const xrhip_group::GateMember &m : g->gate_members)
        if (m.owner == owner) return;
    xrhip_group::GateMember m;
    m.owner = owner;
    m.absent = true;   // counted from its first arrival on
    g->gate_members.push_back(m);
    g->gate_recount_locked();
}
void group_gate_unregister(xrhip_group *g, void *owner) {
    std::lock_guard<std::mutex> lk(g->gate_m);
    for (size_t i = 0; i < g->gate_members.size(); ++i)
        if (g->gate_members[i].owner == owner) {
            g->gate_members.erase(g->gate_members.begin() + (long)i);
            break;
        }
    if (g->gate_ready_locked()) g->gate_open_locked();   // (the others may have been waiting for this one)
    else g->gate_recount_locked();
}
void group_gate_busy(xrhip_group *g, void *owner, bool busy) {
    if (!g) return;
    std::lock_guard<std::mutex> lk(g->gate_m);
    for (xrhip_group::GateMember &m : g->gate_members)
        if (m.owner == owner) m.busy = busy;
    if (busy && g->gate_ready_locked()) g->gate_open_locked();
    else g->gate_recount_locked();
}
void group_gate_arrive(xrhip_group *g, void *owner) {
    if (!g || !g->gate_on.load(std::memory_order_relaxed)) return;
    unsigned my_gen;
    {
        std::lock_guard<std::mutex> lk(g->gate_m);
        xrhip_group::GateMember *me = nullptr;
        for (xrhip_group::GateMember &m : g->gate_members)
            if (m.owner == owner) me = &m;
        if (!me || !g->gate_on.load(std::memory_order_relaxed)) return;
        g->gate_used.store(true, std::memory_order_relaxed);
        me->absent = false;
        my_gen = g->gate_gen.load(std::memory_order_relaxed);
        me->arrived_gen = my_gen;
        if (g->gate_ready_locked()) {
            g->gate_full++;
            g->gate_timeouts_in_a_row = 0;
            g->gate_open_locked();
            return;
        }
        g->gate_recount_locked();
    }
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spin = 1;; ++spin) {
        if (g->gate_gen.load(std::memory_order_acquire) != my_gen) return;
        relax();
        if ((spin & 63) != 0) continue;
        if (std::chrono::steady_clock::now() - t0 < std::chrono::microseconds(g->gate_timeout_us)) continue;
        // Somebody who was expected has not come: stop waiting for them (they rejoin when they arrive).  A group whose gate times
        // out again and again is not being driven by concurrent threads (one thread feeding its sequences in turn): the gate goes.
        std::lock_guard<std::mutex> lk(g->gate_m);
        if (g->gate_gen.load(std::memory_order_relaxed) != my_gen) return;
        for (xrhip_group::GateMember &m : g->gate_members)
            if (!m.busy && !m.absent && m.arrived_gen != my_gen) m.absent = true;
        g->gate_timeouts++;
        if (++g->gate_timeouts_in_a_row >= 16) {
            g->gate_on.store(false);
            g->gate_used.store(false, std::memory_order_relaxed);
            g->linger_us.store(0, std::memory_order_relaxed);
        }
        g->gate_open_locked();
        return;
    }
}

int wait_flag(volatile int *flag, int seq, hipStream_t s, GroupRequest *req, const char *what, hipStream_t s2) {
    if (req) {
        const int rc = group_wait_launched(req);
        if (rc) return rc;
    }
    for (unsigned long spin = 1;; ++spin) {
        if (*flag == seq) return XRHIP_OK;
        if ((spin & 0x3FFF) == 0) {
            hipError_t q = hipStreamQuery(s);
            if (q == hipSuccess && s2) q = hipStreamQuery(s2);
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
    // the frame gate (group.hip.h) lines the members' frames up; behind it a short linger collects the requests of the members that
    // started the frame together (they arrive within microseconds of each other)
    if (const char *e = std::getenv("XRHIP_GROUP_GATE")) g->gate_on.store(std::atoi(e) != 0);
    if (const char *e = std::getenv("XRHIP_GROUP_GATE_TIMEOUT_US")) g->gate_timeout_us = std::max(1, std::atoi(e));
    g->linger_us.store(g->gate_on.load() ? 40 : 0);
    if (const char *e = std::getenv("XRHIP_GROUP_LINGER_US")) g->linger_us.store(std::max(0, std::atoi(e)));
    if (const char *e = std::getenv("XRHIP_GROUP_LINGER_QUEUES")) g->linger_queues = std::atoi(e);
    if (const char *e = std::getenv("XRHIP_GROUP_PER_KIND")) g->per_kind = std::atoi(e) != 0;
    // Hardware queues.  The device runs about four of them side by side; with more in use every kernel of every queue waits 20-27 us
    // for its queue's turn (tools/multiq.hip), with fewer the runtime maps unrelated streams onto one queue and they wait for each
    // other.  The runtime gives every stream PRIORITY LEVEL a pool of GPU_MAX_HW_QUEUES queues of its own -- so with
    // GPU_MAX_HW_QUEUES=2 (set before the process's first HIP call; bench.py does) and the group's streams at high
    // priority the four queues are split by role: two carry the batched per-frame launches of all members, two the members' own
    // window solves and marginalisations (a batch no longer queues behind a member's 100 us factorisation, and nothing is
    // time-sliced).  Measured at 10 sequences: 5690 frames/s against 4960 with four unprioritised queues and 4186 with priorities
    // on top of four (eight in use) -- profiles/r04_multi_sequence.md.  Hence: priorities on exactly when the pool is that small
    // (XRHIP_GROUP_PRIORITY=0/1 overrides).  The inverse split (members' streams high) starves the batches: 4078.
    int prio_low = 0, prio_high = 0;
    const char *qe = std::getenv("GPU_MAX_HW_QUEUES"), *pe = std::getenv("XRHIP_GROUP_PRIORITY");
    const bool want_prio = pe ? std::atoi(pe) != 0 : (qe && std::atoi(qe) > 0 && std::atoi(qe) <= 2);
    const bool use_prio = want_prio && hipDeviceGetStreamPriorityRange(&prio_low, &prio_high) == hipSuccess && prio_high != prio_low;
    // The split above rests on a process-global knob an embedding application may not know about.  A library does not write into its
    // host's stderr unasked: the condition is recorded in the group and can be queried (xrhip_group_queue_split: 1 = the
    // 2 + 2 split is in effect, 0 = the slower unprioritised arrangement; bench.py prints it); XRHIP_GROUP_VERBOSE=1 says it once.
    g->queue_split = use_prio ? 1 : 0;
    if (!use_prio && std::getenv("XRHIP_GROUP_VERBOSE")) {
        static std::atomic<bool> warned{false};
        if (!warned.exchange(true))
            std::fprintf(stderr, "xrslam_hip: instance group created with GPU_MAX_HW_QUEUES=%s: the group's queue split (two hardware queues for "
                                 "the batches, two for the members' window solves) needs GPU_MAX_HW_QUEUES=2 in the environment BEFORE the "
                                 "process's first HIP call; without it grouped throughput is ~15 %% lower (DESIGN.md 4.9)\n", qe ? qe : "unset");
    }
    // XRHIP_GROUP_SIDE_STREAM=1: the front end's Harris passes on a second stream of the KLT queue (nobody waits for them before the
    // next frame's tracks have been digested).  Off by default: one more stream competing for the hardware queues cost more than the
    // overlap gave (10 sequences: 5272 with, 5518 without).
    const bool side_stream = std::getenv("XRHIP_GROUP_SIDE_STREAM") != nullptr;
    for (int k = 0; k < GQ_COUNT; ++k) {
        GroupQueueState &Q = g->qs[k];
        Q.index = k;
        // (the window queue -- the members' batched window rounds, round 5 -- stays at NORMAL priority: the per-frame batches go first)
        hipError_t e = (use_prio && k != GQ_WINDOW) ? hipStreamCreateWithPriority(&Q.stream, hipStreamNonBlocking, prio_high)
                                                    : hipStreamCreateWithFlags(&Q.stream, hipStreamNonBlocking);
        if (e == hipSuccess && k == GQ_KLT && side_stream) e = hipStreamCreateWithFlags(&Q.side, hipStreamNonBlocking);
        if (e != hipSuccess) {   // no thread has been started yet: give back what exists and report
            for (int j = 0; j <= k; ++j) {
                if (g->qs[j].stream) hipStreamDestroy(g->qs[j].stream);
                if (g->qs[j].side) hipStreamDestroy(g->qs[j].side);
            }
            delete g;
            return xr_fail_hip(e, "hipStreamCreate (instance group queue)", __FILE__, __LINE__);
        }
    }
    // XRHIP_GROUP_PREINT_STREAM=klt|chain: the pre-integration queue launches into that queue's stream instead of one of its own (its
    // thread, its batches and its gate stay): with two hardware queues for the group's three streams the runtime pairs two of them
    // anyway -- this picks the pair.
    if (const char *e = std::getenv("XRHIP_GROUP_PREINT_STREAM")) {
        const int to = !std::strcmp(e, "klt") ? GQ_KLT : (!std::strcmp(e, "chain") ? GQ_CHAIN : -1);
        if (to >= 0) {
            hipStreamDestroy(g->qs[GQ_PREINT].stream);
            g->qs[GQ_PREINT].stream = g->qs[to].stream;
            g->preint_shares = to;
        }
    }
    for (int k = 0; k < GQ_COUNT; ++k) {
        GroupQueueState *Q = &g->qs[k];
        Q->th = std::thread([g, Q] { g->run(*Q); });
    }
    *out = g;
    return XRHIP_OK;
}

int xrhip_group_destroy(xrhip_group *g) {
    if (!g) return XRHIP_OK;
    if (g->members.load() != 0) return xr_fail(XRHIP_ESTATE, "xrhip_group_destroy: contexts are still joined to this group");
    g->quit.store(true, std::memory_order_release);
    for (int k = 0; k < GQ_COUNT; ++k) {
        GroupQueueState &Q = g->qs[k];
        Q.submitted.fetch_add(1, std::memory_order_release);
        {
            std::lock_guard<std::mutex> lk(Q.m);
        }
        Q.cv.notify_all();
    }
    for (int k = 0; k < GQ_COUNT; ++k) {
        GroupQueueState &Q = g->qs[k];
        Q.th.join();
        const bool own_stream = !(k == GQ_PREINT && g->preint_shares >= 0);
        if (own_stream) hipStreamSynchronize(Q.stream);
        if (Q.side) hipStreamSynchronize(Q.side);
        for (auto &f : Q.inflight) {
            if (f.begin) hipEventDestroy(f.begin);
            hipEventDestroy(f.end);
        }
        for (hipEvent_t e : Q.free_events) hipEventDestroy(e);
        if (own_stream) hipStreamDestroy(Q.stream);
        if (Q.side) hipStreamDestroy(Q.side);
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
    {
        std::lock_guard<std::mutex> lk(g->stats_m);
        *out = g->stats;
        if (reset) std::memset(&g->stats, 0, sizeof(g->stats));
    }
    // the frame gate's slot -- times it opened / with every expected member present / by timeout
    std::lock_guard<std::mutex> lk(g->gate_m);
    out->batches[GK_GATE_SLOT] = g->gate_opens;
    out->entries[GK_GATE_SLOT] = g->gate_full;
    out->timed[GK_GATE_SLOT] = g->gate_timeouts;
    out->ms[GK_GATE_SLOT] = g->gate_on.load() ? 1.0 : 0.0;
    if (reset) g->gate_opens = g->gate_full = g->gate_timeouts = 0;
    return XRHIP_OK;
}

int xrhip_group_queue_split(xrhip_group *g) {
    if (!g) return xr_fail(XRHIP_EINVAL, "xrhip_group_queue_split: null group");
    return g->queue_split;
}

}   // extern "C"
