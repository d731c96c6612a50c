// group.hip.h -- instance group: one launch serves several sequences.
//
// The reference runs one sequence per process (process-global XRSLAMManager, xrslam-interface/src/XRSLAMManager.cpp:6-9; static id
// counters, utility/identifiable.h:23-30).  Here a process holds many instances, and on one GPU their per-frame kernels are small
// and latency-bound.  What bounds S independent sequences is the hardware-queue scheduler (tools/multiq.hip,
// profiles/r04_multi_sequence.md): with more than ~4 hardware queues busy every kernel of every queue waits 20-27 us for its queue's
// turn, and the device runs about one kernel per busy queue.  Contexts that join an xrhip_group stop launching the per-frame path for
// themselves: every such launch becomes a REQUEST (its argument block, built by the owning context exactly as for a launch of its
// own), and the group's submission threads turn all pending requests of a kind into ONE launch whose blockIdx.z selects the request
// (batch.hip.h: Batch<Args>).
//
//   * Three in-order queues with a stream and a submission thread each: GQ_KLT (frame upload, CLAHE / pyramid, LK + Harris),
//     GQ_CHAIN (the single-launch solves: kb_stage + kb_chain [+ the integration queued behind the solve]), GQ_PREINT
//     (pre-integration batches of every BA context).  Requests of one context reach the device in the order it submitted them;
//     dependencies BETWEEN queues are host-mediated in the pipeline already (a solve is only assembled once the integrations it reads
//     have been collected).
//   * One batch OF A KIND in flight per queue: while it runs, requests of that kind accumulate; when it retires, everything pending
//     of it goes out together.  An idle device launches a lone request at once, a busy one forms bigger batches -- nobody waits
//     for a straggler.  Batches of different kinds follow each other on the queue's in-order stream without a wait (frame ->
//     pyramid -> tracking is one pipeline).
//   * Per-entry arithmetic is untouched (same kernels, same block-to-work mapping, same summation order): results do not depend
//     on the batch a request happened to travel in -- tests/test_instances.py holds grouped runs to the solo runs bit for bit.
//   * Completion stays per context: every kernel publishes into its owner's pinned mailbox as before; the owner's thread spins on it.
//   * Hardware queues: the group's streams are created at high priority when the runtime has two queues per priority level
//     (GPU_MAX_HW_QUEUES=2): two queues for the batches of all members, two for the members' own work (group_api.hip).
// Round 5: a fourth queue, GQ_WINDOW (normal priority), carries the members' window rounds -- a round of refine_window is one request
// (GK_WROUND: kb_stage + kb_prior_lambda of a solve's first round, then kw_lin_all .. kw_solve_try + the first kw_trials_wide with
// blockIdx.z = member), a run of rejected trials another (GK_WTRIALS) -- instead of ~45 launches per keyframe on the member's own stream
// (ba_api.hip: launch_wround_batch; XRHIP_GROUP_NO_WINDOW_BATCH=1 is the round-4 form).  Marginalisations stay per-member launches on
// the members' own streams; the speculative linearisation stays off in a group; localize -> sub-window are two requests.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <functional>

#include "../../include/xrslam_hip.h"

namespace xrhip {

enum GroupQueue { GQ_KLT = 0, GQ_CHAIN, GQ_PREINT, GQ_WINDOW, GQ_COUNT };   // GQ_WINDOW (round 5): the members' window rounds, batched (ba_api.hip)
enum GroupKind { GK_CALL = 0, GK_UPLOAD, GK_PREPROCESS, GK_TRACK, GK_DETECT, GK_CHAIN, GK_PREINT, GK_WROUND, GK_WTRIALS, GK_COUNT };
constexpr int GK_GATE_SLOT = 11;   // statistics slot of the frame gate (xrhip_group_stats has 12)

struct GroupRequest {
    int kind = GK_CALL;
    void *owner = nullptr;     // the submitting context: its requests are launched in submission order
    void *payload = nullptr;   // kind-specific argument block; owned by the submitter, read when the batch is launched
    std::function<int(hipStream_t)> call;   // GK_CALL: anything that is not batched, run alone by the submission thread in queue order
    std::atomic<int> state{0};   // 0 idle, 1 queued, 2 launched (its kernels are on the queue's stream)
    int rc = 0;                  // what the launch returned
    char err[256] = {0};
};

// Turns `n` pending requests of one kind into launches on `s` (chunks of XB entries); registered once per kind by the translation
// unit that owns the kernels.  Returns an XRHIP_* code (the text is taken from the calling thread's xr_err_buf).
// `side`: a second stream of the queue (may be null) for the part of a batch nobody waits for before the queue's next batch -- the
// launch function orders it behind what it depends on with an event (the front end's Harris passes behind the tracking launch).
typedef int (*GroupLaunchFn)(GroupRequest **reqs, int n, hipStream_t s, hipStream_t side);
void group_register(int kind, GroupLaunchFn fn);

// the owner's thread: hand a request over / wait until its kernels are on the stream (returns its rc, error text copied)
int group_submit(xrhip_group *g, int queue, GroupRequest *r);
int group_wait_launched(GroupRequest *r);
// a request that is run alone, in order: fn(stream) on the submission thread; returns when it has been issued
int group_call(xrhip_group *g, int queue, void *owner, std::function<int(hipStream_t)> fn);
// ... and when everything submitted to `queue` before it has completed on the device
int group_drain(xrhip_group *g, int queue, void *owner);
hipStream_t group_stream(xrhip_group *g, int queue);
int group_device(xrhip_group *g);   // the device the group's streams live on: a context of another device must not join
hipStream_t group_side_stream(xrhip_group *g, int queue);
void group_member_add(xrhip_group *g, bool front_end);   // front_end: a KLT context, i.e. one more sequence in the group
void group_member_remove(xrhip_group *g, bool front_end);
// a member enters (+1) / leaves (-1) a stretch of work on its own stream (a window solve): the group does not wait for it
void group_busy_elsewhere(xrhip_group *g, int delta);

// Frame gate (round 5).  Which requests share a launch used to be a matter of coincidence (1.1 - 2 requests per launch at 8 members,
// profiles/r04_multi_sequence.md; a plain linger makes it worse: nothing keeps the members in phase, profiles/r05_multi_sequence.md).
// The members run the same frame loop; if they START their frames together, their uploads, pyramids, tracking launches and solves
// arrive within microseconds of each other and one launch carries all of them.  group_gate_arrive is called by a sequence's front end
// (`owner`: its KLT context) when a new frame is about to be uploaded: it returns once every member that is expected has arrived --
// expected: registered, not marked busy (a keyframe's window solve and marginalisation take several ordinary frames: that member tells
// the group and rejoins at a later frame boundary), not absent (a member that failed to show up within the timeout -- a stream that has
// ended, a driver that is not running -- stops being waited for until it comes back).  Purely a matter of WHEN launches are issued:
// what a member computes does not change (tests/test_instances.py holds members to their solo runs bit for bit).
// MEASURED (profiles/r05_multi_sequence.md): with the gate a launch carries 4.9 frames (pyramid), 3.8 (tracking), 2.9 (solves) instead of
// 1.1 / 1.5 / 2.0 at 8 members, a member's frame drops from 1.32 to 1.09 ms -- and the members then spend 0.42 ms per frame AT the gate:
// 5302 frames/s against 5599 without.  A keyframe (one frame in four) keeps a member away for ~2.8 ms of un-batched window rounds and
// marginalisation, i.e. 44 % of the members are not in the cohort at any moment, and the cohort waits for its slowest member every
// frame.  Off by default (XRHIP_GROUP_GATE=1 switches it on); what would make it pay is batching the window rounds as well.
void group_gate_register(xrhip_group *g, void *owner);
void group_gate_unregister(xrhip_group *g, void *owner);
void group_gate_busy(xrhip_group *g, void *owner, bool busy);
void group_gate_arrive(xrhip_group *g, void *owner);

// Spin until *flag == seq (a kernel's last store into pinned memory).  The stream is polled now and then so that a faulted kernel
// becomes an error instead of a hang -- once the request (if any) is known to be launched: a shared stream may be idle while the
// request still waits in the group's queue.
// (s2: a second stream the publishing kernel may be on)
int wait_flag(volatile int *flag, int seq, hipStream_t s, GroupRequest *req, const char *what, hipStream_t s2 = nullptr);

}   // namespace xrhip
