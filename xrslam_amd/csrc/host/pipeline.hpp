// pipeline.hpp -- per-frame orchestration of the hot path on top of the HIP C ABI.
//
// Host-side mirror of the reference's core/ layer (file:line under /root/reference/xrslam/src/xrslam):
//   XRSLAM::Detail (IMU/camera sync, pose propagation)   core/detail.cpp:15-177
//   FeatureTracker::work                                  core/feature_tracker.cpp:24-153
//   Frame::detect_keypoints / track_keypoints             map/frame.cpp:55-174
//   FrontendWorker::work                                  core/frontend_worker.cpp:28-86
//   SlidingWindowTracker (mirror_frame, localize_newframe, manage_keyframe, track_landmark,
//     refine_window, slide_window, refine_subwindow)      core/sliding_window_tracker.cpp:19-474
//   Map::marginalize_frame                                map/map.cpp:51-63
//   Solver facade (problem assembly)                      estimation/solver.cpp:84-173
// PC semantics (the default): threading off, every worker runs inline in the caller.
// System::set_threading(1) is the reference's XRSLAM_ENABLE_THREADING (utility/worker.h:16-60) with the hand-offs made
// deterministic: the sliding-window tracker of frame t runs on a thread of its own beside the feature tracker of frame t+1.
// All arithmetic of the hot path is delegated to xrhip_* (KLT, pre-integration, BA, marginalisation).
//   Initializer (SfM + visual-inertial alignment)         core/initializer.cpp:22-571   (geometry: two_view.hpp)
//   RD-VIO outlier filters (judge / update_track_status)  core/sliding_window_tracker.cpp:523-790   (parsac.hpp, epnp.hpp)
// Work that does not depend on the rest of the frame is queued early on streams of its own (marginalisation,
// re-integrations, image preprocessing): DESIGN.md section 4.5.
#pragma once
#include <chrono>
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <exception>
#include <functional>
#include <mutex>
#include <optional>
#include <stdexcept>
#include <thread>
#include <tuple>
#include <unordered_map>
#include <unordered_set>

#include "../host_select.hpp"
#include "../hostprof.hpp"
#include "ba_dump.hpp"
#include "config.hpp"
#include "map.hpp"
#include "parsac.hpp"
#include "two_view.hpp"
#include "undistort_map.hpp"

namespace xrh {

struct HipError : std::runtime_error {
    explicit HipError(const std::string &m) : std::runtime_error(m) {}
};
inline void hip_check(int rc, const char *what) {
    if (rc != 0) throw HipError(std::string(what) + ": " + xrhip_last_error());
}

struct StageTimes {   // seconds, accumulated (inspection slots feature_tracker_time / bundle_adjustor_* of the reference)
    double tracker = 0, localize = 0, refine = 0, marginalize = 0, preintegrate = 0;
    long frames = 0, solves = 0, solve_iterations = 0, marginalizations = 0, keyframes = 0;
    double ba_device_ms = 0;
    // host wall-clock seconds spent inside the C-ABI calls, and in the whole per-frame work
    double w_upload = 0, w_preprocess = 0, w_track = 0, w_detect = 0, w_preintegrate = 0, w_solve = 0, w_marginalize = 0,
           w_frame = 0;
    double w_preintegrate_ft = 0;   // pre-integrations of the feature tracker's own context (pipelined mode: another thread)
    double w_join = 0;              // pipelined mode: the feature tracker waiting for the previous frame's backend
    // host wall-clock seconds of whole pipeline stages (device waits included), see SC_* below
    double scope[16] = {0};
};
enum { SC_FT_TRACK = 0, SC_RANSAC_E, SC_RANSAC_R, SC_FT_DETECT, SC_MIRROR, SC_LOCALIZE, SC_MANAGE_KF, SC_TRACK_LANDMARK,
       SC_REFINE_WINDOW, SC_SLIDE_WINDOW, SC_REFINE_SUBWINDOW, SC_INITIALIZE, SC_RD_JUDGED, SC_RD_OUTLIERS, SC_CONST_COPIES, SC_COUNT };
struct WallTimer {   // adds the scope's duration to a StageTimes slot
    double &slot;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    explicit WallTimer(double &s) : slot(s) {}
    ~WallTimer() { slot += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
};

// ------------------------------------------------------------------------------------ worker threads
// One job at a time on a thread of its own (utility/worker.h:7-60 with XRSLAM_ENABLE_THREADING): the backend thread of the
// pipelined mode (System), and the thread that issues a marginalisation's launches (Pipeline::marg_launcher).  A frame is a fraction of
// a millisecond: both sides spin on an atomic before they fall back to the condition variable.
class JobThread {
  public:
    explicit JobThread(int device) : device_(device), th_([this] { loop(); }) {}
    ~JobThread() {
        // a job that is still running would overwrite QUIT with DONE when it finishes (and the loop would never leave): let it
        // finish first.  Callers wait for their jobs by convention; this makes the destructor safe without it.
        {
            const int s = state_.load(std::memory_order_acquire);
            if (s == POSTED) await([this] { return state_.load(std::memory_order_acquire) == DONE; }, 20000);
        }
        state_.store(QUIT, std::memory_order_release);
        {
            std::lock_guard<std::mutex> lk(m_);
        }
        cv_.notify_all();
        th_.join();
    }
    JobThread(const JobThread &) = delete;
    JobThread &operator=(const JobThread &) = delete;
    void post(std::function<void()> job) {   // the previous job must have been waited for
        job_ = std::move(job);
        error_ = nullptr;
        state_.store(POSTED, std::memory_order_release);
        {
            std::lock_guard<std::mutex> lk(m_);
        }
        cv_.notify_all();
    }
    void wait() {   // returns when the posted job has finished; rethrows what it threw
        await([this] { return state_.load(std::memory_order_acquire) == DONE; }, 20000);   // a backend job is a few ms at most
        state_.store(IDLE, std::memory_order_relaxed);
        if (error_) {
            std::exception_ptr e = error_;
            error_ = nullptr;
            std::rethrow_exception(e);
        }
    }

  private:
    enum { IDLE = 0, POSTED, DONE, QUIT };
    static void relax() {
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#else
        std::this_thread::yield();
#endif
    }
    // spin for at most `spin_us` (a sleeping thread costs tens of microseconds to wake), then sleep on the condition variable
    template <class Pred> void await(Pred pred, long spin_us) {
        const auto t0 = std::chrono::steady_clock::now();
        for (;;) {
            for (int spin = 0; spin < 256; ++spin) {
                if (pred()) return;
                relax();
            }
            if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(spin_us)) break;
        }
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, pred);
    }
    void loop() {
        if (device_ >= 0) xrhip_bind_device(device_);
        for (;;) {
            await([this] {
                const int s = state_.load(std::memory_order_acquire);
                return s == POSTED || s == QUIT;
            }, 2000);   // back-to-back frames keep the thread awake; a live 20 Hz stream lets it sleep between frames
            if (state_.load(std::memory_order_acquire) == QUIT) return;
            try {
                job_();
            } catch (...) {
                error_ = std::current_exception();
            }
            state_.store(DONE, std::memory_order_release);
            {
                std::lock_guard<std::mutex> lk(m_);
            }
            cv_.notify_all();
        }
    }
    int device_;
    std::function<void()> job_;
    std::exception_ptr error_;
    std::atomic<int> state_{IDLE};
    std::mutex m_;
    std::condition_variable cv_;
    std::thread th_;   // last: the members above exist before the thread starts
};

struct Pipeline {
    Config config;
    xrhip_klt *klt = nullptr;
    xrhip_ba *ba = nullptr;
    xrhip_ba *ba_marg = nullptr;   // marginalisation has a context (buffers, stream) of its own: it runs beside the next frame
    xrhip_ba *ba_aux = nullptr;    // speculative pre-integration batches (started a frame ahead), same reason
    xrhip_ba *ba_ft = nullptr;     // pipelined mode: the feature tracker's pre-integrations (its thread must not touch `ba`)
    xrhip_ba *ba_sub = nullptr;    // localize_newframe's problem when it is solved together with refine_subwindow's (`ba` holds that one)
    std::mutex pool_mutex;         // image buffers return from whichever thread drops the last reference
    // xrhip_ba_marginalize_begin is ~20 API calls (staging, a memset, sixteen launches, three copies: 0.14 ms) whose result nobody
    // reads for several frames: they are issued by a thread of their own, on the marginalisation's context, while the caller goes
    // on; resolve_marginalization waits for that thread before it waits for the device.  XRSLAM_AMD_SYNC_MARG_LAUNCH=1: inline.
    std::unique_ptr<JobThread> marg_launcher;
    bool marg_launch_pending = false;
    void marg_launch_wait() {
        if (!marg_launch_pending) return;
        marg_launch_pending = false;
        marg_launcher->wait();   // rethrows what the launch threw
    }
    IdSource ids;
    std::vector<xrhip_image *> image_pool;
    double noise36[36];
    StageTimes times;
    unsigned long ba_generation = 0;   // BaBuilder instances stamp frames / tracks with it
    bool undistort_on_device = false;  // frames arrive as the camera recorded them; the KLT context holds the inverse map
    // cv::undistort / ImageUndistorter on the device from now on (model: "cv_undistort", "radtan", "equidistant"), or off (nullptr)
    void set_device_undistort(const char *model) {
        if (!model || !*model) {
            hip_check(xrhip_klt_set_undistort_map(klt, nullptr), "xrhip_klt_set_undistort_map");
            undistort_on_device = false;
            return;
        }
        const double K4[4] = {config.K.fx, config.K.fy, config.K.cx, config.K.cy};
        const std::vector<uint32_t> map = build_undistort_map((int)config.cam_resolution[0], (int)config.cam_resolution[1], K4,
                                                              config.cam_distortion, 4, model);
        hip_check(xrhip_klt_set_undistort_map(klt, map.data()), "xrhip_klt_set_undistort_map");
        undistort_on_device = true;
    }
    SwtLogger swt_log;                 // XRSLAM_AMD_DUMP_SWT=<file>: decisions of the sliding-window tracker (ba_dump.hpp)
    SyncLogger sync_log;               // XRSLAM_AMD_DUMP_SYNC=<file>: IMU samples attached to every frame, poses answered (ba_dump.hpp)
    BaDumper ba_dump;                  // XRSLAM_AMD_DUMP_BA=<dir>: freeze the problems handed to xrhip_ba_solve (ba_dump.hpp)
    InitLogger init_log;               // XRSLAM_AMD_DUMP_INIT=<file>: decisions of the initialiser and of the RD-VIO filters with their inputs (ba_dump.hpp)
    OutLogger out_log;                 // XRSLAM_AMD_DUMP_OUT=<file>: what every frame produces -- key points, track ids, states, landmarks (ba_dump.hpp)

    explicit Pipeline(const Config &c) : config(c) {
        hip_check(xrhip_klt_create((int)c.cam_resolution[0], (int)c.cam_resolution[1],
                                   (int)c.feature_tracker_max_keypoint_detection, &klt),
                  "xrhip_klt_create");
        hip_check(xrhip_ba_create(32, 2048, 16384, &ba), "xrhip_ba_create");
        hip_check(xrhip_ba_create(32, 2048, 16384, &ba_marg), "xrhip_ba_create");
        hip_check(xrhip_ba_create(32, 2048, 16384, &ba_aux), "xrhip_ba_create");
        hip_check(xrhip_ba_create(8, 1024, 4096, &ba_sub), "xrhip_ba_create");
        for (int i = 0; i < 9; ++i) {
            noise36[i] = c.cov_g[i];
            noise36[9 + i] = c.cov_a[i];
            noise36[18 + i] = c.cov_bg[i];
            noise36[27 + i] = c.cov_ba[i];
        }
    }
    ~Pipeline() {
        try {
            marg_launch_wait();
        } catch (...) {
        }
        marg_launcher.reset();
        for (xrhip_image *im : image_pool) xrhip_image_destroy(im);
        if (ba) xrhip_ba_destroy(ba);
        if (ba_marg) xrhip_ba_destroy(ba_marg);
        if (ba_aux) xrhip_ba_destroy(ba_aux);
        if (ba_ft) xrhip_ba_destroy(ba_ft);
        if (ba_sub) xrhip_ba_destroy(ba_sub);
        if (klt) xrhip_klt_destroy(klt);
    }
    void ensure_ft_context() {
        if (ba_ft) return;
        hip_check(xrhip_ba_create(32, 2048, 16384, &ba_ft), "xrhip_ba_create");
        if (group) hip_check(xrhip_ba_join_group(ba_ft, group), "xrhip_ba_join_group");
    }
    // Instance group (include/xrslam_hip.h): the per-frame launches of this sequence -- frame upload, CLAHE / pyramid, LK, Harris,
    // pre-integrations, the single-launch solves -- are issued by the group together with the other members' (one launch per kernel
    // for all of them); window solves stay on `ba`'s own stream, the marginalisation keeps its own context.  Between frames only.
    xrhip_group *group = nullptr;
    void join_group(xrhip_group *g) {
        marg_launch_wait();
        // All or nothing: if a context refuses, the ones already moved return to the group they came from, so that the
        // instance is never half-joined (ensure_ft_context would otherwise create ba_ft outside the group its siblings are in).
        xrhip_group *const old = group;
        bool klt_moved = false;
        std::vector<xrhip_ba *> moved;
        try {
            hip_check(xrhip_klt_join_group(klt, g), "xrhip_klt_join_group");
            klt_moved = true;
            for (xrhip_ba *c : {ba, ba_aux, ba_ft, ba_sub})
                if (c) {
                    hip_check(xrhip_ba_join_group(c, g), "xrhip_ba_join_group");
                    moved.push_back(c);
                }
        } catch (...) {
            for (xrhip_ba *c : moved) xrhip_ba_join_group(c, old);   // best effort: the original error is the one reported
            if (klt_moved) xrhip_klt_join_group(klt, old);
            throw;
        }
        group = g;
    }
    xrhip_image *acquire_image() {
        {
            std::lock_guard<std::mutex> lk(pool_mutex);
            if (!image_pool.empty()) {
                xrhip_image *im = image_pool.back();
                image_pool.pop_back();
                return im;
            }
        }
        xrhip_image *im = nullptr;
        hip_check(xrhip_image_create(klt, &im), "xrhip_image_create");
        return im;
    }
    void recycle_image(xrhip_image *im) {
        std::lock_guard<std::mutex> lk(pool_mutex);
        xrhip_image_release(im);
        image_pool.push_back(im);
    }
    std::shared_ptr<HipImage> make_image(const uint8_t *gray, int stride, double t, bool device_ptr) {
        auto img = std::make_shared<HipImage>();
        img->owner = this;
        img->h = acquire_image();
        img->t = t;
        img->w = (int)config.cam_resolution[0];
        img->hgt = (int)config.cam_resolution[1];
        // a member of an instance group starts its frame together with the other members (timing only: xrslam_hip.h, frame gate)
        if (group) xrhip_klt_frame_gate(klt);
        if (undistort_on_device) hip_check(xrhip_image_upload_distorted(img->h, gray, stride, device_ptr ? 1 : 0), "xrhip_image_upload_distorted");
        else if (device_ptr) hip_check(xrhip_image_upload_device(img->h, gray, stride), "xrhip_image_upload_device");
        else hip_check(xrhip_image_upload(img->h, gray, stride), "xrhip_image_upload");
        // FeatureTracker::work's first step (feature_tracker.cpp:39) depends on nothing but the frame: its launches are queued here,
        // behind the frame's DMA, so the device builds the pyramid while the host is still on its way to the tracker (input
        // synchronisation, frame construction, the interval's pre-integration launch).  Same call, same arguments.
        {
            WallTimer wt_w_preprocess(times.w_preprocess);
            hip_check(xrhip_image_preprocess(img->h, config.feature_tracker_clahe_clip_limit, (int)config.feature_tracker_clahe_width,
                                             (int)config.feature_tracker_clahe_height),
                      "xrhip_image_preprocess");
            img->preprocessed = true;
        }
        return img;
    }
    // PreIntegrator::integrate (preintegrator.cpp:78-95) on the device
    // (`ctx`: the context whose stream runs it; the feature tracker passes its own in pipelined mode)
    double &preintegrate_slot(xrhip_ba *ctx) { return ctx && ctx == ba_ft ? times.w_preintegrate_ft : times.w_preintegrate; }
    bool integrate(PreInt &pre, double t, const V3 &bg, const V3 &ba_, bool jac, bool cov, xrhip_ba *ctx = nullptr) {
        if (pre.data.empty()) return false;
        std::vector<double> smp(pre.data.size() * 7);
        for (size_t i = 0; i < pre.data.size(); ++i) {
            const ImuData &d = pre.data[i];
            double *s = &smp[7 * i];
            s[0] = d.t;
            s[1] = d.w.x; s[2] = d.w.y; s[3] = d.w.z;
            s[4] = d.a.x; s[5] = d.a.y; s[6] = d.a.z;
        }
        const double b1[3] = {bg.x, bg.y, bg.z}, b2[3] = {ba_.x, ba_.y, ba_.z};
        WallTimer wt_w_preintegrate(preintegrate_slot(ctx));
        hip_check(xrhip_ba_preintegrate(ctx ? ctx : ba, smp.data(), (int)pre.data.size(), t, b1, b2, noise36, jac, cov, pre.rec),
                  "xrhip_ba_preintegrate");
        pre.valid = true;
        return true;
    }
    // asynchronous form for one interval: integrate_begin queues the kernel, integrate_end waits and stores the record
    bool integrate_begin(const std::vector<ImuData> &data, double t, const V3 &bg, const V3 &ba_, bool jac, bool cov,
                         xrhip_ba *ctx = nullptr) {
        if (data.empty()) return false;
        std::vector<double> smp(data.size() * 7);
        for (size_t i = 0; i < data.size(); ++i) {
            const ImuData &d = data[i];
            double *s = &smp[7 * i];
            s[0] = d.t;
            s[1] = d.w.x; s[2] = d.w.y; s[3] = d.w.z;
            s[4] = d.a.x; s[5] = d.a.y; s[6] = d.a.z;
        }
        const double b1[3] = {bg.x, bg.y, bg.z}, b2[3] = {ba_.x, ba_.y, ba_.z};
        const int begin = 0, count = (int)data.size();
        WallTimer wt_w_preintegrate(preintegrate_slot(ctx));
        hip_check(xrhip_ba_preintegrate_begin(ctx ? ctx : ba, smp.data(), &begin, &count, &t, b1, b2, 1, noise36, jac, cov),
                  "xrhip_ba_preintegrate_begin");
        return true;
    }
    // one interval whose integration starts from the biases the NEXT solve on `ba` gives frame `frame_index` of its problem:
    // staged now, launched by that solve behind its last kernel (xrhip_ba_preintegrate_after_solve), collected by integrate_end
    bool integrate_after_solve_begin(const std::vector<ImuData> &data, double t, int frame_index, bool jac, bool cov) {
        if (data.empty() || frame_index < 0) return false;
        std::vector<double> smp(data.size() * 7);
        for (size_t i = 0; i < data.size(); ++i) {
            const ImuData &d = data[i];
            double *s = &smp[7 * i];
            s[0] = d.t;
            s[1] = d.w.x; s[2] = d.w.y; s[3] = d.w.z;
            s[4] = d.a.x; s[5] = d.a.y; s[6] = d.a.z;
        }
        const int begin = 0, count = (int)data.size();
        hip_check(xrhip_ba_preintegrate_after_solve(ba, smp.data(), &begin, &count, &t, &frame_index, 1, noise36, jac, cov),
                  "xrhip_ba_preintegrate_after_solve");
        return true;
    }
    void cancel_integrations() noexcept {   // error unwind: no batch stays between begin and end on any context, no solve begun
        for (xrhip_ba *c : {ba, ba_aux, ba_ft})
            if (c) xrhip_ba_preintegrate_cancel(c);
        if (ba_sub) xrhip_ba_solve_abort(ba_sub);
    }
    // the delta (dt, dq, dp, dv) of job 0 of the batch in flight on `ba`, as soon as the kernel has it (the batch stays in flight)
    void integrate_early(PreInt &pre) {
        WallTimer wt_w_preintegrate(times.w_preintegrate);
        hip_check(xrhip_ba_preintegrate_early(ba, 0, pre.rec), "xrhip_ba_preintegrate_early");
        pre.valid = true;
    }
    void integrate_end(PreInt &pre, xrhip_ba *ctx = nullptr) {
        WallTimer wt_w_preintegrate(preintegrate_slot(ctx));
        hip_check(xrhip_ba_preintegrate_end(ctx ? ctx : ba, pre.rec), "xrhip_ba_preintegrate_end");
        pre.valid = true;
    }
    // several integrations in one launch (refine_window re-integrates every keyframe interval, refine_subwindow
    // every subframe interval); entries with no IMU data are skipped and reported as false
    struct IntegrateJob {
        PreInt *pre;
        double t;
        V3 bg, ba;
    };
    std::vector<char> integrate_batch(const std::vector<IntegrateJob> &jobs, bool jac, bool cov) {
        std::vector<char> ok = integrate_batch_begin(jobs, jac, cov);
        integrate_batch_end();
        return ok;
    }
    // asynchronous form: _begin queues the launch and returns which jobs had samples, _end waits and stores the records.
    // Between the two the caller assembles the problem that will read them (BaBuilder copies records at solve time).
    std::vector<char> integrate_batch_begin(const std::vector<IntegrateJob> &jobs, bool jac, bool cov) {
        return batch_begin(ba, pending_pre_, jobs, jac, cov);
    }
    void integrate_batch_end() {
        if (pending_pre_.empty()) return;
        std::vector<double> out = batch_collect(ba, pending_pre_.size());
        for (size_t i = 0; i < pending_pre_.size(); ++i) {
            std::memcpy(pending_pre_[i]->rec, &out[(size_t)XRHIP_IMU_DIM * i], sizeof(double) * XRHIP_IMU_DIM);
            pending_pre_[i]->valid = true;
        }
        pending_pre_.clear();
    }
    // the same on the auxiliary context, for batches whose results may or may not be wanted later: the records come back
    // as one array (jobs without samples are skipped), nothing is written into the PreInt objects
    size_t aux_batch_begin(const std::vector<IntegrateJob> &jobs, bool jac, bool cov) {
        std::vector<PreInt *> which;
        batch_begin(ba_aux, which, jobs, jac, cov);
        return which.size();
    }
    std::vector<double> aux_batch_collect(size_t n_jobs) { return n_jobs ? batch_collect(ba_aux, n_jobs) : std::vector<double>(); }

    std::vector<char> batch_begin(xrhip_ba *ctx, std::vector<PreInt *> &pending, const std::vector<IntegrateJob> &jobs, bool jac,
                                  bool cov) {
        std::vector<char> ok(jobs.size(), 0);
        std::vector<double> smp, t_end, bgs, bas;
        std::vector<int> begin, count;
        pending.clear();
        for (size_t k = 0; k < jobs.size(); ++k) {
            const PreInt &pre = *jobs[k].pre;
            if (pre.data.empty()) continue;
            ok[k] = 1;
            pending.push_back(jobs[k].pre);
            begin.push_back((int)(smp.size() / 7));
            count.push_back((int)pre.data.size());
            t_end.push_back(jobs[k].t);
            for (const ImuData &d : pre.data) {
                const double row[7] = {d.t, d.w.x, d.w.y, d.w.z, d.a.x, d.a.y, d.a.z};
                smp.insert(smp.end(), row, row + 7);
            }
            const double b1[3] = {jobs[k].bg.x, jobs[k].bg.y, jobs[k].bg.z}, b2[3] = {jobs[k].ba.x, jobs[k].ba.y, jobs[k].ba.z};
            bgs.insert(bgs.end(), b1, b1 + 3);
            bas.insert(bas.end(), b2, b2 + 3);
        }
        if (pending.empty()) return ok;
        WallTimer wt_w_preintegrate(preintegrate_slot(ctx));
        hip_check(xrhip_ba_preintegrate_begin(ctx, smp.data(), begin.data(), count.data(), t_end.data(), bgs.data(), bas.data(),
                                              (int)pending.size(), noise36, jac, cov),
                  "xrhip_ba_preintegrate_begin");
        return ok;
    }
    std::vector<double> batch_collect(xrhip_ba *ctx, size_t n_jobs) {
        std::vector<double> out((size_t)XRHIP_IMU_DIM * n_jobs);
        WallTimer wt_w_preintegrate(preintegrate_slot(ctx));
        hip_check(xrhip_ba_preintegrate_end(ctx, out.data()), "xrhip_ba_preintegrate_end");
        return out;
    }
    std::vector<PreInt *> pending_pre_;
};

inline HipImage::~HipImage() {
    if (h && owner) owner->recycle_image(h);
    h = nullptr;
}
inline void HipImage::release_image_buffer() {
    if (h && owner) owner->recycle_image(h);
    h = nullptr;
}

// PreIntegrator::predict (preintegrator.cpp:102-112)
inline void predict(const PreInt &pre, const Frame *o, Frame *n) {
    const V3 gravity{0, 0, -9.80665};
    const double dt = pre.dt();
    n->motion.bg = o->motion.bg;
    n->motion.ba = o->motion.ba;
    n->motion.v = o->motion.v + gravity * dt + o->pose.q * pre.dv();
    n->pose.p = o->pose.p + 0.5 * gravity * dt * dt + o->motion.v * dt + o->pose.q * pre.dp();
    n->pose.q = o->pose.q * pre.dq();
}

// ------------------------------------------------------------------------------------- Frame ops
inline void frame_detect_keypoints(Pipeline &P, Frame *f) {   // frame.cpp:55-72 + opencv_image.cpp:38-73
    WallTimer sc_t(P.times.scope[SC_FT_DETECT]);
    const Config &c = P.config;
    std::vector<double> existing(2 * f->keypoint_num());
    for (size_t i = 0; i < f->keypoint_num(); ++i) {
        V2 px = apply_k(f->bearings[i], f->K);
        existing[2 * i] = px.x;
        existing[2 * i + 1] = px.y;
    }
    const int maxp = (int)c.feature_tracker_max_keypoint_detection;
    std::vector<double> fresh(2 * (size_t)std::max(maxp, 1));
    int n_new = 0;
    WallTimer wt_w_detect(P.times.w_detect);
    hip_check(xrhip_image_detect(f->image->h, existing.data(), (int)f->keypoint_num(), maxp,
                                 c.feature_tracker_min_keypoint_distance, fresh.data(), &n_new),
              "xrhip_image_detect");
    for (int i = 0; i < n_new; ++i) f->append_keypoint(remove_k(V2{fresh[2 * i], fresh[2 * i + 1]}, f->K));
}

// deferred_tag: when given, the rotation-misalignment test that decides the new frame's FT_NO_TRANSLATION tag (an arc cosine per
// inlier and a sort: ~9 us) is not run here but handed back; the caller runs it -- with the window map's copy of the frame, if one has
// been made meanwhile -- before anything reads the tag (manage_keyframe does, after localize_newframe).
inline void frame_track_keypoints(Pipeline &P, Frame *cur, Frame *next, std::function<void(Frame *)> *deferred_tag = nullptr) {   // frame.cpp:74-174
    WallTimer sc_t(P.times.scope[SC_FT_TRACK]);
    const Config &c = P.config;
    const size_t n = cur->keypoint_num();
    std::vector<double> curr_px(2 * n), next_px(2 * n);
    for (size_t i = 0; i < n; ++i) {
        V2 px = apply_k(cur->bearings[i], cur->K);
        curr_px[2 * i] = px.x;
        curr_px[2 * i + 1] = px.y;
    }
    int has_guess = 0;
    if (c.feature_tracker_predict_keypoints) {
        Quat dq = (cur->camera.q_cs.conjugate() * cur->imu.q_cs * next->preintegration.dq() *
                   next->imu.q_cs.conjugate() * next->camera.q_cs)
                      .conjugate();
        for (size_t i = 0; i < n; ++i) {
            V2 px = apply_k(dq * cur->bearings[i], next->K);
            next_px[2 * i] = px.x;
            next_px[2 * i + 1] = px.y;
        }
        has_guess = 1;
    }
    std::vector<uint8_t> st8(std::max<size_t>(n, 1), 0);
    {
        WallTimer wt_w_track(P.times.w_track);
        hip_check(xrhip_image_track(cur->image->h, next->image->h, curr_px.data(), next_px.data(), has_guess,
                                    st8.data(), (int)n),
                  "xrhip_image_track");
    }
    xrhip::HostProfScope hp_post(4, "ft_track: after LK (all)");
    std::optional<xrhip::HostProfScope> hp_a(std::in_place, 5, "ft_track: bearings");
    std::vector<char> status(st8.begin(), st8.begin() + n), mask;
    std::vector<V2> cur_h, next_h;
    std::vector<V3> next_bearings;
    cur_h.reserve(n);
    next_h.reserve(n);
    next_bearings.reserve(n);
    for (size_t i = 0; i < n; ++i) {
        const V3 &b = cur->bearings[i];
        cur_h.push_back({b.x / b.z, b.y / b.z});
        V3 nb = remove_k(V2{next_px[2 * i], next_px[2 * i + 1]}, next->K);
        next_h.push_back({nb.x / nb.z, nb.y / nb.z});
        next_bearings.push_back(nb);
    }
    hp_a.reset();
    {
        WallTimer sc_e(P.times.scope[SC_RANSAC_E]);
        find_essential_matrix(cur_h, next_h, mask, 1.0);
    }
    for (size_t i = 0; i < status.size() && i < mask.size(); ++i)
        if (!mask[i]) status[i] = 0;
    M3 R;
    {
        WallTimer sc_r(P.times.scope[SC_RANSAC_R]);
        R = find_rotation_matrix(cur->bearings, next_bearings, mask, (M_PI / 180.0) * c.rotation_ransac_threshold);
    }
    xrhip::HostProfScope hp_b(6, "ft_track: angles+poisson+append");
    std::optional<xrhip::HostProfScope> hp_c(std::in_place, 13, "ft_track: angles");
    {
        // (R * cur->bearings[i] and next_bearings[i] of the inliers, in keypoint order: everything the test reads, by value)
        std::vector<std::pair<V3, V3>> pairs;
        pairs.reserve(mask.size());
        for (size_t i = 0; i < mask.size(); ++i)
            if (mask[i]) pairs.emplace_back(R * cur->bearings[i], next_bearings[i]);
        const double threshold = c.rotation_misalignment_threshold;
        auto decide = [pairs = std::move(pairs), threshold, next](Frame *copy) {
            std::vector<double> angles;
            angles.reserve(pairs.size());
            for (const auto &pr : pairs) angles.push_back(std::acos(dot(pr.first, pr.second)) * 180 / M_PI);
            std::sort(angles.begin(), angles.end());
            const double misalignment = angles.size() > 0 ? angles[angles.size() * 7 / 10] : 0;
            if (misalignment < threshold) {
                next->tag(FT_NO_TRANSLATION) = true;
                if (copy) copy->tag(FT_NO_TRANSLATION) = true;
            }
        };
        if (deferred_tag) *deferred_tag = std::move(decide);
        else decide(nullptr);
    }

    hp_c.emplace(14, "ft_track: by_length+poisson");
    std::vector<std::pair<size_t, size_t>> by_length;
    by_length.reserve(n);
    for (size_t i = 0; i < n; ++i) {
        if (status[i] == 0) continue;
        Track *t = cur->get_track(i);
        if (t == nullptr) continue;
        by_length.emplace_back(i, t->keypoint_num());
    }
    std::sort(by_length.begin(), by_length.end(), [](const auto &a, const auto &b) { return a.second > b.second; });
    xrhip::PoissonDisk2 filter(c.feature_tracker_min_keypoint_distance, (int)c.cam_resolution[0], (int)c.cam_resolution[1]);
    for (auto &[ki, len] : by_length) {
        (void)len;
        Track *t = cur->get_track(ki);
        if (filter.permit(next_px[2 * ki], next_px[2 * ki + 1]) && (!t || !t->tag(TT_TRASH))) {
            filter.preset(next_px[2 * ki], next_px[2 * ki + 1]);
        } else {
            status[ki] = 0;
        }
    }
    hp_c.reset();
    xrhip::HostProfScope hp_d(15, "ft_track: append");
    for (size_t i = 0; i < n; ++i) {
        if (status[i]) {
            size_t nk = next->keypoint_num();
            next->append_keypoint(next_bearings[i]);
            cur->get_track(i, nullptr)->add_keypoint(next, nk);
        }
    }
}

// Fetches the result of the marginalisation queued by marginalize_frame (it runs on its own context / stream while the
// next frames are tracked and localised); called before the prior is read: by the next window solve, by the next
// marginalisation, at shutdown.
inline void resolve_marginalization(Pipeline &P, MargPrior *prior) {
    if (!prior || !prior->pending) return;
    P.marg_launch_wait();
    const size_t n = prior->frames.size(), R = 15 * n;
    std::vector<double> si(R * R), iv(R), lin(16 * n);
    {
        WallTimer wt_w_marginalize(P.times.w_marginalize);
        hip_check(xrhip_ba_marginalize_end(P.ba_marg, si.data(), iv.data(), lin.data()), "xrhip_ba_marginalize_end");
    }
    prior->sqrt_info.swap(si);
    prior->infovec.swap(iv);
    prior->lin.swap(lin);
    prior->pending = false;
}

// ------------------------------------------------------------------------------- BA problem assembly
// Collects frames / tracks the way Solver::add_frame_states / add_track_states / add_factor do and
// runs xrhip_ba_solve; states are written back in place.
class BaBuilder {
  public:
    // Frame / track -> problem index: stamped on the object with this builder's generation number instead of a
    // hash map (a problem touches ~150 tracks; two problems per frame)
    explicit BaBuilder(Pipeline &P) : P_(P), gen_(++P.ba_generation) {
        obs_tgt_.reserve(2048);
        obs_ref_.reserve(2048);
        obs_lm_.reserve(2048);
        obs_zt_.reserve(3 * 2048);
        obs_zr_.reserve(3 * 2048);
        tracks_.reserve(512);
        lfix_.reserve(512);
    }
    int frame_index(Frame *f, bool as_parameter, bool with_motion = true) {
        if (f->ba_gen != gen_) {
            f->ba_gen = gen_;
            f->ba_index = (int)frames_.size();
            frames_.push_back(f);
            fix_.push_back(XRHIP_FIX_POSE | XRHIP_FIX_MOTION);
        }
        if (as_parameter) {
            uint8_t fx = 0;
            if (f->tag(FT_FIX_POSE)) fx |= XRHIP_FIX_POSE;
            if (!with_motion || f->tag(FT_FIX_MOTION)) fx |= XRHIP_FIX_MOTION;
            fix_[f->ba_index] = fx;
        }
        return f->ba_index;
    }
    void add_frame_states(Frame *f, bool with_motion = true) { frame_index(f, true, with_motion); }
    int landmark_index(Track *t, bool as_parameter) {
        if (t->ba_gen != gen_) {
            t->ba_gen = gen_;
            t->ba_index = (int)tracks_.size();
            tracks_.push_back(t);
            lfix_.push_back(1);
        }
        if (as_parameter) lfix_[t->ba_index] = 0;
        return t->ba_index;
    }
    void add_track_states(Track *t) { landmark_index(t, true); }
    // ReprojectionErrorFactor (parameters: tgt pose, ref pose, inverse depth)
    void add_reprojection_error(Frame *frame, size_t ki) {
        Track *t = frame->get_track(ki);
        auto [ref, kr] = t->first_keypoint();
        push_obs(frame_index(frame, false), frame_index(ref, false), landmark_index(t, false), frame->get_keypoint(ki),
                 ref->get_keypoint(kr));
    }
    // ReprojectionPriorFactor: ref pose and depth are constants whatever their flags
    void add_reprojection_prior(Frame *frame, size_t ki) {
        Track *t = frame->get_track(ki);
        auto [ref, kr] = t->first_keypoint();
        // constants are represented by dedicated constant copies so that a frame/track that is also a
        // parameter elsewhere in the problem is still read as a constant here
        const_obs_.push_back(obs_tgt_.size());
        push_obs(frame_index(frame, false), const_frame(ref), const_landmark(t), frame->get_keypoint(ki),
                 ref->get_keypoint(kr));
    }
    void add_rotation_prior(Frame *frame, size_t ki) {
        Track *t = frame->get_track(ki);
        auto [ref, kr] = t->first_keypoint();
        const_rot_.push_back(rot_tgt_.size());
        rot_tgt_.push_back(frame_index(frame, false));
        rot_ref_.push_back(const_frame(ref));
        push3(rot_zt_, frame->get_keypoint(ki));
        push3(rot_zr_, ref->get_keypoint(kr));
    }
    void add_preintegration_error(Frame *fi, Frame *fj, const PreInt &pre) {
        imu_i_.push_back(frame_index(fi, false));
        imu_j_.push_back(frame_index(fj, false));
        imu_pre_.push_back(&pre);
    }
    void add_preintegration_prior(Frame *fi, Frame *fj, const PreInt &pre) {
        const_imu_.push_back(imu_i_.size());
        imu_i_.push_back(const_frame(fi));
        imu_j_.push_back(frame_index(fj, false));
        imu_pre_.push_back(&pre);
    }
    void add_marginalization(MargPrior *m) { prior_ = m; }
    // An IMU interval to integrate from the biases this solve is about to give `f` (a frame of the problem): queued behind
    // the solve on the device.  chained() tells afterwards whether it was (Pipeline::integrate_end collects it).
    void chain_integration(Frame *f, std::vector<ImuData> samples, double t) {
        chain_f_ = f;
        chain_samples_ = std::move(samples);
        chain_t_ = t;
    }
    bool chained() const { return chain_queued_; }

    size_t factor_num() const { return obs_tgt_.size() + rot_tgt_.size() + imu_i_.size() + (prior_ ? 1 : 0); }
    // Parameter blocks that no factor touches are not part of the program Ceres minimises (it drops them while
    // building the reduced program): a free pose / motion nobody references is passed as a constant.
    void drop_unreferenced_blocks() {
        const size_t F = frames_.size();
        std::vector<char> pose_used(F, 0), motion_used(F, 0);
        for (size_t o = 0; o < obs_tgt_.size(); ++o) pose_used[obs_tgt_[o]] = pose_used[obs_ref_[o]] = 1;
        for (size_t o = 0; o < rot_tgt_.size(); ++o) pose_used[rot_tgt_[o]] = pose_used[rot_ref_[o]] = 1;
        for (size_t o = 0; o < imu_i_.size(); ++o)
            pose_used[imu_i_[o]] = pose_used[imu_j_[o]] = motion_used[imu_i_[o]] = motion_used[imu_j_[o]] = 1;
        if (prior_)
            for (Frame *f : prior_->frames)
                if (f->ba_gen == gen_) pose_used[f->ba_index] = motion_used[f->ba_index] = 1;
        for (size_t f = 0; f < F; ++f) {
            if (!pose_used[f]) fix_[f] |= XRHIP_FIX_POSE;
            if (!motion_used[f]) fix_[f] |= XRHIP_FIX_MOTION;
        }
    }
    // The *Prior factors read their reference frame / landmark as constants (reprojection_factor.h:92-123,
    // rotation_factor.h:23-59, preintegration_factor.h:161-199: the reference object's current value, no Jacobian).  A
    // constant use shares the object's ordinary entry as long as that entry is constant in this problem; when the same
    // frame or track is ALSO a free parameter here (a rotation prior whose reference is a sibling subframe), the prior gets
    // a constant copy of its own, taken at the states the solve starts from.
    void split_aliased_constants() {
        std::unordered_map<int, int> fcopy, lcopy;
        auto frame_copy = [&](int idx) {
            if (fix_[idx] == (XRHIP_FIX_POSE | XRHIP_FIX_MOTION)) return idx;
            auto it = fcopy.find(idx);
            if (it != fcopy.end()) return it->second;
            const int ni = (int)frames_.size();
            frames_.push_back(frames_[idx]);
            fix_.push_back(XRHIP_FIX_POSE | XRHIP_FIX_MOTION);
            fcopy[idx] = ni;
            P_.times.scope[SC_CONST_COPIES] += 1.0;   // a COUNT (XRSLAMAmdTimes.wall_scope[14])
            return ni;
        };
        auto landmark_copy = [&](int idx) {
            if (lfix_[idx]) return idx;
            auto it = lcopy.find(idx);
            if (it != lcopy.end()) return it->second;
            const int ni = (int)tracks_.size();
            tracks_.push_back(tracks_[idx]);
            lfix_.push_back(1);
            lcopy[idx] = ni;
            P_.times.scope[SC_CONST_COPIES] += 1.0;
            return ni;
        };
        for (size_t o : const_obs_) {
            obs_ref_[o] = frame_copy(obs_ref_[o]);
            obs_lm_[o] = landmark_copy(obs_lm_[o]);
        }
        for (size_t o : const_rot_) rot_ref_[o] = frame_copy(rot_ref_[o]);
        for (size_t o : const_imu_) imu_i_[o] = frame_copy(imu_i_[o]);
    }

    // Host work of the caller that does not depend on this solve, run once beside it (xrhip_ba_solve_overlapped); exceptions it
    // throws are re-thrown after the solve has returned (the context is never left with a launch nobody waited for).
    struct Overlap {
        std::function<void()> early;   // the part manage_keyframe must see done (the new frame's FT_NO_TRANSLATION tag): run first
        std::function<void()> fn;
        bool done = false, early_done = false;
        std::exception_ptr error;
        void run_early() {
            if (early_done) return;
            early_done = true;
            try {
                if (early) early();
            } catch (...) {
                error = std::current_exception();
            }
        }
        void run() {
            run_early();
            if (done) return;
            done = true;
            if (error) return;
            try {
                if (fn) fn();
            } catch (...) {
                error = std::current_exception();
            }
        }
        static void trampoline(void *self) { static_cast<Overlap *>(self)->run(); }
    };
    // The problem as xrhip_ba_solve takes it, built from what was added (the arrays it points into live in this builder); states and
    // pre-integration records are read NOW.
    void prepare() {
        const Config &c = P_.config;
        split_aliased_constants();
        drop_unreferenced_blocks();
        const int F = (int)frames_.size(), L = (int)tracks_.size();
        std::vector<double> &state = state_, &depth = depth_;
        state.assign(16 * (size_t)F, 0.0);
        depth.assign(std::max(L, 1), 0.0);
        for (int f = 0; f < F; ++f) pack_state(frames_[f], &state[16 * (size_t)f]);
        for (int l = 0; l < L; ++l) depth[l] = tracks_[l]->landmark.inv_depth;
        xrhip_ba_problem &pb = pb_;
        std::memset(&pb, 0, sizeof(pb));
        pb.n_frames = F;
        pb.frame_state = state.data();
        pb.frame_fix = fix_.data();
        Frame *any = frames_[0];
        const Quat &qc = any->camera.q_cs, &qi = any->imu.q_cs;
        const double cq[4] = {qc.x, qc.y, qc.z, qc.w}, iq[4] = {qi.x, qi.y, qi.z, qi.w};
        std::memcpy(pb.cam_q_bc, cq, sizeof(cq));
        std::memcpy(pb.imu_q_bi, iq, sizeof(iq));
        for (int k = 0; k < 3; ++k) {
            pb.cam_p_bc[k] = any->camera.p_cs[k];
            pb.imu_p_bi[k] = any->imu.p_cs[k];
        }
        pb.sqrt_inv_cov[0] = any->sqrt_inv_cov[0];
        pb.sqrt_inv_cov[1] = any->sqrt_inv_cov[1];
        pb.n_landmarks = L;
        pb.inv_depth = depth.data();
        pb.landmark_fix = lfix_.data();
        pb.n_obs = (int)obs_tgt_.size();
        pb.obs_tgt = obs_tgt_.data();
        pb.obs_ref = obs_ref_.data();
        pb.obs_lm = obs_lm_.data();
        pb.obs_z_tgt = obs_zt_.data();
        pb.obs_z_ref = obs_zr_.data();
        pb.n_rot = (int)rot_tgt_.size();
        pb.rot_tgt = rot_tgt_.data();
        pb.rot_ref = rot_ref_.data();
        pb.rot_z_tgt = rot_zt_.data();
        pb.rot_z_ref = rot_zr_.data();
        pb.n_imu = (int)imu_i_.size();
        pb.imu_i = imu_i_.data();
        pb.imu_j = imu_j_.data();
        // the pre-integration records are read here, not when the factor was added: a batch integration started before
        // the assembly (Pipeline::integrate_batch_begin) only has to be finished by now
        imu_data_.resize((size_t)XRHIP_IMU_DIM * imu_pre_.size());
        for (size_t k = 0; k < imu_pre_.size(); ++k)
            std::memcpy(&imu_data_[(size_t)XRHIP_IMU_DIM * k], imu_pre_[k]->rec, sizeof(double) * XRHIP_IMU_DIM);
        pb.imu_data = imu_data_.data();
        std::vector<int> &pframes = pframes_;
        pframes.clear();
        if (prior_) {
            resolve_marginalization(P_, prior_);
            for (Frame *f : prior_->frames) pframes.push_back(frame_index(f, false));
            // frame_index may have appended constant frames: refresh the arrays that depend on F
            if ((int)frames_.size() != F) throw std::logic_error("prior frame is not part of the problem");
            pb.prior_n = (int)pframes.size();
            pb.prior_frames = pframes.data();
            pb.prior_sqrt_info = prior_->sqrt_info.data();
            pb.prior_infovec = prior_->infovec.data();
            pb.prior_lin = prior_->lin.data();
        }
        pb.max_iterations = (int)c.solver_iteration_limit;
        if (P_.ba_dump.enabled()) P_.ba_dump.maybe_dump(pb, P_.times.frames);
        prepared_ = true;
    }
    // the solved states back into the frames / tracks they came from, the counters of the run
    bool finish(const xrhip_ba_summary &sm, double *elapsed_device_ms = nullptr) {
        const int F = (int)frames_.size(), L = (int)tracks_.size();
        for (int f = 0; f < F; ++f)
            if (fix_[f] != (XRHIP_FIX_POSE | XRHIP_FIX_MOTION)) unpack_state(&state_[16 * (size_t)f], frames_[f], fix_[f]);
        for (int l = 0; l < L; ++l)
            if (!lfix_[l]) tracks_[l]->landmark.inv_depth = depth_[l];
        P_.times.solves++;
        P_.times.solve_iterations += sm.iterations;
        P_.times.ba_device_ms += sm.ms_solve;
        if (elapsed_device_ms) *elapsed_device_ms = sm.ms_solve;
        return sm.usable != 0;
    }
    bool solve(double *elapsed_device_ms = nullptr, Overlap *overlap = nullptr) {
        xrhip::HostProfScope hp_s(10, "BaBuilder::solve (all)");
        if (!prepared_) prepare();
        xrhip_ba_summary sm;
        WallTimer wt_w_solve(P_.times.w_solve);
        if (chain_f_ && chain_f_->ba_gen == gen_)
            chain_queued_ = P_.integrate_after_solve_begin(chain_samples_, chain_t_, chain_f_->ba_index, true, true);
        if (overlap) {
            const int rc = xrhip_ba_solve_overlapped(P_.ba, &pb_, &sm, &Overlap::trampoline, overlap);
            overlap->run();
            if (overlap->error) std::rethrow_exception(overlap->error);
            hip_check(rc, "xrhip_ba_solve_overlapped");
        } else {
            hip_check(xrhip_ba_solve(P_.ba, &pb_, &sm), "xrhip_ba_solve");
        }
        return finish(sm, elapsed_device_ms);
    }
    // Two problems in one go (xrhip_ba_solve_chained): `second` contains frame `link`, which `first` optimises; it starts from the
    // first solve's result for that frame (on the device: one submission, one wait for the pair).  Both builders must be prepared
    // in order -- first, then whatever decides the second's structure, then second; `link_in_first` is link's index in the first
    // problem (its stamp has been overwritten by the second builder since).  The first problem is staged on Pipeline::ba_sub, the
    // second on Pipeline::ba (where an integration queued behind it -- chain_integration -- is collected from).
    // xrhip_ba_solve_begin on `ctx`: true = the solve is running on the device (end() / solve_linked() collect it), false = not a
    // single-launch problem, nothing was queued (solve() it)
    bool begin(xrhip_ba *ctx) {
        if (!prepared_) prepare();
        WallTimer wt_w_solve(P_.times.w_solve);
        const int rc = xrhip_ba_solve_begin(ctx, &pb_);
        if (rc < 0) hip_check(rc, "xrhip_ba_solve_begin");
        return rc == 1;
    }
    bool end(xrhip_ba *ctx, Overlap *overlap = nullptr) {
        xrhip_ba_summary sm;
        WallTimer wt_w_solve(P_.times.w_solve);
        if (overlap) {
            overlap->run();
            if (overlap->error) {
                xrhip_ba_solve_abort(ctx);
                std::rethrow_exception(overlap->error);
            }
        }
        hip_check(xrhip_ba_solve_end(ctx, &sm), "xrhip_ba_solve_end");
        return finish(sm);
    }
    // `first` has been begun on Pipeline::ba_sub; `second` -- prepared while the device worked on it -- contains frame `link`, which
    // `first` optimises: it is solved on Pipeline::ba starting from the first solve's result for that frame, handed over on the
    // device (xrhip_ba_solve_linked); then both are collected.  link_in_first: link's index in the first problem (its stamp has been
    // overwritten by the second builder since).  An integration queued behind the second solve (chain_integration) is collected
    // from Pipeline::ba as after solve().
    static bool solve_linked(BaBuilder &first, int link_in_first, BaBuilder &second, Frame *link, Overlap *overlap) {
        xrhip::HostProfScope hp_s(10, "BaBuilder::solve (all)");
        Pipeline &P = first.P_;
        if (!first.prepared_ || !second.prepared_ || link->ba_gen != second.gen_) throw std::logic_error("solve_linked: builders are not prepared");
        xrhip_ba_summary sm1, sm2;
        WallTimer wt_w_solve(P.times.w_solve);
        if (second.chain_f_ && second.chain_f_->ba_gen == second.gen_)
            second.chain_queued_ = P.integrate_after_solve_begin(second.chain_samples_, second.chain_t_, second.chain_f_->ba_index, true, true);
        const int rc = xrhip_ba_solve_linked(P.ba, &second.pb_, &sm2, link->ba_index, P.ba_sub, link_in_first,
                                             overlap ? &Overlap::trampoline : nullptr, overlap);
        if (overlap) overlap->run();
        if (rc || (overlap && overlap->error)) {
            xrhip_ba_solve_abort(P.ba_sub);
            if (overlap && overlap->error) std::rethrow_exception(overlap->error);
            hip_check(rc, "xrhip_ba_solve_linked");
        }
        hip_check(xrhip_ba_solve_end(P.ba_sub, &sm1), "xrhip_ba_solve_end");
        first.finish(sm1);
        return second.finish(sm2);
    }
    // the starting state of `f` again, from the frame (a problem prepared before another solve moved the frame)
    void repack(Frame *f) {
        if (prepared_ && f->ba_gen == gen_) pack_state(f, &state_[16 * (size_t)f->ba_index]);
    }

    static void pack_state(const Frame *f, double *s) {
        s[0] = f->pose.q.x; s[1] = f->pose.q.y; s[2] = f->pose.q.z; s[3] = f->pose.q.w;
        for (int k = 0; k < 3; ++k) {
            s[4 + k] = f->pose.p[k];
            s[7 + k] = f->motion.v[k];
            s[10 + k] = f->motion.bg[k];
            s[13 + k] = f->motion.ba[k];
        }
    }
    static void unpack_state(const double *s, Frame *f, uint8_t fix) {
        if (!(fix & XRHIP_FIX_POSE)) {
            f->pose.q = Quat{s[0], s[1], s[2], s[3]};
            f->pose.p = {s[4], s[5], s[6]};
        }
        if (!(fix & XRHIP_FIX_MOTION)) {
            f->motion.v = {s[7], s[8], s[9]};
            f->motion.bg = {s[10], s[11], s[12]};
            f->motion.ba = {s[13], s[14], s[15]};
        }
    }

  private:
    // A constant use starts out on the object's ordinary entry (never promoted to a parameter by this call);
    // split_aliased_constants() gives it a constant copy if that entry turns out to be free in this problem.
    int const_frame(Frame *f) { return frame_index(f, false); }
    int const_landmark(Track *t) { return landmark_index(t, false); }
    void push3(std::vector<double> &v, const V3 &a) {
        v.push_back(a.x);
        v.push_back(a.y);
        v.push_back(a.z);
    }
    void push_obs(int ft, int fr, int l, const V3 &zt, const V3 &zr) {
        obs_tgt_.push_back(ft);
        obs_ref_.push_back(fr);
        obs_lm_.push_back(l);
        push3(obs_zt_, zt);
        push3(obs_zr_, zr);
    }
    Pipeline &P_;
    unsigned long gen_;
    std::vector<Frame *> frames_;
    std::vector<Track *> tracks_;
    std::vector<uint8_t> fix_, lfix_;
    std::vector<int> obs_tgt_, obs_ref_, obs_lm_, rot_tgt_, rot_ref_, imu_i_, imu_j_;
    std::vector<double> obs_zt_, obs_zr_, rot_zt_, rot_zr_, imu_data_;
    std::vector<const PreInt *> imu_pre_;
    std::vector<size_t> const_obs_, const_rot_, const_imu_;   // factors whose reference side is a constant use
    MargPrior *prior_ = nullptr;
    std::vector<double> state_, depth_;
    std::vector<int> pframes_;
    xrhip_ba_problem pb_;
    bool prepared_ = false;
    Frame *chain_f_ = nullptr;
    std::vector<ImuData> chain_samples_;
    double chain_t_ = 0;
    bool chain_queued_ = false;
};

// MarginalizationFactor ctor (estimation/marginalization_factor.h:18-33): gauge prior on the first pose
inline std::unique_ptr<MargPrior> create_marginalization_factor(Map *map) {
    auto m = std::make_unique<MargPrior>();
    const size_t n = map->frame_num() - 1;
    m->frames.resize(n);
    m->lin.assign(16 * n, 0.0);
    for (size_t i = 0; i < n; ++i) {
        m->frames[i] = map->get_frame(i);
        BaBuilder::pack_state(map->get_frame(i), &m->lin[16 * i]);
    }
    m->infovec.assign(15 * n, 0.0);
    m->sqrt_info.assign(15 * n * 15 * n, 0.0);
    for (int k = 0; k < 6; ++k) m->sqrt_info[(size_t)k * 15 * n + k] = 1.0e15;
    return m;
}

// CeresMarginalizationFactor::marginalize (ceres/marginalization_factor.h:74-475) via xrhip_ba_marginalize
// The arrays an xrhip_marg_problem points into: kept alive until the call that stages them has been made
struct MargJob {
    xrhip_marg_problem mp;
    std::vector<double> state, imu_data, zt, zr, depth;
    std::vector<int> pframes, imu_i, imu_j, ot, orf, ol;
};

inline void marginalize_frame(Pipeline &P, Map *map, size_t index) {   // Map::marginalize_frame (map.cpp:51-63)
    MargPrior *prior = map->marginalization_factor.get();
    if (!prior) throw std::logic_error("marginalization_factor is not initialized yet");
    resolve_marginalization(P, prior);   // the previous result is this one's input
    const int K = (int)map->frame_num();
    // frame -> index in this problem: BaBuilder's generation stamp on the frame (a hash map here was looked up once per observation,
    // ~2000 times per marginalisation)
    const unsigned long gen = ++P.ba_generation;
    auto fidx_of = [gen](const Frame *f) { return f->ba_gen == gen ? f->ba_index : -1; };
    auto job = std::make_shared<MargJob>();
    std::vector<double> &state = job->state;
    state.resize(16 * (size_t)K);
    for (int i = 0; i < K; ++i) {
        map->get_frame(i)->ba_gen = gen;
        map->get_frame(i)->ba_index = i;
        BaBuilder::pack_state(map->get_frame(i), &state[16 * (size_t)i]);
    }
    xrhip_marg_problem &mp = job->mp;
    std::memset(&mp, 0, sizeof(mp));
    mp.n_frames = K;
    mp.victim = (int)index;
    mp.frame_state = state.data();
    Frame *any = map->get_frame(0);
    const Quat &qc = any->camera.q_cs, &qi = any->imu.q_cs;
    const double cq[4] = {qc.x, qc.y, qc.z, qc.w}, iq[4] = {qi.x, qi.y, qi.z, qi.w};
    std::memcpy(mp.cam_q_bc, cq, sizeof(cq));
    std::memcpy(mp.imu_q_bi, iq, sizeof(iq));
    for (int k = 0; k < 3; ++k) {
        mp.cam_p_bc[k] = any->camera.p_cs[k];
        mp.imu_p_bi[k] = any->imu.p_cs[k];
    }
    mp.sqrt_inv_cov[0] = any->sqrt_inv_cov[0];
    mp.sqrt_inv_cov[1] = any->sqrt_inv_cov[1];
    std::vector<int> &pframes = job->pframes;
    for (Frame *f : prior->frames) {
        if (fidx_of(f) < 0) throw std::out_of_range("marginalize_frame: prior frame is not in the window");
        pframes.push_back(fidx_of(f));
    }
    mp.prior_n = (int)pframes.size();
    mp.prior_frames = pframes.data();
    // (the prior's arrays are read in place: nothing writes them before resolve_marginalization, which waits for the launch)
    mp.prior_sqrt_info = prior->sqrt_info.data();
    mp.prior_infovec = prior->infovec.data();
    mp.prior_lin = prior->lin.data();
    std::vector<int> &imu_i = job->imu_i, &imu_j = job->imu_j;
    std::vector<double> &imu_data = job->imu_data;
    for (size_t j = index; j <= index + 1; ++j) {
        if (j == 0 || j >= (size_t)K) continue;
        Frame *fj = map->get_frame(j);
        imu_i.push_back((int)j - 1);
        imu_j.push_back((int)j);
        imu_data.insert(imu_data.end(), fj->keyframe_preintegration.rec, fj->keyframe_preintegration.rec + XRHIP_IMU_DIM);
    }
    mp.n_imu = (int)imu_i.size();
    mp.imu_i = imu_i.data();
    mp.imu_j = imu_j.data();
    mp.imu_data = imu_data.data();
    std::vector<int> &ot = job->ot, &orf = job->orf, &ol = job->ol;
    std::vector<double> &zt = job->zt, &zr = job->zr, &depth = job->depth;
    Frame *victim = map->get_frame(index);
    ot.reserve(2048); orf.reserve(2048); ol.reserve(2048); zt.reserve(3 * 2048); zr.reserve(3 * 2048); depth.reserve(512);
    for (size_t j = 0; j < victim->keypoint_num(); ++j) {
        Track *track = victim->get_track(j);
        if (!track || !track->tag(TT_VALID)) continue;
        Frame *ref = track->first_frame();
        if (!ref->tag(FT_KEYFRAME)) continue;
        const int fr = fidx_of(ref);   // the reference indexes frame_indices.at(frame_ref) unconditionally
        if (fr < 0) throw std::out_of_range("marginalize_frame: reference frame is not in the window");
        const size_t kr = track->keypoint_refs.at(ref);
        int l = -1;
        for (const auto &[tgt, ki] : track->keypoint_refs) {
            if (tgt == ref) continue;
            const int ft_i = fidx_of(tgt);
            if (ft_i < 0) continue;
            if (l < 0) {
                l = (int)depth.size();
                depth.push_back(track->landmark.inv_depth);
            }
            ot.push_back(ft_i);
            orf.push_back(fr);
            ol.push_back(l);
            const V3 &a = tgt->get_keypoint(ki), &b = ref->get_keypoint(kr);
            zt.insert(zt.end(), {a.x, a.y, a.z});
            zr.insert(zr.end(), {b.x, b.y, b.z});
        }
    }
    mp.n_landmarks = (int)depth.size();
    mp.inv_depth = depth.data();
    mp.n_obs = (int)ot.size();
    mp.obs_tgt = ot.data();
    mp.obs_ref = orf.data();
    mp.obs_lm = ol.data();
    mp.obs_z_tgt = zt.data();
    mp.obs_z_ref = zr.data();
    {   // queued on the marginalisation's own context; the new sqrt_info / infovec / lin are fetched on first use
        WallTimer wt_w_marginalize(P.times.w_marginalize);
        static const bool sync_launch = std::getenv("XRSLAM_AMD_SYNC_MARG_LAUNCH") != nullptr;   // development switch
        if (sync_launch) {
            hip_check(xrhip_ba_marginalize_begin(P.ba_marg, &mp), "xrhip_ba_marginalize_begin");
        } else {
            if (!P.marg_launcher) {
                int dev = -1;
                if (xrhip_get_device(&dev) != 0) dev = -1;
                P.marg_launcher = std::make_unique<JobThread>(dev);
            }
            xrhip_ba *ctx = P.ba_marg;
            P.marg_launcher->post([ctx, job] { hip_check(xrhip_ba_marginalize_begin(ctx, &job->mp), "xrhip_ba_marginalize_begin"); });
            P.marg_launch_pending = true;
        }
    }
    prior->pending = true;
    prior->frames.clear();
    for (int i = 0; i < K; ++i)
        if ((size_t)i != index) prior->frames.push_back(map->get_frame(i));
    // Map::marginalize_frame tail
    for (size_t i = 0; i < victim->keypoint_num(); ++i)
        if (Track *t = victim->get_track(i)) t->remove_keypoint(victim);
    map->frames.erase(map->frames.begin() + index);
    P.times.marginalizations++;
}

using LatestState = std::tuple<double, PoseState, MotionState>;

// ------------------------------------------------------------------------- SlidingWindowTracker
class SlidingWindowTracker {
  public:
    SlidingWindowTracker(Pipeline &P, std::unique_ptr<Map> keyframe_map) : P_(P), map(std::move(keyframe_map)) {
        for (size_t j = 1; j < map->frame_num(); ++j) {
            Frame *fi = map->get_frame(j - 1), *fj = map->get_frame(j);
            P_.integrate(fj->preintegration, fj->image->t, fi->motion.bg, fi->motion.ba, true, true);
        }
    }

    // Early start of mirror_frame's pre-integration for a frame that is about to be tracked (not yet in ft_map): the
    // samples between the last window frame and `incoming`, at the biases of the last window frame.
    // -> true when `tracker_from` is given and the integration queued here IS the feature tracker's own of the incoming frame's
    // interval (FeatureTracker::work integrates it without Jacobians from tracker_from's biases, feature_tracker.cpp:75-77): no frame
    // skipped in between, the same biases bit for bit.  The tracker then reads the delta out of this launch
    // (Pipeline::integrate_early) instead of integrating the interval a second time.
    bool mirror_prepare(Map *ft_map, Frame *incoming, const Frame *tracker_from = nullptr) {
        cancel_prepared();
        Frame *keyframe = map->get_frame(map->frame_num() - 1);
        Frame *new_i = keyframe->subframes.empty() ? keyframe : keyframe->subframes.back().get();
        const size_t idx_i = ft_map->frame_index_by_id(new_i->id);
        if (idx_i == nil()) return false;
        if (tracker_from && !(idx_i + 1 == ft_map->frame_num() && tracker_from->id == new_i->id && same3(tracker_from->motion.bg, new_i->motion.bg) &&
                              same3(tracker_from->motion.ba, new_i->motion.ba)))
            return false;   // not the same integration: nothing queued, the caller goes the two-launch way
        std::vector<ImuData> nd = incoming->preintegration.data;
        for (size_t index = ft_map->frame_num() - 1; index > idx_i; --index) {
            const std::vector<ImuData> &od = ft_map->get_frame(index)->preintegration.data;
            nd.insert(nd.begin(), od.begin(), od.end());
        }
        prepared_ = P_.integrate_begin(nd, incoming->image->t, new_i->motion.bg, new_i->motion.ba, true, true);
        prepared_id_ = incoming->id;
        prepared_from_ = new_i->id;
        prepared_samples_ = nd.size();
        return prepared_ && tracker_from != nullptr;
    }
    void cancel_prepared() {   // a queued integration nobody will collect must still be drained
        if (prepared_id_ != nil() && prepared_) {
            PreInt scratch;
            P_.integrate_end(scratch);
        }
        prepared_id_ = nil();
        prepared_ = false;
    }
    ~SlidingWindowTracker() {
        try {
            cancel_prepared();
            drain_speculation();
        } catch (...) {
        }
        if (std::getenv("XRHIP_HOSTPROF"))
            std::fprintf(stderr, "[hostprof] subframe re-integrations: %ld speculated, %ld taken from mirror_frame, %ld computed in place\n",
                         spec_hits_, memo_hits_, spec_misses_);
        if (std::getenv("XRHIP_HOSTPROF"))
            std::fprintf(stderr, "[hostprof] mirror_frame: %ld of %ld interval integrations were queued ahead (%ld behind a solve, on the device)\n",
                         mirror_prepared_, mirror_total_, mirror_chained_);
    }
    size_t prepared_id_ = nil(), prepared_from_ = nil(), prepared_samples_ = 0;
    bool prepared_ = false;
    long mirror_prepared_ = 0, mirror_total_ = 0, mirror_chained_ = 0;
    // Error unwind (System::recover_after_error): the device side of every queued integration has been cancelled
    // (Pipeline::cancel_integrations); this is the host side -- nothing is "prepared", "chained", "queued" or "speculated" any more,
    // so no later call tries to collect a batch that is gone (xrhip_ba_preintegrate_end: "nothing in flight").
    void forget_queued_work() noexcept {
        prepared_id_ = nil();
        prepared_ = false;
        chained_id_ = nil();
        kf_queued_ = false;
        kf_ok_.clear();
        spec_jobs_ = 0;
        spec_.clear();
        memo_id_ = nil();
        std::lock_guard<std::mutex> lk(hint_mutex_);
        hint_.reset();
    }

    // Pipelined mode (System::set_threading): the window map belongs to the backend thread while the feature tracker works on
    // the next frame, so mirror_prepare cannot be called from there.  The feature tracker posts what it knows -- the samples of
    // the interval that will be mirrored next -- and the backend, at the end of track(), when the biases that integration
    // starts from are final, queues it; mirror_frame then finds it prepared exactly as after mirror_prepare.  A hint that
    // comes too late is dropped (mirror_frame integrates in place): the values are the same either way.
    struct MirrorHint {
        size_t id, from;
        double t;
        std::vector<ImuData> samples;
    };
    void post_mirror_hint(MirrorHint h) {
        std::lock_guard<std::mutex> lk(hint_mutex_);
        hint_ = std::move(h);
    }
    void drop_mirror_hint() {
        std::lock_guard<std::mutex> lk(hint_mutex_);
        hint_.reset();
    }
    void take_mirror_hint() {   // backend thread
        std::optional<MirrorHint> h;
        {
            std::lock_guard<std::mutex> lk(hint_mutex_);
            h.swap(hint_);
        }
        if (!h) return;
        cancel_prepared();
        Frame *keyframe = map->get_frame(map->frame_num() - 1);
        Frame *new_i = keyframe->subframes.empty() ? keyframe : keyframe->subframes.back().get();
        if (new_i->id != h->from) return;
        prepared_ = P_.integrate_begin(h->samples, h->t, new_i->motion.bg, new_i->motion.ba, true, true);
        prepared_id_ = h->id;
        prepared_from_ = h->from;
        prepared_samples_ = h->samples.size();
    }
    // The same, earlier: refine_subwindow is the last solve of a non-keyframe frame and `newest` the frame the next interval
    // starts from -- if the hint is there already, the integration is queued BEHIND that solve on the device and reads the
    // biases where the solve leaves them (BaBuilder::chain_integration), instead of after the host has read them back.
    void chain_mirror_hint(BaBuilder &b, Frame *newest) {
        std::optional<MirrorHint> h;
        {
            std::lock_guard<std::mutex> lk(hint_mutex_);
            if (hint_ && hint_->from == newest->id) h.swap(hint_);
        }
        if (!h) return;
        cancel_prepared();
        chained_id_ = h->id;
        chained_from_ = h->from;
        chained_samples_ = h->samples.size();
        b.chain_integration(newest, std::move(h->samples), h->t);
    }
    void chained_solve_done(const BaBuilder &b) {
        if (chained_id_ == nil()) return;
        prepared_ = b.chained();
        if (prepared_) mirror_chained_++;
        prepared_id_ = chained_id_;
        prepared_from_ = chained_from_;
        prepared_samples_ = chained_samples_;
        chained_id_ = nil();
    }
    size_t chained_id_ = nil(), chained_from_ = nil(), chained_samples_ = 0;
    std::mutex hint_mutex_;
    std::optional<MirrorHint> hint_;

    // mirror_frame (sliding_window_tracker.cpp:31-80) in two halves.  What it reads from the TRACKING map -- the new frame,
    // the IMU samples of the frames it skips, and which keypoint of the last window frame continues as which keypoint of the
    // new one -- is gathered into a packet (by the thread that owns the tracking map: in pipelined mode the feature
    // tracker's, where those frames and tracks are in cache); the window map's half then works from the packet alone.
    struct MirrorPacket {
        struct Link {
            uint32_t ki, kj;      // keypoint of the last window frame -> keypoint of the new frame
            Track *ft_track;      // the tracking map's track (only written to when its trash tag changes)
            bool trash;           // that tag as it stands
        };
        size_t from_id = nil(), frame_id = nil();
        bool valid = false;
        std::unique_ptr<Frame> frame;   // clone of the new frame, its interval extended over the skipped frames
        std::vector<Link> links;
    };
    size_t newest_frame_id() const {
        const Frame *keyframe = map->get_frame(map->frame_num() - 1);
        return keyframe->subframes.empty() ? keyframe->id : keyframe->subframes.back()->id;
    }
    static MirrorPacket make_mirror_packet(Map *ft_map, size_t frame_id, size_t from_id) {
        MirrorPacket pk;
        pk.from_id = from_id;
        pk.frame_id = frame_id;
        const size_t idx_i = ft_map->frame_index_by_id(from_id), idx_j = ft_map->frame_index_by_id(frame_id);
        if (idx_i == nil() || idx_j == nil()) return pk;
        Frame *old_i = ft_map->get_frame(idx_i), *old_j = ft_map->get_frame(idx_j);
        pk.frame = old_j->clone();
        std::vector<ImuData> &nd = pk.frame->preintegration.data;
        for (size_t index = idx_j - 1; index > idx_i; --index) {
            const std::vector<ImuData> &od = ft_map->get_frame(index)->preintegration.data;
            nd.insert(nd.begin(), od.begin(), od.end());
        }
        pk.links.reserve(old_i->keypoint_num());
        for (size_t ki = 0; ki < old_i->keypoint_num(); ++ki) {
            if (Track *track = old_i->get_track(ki)) {
                const size_t kj = track->get_keypoint_index(old_j);
                if (kj != nil()) pk.links.push_back({(uint32_t)ki, (uint32_t)kj, track, track->tag(TT_TRASH)});
            }
        }
        pk.valid = true;
        return pk;
    }
    bool mirror_frame(Map *ft_map, size_t frame_id) {
        feature_tracking_map_ = ft_map;
        return mirror_frame(make_mirror_packet(ft_map, frame_id, newest_frame_id()));
    }
    bool mirror_frame(MirrorPacket pk) {   // true: the frame is now the newest frame of the window map
        xrhip::HostProfScope hp_m(7, "mirror_frame");
        WallTimer sc_t(P_.times.scope[SC_MIRROR]);
        Frame *keyframe = map->get_frame(map->frame_num() - 1);
        Frame *new_i = keyframe;
        if (!keyframe->subframes.empty()) new_i = keyframe->subframes.back().get();
        if (!pk.valid || pk.from_id != new_i->id) {
            cancel_prepared();
            return false;
        }
        const size_t frame_id = pk.frame_id;
        std::unique_ptr<Frame> curr = std::move(pk.frame);
        std::vector<ImuData> &nd = curr->preintegration.data;
        // the pre-integration of the new interval only needs its IMU samples and the biases of the last window frame:
        // mirror_prepare queued it when the frame entered the tracker (it has been running beside the LK kernel);
        // otherwise it is queued now and runs while the track links are copied below
        bool integrating;
        mirror_total_++;
        if (prepared_id_ == frame_id && prepared_from_ == new_i->id && prepared_samples_ == nd.size()) {
            integrating = prepared_;
            prepared_id_ = nil();
            mirror_prepared_++;
        } else {
            cancel_prepared();
            integrating = P_.integrate_begin(nd, curr->image->t, new_i->motion.bg, new_i->motion.ba, true, true);
        }
        map->attach_frame(std::move(curr));
        Frame *new_j = map->get_frame(map->frame_num() - 1);
        const bool log_links = P_.swt_log.enabled();
        std::unique_lock<std::mutex> log_lock(P_.swt_log.mu, std::defer_lock);
        if (log_links) log_lock.lock();
        if (log_links) std::fprintf(P_.swt_log.fp, "{\"mirror\": %zu, \"from\": %zu, \"links\": [", frame_id, new_i->id);
        bool first_link = true;
        for (const MirrorPacket::Link &ln : pk.links) {
            const bool existed = new_i->get_track(ln.ki) != nullptr;
            Track *nt = new_i->get_track(ln.ki, map.get());
            nt->add_keypoint(new_j, ln.kj);
            if (log_links) {
                std::fprintf(P_.swt_log.fp, "%s[%u, %u, %zu, %d]", first_link ? "" : ", ", ln.ki, ln.kj, nt->id, existed ? 0 : 1);
                first_link = false;
            }
            // (written only when it changes: in pipelined mode the tracking map's tracks live in the other thread's
            // caches, and a store would take every one of those lines away from it)
            const bool trash = nt->tag(TT_TRASH) && !nt->tag(TT_STATIC);
            if (ln.trash != trash) ln.ft_track->tag(TT_TRASH) = trash;
        }
        map->prune_tracks([](const Track *t) { return t->tag(TT_TRASH) && !t->tag(TT_STATIC); });
        if (log_links) {   // what the new frame carries after the prune: keypoint -> window-map track
            std::fprintf(P_.swt_log.fp, "], \"after\": [");
            bool first = true;
            for (size_t kj = 0; kj < new_j->keypoint_num(); ++kj)
                if (Track *t = new_j->get_track(kj)) {
                    std::fprintf(P_.swt_log.fp, "%s[%zu, %zu]", first ? "" : ", ", kj, t->id);
                    first = false;
                }
            std::fprintf(P_.swt_log.fp, "]}\n");
            std::fflush(P_.swt_log.fp);
            log_lock.unlock();
        }
        if (integrating) {
            P_.integrate_end(new_j->preintegration);
            memo_id_ = new_j->id;
            memo_samples_ = new_j->preintegration.data.size();
            memo_bg_ = new_i->motion.bg;
            memo_ba_ = new_i->motion.ba;
        } else {
            memo_id_ = nil();
        }
        predict(new_j->preintegration, new_i, new_j);
        return true;
    }

    // `overlap`: host work of the caller that localize_newframe's solve does not depend on (the feature tracker's corner selection
    // for the frame just mirrored), run beside that solve; done by the time this returns.
    bool track(BaBuilder::Overlap *overlap = nullptr) {   // :82-117
        if (P_.config.parsac_flag) {
            if (overlap) {   // (the RD-VIO checks look at every keypoint of the new frame: finish it first)
                overlap->run();
                if (overlap->error) std::rethrow_exception(overlap->error);
                overlap = nullptr;
            }
            struct Trace {   // XRSLAM_AMD_DUMP_INIT: the PARSAC runs of this frame go into the pipeline's decision log
                explicit Trace(InitLogger *l) { parsac_trace() = l; }
                ~Trace() { parsac_trace() = nullptr; }
            } trace(P_.init_log.enabled() ? &P_.init_log : nullptr);
            if (judge_track_status()) update_track_status();
        }
        // localize_newframe, manage_keyframe, then refine_window (keyframe) or refine_subwindow (:96-116).  On a non-keyframe frame
        // the two solves are ONE submission (round 4): manage_keyframe reads tags and counts only (:145-223), and refine_subwindow's
        // problem depends on localize_newframe's result through nothing but the new frame's starting state (its factors, its
        // integrations and the other subframes' states are those of the frames before) -- so both problems are assembled first, the
        // keyframe decision is taken in between, and the second solve picks the new frame's state up on the device where the first
        // leaves it (BaBuilder::solve_chained): one wait instead of two, the sub-window's assembly off the critical path.  Same
        // problems, same results.  Not with the decision log / the problem dump on (their records are written solve by solve), nor
        // with the RD-VIO filters; XRSLAM_AMD_NO_CHAINED_SOLVES=1: one solve after the other.
        static const bool no_pair = std::getenv("XRSLAM_AMD_NO_CHAINED_SOLVES") != nullptr;   // development switch (A/B, parity)
        const bool pair_ok = !no_pair && !P_.config.parsac_flag && !P_.swt_log.enabled() && !P_.ba_dump.enabled();
        size_t log_id = 0, log_mapped = 0;
        bool log_nt = false;
        bool is_kf = false, subwindow_done = false;
        if (pair_ok) {
            std::optional<BaBuilder> a;
            Frame *const fj = map->get_frame(map->frame_num() - 1);
            int link_a = -1;
            bool begun = false;
            {
                WallTimer sc_t(P_.times.scope[SC_LOCALIZE]);
                a.emplace(P_);
                localize_assemble(*a);
                a->prepare();
                link_a = fj->ba_index;
                begun = a->begin(P_.ba_sub);   // the device works on it from here on
            }
            struct AbortBegun {   // (an exception below must not leave the solve "in flight": its problem dies with the builder)
                Pipeline &P;
                bool &armed;
                ~AbortBegun() {
                    if (armed) xrhip_ba_solve_abort(P.ba_sub);
                }
            } abort_begun{P_, begun};
            if (overlap) {   // manage_keyframe reads the new frame's FT_NO_TRANSLATION tag
                overlap->run_early();
                if (overlap->error) std::rethrow_exception(overlap->error);
            }
            is_kf = manage_keyframe();
            BaBuilder b(P_);
            bool have_b = false;
            if (!is_kf) {   // ... and assembles the second problem meanwhile
                WallTimer sc_t(P_.times.scope[SC_REFINE_SUBWINDOW]);
                have_b = refine_subwindow_assemble(b);
                if (have_b) b.prepare();
                subwindow_done = true;
            }
            WallTimer sc_t(P_.times.scope[SC_LOCALIZE]);
            if (have_b && begun) {
                begun = false;   // (solve_linked collects or aborts it)
                BaBuilder::solve_linked(*a, link_a, b, fj, overlap);
                refine_subwindow_finish(b);
            } else {
                if (begun) {
                    begun = false;
                    a->end(P_.ba_sub, overlap);
                } else {
                    a->solve(nullptr, overlap);
                }
                if (have_b) {   // (localize_newframe was not a single-launch problem: the second solve starts from the frame as it is now)
                    b.repack(fj);
                    b.solve();
                    refine_subwindow_finish(b);
                }
            }
        } else {
            localize_newframe(overlap);
            if (P_.swt_log.enabled()) {   // the inputs manage_keyframe is about to look at
                const Frame *nf = map->get_frame(map->frame_num() - 1);
                log_id = nf->id;
                log_nt = nf->tag(FT_NO_TRANSLATION);
                for (size_t k = 0; k < nf->keypoint_num(); ++k)
                    if (Track *t = nf->get_track(k))
                        if (t->all_tagged({TT_VALID, TT_TRIANGULATED, TT_STATIC})) log_mapped++;
            }
            is_kf = manage_keyframe();
        }
        if (is_kf) {
            P_.times.keyframes++;
            // a keyframe's window solve and marginalisation take several ordinary frames: the group's other members do not wait
            // for this one at the frame gate meanwhile (it rejoins them at its next frame)
            struct GroupBusy {
                Pipeline &P;
                explicit GroupBusy(Pipeline &p) : P(p) {
                    if (P.group) xrhip_klt_group_busy(P.klt, 1);
                }
                ~GroupBusy() {
                    if (P.group) xrhip_klt_group_busy(P.klt, 0);
                }
            } group_busy(P_);
            queue_keyframe_integrations();
            track_landmark();
            refine_window();
            take_mirror_hint();   // the newest frame's biases are final: the next interval integrates beside slide_window
            slide_window();
        } else {
            if (!subwindow_done) refine_subwindow();
            take_mirror_hint();
        }
        if (P_.swt_log.enabled()) {
            std::lock_guard<std::mutex> log_lock(P_.swt_log.mu);
            FILE *fp = P_.swt_log.fp;
            std::fprintf(fp, "{\"frame\": %zu, \"no_translation\": %d, \"mapped\": %zu, \"keyframe\": %d, \"window\": [", log_id,
                         log_nt ? 1 : 0, log_mapped, is_kf ? 1 : 0);
            for (size_t i = 0; i < map->frame_num(); ++i) {
                const Frame *f = map->get_frame(i);
                std::fprintf(fp, "%s[%zu, %d, [", i ? ", " : "", f->id, f->tag(FT_NO_TRANSLATION) ? 1 : 0);
                for (size_t j = 0; j < f->subframes.size(); ++j)
                    std::fprintf(fp, "%s[%zu, %d]", j ? ", " : "", f->subframes[j]->id, f->subframes[j]->tag(FT_NO_TRANSLATION) ? 1 : 0);
                std::fprintf(fp, "]]");
            }
            std::fprintf(fp, "]}\n");
            std::fflush(fp);
        }
        {
            xrhip::HostProfScope hp_sp(24, "track: speculate_subframes + hint");
            speculate_subframes();
            take_mirror_hint();
        }
        if (P_.out_log.enabled()) log_outputs(is_kf);
        return true;
    }
    // XRSLAM_AMD_DUMP_OUT: the 'B' record (ba_dump.hpp) -- the newest frame's state, the window, its key points' tracks, every track
    void log_outputs(bool is_kf) {
        const Frame *kf = map->get_frame(map->frame_num() - 1);
        const Frame *nf = kf->subframes.empty() ? kf : kf->subframes.back().get();
        OutLogger::Record r(P_.out_log, 'B');
        r.u64(nf->id);
        r.u32(is_kf ? 1 : 0);
        double st[16];
        BaBuilder::pack_state(nf, st);
        for (double v : st) r.f64(v);
        r.u32((uint32_t)map->frame_num());
        for (size_t i = 0; i < map->frame_num(); ++i) {
            const Frame *f = map->get_frame(i);
            r.u64(f->id);
            r.u32((uint32_t)f->subframes.size());
            for (const auto &sf : f->subframes) r.u64(sf->id);
        }
        r.u32((uint32_t)nf->keypoint_num());
        for (size_t k = 0; k < nf->keypoint_num(); ++k) r.i64(nf->get_track(k) ? (int64_t)nf->get_track(k)->id : -1);
        r.u32((uint32_t)map->track_num());
        for (size_t k = 0; k < map->track_num(); ++k) {
            const Track *t = map->get_track(k);
            uint32_t bits = 0;
            for (int b = 0; b < TT_COUNT; ++b)
                if (t->tags[b]) bits |= 1u << b;
            r.u64(t->id);
            r.u32(bits);
            r.f64(std::isfinite(t->landmark.inv_depth) ? t->landmark.inv_depth : -1.0);
            V3 x{0, 0, 0};
            if (t->tag(TT_TRIANGULATED) && t->keypoint_num() > 0 && std::isfinite(t->landmark.inv_depth) && t->landmark.inv_depth != 0.0)
                x = t->get_landmark_point();
            r.f64(x.x);
            r.f64(x.y);
            r.f64(x.z);
        }
    }

    // ------------------------------------------------------------------ RD-VIO dynamic-object rejection (:523-790)
    // Only runs with parsac.parsac_flag (off in the shipped configurations).  judge_track_status marks the landmarks
    // whose reprojection disagrees with the IMU-aided PnP consensus as TT_OUTLIER / not TT_STATIC, which removes their
    // factors from the solves below; update_track_status runs the 2D-2D PARSAC check against the last keyframes.
    struct Rt {
        M3 R = M3::identity();
        V3 t;
    };
    static Rt rt_mul(const Rt &a, const Rt &b) { return {a.R * b.R, a.R * b.t + a.t}; }
    static Rt rt_inv(const Rt &a) {
        const M3 Rt_ = transpose(a.R);
        return {Rt_, -(Rt_ * a.t)};
    }
    static void predict_RT(Frame *fi, Frame *fj, M3 &R, V3 &t) {   // :549-575, literally (both extrinsics enter)
        const Rt Pwc{to_matrix(fi->camera.q_cs), fi->camera.p_cs}, PwI{to_matrix(fi->imu.q_cs), fi->imu.p_cs};
        const Rt Pwi{to_matrix(fi->pose.q), fi->pose.p}, Pwj{to_matrix(fj->pose.q), fj->pose.p};
        const Rt Pji = rt_mul(rt_inv(Pwj), Pwi);
        const Rt P = rt_mul(rt_mul(rt_mul(rt_mul(rt_inv(Pwc), PwI), Pji), rt_inv(PwI)), Pwc);
        R = P.R;
        t = P.t;
    }
    static double epipolar_dist(const M3 &F, V2 a, V2 b) {   // :490-495
        const V3 l = F * V3{a.x, a.y, 1.0};
        return std::fabs(b.x * l.x + b.y * l.y + l.z) / std::sqrt(l.x * l.x + l.y * l.y);
    }
    static M3 k_inverse(const Intrinsics &K) {
        M3 m;
        m(0, 0) = 1.0 / K.fx;
        m(0, 2) = -K.cx / K.fx;
        m(1, 1) = 1.0 / K.fy;
        m(1, 2) = -K.cy / K.fy;
        m(2, 2) = 1.0;
        return m;
    }

    bool judge_track_status() {   // :577-739
        Frame *curr = map->get_frame(map->frame_num() - 1);
        Frame *keyframe = map->get_frame(map->frame_num() - 2);
        Frame *last = keyframe->subframes.empty() ? keyframe : keyframe->subframes.back().get();
        P_.integrate(curr->preintegration, curr->image->t, last->motion.bg, last->motion.ba, true, true);
        predict(curr->preintegration, last, curr);
        std::vector<V2> P2D;
        std::vector<V3> P3D;
        std::vector<size_t> lens;
        std::vector<int> index(curr->keypoint_num(), -1);
        for (size_t k = 0; k < curr->keypoint_num(); ++k)
            if (Track *t = curr->get_track(k))
                if (t->all_tagged({TT_VALID, TT_TRIANGULATED})) {
                    const V3 &b = curr->get_keypoint(k);
                    P2D.push_back({b.x / b.z, b.y / b.z});
                    P3D.push_back(t->get_landmark_point());
                    lens.push_back(t->m_life);
                    index[k] = (int)P3D.size() - 1;
                }
        if (P2D.size() < 20) return false;
        const PoseState pose = curr->get_pose(curr->camera);
        const M3 Rcw = to_matrix(pose.q.conjugate());
        const V3 tcw = -(pose.q.conjugate() * pose.p);
        std::vector<char> mask;
        find_pnp_matrix_parsac_imu(parsac_, P3D, P2D, lens, Rcw, tcw, 0.20, 1.0, mask, 1.0 / curr->K.fx);
        M3 R;
        V3 t;
        predict_RT(keyframe, curr, R, t);
        M3 tx;
        tx(0, 1) = -t.z; tx(0, 2) = t.y; tx(1, 0) = t.z; tx(1, 2) = -t.x; tx(2, 0) = -t.y; tx(2, 1) = t.x;
        const M3 F = transpose(k_inverse(keyframe->K)) * (tx * R) * k_inverse(curr->K);
        const M3 Ft = transpose(F);
        std::vector<double> d_in, d_out;
        for (size_t i = 0; i < curr->keypoint_num(); ++i) {
            if (index[i] == -1) continue;
            const size_t j = curr->get_track(i)->get_keypoint_index(keyframe);
            if (j == nil()) continue;
            const V2 p1 = apply_k(keyframe->get_keypoint(j), keyframe->K), p2 = apply_k(curr->get_keypoint(i), curr->K);
            const double err = epipolar_dist(F, p1, p2) + epipolar_dist(Ft, p2, p1);
            (mask[index[i]] ? d_in : d_out).push_back(err);
        }
        bool separated = false;
        double th1 = 0, th2 = 0;
        if (d_in.size() >= 20 && d_out.size() >= 20) {
            std::sort(d_in.begin(), d_in.end());
            std::sort(d_out.begin(), d_out.end());
            th1 = d_in[size_t(d_in.size() * 0.5)];
            th2 = d_out[size_t(d_out.size() * 0.5)];
            separated = !(th2 < th1 * 2);   // otherwise the two groups are not separated: ambiguous
        }
        if (P_.init_log.enabled()) {   // what the verdict looked at: the epipolar distances of the PnP consensus' inliers / outliers
            InitLogger::Line ln(P_.init_log, "judge_track_status");
            ln.put("frame", (double)curr->id);
            ln.put("d_in", d_in); ln.put("d_out", d_out);
            ln.put("separated", separated ? 1.0 : 0.0);
            ln.put("threshold", separated ? (th1 + th2) / 2 : 0.0);
            std::vector<double> m(mask.begin(), mask.end());
            ln.put("mask", m);
        }
        if (!separated) return false;
        parsac_th_ = (th1 + th2) / 2;
        for (size_t k = 0; k < curr->keypoint_num(); ++k)
            if (Track *tr = curr->get_track(k))
                if (index[k] != -1) {
                    const bool inlier = mask[index[k]] != 0;
                    tr->tag(TT_OUTLIER) = !inlier;
                    tr->tag(TT_STATIC) = inlier;
                    if (!inlier) P_.times.scope[SC_RD_OUTLIERS] += 1.0;
                }
        P_.times.scope[SC_RD_JUDGED] += 1.0;
        return true;
    }

    bool filter_parsac_2d2d(Frame *fi, Frame *fj, std::vector<char> &mask, std::vector<size_t> &pts_to_index) {   // :523-547
        std::vector<V2> p1, p2;
        for (size_t ki = 0; ki < fi->keypoint_num(); ++ki)
            if (Track *t = fi->get_track(ki)) {
                const size_t kj = t->get_keypoint_index(fj);
                if (kj == 0 || kj == nil()) continue;   // `if (size_t kj = ...)` in the reference also drops index 0
                const V3 &a = fi->get_keypoint(ki), &b = fj->get_keypoint(kj);
                p1.push_back({a.x / a.z, a.y / a.z});
                p2.push_back({b.x / b.z, b.y / b.z});
                pts_to_index.push_back(kj);
            }
        if (p1.size() < 10) return false;
        find_essential_matrix_parsac(parsac_, p1, p2, mask, parsac_th_ / fi->K.fx);
        return true;
    }

    void update_track_status() {   // :741-788
        Frame *curr = map->get_frame(map->frame_num() - 1);
        if (!feature_tracking_map_) return;
        const size_t fidx = feature_tracking_map_->frame_index_by_id(curr->id);
        if (fidx == nil()) return;
        Frame *old_frame = feature_tracking_map_->get_frame(fidx);
        std::vector<size_t> outlier_cnts(curr->keypoint_num(), 0), matches_cnts(curr->keypoint_num(), 0);
        const size_t last = map->frame_num() - 1;
        const size_t start = std::min(last, std::max(last - P_.config.parsac_keyframe_check_size, size_t(0)));   // unsigned, like the reference
        for (size_t i = start; i < last; ++i) {
            std::vector<char> mask;
            std::vector<size_t> to_index;
            if (filter_parsac_2d2d(map->get_frame(i), curr, mask, to_index))
                for (size_t j = 0; j < mask.size(); ++j) {
                    if (!mask[j]) outlier_cnts[to_index[j]] += 1;
                    matches_cnts[to_index[j]] += 1;
                }
        }
        // The reference looks the window's track up in the FEATURE TRACKER's frame (a different map: the lookup never
        // succeeds), so the demotion below is unreachable there as well; kept for the day that lookup is fixed upstream.
        for (size_t i = 0; i < curr->keypoint_num(); ++i)
            if (Track *ct = curr->get_track(i)) {
                const size_t j = ct->get_keypoint_index(old_frame);
                if (j == 0 || j == nil()) continue;
                Track *ot = old_frame->get_track(j);
                if (!ot) continue;
                const size_t outlier_th = map->frame_num() / 2;
                if (outlier_cnts[i] > outlier_th / 2 && outlier_cnts[i] > 0.8 * matches_cnts[i]) ct->tag(TT_STATIC) = false;
                if (!ot->tag(TT_STATIC) || !ct->tag(TT_STATIC)) {
                    ct->tag(TT_STATIC) = false;
                    ot->tag(TT_STATIC) = false;
                }
            }
    }
    ParsacState parsac_;
    double parsac_th_ = 0.0;
    Map *feature_tracking_map_ = nullptr;

    void localize_newframe(BaBuilder::Overlap *overlap = nullptr) {   // :119-143
        WallTimer sc_t(P_.times.scope[SC_LOCALIZE]);
        BaBuilder b(P_);
        localize_assemble(b);
        b.solve(nullptr, overlap);
    }
    void localize_assemble(BaBuilder &b) {   // the problem of :119-143 (called before manage_keyframe: the new frame is the map's last)
        xrhip::HostProfScope hp_a(25, "localize: problem assembly");
        Frame *fi = map->get_frame(map->frame_num() - 2);
        if (!fi->subframes.empty()) fi = fi->subframes.back().get();
        Frame *fj = map->get_frame(map->frame_num() - 1);
        b.add_frame_states(fj);
        b.add_preintegration_prior(fi, fj, fj->preintegration);
        for (size_t k = 0; k < fj->keypoint_num(); ++k)
            if (Track *t = fj->get_track(k))
                if (t->all_tagged({TT_VALID, TT_TRIANGULATED, TT_STATIC})) b.add_reprojection_prior(fj, k);
    }

    bool manage_keyframe() {   // :145-223
        WallTimer sc_t(P_.times.scope[SC_MANAGE_KF]);
        const Config &c = P_.config;
        Frame *kf_i = map->get_frame(map->frame_num() - 2);
        Frame *nf_j = map->get_frame(map->frame_num() - 1);
        if (!kf_i->subframes.empty()) {
            if (kf_i->subframes.back()->tag(FT_NO_TRANSLATION)) {
                if (nf_j->tag(FT_NO_TRANSLATION)) {
                    // fall through to the landmark count
                } else {
                    kf_i->subframes.back()->tag(FT_KEYFRAME) = true;
                    map->attach_frame(std::move(kf_i->subframes.back()), map->frame_num() - 1);
                    kf_i->subframes.pop_back();
                    nf_j->tag(FT_KEYFRAME) = true;
                    return true;
                }
            } else {
                if (nf_j->tag(FT_NO_TRANSLATION)) {
                    std::unique_ptr<Frame> lifted = std::move(kf_i->subframes.back());
                    kf_i->subframes.pop_back();
                    lifted->tag(FT_KEYFRAME) = true;
                    lifted->subframes.emplace_back(map->detach_frame(map->frame_num() - 1));
                    map->attach_frame(std::move(lifted));
                    return true;
                } else if (kf_i->subframes.size() >= c.sliding_window_subframe_size) {
                    nf_j->tag(FT_KEYFRAME) = true;
                    return true;
                }
            }
        }
        size_t mapped = 0;
        for (size_t k = 0; k < nf_j->keypoint_num(); ++k)
            if (Track *t = nf_j->get_track(k))
                if (t->all_tagged({TT_VALID, TT_TRIANGULATED, TT_STATIC})) mapped++;
        if (mapped < c.sliding_window_force_keyframe_landmarks) {
            nf_j->tag(FT_KEYFRAME) = true;
            return true;
        }
        kf_i->subframes.emplace_back(map->detach_frame(map->frame_num() - 1));
        return false;
    }

    // XRSLAM_AMD_DUMP_SWT: what a landmark decision looked at -- every observation of the track as (camera pose q xyzw, p; bearing;
    // keyframe flag; fx fy cx cy), in the track's own order (the first one anchors the inverse depth); tests/swt_model.py
    static void log_track_observations(FILE *fp, const Track *t) {
        std::fprintf(fp, "\"obs\": [");
        bool first = true;
        for (const auto &[f, ki] : t->keypoint_refs) {
            const PoseState pose = f->get_pose(f->camera);
            const V3 z = f->get_keypoint(ki);
            std::fprintf(fp, "%s[%.17g, %.17g, %.17g, %.17g, %.17g, %.17g, %.17g, %.17g, %.17g, %.17g, %d, %.17g, %.17g, %.17g, %.17g, %zu]",
                         first ? "" : ", ", pose.q.x, pose.q.y, pose.q.z, pose.q.w, pose.p.x, pose.p.y, pose.p.z, z.x, z.y, z.z,
                         f->tag(FT_KEYFRAME) ? 1 : 0, f->K.fx, f->K.fy, f->K.cx, f->K.cy, f->id);
            first = false;
        }
        std::fprintf(fp, "]");
    }
    void track_landmark() {   // :225-245
        WallTimer sc_t(P_.times.scope[SC_TRACK_LANDMARK]);
        Frame *nf = map->get_frame(map->frame_num() - 1);
        const bool log = P_.swt_log.enabled();
        for (size_t k = 0; k < nf->keypoint_num(); ++k) {
            if (Track *t = nf->get_track(k)) {
                if (!t->tag(TT_TRIANGULATED)) {
                    std::unique_lock<std::mutex> log_lock(P_.swt_log.mu, std::defer_lock);
                    if (log) {   // the inputs first: set_landmark_point does not move them, but keep the record self-contained
                        log_lock.lock();
                        std::fprintf(P_.swt_log.fp, "{\"triangulate\": %zu, \"track\": %zu, ", nf->id, t->id);
                        log_track_observations(P_.swt_log.fp, t);
                    }
                    if (auto p = t->triangulate()) {
                        t->set_landmark_point(p.value());
                        t->tag(TT_TRIANGULATED) = true;
                        t->tag(TT_VALID) = true;
                        t->tag(TT_STATIC) = true;
                    } else {
                        t->landmark.inv_depth = -1.0;
                        t->tag(TT_TRIANGULATED) = false;
                        t->tag(TT_VALID) = false;
                    }
                    if (log) {
                        std::fprintf(P_.swt_log.fp, ", \"ok\": %d, \"inv_depth\": %.17g}\n", t->tag(TT_TRIANGULATED) ? 1 : 0, t->landmark.inv_depth);
                        std::fflush(P_.swt_log.fp);
                    }
                }
            }
        }
    }

    // refine_window's first step, callable ahead of it: nothing between manage_keyframe and the window solve changes the frames,
    // their samples or their biases (track_landmark triangulates tracks), and the batch (~10 intervals of 20-40 samples, 60-100 us on
    // the device) was still 37 us short of done when the assembly had finished.
    std::vector<char> kf_ok_;
    bool kf_queued_ = false;
    void queue_keyframe_integrations() {
        std::vector<Pipeline::IntegrateJob> kf_jobs;
        for (size_t j = 1; j < map->frame_num(); ++j) {
            Frame *fi = map->get_frame(j - 1), *fj = map->get_frame(j);
            fj->keyframe_preintegration = fj->preintegration;
            if (!fi->subframes.empty()) {
                std::vector<ImuData> extra;
                for (auto &sf : fi->subframes) extra.insert(extra.end(), sf->preintegration.data.begin(), sf->preintegration.data.end());
                fj->keyframe_preintegration.data.insert(fj->keyframe_preintegration.data.begin(), extra.begin(), extra.end());
            }
            kf_jobs.push_back({&fj->keyframe_preintegration, fj->image->t, fi->motion.bg, fi->motion.ba});
        }
        kf_ok_ = P_.integrate_batch_begin(kf_jobs, true, true);
        kf_queued_ = true;
    }

    void refine_window() {   // :247-358
        WallTimer sc_t(P_.times.scope[SC_REFINE_WINDOW]);
        std::optional<xrhip::HostProfScope> hp_a(std::in_place, 17, "refine_window: assembly");
        BaBuilder b(P_);
        if (!map->marginalization_factor) map->marginalization_factor = create_marginalization_factor(map.get());
        // the keyframe intervals are re-integrated at the current biases: queued by track() before track_landmark (the device works on
        // them while the host triangulates and walks the tracks below); the records are read when the problem is handed over
        // (BaBuilder::solve)
        if (!kf_queued_) queue_keyframe_integrations();
        kf_queued_ = false;
        const std::vector<char> &ok = kf_ok_;
        for (size_t i = 0; i < map->frame_num(); ++i) b.add_frame_states(map->get_frame(i));
        const unsigned long visit = ++P_.ba_generation;   // (a stamp on the track instead of a hash set: ~400 tracks per window)
        for (size_t i = 0; i < map->frame_num(); ++i) {
            Frame *f = map->get_frame(i);
            for (size_t j = 0; j < f->keypoint_num(); ++j) {
                Track *t = f->get_track(j);
                if (!t || t->visit_gen == visit) continue;
                t->visit_gen = visit;
                if (!t->tag(TT_VALID) || !t->tag(TT_STATIC)) continue;
                if (!t->first_frame()->tag(FT_KEYFRAME)) continue;
                b.add_track_states(t);
            }
        }
        b.add_marginalization(map->marginalization_factor.get());
        for (size_t i = 0; i < map->frame_num(); ++i) {
            Frame *f = map->get_frame(i);
            for (size_t j = 0; j < f->keypoint_num(); ++j) {
                Track *t = f->get_track(j);
                if (!t || !t->all_tagged({TT_VALID, TT_TRIANGULATED, TT_STATIC})) continue;
                if (!t->first_frame()->tag(FT_KEYFRAME)) continue;
                if (f == t->first_frame()) continue;
                b.add_reprojection_error(f, j);
            }
        }
        {
            for (size_t j = 1; j < map->frame_num(); ++j)
                if (ok[j - 1]) {
                    Frame *fj = map->get_frame(j);
                    b.add_preintegration_error(map->get_frame(j - 1), fj, fj->keyframe_preintegration);
                }
        }
        hp_a.reset();
        {
            xrhip::HostProfScope hp_w(18, "refine_window: batch_end wait");
            P_.integrate_batch_end();
        }
        b.solve();
        xrhip::HostProfScope hp_c(19, "refine_window: landmark sweep");
        const bool log_cull = P_.swt_log.enabled();
        std::vector<std::pair<const Frame *, PoseState>> cam_cache;
        cam_cache.reserve(16);
        for (size_t k = 0; k < map->track_num(); ++k) {
            Track *t = map->get_track(k);
            std::unique_lock<std::mutex> log_lock(P_.swt_log.mu, std::defer_lock);
            if (log_cull) {   // the landmark sweep after the window solve (:325-357): inputs, then the verdict below
                log_lock.lock();
                // (a track that never triangulated can carry a NaN inverse depth: JSON has no NaN, the log says -1 like the reset below)
                std::fprintf(P_.swt_log.fp, "{\"cull\": %zu, \"track\": %zu, \"triangulated\": %d, \"inv_depth\": %.17g, ",
                             map->get_frame(map->frame_num() - 1)->id, t->id, t->tag(TT_TRIANGULATED) ? 1 : 0,
                             std::isfinite(t->landmark.inv_depth) ? t->landmark.inv_depth : -1.0);
                log_track_observations(P_.swt_log.fp, t);
            }
            struct Verdict {   // written when the track's branch below is done
                FILE *fp;
                Track *t;
                ~Verdict() {
                    if (!fp) return;
                    std::fprintf(fp, ", \"valid\": %d, \"inv_depth_after\": %.17g}\n", t->tag(TT_VALID) ? 1 : 0, t->landmark.inv_depth);
                    std::fflush(fp);
                }
            } verdict{log_cull ? P_.swt_log.fp : nullptr, t};
            if (t->tag(TT_TRIANGULATED)) {
                bool valid = true;
                V3 x = t->get_landmark_point();
                double rpe = 0.0, cnt = 0.0;
                for (const auto &[f, ki] : t->keypoint_refs) {
                    if (!f->tag(FT_KEYFRAME)) continue;
                    // the camera pose of a keyframe is the same for every track seen in it: formed once per sweep
                    // (~2000 observations over ~10 keyframes)
                    PoseState pose;
                    {
                        size_t ci = 0;
                        while (ci < cam_cache.size() && cam_cache[ci].first != f) ++ci;
                        if (ci == cam_cache.size()) cam_cache.emplace_back(f, f->get_pose(f->camera));
                        pose = cam_cache[ci].second;
                    }
                    V3 y = pose.q.conjugate() * (x - pose.p);
                    if (y.z <= 1.0e-3 || y.z > 50) {
                        valid = false;
                        break;
                    }
                    V2 a = apply_k(y, f->K), bb = apply_k(f->get_keypoint(ki), f->K);
                    rpe += std::sqrt((a.x - bb.x) * (a.x - bb.x) + (a.y - bb.y) * (a.y - bb.y));
                    cnt += 1.0;
                }
                valid = valid && (rpe / std::max(cnt, 1.0) < 3.0);
                t->tag(TT_VALID) = valid;
            } else {
                t->landmark.inv_depth = -1.0;
            }
        }
        for (size_t k = 0; k < map->track_num(); ++k) {
            Track *t = map->get_track(k);
            if (!t->tag(TT_VALID)) t->tag(TT_TRASH) = true;
        }
    }

    void slide_window() {   // :360-368
        WallTimer sc_t(P_.times.scope[SC_SLIDE_WINDOW]);
        while (map->frame_num() > P_.config.sliding_window_size) {
            Frame *f = map->get_frame(0);
            for (auto &sf : f->subframes) map->untrack_frame(sf.get());
            marginalize_frame(P_, map.get(), 0);
        }
    }

    // (re-)integrate every subframe interval of `frame` with the current biases, one launch
    // A subframe interval is re-integrated whenever the biases it starts from have moved -- after every solve.  Nothing
    // else of the next frame enters those integrations, so they are queued (auxiliary context) as soon as this frame's
    // solve has returned and run beside the next frame's tracker; refine_subwindow then only has to integrate what the
    // speculation did not cover.  A job is reused only if frame, end time, sample count and both biases are identical
    // to what was queued: same kernel, same inputs, bitwise the same record.
    struct SpecJob {
        Frame *sf;
        size_t id, samples;
        double t;
        V3 bg, ba;
    };
    static bool same3(const V3 &a, const V3 &b) { return a.x == b.x && a.y == b.y && a.z == b.z; }
    static bool speculation_enabled() {
        static const bool off = std::getenv("XRSLAM_AMD_NO_SPECULATION") != nullptr;   // development switch
        return !off;
    }
    long spec_hits_ = 0, memo_hits_ = 0, spec_misses_ = 0;
    void speculate_subframes() {
        drain_speculation();
        if (!speculation_enabled()) return;
        Frame *frame = map->get_frame(map->frame_num() - 1);
        if (frame->subframes.empty()) return;
        std::vector<Pipeline::IntegrateJob> jobs;
        for (size_t i = 0; i < frame->subframes.size(); ++i) {
            Frame *sf = frame->subframes[i].get();
            if (sf->preintegration.data.empty()) continue;
            Frame *prev = (i == 0 ? frame : frame->subframes[i - 1].get());
            jobs.push_back({&sf->preintegration, sf->image->t, prev->motion.bg, prev->motion.ba});
            spec_.push_back({sf, sf->id, sf->preintegration.data.size(), sf->image->t, prev->motion.bg, prev->motion.ba});
        }
        spec_jobs_ = P_.aux_batch_begin(jobs, true, true);
    }
    void drain_speculation() {
        if (spec_jobs_) P_.aux_batch_collect(spec_jobs_);
        spec_jobs_ = 0;
        spec_.clear();
    }
    std::vector<SpecJob> spec_;
    size_t spec_jobs_ = 0;
    // what mirror_frame integrated for the frame it added (Jacobians and covariance included)
    size_t memo_id_ = nil(), memo_samples_ = 0;
    V3 memo_bg_, memo_ba_;

    void integrate_subframes_begin(Frame *frame) {   // finished by Pipeline::integrate_batch_end before the solve
        std::vector<double> spec_rec;
        std::vector<SpecJob> spec;
        if (spec_jobs_) {
            spec_rec = P_.aux_batch_collect(spec_jobs_);
            spec.swap(spec_);
            spec_jobs_ = 0;
        }
        std::vector<Pipeline::IntegrateJob> jobs;
        for (size_t i = 0; i < frame->subframes.size(); ++i) {
            Frame *sf = frame->subframes[i].get();
            Frame *prev = (i == 0 ? frame : frame->subframes[i - 1].get());
            const V3 &bg = prev->motion.bg, &ba = prev->motion.ba;
            const size_t ns = sf->preintegration.data.size();
            bool have = false;
            for (size_t k = 0; k < spec.size() && !have; ++k) {
                const SpecJob &j = spec[k];
                if (j.sf == sf && j.id == sf->id && j.samples == ns && j.t == sf->image->t && same3(j.bg, bg) && same3(j.ba, ba)) {
                    std::memcpy(sf->preintegration.rec, &spec_rec[(size_t)XRHIP_IMU_DIM * k], sizeof(double) * XRHIP_IMU_DIM);
                    sf->preintegration.valid = true;
                    have = true;
                    spec_hits_++;
                }
            }
            // the frame mirror_frame has just added was integrated there, from the same samples at the same biases
            if (!have && speculation_enabled() && sf->id == memo_id_ && ns == memo_samples_ && ns > 0 && sf->preintegration.valid &&
                same3(memo_bg_, bg) && same3(memo_ba_, ba)) {
                have = true;
                memo_hits_++;
            }
            if (!have) {
                jobs.push_back({&sf->preintegration, sf->image->t, bg, ba});
                spec_misses_++;
            }
        }
        P_.integrate_batch_begin(jobs, true, true);
    }

    void refine_subwindow() {   // :370-465
        WallTimer sc_t(P_.times.scope[SC_REFINE_SUBWINDOW]);
        BaBuilder b(P_);
        if (!refine_subwindow_assemble(b)) return;
        b.solve();
        refine_subwindow_finish(b);
    }
    void refine_subwindow_finish(BaBuilder &b) {
        chained_solve_done(b);
        Frame *frame = map->get_frame(map->frame_num() - 1);
        frame->tag(FT_FIX_POSE) = false;
        frame->tag(FT_FIX_MOTION) = false;
    }
    // everything of :370-465 up to the solve: the subframe merging of the rotation-only branch, the re-integrations, the factors.
    // false: no subframe, nothing to solve
    bool refine_subwindow_assemble(BaBuilder &b) {
        Frame *frame = map->get_frame(map->frame_num() - 1);
        if (frame->subframes.empty()) return false;
        if (frame->subframes[0]->tag(FT_NO_TRANSLATION)) {
            if (frame->subframes.size() >= 9) {
                for (size_t i = frame->subframes.size() / 3; i > 0; --i) {
                    Frame *tgt = frame->subframes[i * 3 - 1].get();
                    std::vector<ImuData> imu;
                    for (size_t j = i * 3 - 1; j > (i - 1) * 3; --j) {
                        Frame *src = frame->subframes[j - 1].get();
                        imu.insert(imu.begin(), src->preintegration.data.begin(), src->preintegration.data.end());
                        map->untrack_frame(src);
                        frame->subframes.erase(frame->subframes.begin() + (j - 1));
                    }
                    tgt->preintegration.data.insert(tgt->preintegration.data.begin(), imu.begin(), imu.end());
                }
            }
            xrhip::HostProfScope hp_sa(20, "refine_subwindow: assembly");
            frame->tag(FT_FIX_POSE) = true;
            frame->tag(FT_FIX_MOTION) = true;
            b.add_frame_states(frame);
            {
                xrhip::HostProfScope hp_i(21, "refine_subwindow: integrate_subframes_begin");
                integrate_subframes_begin(frame);
            }
            std::optional<xrhip::HostProfScope> hp_l(std::in_place, 22, "refine_subwindow: factor loops");
            for (size_t i = 0; i < frame->subframes.size(); ++i) {
                Frame *sf = frame->subframes[i].get();
                b.add_frame_states(sf);
                Frame *prev = (i == 0 ? frame : frame->subframes[i - 1].get());
                b.add_preintegration_error(prev, sf, sf->preintegration);
            }
            Frame *last = frame->subframes.back().get();
            for (size_t k = 0; k < last->keypoint_num(); ++k) {
                if (Track *t = last->get_track(k)) {
                    if (t->tag(TT_VALID)) {
                        if (t->tag(TT_TRIANGULATED)) {
                            if (t->tag(TT_STATIC)) b.add_reprojection_prior(last, k);
                        } else {
                            b.add_rotation_prior(last, k);
                        }
                    }
                }
            }
            hp_l.reset();
            {
                xrhip::HostProfScope hp_w(23, "refine_subwindow: batch_end wait");
                P_.integrate_batch_end();
            }
            chain_mirror_hint(b, frame->subframes.back().get());
        } else {
            xrhip::HostProfScope hp_sa(20, "refine_subwindow: assembly");
            frame->tag(FT_FIX_POSE) = true;
            frame->tag(FT_FIX_MOTION) = true;
            b.add_frame_states(frame);
            integrate_subframes_begin(frame);
            for (size_t i = 0; i < frame->subframes.size(); ++i) {
                Frame *sf = frame->subframes[i].get();
                b.add_frame_states(sf);
                Frame *prev = (i == 0 ? frame : frame->subframes[i - 1].get());
                b.add_preintegration_error(prev, sf, sf->preintegration);
                for (size_t k = 0; k < sf->keypoint_num(); ++k) {
                    if (Track *t = sf->get_track(k)) {
                        if (t->all_tagged({TT_VALID, TT_TRIANGULATED, TT_STATIC})) {
                            if (t->first_frame()->tag(FT_KEYFRAME)) b.add_reprojection_prior(sf, k);
                            // else-branch of the reference (:453-455) indexes the keyframe's factor array with the
                            // subframe's keypoint index -- a latent bug; never taken on the
                            // supported streams and skipped here.
                        }
                    }
                }
            }
            P_.integrate_batch_end();
            chain_mirror_hint(b, frame->subframes.back().get());
        }
        return true;
    }

    LatestState get_latest_state() const {
        const Frame *f = map->get_frame(map->frame_num() - 1);
        if (!f->subframes.empty()) f = f->subframes.back().get();
        return {f->image->t, f->pose, f->motion};
    }

    Pipeline &P_;
    std::unique_ptr<Map> map;
};

// ------------------------------------------------------------------------------------ initialiser
// Initializer (core/initializer.cpp:22-571): keyframe mirroring (:22-76), two-view + PnP + BA structure from
// motion (:158-384), gyroscope bias / gravity / scale / velocity alignment (:386-571) and the final
// visual-inertial bundle adjustment (:78-140).  When initial states were supplied from outside
// (XRSLAMAmdSetInitialState) they replace SfM + alignment; everything else is the same code.
struct InitialState {
    double t;
    PoseState pose;
    MotionState motion;
};

class Initializer {
  public:
    explicit Initializer(Pipeline &P) : P_(P) {}
    std::vector<InitialState> states;   // sorted by time

    void mirror_keyframe_map(Map *ft_map, size_t init_frame_id) {
        const Config &c = P_.config;
        size_t last = ft_map->frame_index_by_id(init_frame_id);
        size_t gap = c.initializer_keyframe_gap, dist = gap * (c.initializer_keyframe_num - 1);
        map.reset();
        if (last == nil() || last < dist) return;
        size_t first = last - dist;
        std::vector<size_t> idx;
        for (size_t i = 0; i < c.initializer_keyframe_num; ++i) idx.push_back(first + i * gap);
        map = std::make_unique<Map>(&P_.ids);
        for (size_t i : idx) map->attach_frame(ft_map->get_frame(i)->clone());
        for (size_t j = 1; j < map->frame_num(); ++j) {
            Frame *oi = ft_map->get_frame(idx[j - 1]), *oj = ft_map->get_frame(idx[j]);
            Frame *ni = map->get_frame(j - 1), *nj = map->get_frame(j);
            for (size_t ki = 0; ki < oi->keypoint_num(); ++ki)
                if (Track *t = oi->get_track(ki)) {
                    size_t kj = t->get_keypoint_index(oj);
                    if (kj != nil()) ni->get_track(ki, nullptr)->add_keypoint(nj, kj);
                }
            nj->preintegration.data.clear();
            for (size_t f = idx[j - 1]; f < idx[j]; ++f) {
                const auto &od = ft_map->get_frame(f + 1)->preintegration.data;
                nj->preintegration.data.insert(nj->preintegration.data.end(), od.begin(), od.end());
            }
        }
        if (P_.init_log.enabled()) {   // what the pick looked at (the tracking map's frames) and what it made of it
            InitLogger::Line ln(P_.init_log, "keyframes");
            ln.put("init_frame_id", (double)init_frame_id);
            ln.put("keyframe_num", (double)c.initializer_keyframe_num);
            ln.put("keyframe_gap", (double)c.initializer_keyframe_gap);
            std::vector<double> ids, ns, t0, t1, pid, pns, pt0, pt1, links;
            for (size_t i = 0; i < ft_map->frame_num(); ++i) {
                const Frame *f = ft_map->get_frame(i);
                ids.push_back((double)f->id);
                ns.push_back((double)f->preintegration.data.size());
                t0.push_back(f->preintegration.data.empty() ? -1.0 : f->preintegration.data.front().t);
                t1.push_back(f->preintegration.data.empty() ? -1.0 : f->preintegration.data.back().t);
            }
            for (size_t j = 0; j < map->frame_num(); ++j) {
                const Frame *f = map->get_frame(j);
                pid.push_back((double)f->id);
                pns.push_back((double)f->preintegration.data.size());
                pt0.push_back(f->preintegration.data.empty() ? -1.0 : f->preintegration.data.front().t);
                pt1.push_back(f->preintegration.data.empty() ? -1.0 : f->preintegration.data.back().t);
                size_t nl = 0;
                if (j + 1 < map->frame_num())
                    for (size_t k = 0; k < f->keypoint_num(); ++k)
                        if (Track *t = f->get_track(k)) nl += t->has_keypoint(map->get_frame(j + 1)) ? 1 : 0;
                links.push_back((double)nl);
            }
            ln.put("ft_ids", ids); ln.put("ft_samples", ns); ln.put("ft_t0", t0); ln.put("ft_t1", t1);
            ln.put("picked", pid); ln.put("picked_samples", pns); ln.put("picked_t0", pt0); ln.put("picked_t1", pt1);
            ln.put("links_to_next", links);
        }
    }

    const InitialState *lookup(double t) const {
        const InitialState *best = nullptr;
        for (const InitialState &s : states)
            if (std::fabs(s.t - t) < 1e-4 && (!best || std::fabs(s.t - t) < std::fabs(best->t - t))) best = &s;
        return best;
    }

    std::unique_ptr<SlidingWindowTracker> initialize() {
        if (!map) return nullptr;
        if (!states.empty()) {
            if (!apply_external_states()) return nullptr;
        } else {
            WallTimer sc_t(P_.times.scope[SC_INITIALIZE]);
            attempts++;
            const bool verbose = std::getenv("XRSLAM_AMD_DEBUG_INIT") != nullptr;
            const bool ok_sfm = init_sfm();
            if (verbose)
                std::fprintf(stderr, "[init] sfm %d: hypothesis %d, %zu triangulated, %zu tracks\n", (int)ok_sfm,
                             sfm_candidate, sfm_triangulated, map->track_num());
            if (!ok_sfm) return nullptr;
            const bool ok_imu = init_imu();
            if (verbose)
                std::fprintf(stderr, "[init] imu %d: scale %g gravity %g %g %g bg %g %g %g\n", (int)ok_imu, scale, gravity.x,
                             gravity.y, gravity.z, bg.x, bg.y, bg.z);
            if (!ok_imu) return nullptr;
            successes++;
        }
        map->get_frame(0)->tag(FT_FIX_POSE) = true;
        BaBuilder b(P_);
        for (size_t i = 0; i < map->frame_num(); ++i) b.add_frame_states(map->get_frame(i));
        add_valid_tracks(b);
        add_reprojection_factors(b);
        for (size_t j = 1; j < map->frame_num(); ++j) {
            Frame *fi = map->get_frame(j - 1), *fj = map->get_frame(j);
            if (P_.integrate(fj->preintegration, fj->image->t, fi->motion.bg, fi->motion.ba, true, true))
                b.add_preintegration_error(fi, fj, fj->preintegration);
        }
        b.solve();
        for (size_t i = 0; i < map->frame_num(); ++i) map->get_frame(i)->tag(FT_KEYFRAME) = true;
        auto swt = std::make_unique<SlidingWindowTracker>(P_, std::move(map));
        return swt;
    }

    // quantities of the last IMU alignment (also read by tests)
    V3 bg, ba, gravity;
    double scale = 1;
    std::vector<V3> velocities;
    long attempts = 0, successes = 0;   // initialise() calls that reached SfM / that produced a window
    int sfm_candidate = -1;             // which of the 8 (R, T) hypotheses won the triangulation vote
    size_t sfm_triangulated = 0;

  private:
    static constexpr int kRansacSeed = 648;   // Config::random() (config.cpp:66)
    static constexpr double kGravityNominal = 9.80665;

    bool apply_external_states() {
        for (size_t i = 0; i < map->frame_num(); ++i) {
            Frame *f = map->get_frame(i);
            const InitialState *s = lookup(f->image->t);
            if (!s) return false;
            f->pose = s->pose;
            f->motion = s->motion;
        }
        return retriangulate() >= P_.config.initializer_min_landmarks;
    }

    // every track is re-triangulated from the current poses; the ones that fail become invalid (:548-568)
    size_t retriangulate() {
        size_t ok = 0;
        for (size_t k = 0; k < map->track_num(); ++k) {
            Track *t = map->get_track(k);
            if (auto p = t->triangulate()) {
                t->set_landmark_point(p.value());
                t->tag(TT_VALID) = true;
                t->tag(TT_TRIANGULATED) = true;
                ok++;
            } else {
                t->tag(TT_VALID) = false;
            }
        }
        return ok;
    }

    void add_valid_tracks(BaBuilder &b) {
        std::unordered_set<Track *> visited;
        for (size_t i = 0; i < map->frame_num(); ++i) {
            Frame *f = map->get_frame(i);
            for (size_t j = 0; j < f->keypoint_num(); ++j) {
                Track *t = f->get_track(j);
                if (!t || !t->tag(TT_VALID) || visited.count(t)) continue;
                visited.insert(t);
                b.add_track_states(t);
            }
        }
    }
    void add_reprojection_factors(BaBuilder &b) {
        for (size_t i = 0; i < map->frame_num(); ++i) {
            Frame *f = map->get_frame(i);
            for (size_t j = 0; j < f->keypoint_num(); ++j) {
                Track *t = f->get_track(j);
                if (!t || !t->all_tagged({TT_VALID, TT_TRIANGULATED})) continue;
                if (f == t->first_frame()) continue;
                b.add_reprojection_error(f, j);
            }
        }
    }

    // ---------------------------------------------------------------- structure from motion (:158-384)
    bool init_sfm() {
        const Config &c = P_.config;
        Frame *fi = map->get_frame(0), *fj = map->get_frame(map->frame_num() - 1);
        std::vector<V2> pi, pj;
        std::vector<std::pair<size_t, size_t>> matches;
        double parallax = 0;
        for (size_t ki = 0; ki < fi->keypoint_num(); ++ki) {
            Track *t = fi->get_track(ki);
            if (!t) continue;
            size_t kj = t->get_keypoint_index(fj);
            if (kj == nil()) continue;
            const V3 &a = fi->get_keypoint(ki), &bq = fj->get_keypoint(kj);
            pi.push_back({a.x / a.z, a.y / a.z});
            pj.push_back({bq.x / bq.z, bq.y / bq.z});
            matches.emplace_back(ki, kj);
            V2 ua = apply_k(a, fi->K), ub = apply_k(bq, fj->K);
            parallax += std::hypot(ua.x - ub.x, ua.y - ub.y);
        }
        const int common = (int)matches.size();
        if (common < (int)c.initializer_min_matches) return false;
        parallax /= std::max(common, 1);
        if (parallax < c.initializer_min_parallax) return false;

        // eight (R, T) hypotheses: two homography decompositions and the essential twisted pair, each with +-T
        std::vector<M3> Rs;
        std::vector<V3> Ts;
        std::vector<char> mask;
        M3 RH1, RH2, RE1, RE2;
        V3 TH1, TH2, nH1, nH2, TE;
        M3 H = find_homography_matrix(pi, pj, mask, 0.7 / fi->K.fx, 0.999, 1000, kRansacSeed);
        if (!decompose_homography(H, RH1, RH2, TH1, TH2, nH1, nH2)) return false;   // pure rotation
        TH1 = normalized(TH1);
        TH2 = normalized(TH2);
        Rs.insert(Rs.end(), {RH1, RH1, RH2, RH2});
        Ts.insert(Ts.end(), {TH1, -TH1, TH2, -TH2});
        M3 E = find_essential_matrix(pi, pj, mask, 0.7 / fi->K.fx, 0.999, 1000, kRansacSeed);
        decompose_essential(E, RE1, RE2, TE);
        TE = normalized(TE);
        Rs.insert(Rs.end(), {RE1, RE1, RE2, RE2});
        Ts.insert(Ts.end(), {TE, -TE, TE, -TE});

        const size_t n = pi.size();
        std::vector<std::vector<V3>> pts(Rs.size());
        std::vector<std::vector<char>> status(Rs.size());
        std::vector<size_t> counts(Rs.size(), 0);
        std::vector<double> scores(Rs.size(), 0.0);
        size_t best = 0;
        for (size_t h = 0; h < Rs.size(); ++h) {
            pts[h].resize(n);
            status[h].assign(n, 0);
            P34 P1{{1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0}}, P2;
            for (int r = 0; r < 3; ++r) {
                for (int cc = 0; cc < 3; ++cc) P2.m[4 * r + cc] = Rs[h](r, cc);
                P2.m[4 * r + 3] = Ts[h][r];
            }
            for (size_t k = 0; k < n; ++k) {
                auto q = triangulate_point(P1, P2, V3{pi[k].x, pi[k].y, 1.0}, V3{pj[k].x, pj[k].y, 1.0});
                V3 q1{q[0], q[1], q[2]};
                V3 q2 = Rs[h] * q1 + Ts[h] * q[3];
                if (q1.z * q[3] > 0 && q2.z * q[3] > 0 && q1.z / q[3] < 100 && q2.z / q[3] < 100) {
                    pts[h][k] = q1 / q[3];
                    status[h][k] = 1;
                    counts[h]++;
                    const double ax = q1.x / q1.z - pi[k].x, ay = q1.y / q1.z - pi[k].y;
                    const double bx = q2.x / q2.z - pj[k].x, by = q2.y / q2.z - pj[k].y;
                    scores[h] += 0.5 * ((ax * ax + ay * ay) + (bx * bx + by * by));
                }
            }
            // the vote as the reference casts it (:254-260): a better score wins among well-triangulated
            // hypotheses, otherwise more points win
            if (counts[h] > c.initializer_min_triangulation && scores[h] < scores[best]) best = h;
            else if (counts[h] > counts[best]) best = h;
        }
        sfm_candidate = (int)best;
        sfm_triangulated = counts[best];
        if (P_.init_log.enabled()) {   // the two-view stage: matches, models, the eight hypotheses, the vote
            InitLogger::Line ln(P_.init_log, "sfm_vote");
            std::vector<double> a, b2, rs, ts, cn, sc, pb, sb;
            for (size_t k = 0; k < n; ++k) {
                a.push_back(pi[k].x); a.push_back(pi[k].y);
                b2.push_back(pj[k].x); b2.push_back(pj[k].y);
            }
            for (size_t h = 0; h < Rs.size(); ++h) {
                rs.insert(rs.end(), Rs[h].m, Rs[h].m + 9);
                for (int r = 0; r < 3; ++r) ts.push_back(Ts[h][r]);
                cn.push_back((double)counts[h]);
                sc.push_back(scores[h]);
            }
            for (size_t k = 0; k < n; ++k) {
                sb.push_back(status[best][k] ? 1.0 : 0.0);
                for (int r = 0; r < 3; ++r) pb.push_back(status[best][k] ? pts[best][k][r] : 0.0);
            }
            ln.put("pi", a); ln.put("pj", b2);
            ln.put("parallax", parallax);
            ln.put("fx", fi->K.fx);
            ln.put("min_matches", (double)c.initializer_min_matches);
            ln.put("min_parallax", c.initializer_min_parallax);
            ln.put("min_triangulation", (double)c.initializer_min_triangulation);
            ln.put("H", H.m, 9); ln.put("E", E.m, 9);
            const double nn[6] = {nH1.x, nH1.y, nH1.z, nH2.x, nH2.y, nH2.z};
            ln.put("nH", nn, 6);
            ln.put("Rs", rs); ln.put("Ts", ts); ln.put("counts", cn); ln.put("scores", sc);
            ln.put("best", (double)best);
            ln.put("points_best", pb); ln.put("status_best", sb);
        }
        if (std::getenv("XRSLAM_AMD_DEBUG_INIT")) {   // development aid: the vote over the eight (R, T) hypotheses
            std::fprintf(stderr, "[init] %d matches, parallax %.1f px; (count, sum of errors) per hypothesis:", common, parallax);
            for (size_t h = 0; h < Rs.size(); ++h) std::fprintf(stderr, " %zu:(%zu, %.3g)", h, counts[h], scores[h]);
            std::fprintf(stderr, " -> %zu\n", best);
        }
        if (counts[best] < c.initializer_min_triangulation) return false;

        PoseState pose;
        pose.q = Quat{0, 0, 0, 1};
        pose.p = V3{0, 0, 0};
        fi->set_pose(fi->camera, pose);
        M3 Rt = transpose(Rs[best]);
        pose.q = quat_from_matrix(Rt);
        pose.p = -(Rt * Ts[best]);
        fj->set_pose(fj->camera, pose);
        for (size_t k = 0; k < n; ++k) {
            if (!status[best][k]) continue;
            Track *t = fi->get_track(matches[k].first);
            t->set_landmark_point(pts[best][k]);
            t->tag(TT_VALID) = true;
            t->tag(TT_TRIANGULATED) = true;
        }

        // the frames in between: pose-only adjustment against the two-view landmarks, seeded with the previous pose
        for (size_t j = 1; j + 1 < map->frame_num(); ++j) {
            Frame *prev = map->get_frame(j - 1), *cur = map->get_frame(j);
            cur->set_pose(cur->camera, prev->get_pose(prev->camera));
            BaBuilder b(P_);
            b.add_frame_states(cur);
            for (size_t k = 0; k < cur->keypoint_num(); ++k) {
                Track *t = cur->get_track(k);
                if (!t || !t->has_keypoint(map->get_frame(0))) continue;
                if (t->tag(TT_VALID) && t->tag(TT_TRIANGULATED)) b.add_reprojection_prior(cur, k);
            }
            if (b.factor_num() > 0) b.solve();
        }

        for (size_t k = 0; k < map->track_num(); ++k) {
            Track *t = map->get_track(k);
            if (t->tag(TT_VALID)) continue;
            if (auto p = t->triangulate()) {
                t->set_landmark_point(p.value());
                t->tag(TT_VALID) = true;
                t->tag(TT_TRIANGULATED) = true;
            }
        }

        map->get_frame(0)->tag(FT_FIX_POSE) = true;
        BaBuilder b(P_);
        for (size_t i = 0; i < map->frame_num(); ++i) b.add_frame_states(map->get_frame(i), false);
        add_valid_tracks(b);
        add_reprojection_factors(b);
        if (!b.solve()) return false;
        // landmark.reprojection_error is never written by the reference, so its "> 3.0" clause never fires (:376-380)
        map->prune_tracks([](const Track *t) { return !t->tag(TT_VALID) || t->landmark.reprojection_error > 3.0; });
        return true;
    }

    // -------------------------------------------------------------------- IMU alignment (:386-571)
    bool init_imu() {
        const bool ok = init_imu_steps();
        if (P_.init_log.enabled()) {
            InitLogger::Line ln(P_.init_log, "imu_result");
            ln.put("ok", ok ? 1.0 : 0.0);
            ln.put("scale", scale);
            ln.put("refine_imu", P_.config.initializer_refine_imu ? 1.0 : 0.0);
            ln.put("min_landmarks", (double)P_.config.initializer_min_landmarks);
            ln.put("final_points", (double)final_points_);
        }
        return ok;
    }
    bool init_imu_steps() {
        bg = ba = gravity = V3{0, 0, 0};
        scale = 1;
        final_points_ = 0;
        velocities.assign(map->frame_num(), V3{0, 0, 0});
        solve_gyro_bias();
        solve_gravity_scale_velocity();
        if (scale < 0.001 || scale > 1.0) return false;
        if (!P_.config.initializer_refine_imu) return apply_init();
        refine_scale_velocity_via_gravity();
        if (scale < 0.001 || scale > 1.0) return false;
        return apply_init();
    }
    size_t final_points_ = 0;

    void preintegrate() {
        std::vector<Pipeline::IntegrateJob> jobs;
        for (size_t j = 1; j < map->frame_num(); ++j) {
            Frame *f = map->get_frame(j);
            jobs.push_back({&f->preintegration, f->image->t, bg, ba});
        }
        P_.integrate_batch(jobs, true, false);
    }

    void solve_gyro_bias() {   // sum over intervals of |log((q_i dq)^-1 q_j) - dq_dbg bg|^2
        preintegrate();
        M3 A;
        V3 rhs{0, 0, 0};
        for (size_t j = 1; j < map->frame_num(); ++j) {
            const Frame *fi = map->get_frame(j - 1), *fj = map->get_frame(j);
            const PoseState pi = fi->get_pose(fi->imu), pj = fj->get_pose(fj->imu);
            M3 J;
            for (int k = 0; k < 9; ++k) J.m[k] = fj->preintegration.rec[11 + k];   // dq_dbg
            M3 Jt = transpose(J);
            M3 JtJ = Jt * J;
            for (int k = 0; k < 9; ++k) A.m[k] += JtJ.m[k];
            rhs = rhs + Jt * logmap((pi.q * fj->preintegration.dq()).conjugate() * pj.q);
        }
        bg = svd_solve3(A, rhs);
        if (P_.init_log.enabled()) {   // per interval: IMU poses of its two frames, dq, dq_dbg (integrated at zero biases) -> the bias
            InitLogger::Line ln(P_.init_log, "gyro_bias");
            std::vector<double> qi, qj, dq, jac;
            for (size_t j = 1; j < map->frame_num(); ++j) {
                const Frame *fi = map->get_frame(j - 1), *fj = map->get_frame(j);
                const PoseState pi = fi->get_pose(fi->imu), pj = fj->get_pose(fj->imu);
                const Quat d = fj->preintegration.dq();
                for (double v : {pi.q.x, pi.q.y, pi.q.z, pi.q.w}) qi.push_back(v);
                for (double v : {pj.q.x, pj.q.y, pj.q.z, pj.q.w}) qj.push_back(v);
                for (double v : {d.x, d.y, d.z, d.w}) dq.push_back(v);
                for (int k = 0; k < 9; ++k) jac.push_back(fj->preintegration.rec[11 + k]);
            }
            ln.put("q_i", qi); ln.put("q_j", qj); ln.put("dq", dq); ln.put("dq_dbg", jac);
            const double out[3] = {bg.x, bg.y, bg.z};
            ln.put("bg", out, 3);
        }
    }

    // rows per interval (i, j = i+1), unknowns [g | s | v_0 .. v_{N-1}] (or [dg(2) | s | v] with the gravity
    // direction restricted to its tangent plane):
    //   -dt^2/2 g + s (pc_j - pc_i) - dt v_i      = q_i dp + (q_j - q_i) p_bc
    //   -dt g               - v_i + v_j           = q_i dv
    void fill_alignment(Dense &A, std::vector<double> &rhs, bool tangent) {
        const int N = (int)map->frame_num(), g_cols = tangent ? 2 : 3, s_col = g_cols, v0 = g_cols + 1;
        A = Dense((N - 1) * 6, g_cols + 1 + 3 * N);
        rhs.assign((size_t)(N - 1) * 6, 0.0);
        V3 t1, t2;
        if (tangent) s2_tangential_basis(gravity, t1, t2);
        for (int j = 1; j < N; ++j) {
            const int i = j - 1;
            const Frame *fi = map->get_frame(i), *fj = map->get_frame(j);
            const PreInt &d = fj->preintegration;
            const double dt = d.dt();
            const PoseState ci = fi->get_pose(fi->camera), cj = fj->get_pose(fj->camera);
            const V3 dpc = cj.p - ci.p;
            V3 bp = fi->pose.q * d.dp() + (fj->pose.q * fj->camera.p_cs - fi->pose.q * fi->camera.p_cs);
            V3 bv = fi->pose.q * d.dv();
            if (tangent) {
                bp = bp + gravity * (0.5 * dt * dt);
                bv = bv + gravity * dt;
            }
            for (int r = 0; r < 3; ++r) {
                if (tangent) {
                    A(i * 6 + r, 0) = -0.5 * dt * dt * t1[r];
                    A(i * 6 + r, 1) = -0.5 * dt * dt * t2[r];
                    A(i * 6 + 3 + r, 0) = -dt * t1[r];
                    A(i * 6 + 3 + r, 1) = -dt * t2[r];
                } else {
                    A(i * 6 + r, r) = -0.5 * dt * dt;
                    A(i * 6 + 3 + r, r) = -dt;
                }
                A(i * 6 + r, s_col) = dpc[r];
                A(i * 6 + r, v0 + i * 3 + r) = -dt;
                A(i * 6 + 3 + r, v0 + i * 3 + r) = -1.0;
                A(i * 6 + 3 + r, v0 + j * 3 + r) = 1.0;
                rhs[i * 6 + r] = bp[r];
                rhs[i * 6 + 3 + r] = bv[r];
            }
        }
    }

    void solve_gravity_scale_velocity() {
        preintegrate();
        Dense A;
        std::vector<double> rhs;
        fill_alignment(A, rhs, false);
        std::vector<double> x = lstsq_qr(A, rhs);
        gravity = normalized(V3{x[0], x[1], x[2]}) * kGravityNominal;
        scale = x[3];
        for (size_t i = 0; i < map->frame_num(); ++i) velocities[i] = V3{x[4 + 3 * i], x[5 + 3 * i], x[6 + 3 * i]};
        log_alignment("gravity_scale_velocity", V3{0, 0, 0});
    }
    // XRSLAM_AMD_DUMP_INIT: what an alignment solve looked at, per interval -- dt, dp, dv (integrated at the gyroscope bias found
    // before), the camera positions and body rotations of its two frames, the camera lever arm -- and what it answered
    void log_alignment(const char *what, V3 gravity_before) {
        if (!P_.init_log.enabled()) return;
        InitLogger::Line ln(P_.init_log, what);
        std::vector<double> dt, dp, dv, ci, cj, qi, qj, vel;
        for (size_t j = 1; j < map->frame_num(); ++j) {
            const Frame *fi = map->get_frame(j - 1), *fj = map->get_frame(j);
            const PreInt &d = fj->preintegration;
            const PoseState a = fi->get_pose(fi->camera), b = fj->get_pose(fj->camera);
            dt.push_back(d.dt());
            for (int r = 0; r < 3; ++r) {
                dp.push_back(d.dp()[r]);
                dv.push_back(d.dv()[r]);
                ci.push_back(a.p[r]);
                cj.push_back(b.p[r]);
            }
            for (double v : {fi->pose.q.x, fi->pose.q.y, fi->pose.q.z, fi->pose.q.w}) qi.push_back(v);
            for (double v : {fj->pose.q.x, fj->pose.q.y, fj->pose.q.z, fj->pose.q.w}) qj.push_back(v);
        }
        for (const V3 &v : velocities)
            for (int r = 0; r < 3; ++r) vel.push_back(v[r]);
        const Frame *f0 = map->get_frame(0);
        const double pcs[3] = {f0->camera.p_cs.x, f0->camera.p_cs.y, f0->camera.p_cs.z};
        const double gb[3] = {gravity_before.x, gravity_before.y, gravity_before.z}, ga[3] = {gravity.x, gravity.y, gravity.z};
        ln.put("dt", dt); ln.put("dp", dp); ln.put("dv", dv); ln.put("cam_p_i", ci); ln.put("cam_p_j", cj);
        ln.put("body_q_i", qi); ln.put("body_q_j", qj); ln.put("p_cs", pcs, 3);
        ln.put("gravity_before", gb, 3); ln.put("gravity", ga, 3); ln.put("scale", scale); ln.put("velocities", vel);
    }

    void refine_scale_velocity_via_gravity() {   // one damped step of the gravity direction on the sphere |g| = 9.80665
        const double damp = 0.1;
        preintegrate();
        Dense A;
        std::vector<double> rhs;
        fill_alignment(A, rhs, true);
        std::vector<double> x = lstsq_qr(A, rhs);
        V3 t1, t2;
        s2_tangential_basis(gravity, t1, t2);
        const V3 gravity_before = gravity;
        gravity = normalized(gravity + (t1 * x[0] + t2 * x[1]) * damp) * kGravityNominal;
        scale = x[2];
        for (size_t i = 0; i < map->frame_num(); ++i) velocities[i] = V3{x[3 + 3 * i], x[4 + 3 * i], x[5 + 3 * i]};
        log_alignment("refine_via_gravity", gravity_before);
    }

    bool apply_init() {   // rotate gravity onto -z, apply the metric scale, set velocities and the gyroscope bias
        Quat q = quat_from_two_vectors(gravity, V3{0, 0, -kGravityNominal});
        std::vector<double> before, after, vel;
        for (size_t i = 0; i < map->frame_num(); ++i) {
            Frame *f = map->get_frame(i);
            PoseState ip = f->get_pose(f->imu);
            for (double v : {ip.q.x, ip.q.y, ip.q.z, ip.q.w, ip.p.x, ip.p.y, ip.p.z}) before.push_back(v);
            ip.q = q * ip.q;
            ip.p = (q * ip.p) * scale;
            f->set_pose(f->imu, ip);
            f->motion.v = q * velocities[i];
            f->motion.bg = bg;
            f->motion.ba = ba;
            const PoseState np = f->get_pose(f->imu);
            for (double v : {np.q.x, np.q.y, np.q.z, np.q.w, np.p.x, np.p.y, np.p.z}) after.push_back(v);
            for (int r = 0; r < 3; ++r) vel.push_back(f->motion.v[r]);
        }
        final_points_ = retriangulate();
        if (P_.init_log.enabled()) {
            InitLogger::Line ln(P_.init_log, "apply_init");
            const double g[3] = {gravity.x, gravity.y, gravity.z};
            std::vector<double> vin;
            for (const V3 &v : velocities)
                for (int r = 0; r < 3; ++r) vin.push_back(v[r]);
            ln.put("gravity", g, 3); ln.put("scale", scale); ln.put("velocities", vin);
            ln.put("imu_pose_before", before); ln.put("imu_pose_after", after); ln.put("v_after", vel);
        }
        return final_points_ >= P_.config.initializer_min_landmarks;
    }

  public:
    Pipeline &P_;
    std::unique_ptr<Map> map;
};

// ------------------------------------------------------------------------------------ the system
enum SysState { SYS_INITIALIZING = 0, SYS_TRACKING, SYS_CRASH, SYS_UNKNOWN };

enum Threading { THREADING_OFF = 0, THREADING_PIPELINED = 1 };

class System {   // XRSLAM::Detail + FeatureTracker + FrontendWorker: inline, or the backend on a thread (set_threading)
  public:
    explicit System(const Config &c) : P(c), ft_map(std::make_unique<Map>(&P.ids)), init(P) {
        if (const char *e = std::getenv("XRSLAM_AMD_THREADING"))
            if (!std::strcmp(e, "1") || !std::strcmp(e, "pipelined")) set_threading(THREADING_PIPELINED);
    }
    ~System() {
        try {
            sync();
        } catch (...) {
        }
        worker.reset();
        try {   // the launches of the last marginalisation read the prior's arrays, which die with `swt` (before `P`)
            P.marg_launch_wait();
        } catch (...) {
        }
    }

    // -------- threading.  THREADING_OFF is the PC build of the reference: the feature tracker calls the backend inline.
    // THREADING_PIPELINED is its XRSLAM_ENABLE_THREADING build with fixed hand-off points, so that a run is reproducible:
    // the backend (SlidingWindowTracker::track) of frame t runs on `worker` while this thread tracks the features of
    // frame t+1; the feature tracker of frame t+1 therefore sees the state the backend published for frame t-1 (one frame
    // older than inline) and propagates it over the frames in between exactly as FeatureTracker::work does whenever the
    // backend lags (feature_tracker.cpp:44-66).  Hand-offs: at the end of its frame t the feature tracker waits for the job
    // of frame t-1, publishes its state (FrontendWorker's latest_state), gathers what mirror_frame reads from the tracking map
    // into a packet and posts the job of frame t = mirror_frame (from the packet) + track.  mirror_frame is the only step that
    // touches both maps (the reference holds the tracking map's lock for it): what it may still write there -- a track's
    // trash tag when it changes -- the feature tracker of frame t+1 does not look at before the backend has flagged it done.
    // Otherwise the two threads share nothing: the feature tracker owns `ft_map`, the KLT context and `P.ba_ft`; the
    // backend owns the window map and the other BA contexts.  Poses lag one frame more than inline.
    // RD-VIO's update_track_status reads the tracking map from inside the backend (:741-788): with parsac_flag the
    // frames stay inline.
    void set_threading(int mode) {
        publish_backend_state();   // a mode switch is a deterministic point of the caller's sequence: nothing is lost across it
        threading = mode == THREADING_PIPELINED ? THREADING_PIPELINED : THREADING_OFF;
        if (threading == THREADING_PIPELINED) {
            P.ensure_ft_context();
            if (!worker) {
                int dev = -1;
                if (xrhip_get_device(&dev) != 0) dev = -1;
                worker = std::make_unique<JobThread>(dev);
            }
        }
    }
    bool pipelined() const { return threading == THREADING_PIPELINED && !P.config.parsac_flag; }
    // The backend job in flight (if any) has finished when this returns.  It only WAITS: the job's state stays pending and is
    // published (FrontendWorker's latest_state) at the deterministic hand-off of the next frame (frontend_work) or at a mode
    // switch -- never by a getter / Flush between two frames, so the trajectory does not depend on which statistics a caller
    // reads in between (the feature tracker of frame t+1 always starts from the state of frame t-1).
    void sync() {
        if (inflight_id_ == nil() || inflight_joined_) return;
        inflight_joined_ = true;   // set first: if the job threw, wait() rethrows and the job must not be waited for twice
        WallTimer wt(P.times.w_join);
        try {
            worker->wait();
        } catch (...) {
            backend_failed_ = true;   // the window map may be half way through a frame: recover_after_error() goes back to the initialiser
            throw;
        }
    }
    // The unwind of the C API's catch-all (xrslam_api.cpp: guarded()).  An exception has left the pipeline somewhere inside a frame:
    //  * pipelined mode: the backend's contexts (P.ba, P.ba_aux) belong to its thread while a job runs -- JOIN it before anything
    //    of its is touched (an error on the API thread, e.g. an unsupported image, must not race the running job);
    //  * no pre-integration batch stays between begin and end on any context (device side), and nothing on the host still believes
    //    one is (prepared / chained / queued / speculated batches, Pipeline::pending_pre_): the frames that follow are not refused;
    //  * if the error came out of the sliding-window tracker (mirror_frame / track, inline or as the backend job), its map may have
    //    taken half a frame: FrontendWorker::work's failure branch (core/frontend_worker.cpp:75-81) -- the tracker is dropped, a fresh
    //    initialiser takes over, the published state is cleared.  Externally supplied initial states are kept.
    // tests/test_error_recovery.py injects failures through the CPU shim and requires the stream to go on tracking.
    // Joining / leaving a group mid-sequence: nothing of this instance stays on the device across it.
    void resolve_device_work() {
        sync();
        if (swt && swt->map && swt->map->marginalization_factor) resolve_marginalization(P, swt->map->marginalization_factor.get());
    }
    void recover_after_error() noexcept {
        if (inflight_id_ != nil() && !inflight_joined_) {
            try {
                sync();
            } catch (...) {
            }
        }
        try {
            P.marg_launch_wait();
        } catch (...) {
            backend_failed_ = true;   // a prior whose marginalisation was never launched cannot be resolved
        }
        P.cancel_integrations();
        P.pending_pre_.clear();
        if (swt) swt->forget_queued_work();
        mirror_done_.store(true, std::memory_order_release);
        if (backend_failed_) {
            backend_failed_ = false;
            inflight_id_ = nil();
            inflight_joined_ = false;
            inflight_ok_ = false;
            last_mirrored_id_ = nil();
            frontend_latest_state = {0.0, nil(), PoseState{}, MotionState{}};
            if (swt && swt->map && swt->map->marginalization_factor) {   // a marginalisation still on the device: drain its context
                try {
                    resolve_marginalization(P, swt->map->marginalization_factor.get());
                } catch (...) {
                }
            }
            swt.reset();
            init.map.reset();
        }
    }
    // marks the region in which an exception means "the window map is suspect" (inline mode; the pipelined job reports through sync())
    struct BackendScope {
        System &s;
        int pending = std::uncaught_exceptions();
        explicit BackendScope(System &sys) : s(sys) {}
        ~BackendScope() {
            if (std::uncaught_exceptions() > pending) s.backend_failed_ = true;
        }
    };
    bool backend_failed_ = false;
    // join + publish: the hand-off proper
    void publish_backend_state() {
        if (inflight_id_ == nil()) return;
        const size_t id = inflight_id_;
        struct Clear {
            System &s;
            ~Clear() {
                s.inflight_id_ = nil();
                s.inflight_joined_ = false;
            }
        } clear{*this};
        sync();
        if (inflight_ok_) {
            auto [t, pose, motion] = swt->get_latest_state();
            frontend_latest_state = {t, id, pose, motion};
        }
    }

    // -------- Detail (core/detail.cpp:46-177)
    // want_pose: Detail::track_gyroscope / track_accelerometer return the propagated pose (detail.cpp:46-100), which
    // XRSLAMPushSensorData throws away for both sensors (XRSLAMManager.cpp:139-146): the C API passes false and the propagation over the
    // samples since the last state -- an exponential map per sample, on every one of ~20 pushes per frame -- is not run.  (Its only
    // side effect, dropping samples older than the state, happens in the next propagation: track_camera's.)
    PoseState track_gyroscope(double t, double x, double y, double z, bool want_pose = true) {
        if (!accelerometers.empty()) {
            if (t < accelerometers.front().t) {
                gyroscopes.clear();
            } else {
                while (!accelerometers.empty() && t >= accelerometers.front().t) {
                    const auto acc = accelerometers.front();
                    double lambda = (acc.t - gyroscopes[0].t) / (t - gyroscopes[0].t);
                    V3 w = gyroscopes[0].w + lambda * (V3{x, y, z} - gyroscopes[0].w);
                    track_imu({acc.t, w, acc.a});
                    accelerometers.pop_front();
                }
                if (!accelerometers.empty())
                    while (!gyroscopes.empty() && gyroscopes.front().t < t) gyroscopes.pop_front();
            }
        }
        gyroscopes.push_back({t, {x, y, z}});
        return want_pose ? predict_pose(t) : PoseState{};
    }
    PoseState track_accelerometer(double t, double x, double y, double z, bool want_pose = true) {
        if (!gyroscopes.empty() && t >= gyroscopes.front().t) {
            if (t > gyroscopes.back().t) {
                while (gyroscopes.size() > 1) gyroscopes.pop_front();
                accelerometers.push_back({t, {x, y, z}});
            } else if (t == gyroscopes.back().t) {
                while (gyroscopes.size() > 1) gyroscopes.pop_front();
                track_imu({t, gyroscopes.front().w, {x, y, z}});
            } else {
                while (t >= gyroscopes[1].t) gyroscopes.pop_front();
                double lambda = (t - gyroscopes[0].t) / (gyroscopes[1].t - gyroscopes[0].t);
                V3 w = gyroscopes[0].w + lambda * (gyroscopes[1].w - gyroscopes[0].w);
                track_imu({t, w, {x, y, z}});
            }
        }
        return want_pose ? predict_pose(t) : PoseState{};
    }
    PoseState track_camera(std::shared_ptr<HipImage> image) {
        const Config &c = P.config;
        auto f = std::make_unique<Frame>();
        f->id = ++P.ids.frame;
        f->K = c.K;
        f->image = image;
        f->sqrt_inv_cov[0] = c.K.fx / std::sqrt(c.keypoint_noise_cov[0]);
        f->sqrt_inv_cov[1] = c.K.fy / std::sqrt(c.keypoint_noise_cov[3]);
        f->camera = {c.q_bc, c.p_bc};
        f->imu = {c.q_bi, c.p_bi};
        frames.emplace_back(std::move(f));
        PoseState out = predict_pose(image->t);
        if (P.sync_log.enabled()) {
            std::fprintf(P.sync_log.fp, "{\"pose_t\": %.17g, \"q\": [%.17g, %.17g, %.17g, %.17g], \"p\": [%.17g, %.17g, %.17g]", image->t, out.q.x,
                         out.q.y, out.q.z, out.q.w, out.p.x, out.p.y, out.p.z);
            if (ft_latest_state) {
                const auto &[st, sp, sm] = ft_latest_state.value();
                std::fprintf(P.sync_log.fp,
                             ", \"state_t\": %.17g, \"sq\": [%.17g, %.17g, %.17g, %.17g], \"sp\": [%.17g, %.17g, %.17g], \"sv\": [%.17g, %.17g, %.17g], "
                             "\"sbg\": [%.17g, %.17g, %.17g], \"sba\": [%.17g, %.17g, %.17g]",
                             st, sp.q.x, sp.q.y, sp.q.z, sp.q.w, sp.p.x, sp.p.y, sp.p.z, sm.v.x, sm.v.y, sm.v.z, sm.bg.x, sm.bg.y, sm.bg.z,
                             sm.ba.x, sm.ba.y, sm.ba.z);
            }
            std::fprintf(P.sync_log.fp, "}\n");
            std::fflush(P.sync_log.fp);
        }
        if (image->t > latest_timestamp) {
            latest_pose = out;
            latest_timestamp = image->t;
        }
        return out;
    }
    void track_imu(const ImuData &imu) {
        frontal_imus.push_back(imu);
        imus.push_back(imu);
        while (!imus.empty() && !frames.empty()) {
            if (imus.front().t <= frames.front()->image->t) {
                frames.front()->preintegration.data.push_back(imus.front());
                imus.pop_front();
            } else {
                std::unique_ptr<Frame> f = std::move(frames.front());
                frames.pop_front();
                feature_tracker_work(std::move(f));
            }
        }
    }
    PoseState predict_pose(double t) {
        PoseState out;
        if (ft_latest_state) {
            auto [st, sp, sm] = ft_latest_state.value();
            while (!frontal_imus.empty() && frontal_imus.front().t <= st) frontal_imus.pop_front();
            const V3 gravity{0, 0, -9.80665};
            for (const ImuData &imu : frontal_imus) {
                if (imu.t <= t) {
                    double dt = imu.t - st;
                    sp.p = sp.p + dt * sm.v + 0.5 * dt * dt * (gravity + sp.q * (imu.a - sm.ba));
                    sm.v = sm.v + dt * (gravity + sp.q * (imu.a - sm.ba));
                    sp.q = (sp.q * expmap((imu.w - sm.bg) * dt)).normalized();
                    st = imu.t;
                }
            }
            out.q = sp.q * P.config.q_bo;
            out.p = sp.p + sp.q * P.config.p_bo;
        } else {
            out.q = Quat{0, 0, 0, 0};
            out.p = V3{0, 0, 0};
        }
        return out;
    }
    SysState get_system_state() const { return swt ? SYS_TRACKING : SYS_INITIALIZING; }

    // -------- FeatureTracker::work (core/feature_tracker.cpp:24-153)
    void feature_tracker_work(std::unique_ptr<Frame> frame) {
        xrhip::HostProfScope hp_f(11, "feature_tracker_work (all)");
        WallTimer wt_frame(P.times.w_frame);
        const Config &c = P.config;
        if (P.sync_log.enabled()) {   // what Detail::track_imu attached to this frame (before the tracker adds its boundary sample)
            FILE *fp = P.sync_log.fp;
            std::fprintf(fp, "{\"frame\": %zu, \"t\": %.17g, \"imu\": [", frame->id, frame->image->t);
            for (size_t i = 0; i < frame->preintegration.data.size(); ++i) {
                const ImuData &d = frame->preintegration.data[i];
                std::fprintf(fp, "%s[%.17g, %.17g, %.17g, %.17g, %.17g, %.17g, %.17g]", i ? ", " : "", d.t, d.w.x, d.w.y, d.w.z, d.a.x,
                             d.a.y, d.a.z);
            }
            std::fprintf(fp, "]}\n");
            std::fflush(fp);
        }
        // CLAHE / pyramid / gradients of the new image depend on nothing else: their launches are issued while the
        // pre-integration of the new interval runs on the BA stream (below), or right away when there is none to wait for
        bool preprocessed = false;
        auto preprocess = [&] {
            if (preprocessed) return;
            preprocessed = true;
            if (frame->image->preprocessed) return;   // queued when the frame arrived (Pipeline::make_image)
            WallTimer wt_w_preprocess(P.times.w_preprocess);
            hip_check(xrhip_image_preprocess(frame->image->h, c.feature_tracker_clahe_clip_limit,
                                             (int)c.feature_tracker_clahe_width, (int)c.feature_tracker_clahe_height),
                      "xrhip_image_preprocess");
        };
        auto [opt_t, opt_id, opt_pose, opt_motion] = frontend_latest_state;
        (void)opt_t;
        xrhip_ba *ft_ctx = pipelined() ? P.ba_ft : nullptr;
        bool is_initialized = opt_id != nil();
        bool swt_tag = !is_initialized || frame->id % c.sliding_window_tracker_frequent == 0;
        Map *map = ft_map.get();
        static const bool no_overlap = std::getenv("XRSLAM_AMD_NO_DETECT_OVERLAP") != nullptr;   // development switch (A/B, parity)
        const bool overlap_detect = swt_tag && swt && !pipelined() && !P.swt_log.enabled() && !no_overlap;   // see below, at the detection
        std::function<void(Frame *)> deferred_tag;
        if (map->frame_num() > 0) {
            Frame *last = map->get_frame(map->frame_num() - 1);
            if (!last->preintegration.data.empty()) {
                if (frame->preintegration.data.empty() ||
                    (frame->preintegration.data.front().t - last->image->t > 1.0e-5)) {
                    ImuData imu = last->preintegration.data.back();
                    imu.t = last->image->t;
                    frame->preintegration.data.insert(frame->preintegration.data.begin(), imu);
                }
            }
            // pipelined mode: the backend queues the integration of the interval it will mirror next itself, once the
            // biases it starts from are final (SlidingWindowTracker::take_mirror_hint)
            if (swt && swt_tag && is_initialized && pipelined()) {
                if (const size_t from = map->frame_index_by_id(last_mirrored_id_); from != nil()) {
                    SlidingWindowTracker::MirrorHint h{frame->id, last_mirrored_id_, frame->image->t, frame->preintegration.data};
                    for (size_t index = map->frame_num() - 1; index > from; --index) {
                        const std::vector<ImuData> &od = map->get_frame(index)->preintegration.data;
                        h.samples.insert(h.samples.begin(), od.begin(), od.end());
                    }
                    swt->post_mirror_hint(std::move(h));
                }
            }
            bool integrated = false;
            if (is_initialized) {
                size_t oi = map->frame_index_by_id(opt_id);
                if (oi != nil()) {
                    Frame *of = map->get_frame(oi);
                    of->pose = opt_pose;
                    of->motion = opt_motion;
                    if (ft_ctx && oi + 1 < map->frame_num()) {
                        // pipelined mode: the published state is (at least) one frame old.  predict() hands the biases on
                        // unchanged, so every interval behind it -- and the new frame's -- integrates from the published
                        // biases: one launch for all of them instead of one launch and one wait per interval
                        std::vector<Pipeline::IntegrateJob> jobs;
                        for (size_t j = oi + 1; j < map->frame_num(); ++j) {
                            Frame *fj = map->get_frame(j);
                            jobs.push_back({&fj->preintegration, fj->image->t, opt_motion.bg, opt_motion.ba});
                        }
                        jobs.push_back({&frame->preintegration, frame->image->t, opt_motion.bg, opt_motion.ba});
                        std::vector<PreInt *> pending;
                        P.batch_begin(ft_ctx, pending, jobs, false, false);
                        preprocess();
                        if (!pending.empty()) {
                            const std::vector<double> rec = P.batch_collect(ft_ctx, pending.size());
                            for (size_t k = 0; k < pending.size(); ++k) {
                                std::memcpy(pending[k]->rec, &rec[(size_t)XRHIP_IMU_DIM * k], sizeof(double) * XRHIP_IMU_DIM);
                                pending[k]->valid = true;
                            }
                        }
                        for (size_t j = oi + 1; j < map->frame_num(); ++j)
                            predict(map->get_frame(j)->preintegration, map->get_frame(j - 1), map->get_frame(j));
                        integrated = true;
                    } else {
                        for (size_t j = oi + 1; j < map->frame_num(); ++j) {
                            Frame *fi = map->get_frame(j - 1), *fj = map->get_frame(j);
                            P.integrate(fj->preintegration, fj->image->t, fi->motion.bg, fi->motion.ba, false, false, ft_ctx);
                            predict(fj->preintegration, fi, fj);
                        }
                    }
                } else {
                    ft_latest_state.reset();   // "SWT cannot catch up."
                }
            }
            // Inline mode: the backend will pre-integrate the same interval with Jacobians and covariance when it mirrors this frame
            // (mirror_frame: same samples, the biases of the window's newest frame -- which are `last`'s, just published).  That
            // integration is queued FIRST and the tracker takes its delta out of it as soon as the kernel has it: one launch and
            // one wait per frame instead of two of each (round 4; XRSLAM_AMD_NO_SHARED_INTEGRATION=1: the two-launch form).  The
            // rest of the record is collected by mirror_frame: it runs beside the LK kernel and the RANSAC gates.
            static const bool no_share = std::getenv("XRSLAM_AMD_NO_SHARED_INTEGRATION") != nullptr;   // development switch (A/B, parity)
            const bool backend_mirrors = swt && swt_tag && is_initialized && !pipelined();
            bool shared = false;
            if (!integrated && backend_mirrors && !no_share && !frame->preintegration.data.empty()) {
                shared = swt->mirror_prepare(map, frame.get(), last);
                if (shared) {
                    preprocess();
                    P.integrate_early(frame->preintegration);
                }
            }
            if (!integrated && !shared &&
                P.integrate_begin(frame->preintegration.data, frame->image->t, last->motion.bg, last->motion.ba, false, false,
                                  ft_ctx)) {
                preprocess();
                P.integrate_end(frame->preintegration, ft_ctx);
            }
            preprocess();
            if (swt_tag) hip_check(xrhip_image_prefetch_detect(frame->image->h), "xrhip_image_prefetch_detect");
            // (otherwise the backend's integration is queued now: inline mode only -- in pipelined mode the window map belongs to
            // the backend thread until the hand-off)
            if (backend_mirrors && !shared) swt->mirror_prepare(map, frame.get());
            // pipelined mode: the backend may still be writing changed trash tags into this map's tracks (its half of
            // mirror_frame); nothing above looks at tracks, everything from here on does.  The wait also orders the track-id
            // counter the two maps share (P.ids): mirror_frame's new window-map tracks take their ids before this frame's new
            // tracking-map tracks do, as inline -- ids are reproducible and the counter is never touched by both threads at once.
            if (pipelined()) wait_mirror();
            frame_track_keypoints(P, last, frame.get(), overlap_detect ? &deferred_tag : nullptr);
            if (is_initialized) {
                predict(frame->preintegration, last, frame.get());
                ft_latest_state = LatestState{frame->image->t, frame->pose, frame->motion};
            }
            last->image->release_image_buffer();
        }
        preprocess();
        // Inline mode, tracking: the detection's host half -- waiting for the selection kernel, the greedy min-distance pass, the
        // Poisson-disk filter: ~30 us -- runs beside localize_newframe's solve instead of in front of it.  That solve reads the
        // keypoints of the new frame that carry triangulated tracks; the detection only APPENDS keypoints without tracks (their
        // tracks are created when the next frame is tracked: no id is taken here), so the frame is mirrored first and the new
        // keypoints are appended to both copies afterwards: the same frames, tracks and ids as detecting first.  (Not with the
        // decision log on -- its records are written in the reference's order.)
        // (The misalignment test that sets the frame's FT_NO_TRANSLATION tag -- read by manage_keyframe, after the solve -- rides along.)
        if (swt_tag && !overlap_detect) frame_detect_keypoints(P, frame.get());
        Frame *const attached = frame.get();
        map->attach_frame(std::move(frame));
        if (P.swt_log.enabled()) {   // the tracking map's side of mirror_frame: which track every keypoint of the last two frames is on
            std::lock_guard<std::mutex> log_lock(P.swt_log.mu);
            FILE *fp = P.swt_log.fp;
            std::fprintf(fp, "{\"ft\": [");
            const size_t nf = map->frame_num();
            for (size_t j = nf >= 2 ? nf - 2 : 0; j < nf; ++j) {
                const Frame *f = map->get_frame(j);
                std::fprintf(fp, "%s{\"id\": %zu, \"tracks\": [", j + 1 < nf ? "" : (nf >= 2 ? ", " : ""), f->id);
                for (size_t k = 0; k < f->keypoint_num(); ++k)
                    std::fprintf(fp, "%s%ld", k ? ", " : "", f->get_track(k) ? (long)f->get_track(k)->id : -1L);
                std::fprintf(fp, "]}");
            }
            std::fprintf(fp, "]}\n");
            std::fflush(fp);
        }
        size_t max_frames = is_initialized ? c.feature_tracker_max_frames : c.feature_tracker_max_init_frames;
        while (map->frame_num() > max_frames && map->get_frame(0)->id < opt_id) map->erase_frame(0);
        P.times.frames++;
        if (swt_tag) frontend_work(map->get_frame(map->frame_num() - 1)->id, overlap_detect ? attached : nullptr, std::move(deferred_tag));
        // XRSLAM_AMD_DUMP_OUT: the 'F' record (ba_dump.hpp) -- the key points this frame carries out of the tracker (tracked and newly detected, in
        // key-point order) and the tracks they are on (new tracks are created when the NEXT frame continues a point)
        if (P.out_log.enabled()) {
            OutLogger::Record r(P.out_log, 'F');
            r.u64(attached->id);
            r.f64(attached->image->t);
            r.u32((uint32_t)attached->keypoint_num());
            for (size_t k = 0; k < attached->keypoint_num(); ++k) {
                const V2 px = apply_k(attached->get_keypoint(k), attached->K);
                r.f64(px.x);
                r.f64(px.y);
                r.i64(attached->get_track(k) ? (int64_t)attached->get_track(k)->id : -1);
            }
        }
    }

    // -------- FrontendWorker::work (core/frontend_worker.cpp:28-86)
    // detect_later: inline tracking mode -- the frame whose detection has not run yet (see feature_tracker_work)
    void frontend_work(size_t pending_frame_id, Frame *detect_later = nullptr, std::function<void(Frame *)> deferred_tag = nullptr) {
        if (!swt) {
            init.mirror_keyframe_map(ft_map.get(), pending_frame_id);
            if ((swt = init.initialize())) {
                auto [t, pose, motion] = swt->get_latest_state();
                frontend_latest_state = {t, pending_frame_id, pose, motion};
            }
        } else if (pipelined()) {
            publish_backend_state();
            swt->drop_mirror_hint();   // a hint the backend did not get to is stale from here on
            inflight_ok_ = false;
            // The tracking map's half of mirror_frame here (its frames and tracks are in this core's caches), the window map's
            // half on the backend thread.  Only a trash tag that CHANGES is written back into the tracking map (rare):
            // the feature tracker of the next frame does not read those tags before mirror_done_.
            swt->feature_tracking_map_ = ft_map.get();
            // everything that can throw (the packet, its control block, the job's closure) comes BEFORE the in-flight markers:
            // a failed hand-off must not leave a job "in flight" that was never posted (sync() / wait_mirror() would wait forever)
            auto packet = std::make_shared<SlidingWindowTracker::MirrorPacket>(
                SlidingWindowTracker::make_mirror_packet(ft_map.get(), pending_frame_id, swt->newest_frame_id()));
            std::function<void()> job = [this, packet] {
                {
                    struct Release {
                        std::atomic<bool> &flag;
                        ~Release() { flag.store(true, std::memory_order_release); }
                    } release{mirror_done_};
                    swt->mirror_frame(std::move(*packet));
                }
                inflight_ok_ = swt->track();
            };
            if (packet->valid) last_mirrored_id_ = pending_frame_id;
            mirror_done_.store(false, std::memory_order_relaxed);
            try {
                worker->post(std::move(job));
            } catch (...) {
                mirror_done_.store(true, std::memory_order_release);
                throw;
            }
            inflight_id_ = pending_frame_id;
            inflight_joined_ = false;
        } else if (detect_later) {
            BackendScope backend_scope(*this);
            const bool mirrored = swt->mirror_frame(ft_map.get(), pending_frame_id);
            Frame *const copy = mirrored ? swt->map->get_frame(swt->map->frame_num() - 1) : nullptr;
            BaBuilder::Overlap ov;
            ov.early = [copy, &deferred_tag] {
                if (deferred_tag) deferred_tag(copy);
            };
            ov.fn = [this, detect_later, copy] {
                const size_t n0 = detect_later->keypoint_num();
                frame_detect_keypoints(P, detect_later);
                if (copy)
                    for (size_t k = n0; k < detect_later->keypoint_num(); ++k) copy->append_keypoint(detect_later->get_keypoint(k));
            };
            const bool ok = swt->track(&ov);
            ov.run();   // (a path of track() that never reached the solve)
            if (ov.error) std::rethrow_exception(ov.error);
            if (ok) {
                auto [t, pose, motion] = swt->get_latest_state();
                frontend_latest_state = {t, pending_frame_id, pose, motion};
            }
        } else {
            BackendScope backend_scope(*this);
            swt->mirror_frame(ft_map.get(), pending_frame_id);
            if (swt->track()) {
                auto [t, pose, motion] = swt->get_latest_state();
                frontend_latest_state = {t, pending_frame_id, pose, motion};
            }
        }
    }

    struct Gyro {
        double t;
        V3 w;
    };
    struct Acc {
        double t;
        V3 a;
    };
    Pipeline P;
    std::unique_ptr<Map> ft_map;
    Initializer init;
    std::unique_ptr<SlidingWindowTracker> swt;
    std::deque<Gyro> gyroscopes;
    std::deque<Acc> accelerometers;
    std::deque<ImuData> imus, frontal_imus;
    std::deque<std::unique_ptr<Frame>> frames;
    std::optional<LatestState> ft_latest_state;
    std::tuple<double, size_t, PoseState, MotionState> frontend_latest_state{0.0, nil(), PoseState{}, MotionState{}};
    PoseState latest_pose;
    double latest_timestamp = 0;
    int threading = THREADING_OFF;
    size_t inflight_id_ = nil();   // frame whose backend job is running on `worker` (or finished, its state not yet published)
    bool inflight_joined_ = false; // that job has been waited for (sync()); publication happens at the next hand-off
    size_t last_mirrored_id_ = nil();
    std::atomic<bool> mirror_done_{true};   // false while the backend thread copies the new frame out of the tracking map
    void wait_mirror() {
        if (mirror_done_.load(std::memory_order_acquire)) return;
        WallTimer wt(P.times.w_join);
        while (!mirror_done_.load(std::memory_order_acquire)) {
#if defined(__x86_64__) || defined(__i386__)
            __builtin_ia32_pause();
#else
            std::this_thread::yield();
#endif
        }
    }
    bool inflight_ok_ = false;
    std::unique_ptr<JobThread> worker;
};

}   // namespace xrh
