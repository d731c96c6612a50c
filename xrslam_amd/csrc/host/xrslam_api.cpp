// xrslam_api.cpp -- the outer C ABI (include/XRSLAM.h) on top of the host pipeline.
// Mirrors XRSLAMManager (reference xrslam-interface/src/XRSLAMManager.cpp:85-242) and
// XRSLAMInternal.cpp:4-90: deep-copied images, synchronous calls.  The six reference symbols keep the reference's
// process-global instance (XRSLAMManager.cpp:6-9); every entry point is a thin wrapper over the same functions on an
// `Instance`, and the additive XRSLAMAmdInstance* symbols expose those directly, so that one process can run several
// independent sequences on one GPU (the reference's globals force one instance per process; here all
// state -- id counters, CLAHE / Harris scratch, solver configuration, RD-VIO bin confidences -- lives in the instance).
// An instance may be driven by one thread at a time; different instances may be driven by different threads.
#include "../../../include/XRSLAM.h"

#include <cstring>
#include <memory>
#include <mutex>
#include <string>

#include "pipeline.hpp"

namespace {

struct Manager {
    std::unique_ptr<xrh::System> sys;
    xrh::Config config;
    std::shared_ptr<xrh::HipImage> cur_image;
    std::mutex input_mutex;
    std::string last_error;
    std::vector<uint8_t> gray;
    int device = -1;   // HIP device the instance was created on (the current device is a per-thread setting)
    bool bound_by_replay = false;   // inside XRSLAMAmdInstanceReplay: the device is current for the whole loop
};

Manager &mgr() {
    static Manager m;
    return m;
}

// The thread that drives an instance need not be the one that created it: make the instance's device current.
void bind_device(const Manager &m) {
    if (m.bound_by_replay) return;   // XRSLAMAmdInstanceReplay made the device current once for all the pushes of its loop
    if (m.device >= 0) xrhip_bind_device(m.device);
}

template <class F> void guarded(Manager &m, F &&f) {
    try {
        f();
    } catch (const std::exception &e) {
        m.last_error = e.what();
        std::fprintf(stderr, "[xrslam_hip] %s\n", e.what());
        // whoever threw may have left work between a begin and its end, on the device and in the host's bookkeeping, and -- in
        // pipelined mode -- a backend job running: System::recover_after_error joins it, unwinds both sides and, if the sliding-window
        // tracker was interrupted, falls back to the initialiser like the reference's tracking-failure branch
        if (m.sys) m.sys->recover_after_error();
    }
}

void fill_pose(const xrh::PoseState &latest, const xrh::Quat &q_ext, const xrh::V3 &p_ext, double t, XRSLAMPose *pose) {
    xrh::Quat q = latest.q * q_ext;
    xrh::V3 p = latest.p + latest.q * p_ext;
    pose->timestamp = t;
    pose->quaternion[0] = q.x;
    pose->quaternion[1] = q.y;
    pose->quaternion[2] = q.z;
    pose->quaternion[3] = q.w;
    pose->translation[0] = p.x;
    pose->translation[1] = p.y;
    pose->translation[2] = p.z;
}


int impl_create(Manager &m, const char *slam_config_path, const char *device_config_path, void **config) {
    int ok = 0;
    if (xrhip_get_device(&m.device) != 0) m.device = -1;
    m.last_error.clear();   // XRSLAMAmdLastError speaks about the instance being created, not about an earlier one
    guarded(m, [&] {
        {   // a second Create without Destroy: drop the previous instance first, in XRSLAMDestroy's order -- the pending
            // image returns its device buffer to the Pipeline that owns it, which must still be alive at that point
            std::lock_guard<std::mutex> lk(m.input_mutex);
            m.cur_image.reset();
            m.sys.reset();
        }
        m.config = xrh::load_config(slam_config_path, device_config_path);
        m.sys = std::make_unique<xrh::System>(m.config);
        if (config) *config = static_cast<void *>(&m.config);
        ok = 1;
    });
    return ok;
}

void impl_push(Manager &m, XRSLAMSensorType type, void *data) {
    if (!m.sys || !data) return;
    bind_device(m);
    guarded(m, [&] {
        switch (type) {
        case XRSLAM_SENSOR_CAMERA: {
            auto *im = static_cast<XRSLAMImage *>(data);
            if (im->camera_id != 0) break;
            const int cols = (int)m.config.cam_resolution[0], rows = (int)m.config.cam_resolution[1];
            const uint8_t *src = im->data;
            int stride = im->stride;
            if (im->channel == 3 || im->channel == 4) {   // cv::cvtColor BGR(A)2GRAY: (B*1868 + G*9617 + R*4899 + 8192) >> 14
                m.gray.resize((size_t)cols * rows);
                for (int y = 0; y < rows; ++y)
                    for (int x = 0; x < cols; ++x) {
                        const uint8_t *px = im->data + (size_t)y * im->stride + (size_t)x * im->channel;
                        m.gray[(size_t)y * cols + x] = (uint8_t)((px[0] * 1868 + px[1] * 9617 + px[2] * 4899 + 8192) >> 14);
                    }
                src = m.gray.data();
                stride = cols;
            } else if (im->channel != 1) {
                throw std::runtime_error("Image channel is not supported!");
            }
            try {
                std::lock_guard<std::mutex> lk(m.input_mutex);
                m.cur_image = m.sys->P.make_image(src, stride, im->timeStamp, false);
            } catch (...) {   // this frame did not arrive: XRSLAMRunOneFrame must not track the previous image a second time
                std::lock_guard<std::mutex> lk(m.input_mutex);
                m.cur_image.reset();
                throw;
            }
            break;
        }
        case XRSLAM_SENSOR_ACCELERATION: {
            auto *a = static_cast<XRSLAMAcceleration *>(data);
            m.sys->track_accelerometer(a->timestamp, a->data[0], a->data[1], a->data[2], false);
            break;
        }
        case XRSLAM_SENSOR_GYROSCOPE: {
            auto *g = static_cast<XRSLAMGyroscope *>(data);
            m.sys->track_gyroscope(g->timestamp, g->data[0], g->data[1], g->data[2], false);
            break;
        }
        default:
            break;
        }
    });
}

void impl_run(Manager &m) {
    if (!m.sys) return;
    bind_device(m);
    guarded(m, [&] {
        std::lock_guard<std::mutex> lk(m.input_mutex);
        if (m.cur_image) m.sys->track_camera(m.cur_image);
    });
}

void impl_get_result(Manager &m, XRSLAMResultType type, void *out) {
    if (!m.sys || !out) return;
    guarded(m, [&] {
        switch (type) {
        case XRSLAM_RESULT_BODY_POSE:
            fill_pose(m.sys->latest_pose, m.config.q_bi, m.config.p_bi, m.sys->latest_timestamp, static_cast<XRSLAMPose *>(out));
            break;
        case XRSLAM_RESULT_CAMERA_POSE:
            fill_pose(m.sys->latest_pose, m.config.q_bc, m.config.p_bc, m.sys->latest_timestamp, static_cast<XRSLAMPose *>(out));
            break;
        case XRSLAM_RESULT_STATE: {
            auto st = m.sys->get_system_state();
            *static_cast<XRSLAMState *>(out) = st == xrh::SYS_INITIALIZING ? XRSLAM_STATE_INITIALIZING
                                               : st == xrh::SYS_TRACKING   ? XRSLAM_STATE_TRACKING_SUCCESS
                                                                           : XRSLAM_STATE_TRACKING_FAIL;
            break;
        }
        case XRSLAM_RESULT_LANDMARKS: {
            m.sys->sync();   // pipelined mode: the window map belongs to the backend thread while a frame is in flight
            auto *lm = static_cast<XRSLAMLandmarks *>(out);
            lm->num_landmarks = 0;
            lm->landmarks = nullptr;
            if (m.sys->swt) {
                xrh::Map *map = m.sys->swt->map.get();
                std::vector<XRSLAMLandmark> pts;
                for (size_t i = 0; i < map->track_num(); ++i) {
                    xrh::Track *t = map->get_track(i);
                    if (t->tag(xrh::TT_VALID)) {
                        xrh::V3 p = t->get_landmark_point();
                        pts.push_back({p.x, p.y, p.z});
                    }
                }
                lm->num_landmarks = (int)pts.size();
                lm->landmarks = new XRSLAMLandmark[pts.size() + 1];   // caller-owned, like the reference (:207)
                std::memcpy(lm->landmarks, pts.data(), sizeof(XRSLAMLandmark) * pts.size());
            }
            break;
        }
        case XRSLAM_RESULT_BIAS: {
            auto *b = static_cast<XRSLAMIMUBias *>(out);
            m.sys->sync();
            if (m.sys->swt) {
                auto [t, pose, motion] = m.sys->swt->get_latest_state();
                (void)t;
                (void)pose;
                for (int i = 0; i < 3; ++i) {
                    b->acc_bias.data[i] = motion.ba[i];
                    b->gyr_bias.data[i] = motion.bg[i];
                }
            }
            break;
        }
        case XRSLAM_RESULT_VERSION: {
            auto *s = static_cast<XRSLAMStringOutput *>(out);
            static const char ver[] = "0.1.0";
            s->str_length = (int)std::strlen(ver);
            s->data = new char[s->str_length + 5];
            std::strcpy(s->data, ver);
            break;
        }
        case XRSLAM_INFO_INTRINSICS: {
            auto *k = static_cast<XRSLAMIntrinsics *>(out);
            k->fx = m.config.K.fx;
            k->fy = m.config.K.fy;
            k->cx = m.config.K.cx;
            k->cy = m.config.K.cy;
            break;
        }
        default:
            break;
        }
    });
}

void impl_destroy(Manager &m) {
    bind_device(m);
    guarded(m, [&] {
        m.cur_image.reset();
        m.sys.reset();
    });
}

void impl_set_initial_state(Manager &m, double t, const double q[4], const double p[3], const double v[3],
                            const double bg[3], const double ba[3]) {
    if (!m.sys) return;
    xrh::InitialState s;
    s.t = t;
    s.pose.q = xrh::Quat{q[0], q[1], q[2], q[3]}.normalized();
    s.pose.p = {p[0], p[1], p[2]};
    s.motion.v = {v[0], v[1], v[2]};
    s.motion.bg = {bg[0], bg[1], bg[2]};
    s.motion.ba = {ba[0], ba[1], ba[2]};
    m.sys->init.states.push_back(s);
}

void impl_push_image_device(Manager &m, const void *gray_dev, int stride, double timestamp) {
    if (!m.sys) return;
    bind_device(m);
    guarded(m, [&] {
        try {
            std::lock_guard<std::mutex> lk(m.input_mutex);
            m.cur_image = m.sys->P.make_image(static_cast<const uint8_t *>(gray_dev), stride, timestamp, true);
        } catch (...) {
            std::lock_guard<std::mutex> lk(m.input_mutex);
            m.cur_image.reset();
            throw;
        }
    });
}

void impl_get_camera_config(Manager &m, XRSLAMAmdCameraConfig *out) {
    if (!out) return;
    std::memset(out, 0, sizeof(*out));
    if (!m.sys) return;
    const xrh::Config &c = m.config;
    out->time_offset = c.cam_time_offset;
    out->distortion_flag = (int)c.cam_distortion_flag;
    for (int i = 0; i < 4; ++i) out->distortion[i] = c.cam_distortion[i];
    out->intrinsics[0] = c.K.fx;
    out->intrinsics[1] = c.K.fy;
    out->intrinsics[2] = c.K.cx;
    out->intrinsics[3] = c.K.cy;
    out->resolution[0] = (int)c.cam_resolution[0];
    out->resolution[1] = (int)c.cam_resolution[1];
}

int impl_describe_config(Manager &m, char *buf, int cap) {
    if (!m.sys) return 0;
    const std::string text = xrh::describe_config(m.config);
    if (buf && cap > 0) {
        const size_t n = std::min(text.size(), (size_t)cap - 1);
        std::memcpy(buf, text.data(), n);
        buf[n] = 0;
    }
    return (int)text.size();
}

void impl_set_device_undistort(Manager &m, const char *model) {
    if (!m.sys) return;
    bind_device(m);
    guarded(m, [&] { m.sys->P.set_device_undistort(model); });
}

void impl_flush(Manager &m) {
    if (!m.sys) return;
    bind_device(m);
    guarded(m, [&] { m.sys->sync(); });
}

void impl_set_threading(Manager &m, int mode) {
    if (!m.sys) return;
    bind_device(m);
    guarded(m, [&] { m.sys->set_threading(mode); });
}

void impl_get_times(Manager &m, XRSLAMAmdTimes *out) {
    if (!out) return;
    std::memset(out, 0, sizeof(*out));
    if (!m.sys) return;
    impl_flush(m);
    const xrh::StageTimes &t = m.sys->P.times;
    out->frames = t.frames;
    out->solves = t.solves;
    out->solve_iterations = t.solve_iterations;
    out->marginalizations = t.marginalizations;
    out->keyframes = t.keyframes;
    out->ba_device_ms = t.ba_device_ms;
    out->wall_preprocess = t.w_preprocess;
    out->wall_track = t.w_track;
    out->wall_detect = t.w_detect;
    out->wall_preintegrate = t.w_preintegrate + t.w_preintegrate_ft;
    out->wall_solve = t.w_solve;
    out->wall_marginalize = t.w_marginalize;
    out->wall_frame = t.w_frame;
    for (int i = 0; i < 15; ++i) out->wall_scope[i] = t.scope[i];
    out->wall_scope[15] = t.w_join;
}

void impl_set_profiling(Manager &m, int enable) {
    if (m.sys) {
        impl_flush(m);   // pipelined mode: the backend thread may be inside xrhip_ba_solve, which reads the profiling switch
        bind_device(m);
        xrhip_klt_set_profiling(m.sys->P.klt, enable);
        xrhip_ba_set_profiling(m.sys->P.ba, enable);
        xrhip_ba_set_profiling(m.sys->P.ba_sub, enable);   // (localize_newframe's problem when it is solved together with refine_subwindow's)
    }
}

void impl_get_ba_stats(Manager &m, void *out, int reset) {
    if (!m.sys || !out) return;
    impl_flush(m);
    bind_device(m);
    guarded(m, [&] {
        xrhip_ba_stats *o = static_cast<xrhip_ba_stats *>(out), sub;
        xrh::hip_check(xrhip_ba_get_stats(m.sys->P.ba, o, reset), "xrhip_ba_get_stats");
        xrh::hip_check(xrhip_ba_get_stats(m.sys->P.ba_sub, &sub, reset), "xrhip_ba_get_stats");
        o->n_tiny += sub.n_tiny;   // the single-launch solves of the pair are booked on the context that began the first
        o->n_chain_timed += sub.n_chain_timed;
        o->ms_chain += sub.ms_chain;
        o->bytes_chain += sub.bytes_chain;
    });
}

void impl_get_klt_stats(Manager &m, void *out, int reset) {
    if (!m.sys || !out) return;
    bind_device(m);
    guarded(m, [&] { xrh::hip_check(xrhip_klt_get_stats(m.sys->P.klt, static_cast<xrhip_klt_stats *>(out), reset), "xrhip_klt_get_stats"); });
}

void impl_get_init_report(Manager &m, XRSLAMAmdInitReport *out) {
    if (!m.sys || !out) return;
    const xrh::Initializer &in = m.sys->init;
    out->attempts = in.attempts;
    out->successes = in.successes;
    out->sfm_candidate = in.sfm_candidate;
    out->sfm_triangulated = (int)in.sfm_triangulated;
    out->scale = in.scale;
    for (int k = 0; k < 3; ++k) {
        out->gravity[k] = in.gravity[k];
        out->bg[k] = in.bg[k];
    }
}

}   // namespace

// An opaque handle for the instance-scoped entry points
struct XRSLAMAmdInstance {
    Manager m;
};
// ... and for a group of instances on one GPU whose per-frame launches are issued together (xrslam_hip.h: xrhip_group)
struct XRSLAMAmdGroup {
    xrhip_group *g = nullptr;
    int device = -1;
};

extern "C" {

// ---- the reference's six symbols: one process-global instance (XRSLAMManager.cpp:6-9)
int XRSLAMCreate(const char *slam_config_path, const char *device_config_path, const char *, const char *, void **config) {
    return impl_create(mgr(), slam_config_path, device_config_path, config);
}
void XRSLAMPushSensorData(XRSLAMSensorType type, void *data) { impl_push(mgr(), type, data); }
void XRSLAMRunOneFrame() { impl_run(mgr()); }
void XRSLAMSetViewer(void *) {}
void XRSLAMGetResult(XRSLAMResultType type, void *out) { impl_get_result(mgr(), type, out); }
void XRSLAMDestroy() { impl_destroy(mgr()); }

// ---- additive entry points on the process-global instance
void XRSLAMAmdSetInitialState(double t, const double q[4], const double p[3], const double v[3], const double bg[3],
                              const double ba[3]) {
    impl_set_initial_state(mgr(), t, q, p, v, bg, ba);
}
void XRSLAMAmdPushImageDevice(const void *gray_dev, int stride, double timestamp) {
    impl_push_image_device(mgr(), gray_dev, stride, timestamp);
}
void XRSLAMAmdGetCameraConfig(XRSLAMAmdCameraConfig *out) { impl_get_camera_config(mgr(), out); }
int XRSLAMAmdDescribeConfig(char *buf, int cap) { return impl_describe_config(mgr(), buf, cap); }
void XRSLAMAmdSetDeviceUndistort(const char *model) { impl_set_device_undistort(mgr(), model); }
void XRSLAMAmdGetTimes(XRSLAMAmdTimes *out) { impl_get_times(mgr(), out); }
void XRSLAMAmdSetProfiling(int enable) { impl_set_profiling(mgr(), enable); }
void XRSLAMAmdGetBaStats(void *out, int reset) { impl_get_ba_stats(mgr(), out, reset); }
void XRSLAMAmdGetKltStats(void *out, int reset) { impl_get_klt_stats(mgr(), out, reset); }
void XRSLAMAmdGetInitReport(XRSLAMAmdInitReport *out) { impl_get_init_report(mgr(), out); }
const char *XRSLAMAmdLastError(void) { return mgr().last_error.c_str(); }
void XRSLAMAmdSetThreading(int mode) { impl_set_threading(mgr(), mode); }
void XRSLAMAmdFlush(void) { impl_flush(mgr()); }

// ---- the same entry points on caller-owned instances (several sequences per process / per GPU)
int XRSLAMAmdInstanceCreate(const char *slam_config_path, const char *device_config_path, XRSLAMAmdInstance **out,
                            void **config) {
    if (!out) return 0;
    *out = nullptr;
    XRSLAMAmdInstance *inst = new (std::nothrow) XRSLAMAmdInstance();
    if (!inst) return 0;
    if (impl_create(inst->m, slam_config_path, device_config_path, config) != 1) {
        // the error text must outlive the failed instance: park it in the global one, where XRSLAMAmdLastError reads
        mgr().last_error = inst->m.last_error;
        delete inst;
        return 0;
    }
    *out = inst;
    return 1;
}
void XRSLAMAmdInstanceDestroy(XRSLAMAmdInstance *inst) {
    if (!inst) return;
    impl_destroy(inst->m);
    delete inst;
}
void XRSLAMAmdInstancePushSensorData(XRSLAMAmdInstance *inst, XRSLAMSensorType type, void *data) {
    if (inst) impl_push(inst->m, type, data);
}
void XRSLAMAmdInstanceRunOneFrame(XRSLAMAmdInstance *inst) {
    if (inst) impl_run(inst->m);
}
void XRSLAMAmdInstanceGetResult(XRSLAMAmdInstance *inst, XRSLAMResultType type, void *out) {
    if (inst) impl_get_result(inst->m, type, out);
}
void XRSLAMAmdInstanceSetInitialState(XRSLAMAmdInstance *inst, double t, const double q[4], const double p[3],
                                      const double v[3], const double bg[3], const double ba[3]) {
    if (inst) impl_set_initial_state(inst->m, t, q, p, v, bg, ba);
}
void XRSLAMAmdInstancePushImageDevice(XRSLAMAmdInstance *inst, const void *gray_dev, int stride, double timestamp) {
    if (inst) impl_push_image_device(inst->m, gray_dev, stride, timestamp);
}
void XRSLAMAmdInstanceGetCameraConfig(XRSLAMAmdInstance *inst, XRSLAMAmdCameraConfig *out) {
    if (inst) impl_get_camera_config(inst->m, out);
}
int XRSLAMAmdInstanceDescribeConfig(XRSLAMAmdInstance *inst, char *buf, int cap) {
    return inst ? impl_describe_config(inst->m, buf, cap) : 0;
}
void XRSLAMAmdInstanceSetDeviceUndistort(XRSLAMAmdInstance *inst, const char *model) {
    if (inst) impl_set_device_undistort(inst->m, model);
}
void XRSLAMAmdInstanceGetTimes(XRSLAMAmdInstance *inst, XRSLAMAmdTimes *out) {
    if (inst) impl_get_times(inst->m, out);
}
void XRSLAMAmdInstanceSetProfiling(XRSLAMAmdInstance *inst, int enable) {
    if (inst) impl_set_profiling(inst->m, enable);
}
void XRSLAMAmdInstanceGetBaStats(XRSLAMAmdInstance *inst, void *out, int reset) {
    if (inst) impl_get_ba_stats(inst->m, out, reset);
}
void XRSLAMAmdInstanceGetKltStats(XRSLAMAmdInstance *inst, void *out, int reset) {
    if (inst) impl_get_klt_stats(inst->m, out, reset);
}
void XRSLAMAmdInstanceGetInitReport(XRSLAMAmdInstance *inst, XRSLAMAmdInitReport *out) {
    if (inst) impl_get_init_report(inst->m, out);
}
const char *XRSLAMAmdInstanceLastError(XRSLAMAmdInstance *inst) { return inst ? inst->m.last_error.c_str() : ""; }
void XRSLAMAmdInstanceSetThreading(XRSLAMAmdInstance *inst, int mode) {
    if (inst) impl_set_threading(inst->m, mode);
}
void XRSLAMAmdInstanceFlush(XRSLAMAmdInstance *inst) {
    if (inst) impl_flush(inst->m);
}

// ---- instance groups
int XRSLAMAmdGroupCreate(XRSLAMAmdGroup **out) {
    if (!out) return 0;
    *out = nullptr;
    XRSLAMAmdGroup *grp = new (std::nothrow) XRSLAMAmdGroup();
    if (!grp) return 0;
    if (xrhip_get_device(&grp->device) != 0) grp->device = -1;
    if (xrhip_group_create(&grp->g) != 0) {
        mgr().last_error = xrhip_last_error();
        delete grp;
        return 0;
    }
    *out = grp;
    return 1;
}
int XRSLAMAmdGroupDestroy(XRSLAMAmdGroup *grp) {
    if (!grp) return 1;
    if (grp->device >= 0) xrhip_bind_device(grp->device);
    if (xrhip_group_destroy(grp->g) != 0) {   // instances are still joined: the group stays
        mgr().last_error = xrhip_last_error();
        return 0;
    }
    delete grp;
    return 1;
}
int XRSLAMAmdInstanceJoinGroup(XRSLAMAmdInstance *inst, XRSLAMAmdGroup *grp) {
    if (!inst || !inst->m.sys) return 0;
    int ok = 0;
    bind_device(inst->m);
    guarded(inst->m, [&] {
        inst->m.sys->resolve_device_work();
        inst->m.sys->P.join_group(grp ? grp->g : nullptr);
        ok = 1;
    });
    return ok;
}
void XRSLAMAmdGroupSetProfiling(XRSLAMAmdGroup *grp, int enable) {
    if (grp) xrhip_group_set_profiling(grp->g, enable);
}
void XRSLAMAmdGroupGetStats(XRSLAMAmdGroup *grp, void *out, int reset) {
    if (grp && out) xrhip_group_get_stats(grp->g, static_cast<xrhip_group_stats *>(out), reset);
}

// The player's loop (xrslam-pc/player/src/main.cpp:116-169) for n_steps camera frames of a pre-staged sequence, without a
// host-language round trip per sensor sample: at equal timestamps gyroscope, then accelerometer, then camera
// (IO/async_dataset_reader.cpp:41-48); RunOneFrame and the state / pose query after every image.
int XRSLAMAmdInstanceReplay(XRSLAMAmdInstance *inst, const double *imu7, int n_imu, const double *cam_t, int n_frames,
                            const void *frames, size_t frame_bytes, int stride, int on_device, int *imu_cursor,
                            int *frame_cursor, int n_steps, double *poses_out8) {
    if (!inst || !imu7 || !cam_t || !frames || !imu_cursor || !frame_cursor) return -1;
    Manager &m = inst->m;
    int n_poses = 0;
    // one hipSetDevice for the loop instead of one per pushed sample (~22 per frame); nothing in between runs foreign code on this thread
    bind_device(m);
    struct Bound {
        Manager &m;
        explicit Bound(Manager &mm) : m(mm) { m.bound_by_replay = true; }
        ~Bound() { m.bound_by_replay = false; }
    } bound(m);
    for (int s = 0; s < n_steps && *frame_cursor < n_frames; ++s) {
        const int fk = *frame_cursor;
        const double t = cam_t[fk], lim = t + 1e-9;
        int k = *imu_cursor;
        while (k < n_imu && imu7[7 * (size_t)k] <= lim) {
            const double *r = imu7 + 7 * (size_t)k;
            XRSLAMGyroscope g;
            g.data[0] = r[1]; g.data[1] = r[2]; g.data[2] = r[3]; g.timestamp = r[0];
            XRSLAMAcceleration a;
            a.data[0] = r[4]; a.data[1] = r[5]; a.data[2] = r[6]; a.timestamp = r[0];
            impl_push(m, XRSLAM_SENSOR_GYROSCOPE, &g);
            impl_push(m, XRSLAM_SENSOR_ACCELERATION, &a);
            ++k;
        }
        *imu_cursor = k;
        const unsigned char *img = static_cast<const unsigned char *>(frames) + (size_t)fk * frame_bytes;
        if (on_device) {
            impl_push_image_device(m, img, stride, t);
        } else {
            XRSLAMImage im;
            im.data = const_cast<unsigned char *>(img);
            im.timeStamp = t;
            im.stride = stride;
            im.camera_id = 0;
            im.channel = 1;
            im.ext = nullptr;
            impl_push(m, XRSLAM_SENSOR_CAMERA, &im);
        }
        impl_run(m);
        XRSLAMState state = XRSLAM_STATE_INITIALIZING;
        impl_get_result(m, XRSLAM_RESULT_STATE, &state);
        if (state == XRSLAM_STATE_TRACKING_SUCCESS && poses_out8) {
            XRSLAMPose pose;
            impl_get_result(m, XRSLAM_RESULT_BODY_POSE, &pose);
            double *o = poses_out8 + 8 * (size_t)n_poses++;
            o[0] = pose.timestamp;
            for (int i = 0; i < 3; ++i) o[1 + i] = pose.translation[i];
            for (int i = 0; i < 4; ++i) o[4 + i] = pose.quaternion[i];
        }
        *frame_cursor = fk + 1;
    }
    return n_poses;
}

}   // extern "C"
