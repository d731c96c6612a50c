// ba_dump.hpp -- development / test aid: writes the problem a BaBuilder hands to xrhip_ba_solve to a file, so that the
// solves of a run can be frozen and replayed (SURVEY.md 8d "S4": frozen refine_window snapshots; tests/golden/ba_snapshots).
// Enabled by the environment:  XRSLAM_AMD_DUMP_BA=<directory>  [XRSLAM_AMD_DUMP_MIN_OBS=<n>]  [XRSLAM_AMD_DUMP_EVERY=<k>]
// File layout (little endian): "XRBA1\0\0\0", 8 int32 counts {F, L, M, MR, NI, NP, max_iterations, 0}, then the arrays
// of xrhip_ba_problem in declaration order (include/xrslam_hip.h), doubles as f64, indices as i32, flags as u8.
// tests/ba_snapshots.py reads it back.  No HIP dependency.
#pragma once
#include <mutex>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../../include/xrslam_hip.h"

namespace xrh {

struct BaDumper {
    std::string dir;
    int min_obs = 0, every = 1;
    long seen = 0, written = 0;
    BaDumper() {
        if (const char *d = std::getenv("XRSLAM_AMD_DUMP_BA")) dir = d;
        if (const char *m = std::getenv("XRSLAM_AMD_DUMP_MIN_OBS")) min_obs = std::atoi(m);
        if (const char *e = std::getenv("XRSLAM_AMD_DUMP_EVERY")) every = std::max(1, std::atoi(e));
    }
    bool enabled() const { return !dir.empty(); }
    void maybe_dump(const xrhip_ba_problem &pb, long frame_count) {
        if (!enabled() || pb.n_obs < min_obs) return;
        if (seen++ % every) return;
        char name[512];
        std::snprintf(name, sizeof name, "%s/ba_%05ld_f%05ld_F%d_L%d_M%d.xrba", dir.c_str(), written++, frame_count, pb.n_frames,
                      pb.n_landmarks, pb.n_obs);
        FILE *fp = std::fopen(name, "wb");
        if (!fp) return;
        const char magic[8] = {'X', 'R', 'B', 'A', '1', 0, 0, 0};
        const int32_t hdr[8] = {pb.n_frames, pb.n_landmarks, pb.n_obs, pb.n_rot, pb.n_imu, pb.prior_n, pb.max_iterations, 0};
        std::fwrite(magic, 1, 8, fp);
        std::fwrite(hdr, sizeof(int32_t), 8, fp);
        auto wd = [&](const double *p, size_t n) { if (n) std::fwrite(p, sizeof(double), n, fp); };
        auto wi = [&](const int *p, size_t n) { if (n) std::fwrite(p, sizeof(int), n, fp); };
        auto wb = [&](const uint8_t *p, size_t n) { if (n) std::fwrite(p, 1, n, fp); };
        const size_t F = pb.n_frames, L = pb.n_landmarks, M = pb.n_obs, MR = pb.n_rot, NI = pb.n_imu, NP = pb.prior_n;
        wd(pb.frame_state, 16 * F);
        wb(pb.frame_fix, F);
        wd(pb.cam_q_bc, 4); wd(pb.cam_p_bc, 3); wd(pb.imu_q_bi, 4); wd(pb.imu_p_bi, 3); wd(pb.sqrt_inv_cov, 2);
        wd(pb.inv_depth, L);
        wb(pb.landmark_fix, L);
        wi(pb.obs_tgt, M); wi(pb.obs_ref, M); wi(pb.obs_lm, M);
        wd(pb.obs_z_tgt, 3 * M); wd(pb.obs_z_ref, 3 * M);
        wi(pb.rot_tgt, MR); wi(pb.rot_ref, MR);
        wd(pb.rot_z_tgt, 3 * MR); wd(pb.rot_z_ref, 3 * MR);
        wi(pb.imu_i, NI); wi(pb.imu_j, NI);
        wd(pb.imu_data, (size_t)XRHIP_IMU_DIM * NI);
        wi(pb.prior_frames, NP);
        wd(pb.prior_sqrt_info, 225 * NP * NP); wd(pb.prior_infovec, 15 * NP); wd(pb.prior_lin, 16 * NP);
        std::fclose(fp);
    }
};

// Decision log of the sliding-window tracker (development / test aid): XRSLAM_AMD_DUMP_SWT=<file> makes
// SlidingWindowTracker::track() append one JSON line per frame -- the inputs of manage_keyframe (new frame id, its
// FT_NO_TRANSLATION tag, the number of mapped landmarks it observes) and the outcome (keyframe or subframe, the window of
// keyframes with their subframe lists afterwards).  tests/test_swt_model.py replays the inputs through an independent
// Python model of core/sliding_window_tracker.cpp:145-223,360-393 and requires the same outcome at every frame.
struct SwtLogger {
    FILE *fp = nullptr;
    std::mutex mu;   // a record is several fprintf calls; in pipelined mode two threads write records
    SwtLogger() {
        if (const char *p = std::getenv("XRSLAM_AMD_DUMP_SWT")) fp = std::fopen(p, "w");
    }
    ~SwtLogger() {
        if (fp) std::fclose(fp);
    }
    SwtLogger(const SwtLogger &) = delete;
    SwtLogger &operator=(const SwtLogger &) = delete;
    bool enabled() const { return fp != nullptr; }
};

// Log of what a frame PRODUCES (test aid): XRSLAM_AMD_DUMP_OUT=<file> appends binary records (little endian) --
//   'F' (feature tracker, when a frame joins the tracking map):  u64 frame id, f64 t, u32 n, n x {f64 px, f64 py, i64 track id | -1}
//   'B' (sliding-window tracker, at the end of track()):  u64 newest frame id, u32 keyframe?, f64 state[16] (q xyzw, p, v, bg, ba),
//        u32 window frames, per frame {u64 id, u32 n_sub, n_sub x u64}, u32 n_kp, n_kp x i64 window-map track id | -1 (newest frame),
//        u32 n_tracks, n_tracks x {u64 id, u32 tag bits (1 << TrackTag), f64 inv_depth, f64 x, y, z (0 unless triangulated)}
// each preceded by {u8 tag, u32 payload bytes}.  Unlike the decision log above it changes nothing about the path a frame takes
// (the overlaps of the inline mode stay on): tests/test_bench_stream_parity.py compares the GPU library's file with the CPU
// reference pipeline's -- ids and indices exactly, pixels / states / landmarks at north_star's tolerance.
struct OutLogger {
    FILE *fp = nullptr;
    std::mutex mu;   // pipelined mode: 'F' records come from the caller's thread, 'B' records from the backend thread
    std::vector<unsigned char> buf;
    OutLogger() {
        if (const char *p = std::getenv("XRSLAM_AMD_DUMP_OUT")) fp = std::fopen(p, "wb");
    }
    ~OutLogger() {
        if (fp) std::fclose(fp);
    }
    OutLogger(const OutLogger &) = delete;
    OutLogger &operator=(const OutLogger &) = delete;
    bool enabled() const { return fp != nullptr; }
    struct Record {   // one record under the lock; written out when it goes out of scope
        OutLogger &log;
        std::unique_lock<std::mutex> lk;
        unsigned char tag;
        Record(OutLogger &l, unsigned char t) : log(l), lk(l.mu), tag(t) { log.buf.clear(); }
        template <class T> void put(const T &v) {
            const unsigned char *p = reinterpret_cast<const unsigned char *>(&v);
            log.buf.insert(log.buf.end(), p, p + sizeof(T));
        }
        void u32(uint32_t v) { put(v); }
        void u64(uint64_t v) { put(v); }
        void i64(int64_t v) { put(v); }
        void f64(double v) { put(v); }
        ~Record() {
            const uint32_t n = (uint32_t)log.buf.size();
            std::fwrite(&tag, 1, 1, log.fp);
            std::fwrite(&n, sizeof n, 1, log.fp);
            if (n) std::fwrite(log.buf.data(), 1, n, log.fp);
            std::fflush(log.fp);
        }
    };
};

// Decision log of the initialiser and of the RD-VIO filters (test aid): XRSLAM_AMD_DUMP_INIT=<file> appends one JSON line per
// decision with everything the decision looked at (%.17g) -- the key frames Initializer::mirror_keyframe_map picked, the two-view
// matches / model matrices / eight (R, T) hypotheses / triangulation vote of init_sfm, the per-interval quantities of the three
// alignment solves of init_imu with their answers, the scale gates, apply_init's rotation; and for parsac.parsac_flag the inputs and
// verdicts of judge_track_status and of every PARSAC run.  tests/init_model.py and tests/parsac_model.py re-derive every answer from
// the inputs in numpy, written from core/initializer.cpp:22-571, utility/parsac.h and utility/imu_parsac.h (rows f3 / f4).
struct InitLogger {
    FILE *fp = nullptr;
    std::mutex mu;
    InitLogger() {
        if (const char *p = std::getenv("XRSLAM_AMD_DUMP_INIT")) fp = std::fopen(p, "w");
    }
    ~InitLogger() {
        if (fp) std::fclose(fp);
    }
    InitLogger(const InitLogger &) = delete;
    InitLogger &operator=(const InitLogger &) = delete;
    bool enabled() const { return fp != nullptr; }
    // a record under construction: key / value pairs appended as JSON, written as one line when it goes out of scope
    struct Line {
        InitLogger &log;
        std::unique_lock<std::mutex> lk;
        std::string s;
        Line(InitLogger &l, const char *what) : log(l), lk(l.mu) {
            s = "{\"what\": \"";
            s += what;
            s += "\"";
        }
        void key(const char *k) {
            s += ", \"";
            s += k;
            s += "\": ";
        }
        void num(double v) {
            char b[40];
            if (std::isfinite(v)) std::snprintf(b, sizeof b, "%.17g", v);
            else std::snprintf(b, sizeof b, "null");
            s += b;
        }
        void put(const char *k, double v) {
            key(k);
            num(v);
        }
        void put(const char *k, const double *v, size_t n) {
            key(k);
            s += "[";
            for (size_t i = 0; i < n; ++i) {
                if (i) s += ", ";
                num(v[i]);
            }
            s += "]";
        }
        void put(const char *k, const std::vector<double> &v) { put(k, v.data(), v.size()); }
        ~Line() {
            s += "}\n";
            std::fwrite(s.data(), 1, s.size(), log.fp);
            std::fflush(log.fp);
        }
    };
};

// Log of the sensor synchronisation (development / test aid): XRSLAM_AMD_DUMP_SYNC=<file> makes System::feature_tracker_work
// append one JSON line per frame -- the frame's id and time and the IMU samples Detail::track_imu attached to it (time, angular
// rate, acceleration, %.17g) -- and System::track_camera one line per answered pose.  tests/test_sync_model.py replays the same
// sensor events through an independent Python model of core/detail.cpp:46-177 and requires identical samples and poses.
struct SyncLogger {
    FILE *fp = nullptr;
    SyncLogger() {
        if (const char *p = std::getenv("XRSLAM_AMD_DUMP_SYNC")) fp = std::fopen(p, "w");
    }
    ~SyncLogger() {
        if (fp) std::fclose(fp);
    }
    SyncLogger(const SyncLogger &) = delete;
    SyncLogger &operator=(const SyncLogger &) = delete;
    bool enabled() const { return fp != nullptr; }
};

}   // namespace xrh
