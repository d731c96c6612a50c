// xrslam_player.cpp -- headless EuRoC player over XRSLAM.h.
//
// The GUI-free equivalent of the loop in xrslam-pc/player/src/main.cpp:116-169: reads an ASL/EuRoC directory,
// feeds gyroscope / accelerometer / camera events in the order of the reference's asynchronous reader, undistorts
// every image like EurocDatasetReader::read_image, writes the body poses in TUM format and, when the ground truth
// is present, reports the ATE.  It links against libxrslam_hip.so only through XRSLAM.h.
//
//   xrslam-player --slam configs/euroc_slam.yaml --device configs/euroc_sensor.yaml --euroc <dir>/mav0
//                 [--out traj.tum] [--bootstrap-frames N] [--max-frames N] [--no-undistort | --host-undistort] [--pipelined]
//
// (--pipelined: XRSLAMAmdSetThreading(1), the reference's XRSLAM_ENABLE_THREADING build with deterministic hand-offs.)
// The reference player's own command line (main.cpp:57-79) is accepted as well, so its invocations carry over:
//
//   xrslam-player -sc configs/euroc_slam.yaml -dc configs/euroc_sensor.yaml [-lc license] [--tum traj.tum]
//                 [--csv traj.csv] [-p] euroc://<dir>/mav0 | tum://<dir>/mav0
//
// (-p / --play and the license are accepted and ignored: there is no viewer to start.)  `tum://` is the reference's
// TUM-VI reader (IO/tum_dataset_reader.cpp): same ASL directory layout, images undistorted with the equidistant
// (fisheye) model of xrslam::extra::ImageUndistorter instead of cv::undistort.
//
// By default the library initialises itself (SfM + IMU alignment, like the reference).  With --bootstrap-frames N
// the first N camera frames are instead seeded from state_groundtruth_estimate0 through XRSLAMAmdSetInitialState
// (N >= 36 covers the first window), which takes the initialiser out of an accuracy / throughput comparison.
#include <chrono>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <string>

#include "XRSLAM.h"
#include "euroc_io.hpp"
#include "../host/config.hpp"

using namespace xrplayer;

static const TruthRow *nearest_truth(const std::vector<TruthRow> &gt, double t, double tol) {
    if (gt.empty()) return nullptr;
    size_t lo = 0, hi = gt.size();
    while (lo + 1 < hi) {
        const size_t mid = (lo + hi) / 2;
        if (gt[mid].t <= t) lo = mid;
        else hi = mid;
    }
    const TruthRow *best = &gt[lo];
    if (lo + 1 < gt.size() && std::fabs(gt[lo + 1].t - t) < std::fabs(best->t - t)) best = &gt[lo + 1];
    return std::fabs(best->t - t) <= tol ? best : nullptr;
}

int main(int argc, char **argv) {
    std::map<std::string, std::string> opt;
    bool undistort = true, host_undistort = false, pipelined = false;
    // the reference's option names (main.cpp:57-71) map onto ours
    const std::map<std::string, std::string> alias = {{"-sc", "slam"}, {"--slamconfig", "slam"}, {"-dc", "device"},
                                                      {"--deviceconfig", "device"}, {"-lc", "license"}, {"--license", "license"},
                                                      {"--tum", "out"}};
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        if (a == "--no-undistort") undistort = false;
        else if (a == "--host-undistort") host_undistort = true;   // the reference's arrangement: the reader rectifies on the host
        else if (a == "--pipelined") pipelined = true;
        else if (a == "-p" || a == "--play") continue;
        else if (alias.count(a) && i + 1 < argc) opt[alias.at(a)] = argv[++i];
        else if (a.rfind("--", 0) == 0 && i + 1 < argc) opt[a.substr(2)] = argv[++i];
        else if (a[0] != '-' && !opt.count("input")) opt["input"] = a;
        else {
            std::fprintf(stderr, "unknown argument %s\n", a.c_str());
            return 2;
        }
    }
    std::string model = "cv_undistort";   // EurocDatasetReader::read_image: cv::undistort, radial-tangential
    if (opt.count("input")) {
        const auto [scheme, path] = split_dataset_url(opt["input"]);
        if (scheme.empty()) {
            std::fprintf(stderr, "Cannot open \"%s\"\n", opt["input"].c_str());   // main.cpp:99-103
            return 1;
        }
        opt["euroc"] = path;
        if (scheme == "tum") model = "equidistant";
    }
    if (!opt.count("slam") || !opt.count("device") || !opt.count("euroc")) {
        std::fprintf(stderr, "usage: xrslam-player --slam cfg.yaml --device sensor.yaml --euroc <dir>/mav0 [--out traj.tum] "
                             "[--csv traj.csv] [--bootstrap-frames N] [--max-frames N] [--no-undistort | --host-undistort] [--pipelined]\n"
                             "   or: xrslam-player -sc cfg.yaml -dc sensor.yaml [--tum traj.tum] [--csv traj.csv] [-p] "
                             "euroc://<dir>/mav0 | tum://<dir>/mav0\n");
        return 2;
    }
    const std::string root = opt["euroc"];
    const size_t bootstrap = opt.count("bootstrap-frames") ? (size_t)std::atol(opt["bootstrap-frames"].c_str()) : 0;
    const size_t max_frames = opt.count("max-frames") ? (size_t)std::atol(opt["max-frames"].c_str()) : (size_t)-1;

    // the same configuration surface the library parses (camera offset, intrinsics and distortion for the reader)
    xrh::Config cfg;
    try {
        cfg = xrh::load_config(opt["slam"], opt["device"]);
    } catch (const std::exception &e) {
        std::fprintf(stderr, "configuration: %s\n", e.what());
        return 1;
    }
    const std::vector<CameraRow> cam = load_camera_csv(root + "/cam0/data.csv");
    const std::vector<ImuRow> imu = load_imu_csv(root + "/imu0/data.csv");
    const std::vector<TruthRow> gt = load_groundtruth_csv(root + "/state_groundtruth_estimate0/data.csv");
    if (cam.empty() || imu.empty()) {
        std::fprintf(stderr, "no camera or IMU data under %s\n", root.c_str());
        return 1;
    }
    const std::vector<Event> events = merge_events(cam, imu, cfg.cam_time_offset);

    void *handle = nullptr;
    if (XRSLAMCreate(opt["slam"].c_str(), opt["device"].c_str(), "", "xrslam-player", &handle) != 1) {
        std::fprintf(stderr, "XRSLAMCreate failed: %s\n", XRSLAMAmdLastError());
        return 1;
    }
    // Frames are rectified on the GPU by default (one upload of the frame as recorded, no host pass over the pixels);
    // --host-undistort keeps the reference's arrangement, where the reader thread runs cv::undistort.  Same pixels
    // either way: the library builds the same 1/32-pixel map (host/undistort_map.hpp).
    const bool device_undistort = undistort && cfg.cam_distortion_flag && !host_undistort;
    if (device_undistort) XRSLAMAmdSetDeviceUndistort(model.c_str());
    if (pipelined) XRSLAMAmdSetThreading(1);
    size_t seeded = 0;
    for (size_t i = 0; i < cam.size() && i < bootstrap; ++i) {
        const double t = cam[i].t + cfg.cam_time_offset;
        if (const TruthRow *g = nearest_truth(gt, t, 2.6e-3)) {
            const double q[4] = {g->q[1], g->q[2], g->q[3], g->q[0]};   // ASL stores w first
            XRSLAMAmdSetInitialState(t, q, g->p, g->v, g->bg, g->ba);
            ++seeded;
        }
    }
    FILE *out = opt.count("out") ? std::fopen(opt["out"].c_str(), "w") : nullptr;
    FILE *csv = opt.count("csv") ? std::fopen(opt["csv"].c_str(), "w") : nullptr;
    if ((opt.count("out") && !out) || (opt.count("csv") && !csv)) {
        std::fprintf(stderr, "Cannot open file\n");   // trajectory_writer.h:37,62
        return 1;
    }
    const double K4[4] = {cfg.K.fx, cfg.K.fy, cfg.K.cx, cfg.K.cy};
    std::unique_ptr<Undistorter> und;
    std::vector<uint8_t> rectified;
    std::vector<xrh::V3> est, ref;
    bool has_gyro = false, has_acc = false;
    size_t frames = 0, tracked = 0;
    // a frame is processed when the next IMU sample arrives (detail.cpp:134), so the loop is timed as a whole;
    // PNG decoding and undistortion run in the reference's reader thread and are timed separately here
    double io_seconds = 0.0;
    const auto loop_begin = std::chrono::steady_clock::now();
    for (const Event &ev : events) {
        if (ev.type == EV_GYROSCOPE) {
            has_gyro = true;
            XRSLAMGyroscope g{{imu[ev.index].w[0], imu[ev.index].w[1], imu[ev.index].w[2]}, ev.t};
            XRSLAMPushSensorData(XRSLAM_SENSOR_GYROSCOPE, &g);
        } else if (ev.type == EV_ACCELEROMETER) {
            has_acc = true;
            XRSLAMAcceleration a{{imu[ev.index].a[0], imu[ev.index].a[1], imu[ev.index].a[2]}, ev.t};
            XRSLAMPushSensorData(XRSLAM_SENSOR_ACCELERATION, &a);
        } else {
            if (frames >= max_frames) break;
            const auto io0 = std::chrono::steady_clock::now();
            GrayImage img;
            try {
                img = decode_png(read_file(root + "/cam0/data/" + cam[ev.index].filename));
            } catch (const std::exception &e) {
                std::fprintf(stderr, "%s: %s\n", cam[ev.index].filename.c_str(), e.what());
                break;
            }
            // the library copies cam0.resolution rows x columns out of the buffer it is handed
            // (XRSLAMManager.cpp:113-131 does the same with the configured size): a frame of any other size is an error
            if (img.w != (int)cfg.cam_resolution[0] || img.h != (int)cfg.cam_resolution[1]) {
                std::fprintf(stderr, "%s: image is %dx%d, the device configuration says %dx%d\n", cam[ev.index].filename.c_str(), img.w,
                             img.h, (int)cfg.cam_resolution[0], (int)cfg.cam_resolution[1]);
                break;
            }
            const uint8_t *pixels = img.px.data();
            if (undistort && cfg.cam_distortion_flag && !device_undistort) {
                if (!und) {
                    if (model == "cv_undistort") und.reset(new Undistorter(img.w, img.h, K4, cfg.cam_distortion));
                    else und.reset(new Undistorter(img.w, img.h, K4, std::vector<double>(cfg.cam_distortion, cfg.cam_distortion + 4), model));
                }
                rectified.resize((size_t)img.w * img.h);
                und->apply(img.px.data(), img.w, rectified.data(), img.w);
                pixels = rectified.data();
            }
            io_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - io0).count();
            XRSLAMImage xi;
            std::memset(&xi, 0, sizeof(xi));
            xi.camera_id = 0;
            xi.timeStamp = ev.t;
            xi.data = const_cast<uint8_t *>(pixels);
            xi.channel = 1;
            xi.stride = img.w;
            XRSLAMPushSensorData(XRSLAM_SENSOR_CAMERA, &xi);
            if (has_gyro && has_acc) {
                XRSLAMRunOneFrame();
                XRSLAMState state;
                XRSLAMGetResult(XRSLAM_RESULT_STATE, &state);
                if (state == XRSLAM_STATE_TRACKING_SUCCESS) {
                    XRSLAMPose pose;
                    XRSLAMGetResult(XRSLAM_RESULT_BODY_POSE, &pose);
                    // before the first tracked frame the library reports an all-zero quaternion (detail.cpp:165-168);
                    // such a pose carries no estimate and is kept out of the trajectory
                    const bool valid = pose.quaternion[0] != 0 || pose.quaternion[1] != 0 || pose.quaternion[2] != 0 || pose.quaternion[3] != 0;
                    if (pose.timestamp > 0 && valid) {
                        ++tracked;
                        if (out) write_tum_pose(out, pose.timestamp, pose.translation, pose.quaternion);
                        if (csv) write_csv_pose(csv, pose.timestamp, pose.translation, pose.quaternion);
                        if (const TruthRow *g = nearest_truth(gt, pose.timestamp, 2.6e-3)) {
                            est.push_back({pose.translation[0], pose.translation[1], pose.translation[2]});
                            ref.push_back({g->p[0], g->p[1], g->p[2]});
                        }
                    }
                }
            }
            ++frames;
        }
    }
    XRSLAMAmdFlush();   // pipelined mode: the backend of the last frame belongs to the run
    const double busy = std::chrono::duration<double>(std::chrono::steady_clock::now() - loop_begin).count() - io_seconds;
    if (out) std::fclose(out);
    if (csv) std::fclose(csv);
    const char *err = XRSLAMAmdLastError();
    XRSLAMAmdInitReport rep;
    std::memset(&rep, 0, sizeof(rep));
    XRSLAMAmdGetInitReport(&rep);
    std::printf("{\"frames\": %zu, \"tracked\": %zu, \"bootstrap_states\": %zu, \"init_attempts\": %ld, \"init_scale\": %.6f, "
                "\"ms_per_frame\": %.4f, \"io_ms_per_frame\": %.4f, \"ate_rmse_m\": %.6f, \"error\": \"%s\"}\n",
                frames, tracked, seeded, rep.attempts, rep.successes ? rep.scale : 0.0, frames ? 1e3 * busy / frames : 0.0,
                frames ? 1e3 * io_seconds / frames : 0.0, est.size() >= 3 ? ate_rmse(est, ref) : -1.0, err ? err : "");
    XRSLAMDestroy();
    return (err && *err) ? 1 : 0;
}
