// config.hpp -- configuration surface of the reference kept byte-compatible:
//   xrslam::Config defaults            /root/reference/xrslam/src/xrslam/config.cpp:7-78
//   YamlConfig (keys, mandatory/optional) xrslam-extra/src/xrslam/extra/yaml_config.cpp:152-362
//   configs/euroc_slam.yaml, configs/euroc_sensor.yaml ("%YAML:1.0", nested maps, flow sequences)
// yaml-cpp is not available here; the two files only use block maps (2-space indent), scalars and
// flow sequences (possibly spanning lines), which this small reader handles.
#pragma once
#include <cctype>
#include <cstdio>
#include <initializer_list>
#include <cstdlib>
#include <fstream>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "geometry.hpp"

namespace xrh {

struct Config {
    // device (mandatory in the sensor yaml)
    double cam_resolution[2] = {752, 480};
    Intrinsics K;
    double cam_distortion[4] = {0, 0, 0, 0};
    size_t cam_distortion_flag = 0;
    double cam_time_offset = 0;
    Quat q_bc;
    V3 p_bc;
    Quat q_bi;
    V3 p_bi;
    double keypoint_noise_cov[4] = {0.5, 0, 0, 0.5};
    double cov_g[9] = {0}, cov_a[9] = {0}, cov_bg[9] = {0}, cov_ba[9] = {0};
    // slam (optional, defaults from config.cpp)
    Quat q_bo;
    V3 p_bo;
    size_t sliding_window_size = 10, sliding_window_subframe_size = 3, sliding_window_force_keyframe_landmarks = 35;
    size_t sliding_window_tracker_frequent = 1;
    double feature_tracker_min_keypoint_distance = 20.0;
    size_t feature_tracker_max_keypoint_detection = 150, feature_tracker_max_init_frames = 60,
           feature_tracker_max_frames = 200;
    double feature_tracker_clahe_clip_limit = 6.0;
    size_t feature_tracker_clahe_width = 8, feature_tracker_clahe_height = 8;
    bool feature_tracker_predict_keypoints = true;
    size_t initializer_keyframe_num = 8, initializer_keyframe_gap = 5, initializer_min_matches = 50;
    double initializer_min_parallax = 10;
    size_t initializer_min_triangulation = 50, initializer_min_landmarks = 30;
    bool initializer_refine_imu = true;
    size_t solver_iteration_limit = 10;
    double solver_time_limit = 1.0e6;
    double rotation_misalignment_threshold = 0.1, rotation_ransac_threshold = 10;
    bool parsac_flag = false;
    double parsac_dynamic_probability = 0.0, parsac_threshold = 3.0, parsac_norm_scale = 1.0;   // config.cpp:70-74 (unused by the tracker, like there)
    size_t parsac_keyframe_check_size = 3;
};

class ConfigError : public std::runtime_error {
  public:
    explicit ConfigError(const std::string &m) : std::runtime_error(m) {}
};

// flat "a.b.c" -> raw scalar / flow-sequence text
inline std::map<std::string, std::string> read_yaml_subset(const std::string &path) {
    std::ifstream f(path);
    if (!f) throw ConfigError("cannot load config " + path);
    std::map<std::string, std::string> out;
    std::vector<std::pair<int, std::string>> stack;
    std::string line, pending_key, pending_val;
    int depth = 0;
    auto strip = [](std::string s) {
        size_t h = std::string::npos;
        bool inq = false;
        for (size_t i = 0; i < s.size(); ++i) {
            if (s[i] == '"') inq = !inq;
            if (s[i] == '#' && !inq) {
                h = i;
                break;
            }
        }
        if (h != std::string::npos) s = s.substr(0, h);
        while (!s.empty() && std::isspace((unsigned char)s.back())) s.pop_back();
        return s;
    };
    while (std::getline(f, line)) {
        if (line.rfind("%YAML", 0) == 0 || line.rfind("---", 0) == 0) continue;
        line = strip(line);
        if (line.find_first_not_of(" \t") == std::string::npos) continue;
        if (depth > 0) {   // continuation of a multi-line flow sequence
            pending_val += " " + line;
            for (char ch : line) depth += (ch == '[') - (ch == ']');
            if (depth == 0) out[pending_key] = pending_val;
            continue;
        }
        int indent = (int)line.find_first_not_of(' ');
        std::string body = line.substr(indent);
        size_t colon = body.find(':');
        if (colon == std::string::npos) continue;
        std::string key = body.substr(0, colon), val = body.substr(colon + 1);
        size_t b = val.find_first_not_of(" \t");
        val = b == std::string::npos ? "" : val.substr(b);
        while (!stack.empty() && stack.back().first >= indent) stack.pop_back();
        std::string full;
        for (auto &s : stack) full += s.second + ".";
        full += key;
        if (val.empty()) {
            stack.emplace_back(indent, key);
            continue;
        }
        for (char ch : val) depth += (ch == '[') - (ch == ']');
        if (depth > 0) {
            pending_key = full;
            pending_val = val;
        } else {
            out[full] = val;
        }
    }
    return out;
}

inline std::vector<double> yaml_numbers(const std::string &v) {
    std::vector<double> r;
    std::string t;
    for (char ch : v) t += (ch == '[' || ch == ']' || ch == ',') ? ' ' : ch;
    std::stringstream ss(t);
    std::string tok;
    while (ss >> tok) r.push_back(std::strtod(tok.c_str(), nullptr));
    return r;
}

inline Config load_config(const std::string &slam_path, const std::string &device_path) {
    Config c;
    auto dev = read_yaml_subset(device_path);
    auto need = [&](const std::string &k, size_t n) {
        auto it = dev.find(k);
        if (it == dev.end()) throw ConfigError("config missing: " + k);
        auto v = yaml_numbers(it->second);
        if (v.size() != n) throw ConfigError("config type error: " + k);
        return v;
    };
    auto v = need("cam0.resolution", 2);
    c.cam_resolution[0] = v[0];
    c.cam_resolution[1] = v[1];
    v = need("cam0.intrinsics", 4);
    c.K = {v[0], v[1], v[2], v[3]};
    v = need("cam0.distortion", 4);
    for (int i = 0; i < 4; ++i) c.cam_distortion[i] = v[i];
    c.cam_distortion_flag = (size_t)need("cam0.camera_distortion_flag", 1)[0];
    c.cam_time_offset = need("cam0.time_offset", 1)[0];
    v = need("cam0.extrinsic.q_bc", 4);
    c.q_bc = Quat{v[0], v[1], v[2], v[3]}.normalized();
    v = need("cam0.extrinsic.p_bc", 3);
    c.p_bc = {v[0], v[1], v[2]};
    v = need("cam0.noise", 4);
    for (int i = 0; i < 4; ++i) c.keypoint_noise_cov[i] = v[i];
    v = need("imu.extrinsic.q_bi", 4);
    c.q_bi = Quat{v[0], v[1], v[2], v[3]}.normalized();
    v = need("imu.extrinsic.p_bi", 3);
    c.p_bi = {v[0], v[1], v[2]};
    v = need("imu.noise.cov_g", 9);
    for (int i = 0; i < 9; ++i) c.cov_g[i] = v[i];
    v = need("imu.noise.cov_a", 9);
    for (int i = 0; i < 9; ++i) c.cov_a[i] = v[i];
    v = need("imu.noise.cov_bg", 9);
    for (int i = 0; i < 9; ++i) c.cov_bg[i] = v[i];
    v = need("imu.noise.cov_ba", 9);
    for (int i = 0; i < 9; ++i) c.cov_ba[i] = v[i];

    auto slam = read_yaml_subset(slam_path);
    auto num = [&](const std::string &k, double &dst) {
        auto it = slam.find(k);
        if (it != slam.end()) dst = yaml_numbers(it->second).at(0);
    };
    auto siz = [&](const std::string &k, size_t &dst) {
        double d = (double)dst;
        num(k, d);
        dst = (size_t)d;
    };
    auto boo = [&](const std::string &k, bool &dst) {
        auto it = slam.find(k);
        if (it != slam.end()) dst = (it->second == "true" || it->second == "True" || it->second == "1");
    };
    if (slam.count("output.q_bo")) {
        auto q = yaml_numbers(slam["output.q_bo"]);
        if (q.size() == 4) c.q_bo = Quat{q[0], q[1], q[2], q[3]}.normalized();
    }
    if (slam.count("output.p_bo")) {
        auto p = yaml_numbers(slam["output.p_bo"]);
        if (p.size() == 3) c.p_bo = {p[0], p[1], p[2]};
    }
    siz("sliding_window.size", c.sliding_window_size);
    siz("sliding_window.subframe_size", c.sliding_window_subframe_size);
    siz("sliding_window.force_keyframe_landmarks", c.sliding_window_force_keyframe_landmarks);
    siz("sliding_window.tracker_frequent", c.sliding_window_tracker_frequent);
    num("feature_tracker.min_keypoint_distance", c.feature_tracker_min_keypoint_distance);
    siz("feature_tracker.max_keypoint_detection", c.feature_tracker_max_keypoint_detection);
    siz("feature_tracker.max_init_frames", c.feature_tracker_max_init_frames);
    siz("feature_tracker.max_frames", c.feature_tracker_max_frames);
    num("feature_tracker.clahe_clip_limit", c.feature_tracker_clahe_clip_limit);
    siz("feature_tracker.clahe_width", c.feature_tracker_clahe_width);
    siz("feature_tracker.clahe_height", c.feature_tracker_clahe_height);
    boo("feature_tracker.predict_keypoints", c.feature_tracker_predict_keypoints);
    siz("initializer.keyframe_num", c.initializer_keyframe_num);
    siz("initializer.keyframe_gap", c.initializer_keyframe_gap);
    siz("initializer.min_matches", c.initializer_min_matches);
    num("initializer.min_parallax", c.initializer_min_parallax);
    siz("initializer.min_triangulation", c.initializer_min_triangulation);
    siz("initializer.min_landmarks", c.initializer_min_landmarks);
    boo("initializer.refine_imu", c.initializer_refine_imu);
    siz("solver.iteration_limit", c.solver_iteration_limit);
    num("solver.time_limit", c.solver_time_limit);
    num("rotation.misalignment_threshold", c.rotation_misalignment_threshold);
    num("rotation.ransac_threshold", c.rotation_ransac_threshold);
    boo("parsac.parsac_flag", c.parsac_flag);
    num("parsac.dynamic_probability", c.parsac_dynamic_probability);
    num("parsac.threshold", c.parsac_threshold);
    num("parsac.norm_scale", c.parsac_norm_scale);
    siz("parsac.keyframe_check_size", c.parsac_keyframe_check_size);
    return c;
}

// "key = value" lines of everything load_config resolved (XRSLAMAmdDescribeConfig), keys as in the yaml files
inline std::string describe_config(const Config &c) {
    std::string out;
    char b[64];
    auto put = [&](const char *key, std::initializer_list<double> v) {
        out += key;
        out += " =";
        for (double x : v) {
            std::snprintf(b, sizeof b, " %.17g", x);
            out += b;
        }
        out += "\n";
    };
    auto arr = [&](const char *key, const double *v, int n) {
        out += key;
        out += " =";
        for (int i = 0; i < n; ++i) {
            std::snprintf(b, sizeof b, " %.17g", v[i]);
            out += b;
        }
        out += "\n";
    };
    arr("cam0.resolution", c.cam_resolution, 2);
    put("cam0.intrinsics", {c.K.fx, c.K.fy, c.K.cx, c.K.cy});
    arr("cam0.distortion", c.cam_distortion, 4);
    put("cam0.camera_distortion_flag", {(double)c.cam_distortion_flag});
    put("cam0.time_offset", {c.cam_time_offset});
    put("cam0.extrinsic.q_bc", {c.q_bc.x, c.q_bc.y, c.q_bc.z, c.q_bc.w});
    put("cam0.extrinsic.p_bc", {c.p_bc.x, c.p_bc.y, c.p_bc.z});
    arr("cam0.noise", c.keypoint_noise_cov, 4);
    put("imu.extrinsic.q_bi", {c.q_bi.x, c.q_bi.y, c.q_bi.z, c.q_bi.w});
    put("imu.extrinsic.p_bi", {c.p_bi.x, c.p_bi.y, c.p_bi.z});
    arr("imu.noise.cov_g", c.cov_g, 9);
    arr("imu.noise.cov_a", c.cov_a, 9);
    arr("imu.noise.cov_bg", c.cov_bg, 9);
    arr("imu.noise.cov_ba", c.cov_ba, 9);
    put("output.q_bo", {c.q_bo.x, c.q_bo.y, c.q_bo.z, c.q_bo.w});
    put("output.p_bo", {c.p_bo.x, c.p_bo.y, c.p_bo.z});
    put("sliding_window.size", {(double)c.sliding_window_size});
    put("sliding_window.subframe_size", {(double)c.sliding_window_subframe_size});
    put("sliding_window.force_keyframe_landmarks", {(double)c.sliding_window_force_keyframe_landmarks});
    put("sliding_window.tracker_frequent", {(double)c.sliding_window_tracker_frequent});
    put("feature_tracker.min_keypoint_distance", {c.feature_tracker_min_keypoint_distance});
    put("feature_tracker.max_keypoint_detection", {(double)c.feature_tracker_max_keypoint_detection});
    put("feature_tracker.max_init_frames", {(double)c.feature_tracker_max_init_frames});
    put("feature_tracker.max_frames", {(double)c.feature_tracker_max_frames});
    put("feature_tracker.clahe_clip_limit", {c.feature_tracker_clahe_clip_limit});
    put("feature_tracker.clahe_width", {(double)c.feature_tracker_clahe_width});
    put("feature_tracker.clahe_height", {(double)c.feature_tracker_clahe_height});
    put("feature_tracker.predict_keypoints", {c.feature_tracker_predict_keypoints ? 1.0 : 0.0});
    put("initializer.keyframe_num", {(double)c.initializer_keyframe_num});
    put("initializer.keyframe_gap", {(double)c.initializer_keyframe_gap});
    put("initializer.min_matches", {(double)c.initializer_min_matches});
    put("initializer.min_parallax", {c.initializer_min_parallax});
    put("initializer.min_triangulation", {(double)c.initializer_min_triangulation});
    put("initializer.min_landmarks", {(double)c.initializer_min_landmarks});
    put("initializer.refine_imu", {c.initializer_refine_imu ? 1.0 : 0.0});
    put("solver.iteration_limit", {(double)c.solver_iteration_limit});
    put("solver.time_limit", {c.solver_time_limit});
    put("rotation.misalignment_threshold", {c.rotation_misalignment_threshold});
    put("rotation.ransac_threshold", {c.rotation_ransac_threshold});
    put("parsac.parsac_flag", {c.parsac_flag ? 1.0 : 0.0});
    put("parsac.dynamic_probability", {c.parsac_dynamic_probability});
    put("parsac.threshold", {c.parsac_threshold});
    put("parsac.norm_scale", {c.parsac_norm_scale});
    put("parsac.keyframe_check_size", {(double)c.parsac_keyframe_check_size});
    return out;
}

}   // namespace xrh
