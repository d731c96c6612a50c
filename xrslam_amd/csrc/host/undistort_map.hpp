// undistort_map.hpp -- the inverse map of cv::undistort(src, dst, K, D) / xrslam::extra::ImageUndistorter in the packed
// form the device remap (k_undistort, klt_kernels.hip.h) and the host remap (player/euroc_io.hpp) both consume.
//
// The map depends on (K, D, model, size) only: it is computed ONCE, on the host, in double precision exactly like the
// reference's OpenCV path does -- inverse distortion per pixel, the source position passed through float32, rounded to
// 1/32 pixel (cv::convertMaps) -- so that the per-frame work left for the device is pure integer arithmetic (bilinear
// remap with 15-bit weights), bit-exact by construction.  Two words per pixel:
//   word 0 = (uint16)sx | (uint16)sy << 16      integer source position (int16, saturated like CV_16SC2)
//   word 1 = ax | ay << 8                        5-bit fractions
// Models: "cv_undistort" = cv::undistort with (k1 k2 p1 p2), K and D rounded through float32 because the reference
// builds CV_32F matrices (xrslam-pc/player/src/IO/euroc_dataset_reader.cpp:62-69); "radtan" / "equidistant" =
// xrslam::extra::ImageUndistorter (xrslam-extra/include/xrslam/extra/image_undistorter.h:14-92, used by
// IO/tum_dataset_reader.cpp:67-76).  No dependencies.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

namespace xrh {

inline std::vector<uint32_t> build_undistort_map(int w, int h, const double K4[4], const double *D, int nD, const std::string &model) {
    if (model != "cv_undistort" && model != "radtan" && model != "equidistant") throw std::runtime_error("unknown model: " + model);
    if (nD < 4) throw std::runtime_error("distortion model needs at least 4 coefficients");
    std::vector<uint32_t> map((size_t)2 * w * h);
    const bool cv = model == "cv_undistort";
    const double fx = cv ? (double)(float)K4[0] : K4[0], fy = cv ? (double)(float)K4[1] : K4[1];
    const double cx = cv ? (double)(float)K4[2] : K4[2], cy = cv ? (double)(float)K4[3] : K4[3];
    for (int i = 0; i < h; ++i)
        for (int j = 0; j < w; ++j) {
            const double x = (j - cx) / fx, y = (i - cy) / fy;
            double u = j, v = i;
            if (cv) {
                const double k1 = (double)(float)D[0], k2 = (double)(float)D[1], p1 = (double)(float)D[2], p2 = (double)(float)D[3];
                const double x2 = x * x, y2 = y * y, r2 = x2 + y2, xy2 = 2 * x * y;
                const double kr = 1 + ((0.0 * r2 + k2) * r2 + k1) * r2;
                const double xd = x * kr + p1 * xy2 + p2 * (r2 + 2 * x2);
                const double yd = y * kr + p1 * (r2 + 2 * y2) + p2 * xy2;
                u = fx * xd + cx;
                v = fy * yd + cy;
            } else if (model == "radtan") {
                const double k3 = nD > 4 ? D[4] : 0.0;
                const double r2 = x * x + y * y, r4 = r2 * r2, r6 = r4 * r2;
                const double kr = 1.0 + D[0] * r2 + D[1] * r4 + k3 * r6;
                u = fx * (x * kr + 2.0 * D[2] * x * y + D[3] * (r2 + 2.0 * x * x)) + cx;
                v = fy * (y * kr + 2.0 * D[3] * x * y + D[2] * (r2 + 2.0 * y * y)) + cy;
            } else {
                const double r = std::sqrt(x * x + y * y);
                if (r >= 1e-10) {   // the principal point maps to itself
                    const double th = std::atan(r), th2 = th * th, th4 = th2 * th2, th6 = th2 * th4, th8 = th4 * th4;
                    const double thd = th * (1 + D[0] * th2 + D[1] * th4 + D[2] * th6 + D[3] * th8);
                    const double sc = (r > 1e-8) ? thd / r : 1.0;
                    u = fx * (x * sc) + cx;
                    v = fy * (y * sc) + cy;
                }
            }
            // the source position passes through float32 before the 1/32-pixel rounding (cvRound(float * 32.f), like
            // cv::convertMaps): this form -- not the rounding straight from the double -- reproduces the known answers of
            // the reference's test_feature_track on its two EuRoC frames (oracle/undistort.py)
            const long iu = std::lrintf((float)u * 32.0f), iv = std::lrintf((float)v * 32.0f);
            const long sx = std::min(32767L, std::max(-32768L, iu >> 5)), sy = std::min(32767L, std::max(-32768L, iv >> 5));
            map[2 * ((size_t)i * w + j)] = (uint32_t)(uint16_t)(int16_t)sx | ((uint32_t)(uint16_t)(int16_t)sy << 16);
            map[2 * ((size_t)i * w + j) + 1] = (uint32_t)(iu & 31) | ((uint32_t)(iv & 31) << 8);
        }
    return map;
}

// cv::remap(INTER_LINEAR, BORDER_CONSTANT 0) on the packed map: the host form of k_undistort
inline void remap_packed(const uint32_t *map, int w, int h, const uint8_t *src, int src_stride, uint8_t *dst, int dst_stride) {
    auto tap = [&](int y, int x) -> int { return (x >= 0 && x < w && y >= 0 && y < h) ? src[(size_t)y * src_stride + x] : 0; };
    for (int i = 0; i < h; ++i)
        for (int j = 0; j < w; ++j) {
            const uint32_t m0 = map[2 * ((size_t)i * w + j)], m1 = map[2 * ((size_t)i * w + j) + 1];
            const int sx = (int16_t)(m0 & 0xffff), sy = (int16_t)(m0 >> 16), ax = (int)(m1 & 31), ay = (int)((m1 >> 8) & 31);
            const int w00 = (32 - ay) * (32 - ax) * 32, w01 = (32 - ay) * ax * 32, w10 = ay * (32 - ax) * 32, w11 = ay * ax * 32;
            const int acc = tap(sy, sx) * w00 + tap(sy, sx + 1) * w01 + tap(sy + 1, sx) * w10 + tap(sy + 1, sx + 1) * w11;
            const int v = (acc + (1 << 14)) >> 15;
            dst[(size_t)i * dst_stride + j] = (uint8_t)std::min(255, std::max(0, v));
        }
}

}   // namespace xrh
