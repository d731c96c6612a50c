// player_loop_host.cpp -- conformance check of include/XRSLAM.h against the call sequence of the reference's player.
//
// A host application written against the reference's interface (xrslam-interface/include/XRSLAM.h:19-229) uses exactly
// these names, struct fields and enum constants in its per-sensor loop (xrslam-pc/player/src/main.cpp:80-169): Create with
// a `void *` configuration out-parameter, the three sensor pushes, RunOneFrame, the state / body-pose queries, Destroy.
// This file restates that loop against OUR header; tests/test_abi.py compiles it (that is the check that the header is
// source-compatible), links it against the CPU reference build of the library and runs it on a synthetic ASL directory.
// The reader below takes the five camera values it needs from XRSLAMAmdGetCameraConfig -- the patch INTEGRATION.md
// section 1 prescribes for euroc_dataset_reader.cpp:4-7 -- instead of calling YamlConfig virtuals on the handle.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/XRSLAM.h"

int main(int argc, char **argv) {
    if (argc < 4) {
        std::fprintf(stderr, "usage: %s slam.yaml device.yaml frames.bin\n", argv[0]);
        return 2;
    }
    // frames.bin: int32 n_frames, w, h, n_imu; double cam_t[n_frames]; double imu[n_imu][7]; uint8 frames[n_frames][h][w]
    FILE *f = std::fopen(argv[3], "rb");
    if (!f) return 2;
    int hdr[4];
    if (std::fread(hdr, sizeof(int), 4, f) != 4) return 2;
    const int n_frames = hdr[0], w = hdr[1], h = hdr[2], n_imu = hdr[3];
    std::vector<double> cam_t(n_frames), imu((size_t)7 * n_imu);
    std::vector<unsigned char> frames((size_t)n_frames * w * h);
    if (std::fread(cam_t.data(), sizeof(double), n_frames, f) != (size_t)n_frames) return 2;
    if (std::fread(imu.data(), sizeof(double), imu.size(), f) != imu.size()) return 2;
    if (std::fread(frames.data(), 1, frames.size(), f) != frames.size()) return 2;
    std::fclose(f);

    void *yaml_config = nullptr;                                            // main.cpp:82
    int create_succ = XRSLAMCreate(argv[1], argv[2], "", "XRSLAM PC", &yaml_config);   // main.cpp:83-86
    if (create_succ != 1) {
        std::fprintf(stderr, "create failed: %s\n", XRSLAMAmdLastError());
        return 1;
    }
    XRSLAMAmdCameraConfig cc;                                               // the reader's view of the configuration
    XRSLAMAmdGetCameraConfig(&cc);
    if (cc.resolution[0] != w || cc.resolution[1] != h) return 3;

    bool has_gyroscope = false, has_accelerometer = false;                  // main.cpp:104
    int k = 0, tracked = 0;
    XRSLAMPose last{};
    for (int i = 0; i < n_frames; ++i) {
        const double t_img = cam_t[i] + cc.time_offset;                     // euroc_dataset_reader.cpp:16
        while (k < n_imu && imu[7 * (size_t)k] <= t_img + 1e-9) {
            const double *r = &imu[7 * (size_t)k];
            XRSLAMGyroscope gyro = {{r[1], r[2], r[3]}, r[0]};              // euroc_dataset_reader.cpp:23
            has_gyroscope = true;
            XRSLAMPushSensorData(XRSLAM_SENSOR_GYROSCOPE, &gyro);           // main.cpp:126
            XRSLAMAcceleration acc = {{r[4], r[5], r[6]}, r[0]};            // euroc_dataset_reader.cpp:26
            has_accelerometer = true;
            XRSLAMPushSensorData(XRSLAM_SENSOR_ACCELERATION, &acc);         // main.cpp:131
            ++k;
        }
        XRSLAMImage image;                                                  // main.cpp:142-149
        image.camera_id = 0;
        image.timeStamp = t_img;
        image.ext = nullptr;
        image.data = &frames[(size_t)i * w * h];
        image.channel = 1;
        image.stride = w;
        XRSLAMPushSensorData(XRSLAM_SENSOR_CAMERA, &image);
        if (has_accelerometer && has_gyroscope) {
            XRSLAMRunOneFrame();                                            // main.cpp:151
            XRSLAMState state;
            XRSLAMGetResult(XRSLAM_RESULT_STATE, &state);                   // main.cpp:152-153
            if (state == XRSLAM_STATE_TRACKING_SUCCESS) {
                XRSLAMPose pose_b;
                XRSLAMGetResult(XRSLAM_RESULT_BODY_POSE, &pose_b);          // main.cpp:158-159
                if (pose_b.timestamp > 0) {                                 // main.cpp:160
                    ++tracked;
                    last = pose_b;
                }
            }
        }
    }
    XRSLAMIntrinsics K;                                                     // the reference's own query for K (XRSLAM.h:168-176)
    XRSLAMGetResult(XRSLAM_INFO_INTRINSICS, &K);
    std::printf("{\"tracked\": %d, \"t\": %.9f, \"p\": [%.9f, %.9f, %.9f], \"fx\": %.6f, \"distortion_flag\": %d}\n", tracked,
                last.timestamp, last.translation[0], last.translation[1], last.translation[2], K.fx, cc.distortion_flag);
    XRSLAMDestroy();                                                        // main.cpp:172
    return 0;
}
