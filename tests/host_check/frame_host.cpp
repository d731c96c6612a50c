// ctypes hook for tests/test_oracle_klt.py: the reference's test_feature_track (xrslam-test/test/src/
// test_feature_track.cpp:24-65) at frame level -- preprocess + Frame::detect_keypoints on the first image,
// Frame::track_keypoints (forward/backward LK, 5-point and 2-point RANSAC gates, Poisson thinning) onto the second --
// through the product's host pipeline sources linked against the CPU oracle (oracle/_build/*.o, the xrhip shim).
// No extrinsics and no IMU data are set, like in that test: the keypoint prediction is the identity.
#include <cstdint>
#include <memory>

#include "../../xrslam_amd/csrc/host/pipeline.hpp"

extern "C" int fh_feature_track(const uint8_t *a, const uint8_t *b, int w, const char *slam_yaml, const char *sensor_yaml, int *out3) {
    try {
        xrh::Config cfg = xrh::load_config(slam_yaml, sensor_yaml);
        xrh::Pipeline P(cfg);
        auto map = std::make_unique<xrh::Map>(&P.ids);
        auto pre = [&](xrh::Frame *f) {
            xrh::hip_check(xrhip_image_preprocess(f->image->h, cfg.feature_tracker_clahe_clip_limit, (int)cfg.feature_tracker_clahe_width,
                                                  (int)cfg.feature_tracker_clahe_height),
                           "xrhip_image_preprocess");
        };
        auto f1 = std::make_unique<xrh::Frame>();
        f1->K = cfg.K;
        f1->image = P.make_image(a, w, 0.0, false);
        pre(f1.get());
        xrh::frame_detect_keypoints(P, f1.get());
        out3[0] = (int)f1->keypoint_num();
        map->attach_frame(std::move(f1));
        xrh::Frame *last = map->get_frame(0);
        auto f2 = std::make_unique<xrh::Frame>();
        f2->K = cfg.K;
        f2->image = P.make_image(b, w, 0.05, false);
        pre(f2.get());
        xrh::frame_track_keypoints(P, last, f2.get());
        out3[1] = f2->tag(xrh::FT_NO_TRANSLATION) ? 1 : 0;
        int count = 0;
        for (size_t i = 0; i < f2->keypoint_num(); ++i)
            if (f2->get_track(i)) ++count;
        out3[2] = count;
        return 0;
    } catch (const std::exception &e) {
        std::fprintf(stderr, "fh_feature_track: %s\n", e.what());
        return 1;
    }
}
