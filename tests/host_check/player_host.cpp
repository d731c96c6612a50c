// ctypes hooks for tests/test_player_io.py: the host-side I/O of the headless player (no HIP, no GPU).
#include <cstring>

#include "../../xrslam_amd/csrc/player/euroc_io.hpp"

using namespace xrplayer;

extern "C" {

// decodes a PNG file; returns 0 and fills w/h (+ pixels when out != nullptr and cap is large enough)
int ph_decode_png(const char *path, int *w, int *h, unsigned char *out, long cap) {
    try {
        GrayImage img = decode_png(read_file(path));
        *w = img.w;
        *h = img.h;
        if (out && cap >= (long)img.px.size()) std::memcpy(out, img.px.data(), img.px.size());
        return 0;
    } catch (const std::exception &) {
        return 1;
    }
}

void ph_undistort(const unsigned char *src, int w, int h, const double *K4, const double *D4, unsigned char *dst) {
    Undistorter u(w, h, K4, D4);
    u.apply(src, w, dst, w);
}

// the reference's ImageUndistorter maps ("radtan" / "equidistant"); returns 1 for an unknown model
int ph_undistort_model(const unsigned char *src, int w, int h, const double *K4, const double *D, int nd, const char *model,
                       unsigned char *dst) {
    try {
        Undistorter u(w, h, K4, std::vector<double>(D, D + nd), model);
        u.apply(src, w, dst, w);
        return 0;
    } catch (const std::exception &) {
        return 1;
    }
}

// event order of a directory: writes up to cap (type, index) pairs, returns the number of events
long ph_merge(const char *root, double camera_offset, int *types, long *index, double *times, long cap) {
    const std::vector<CameraRow> cam = load_camera_csv(std::string(root) + "/cam0/data.csv");
    const std::vector<ImuRow> imu = load_imu_csv(std::string(root) + "/imu0/data.csv");
    const std::vector<Event> ev = merge_events(cam, imu, camera_offset);
    for (size_t i = 0; i < ev.size() && (long)i < cap; ++i) {
        types[i] = (int)ev[i].type;
        index[i] = (long)ev[i].index;
        times[i] = ev[i].t;
    }
    return (long)ev.size();
}

long ph_load_truth(const char *path, double *rows17, long cap) {
    const std::vector<TruthRow> gt = load_groundtruth_csv(path);
    for (size_t i = 0; i < gt.size() && (long)i < cap; ++i) {
        double *r = rows17 + 17 * i;
        r[0] = gt[i].t;
        for (int k = 0; k < 3; ++k) r[1 + k] = gt[i].p[k];
        for (int k = 0; k < 4; ++k) r[4 + k] = gt[i].q[k];
        for (int k = 0; k < 3; ++k) {
            r[8 + k] = gt[i].v[k];
            r[11 + k] = gt[i].bg[k];
            r[14 + k] = gt[i].ba[k];
        }
    }
    return (long)gt.size();
}

double ph_ate(const double *est, const double *ref, long n) {
    std::vector<xrh::V3> a, b;
    for (long i = 0; i < n; ++i) {
        a.push_back({est[3 * i], est[3 * i + 1], est[3 * i + 2]});
        b.push_back({ref[3 * i], ref[3 * i + 1], ref[3 * i + 2]});
    }
    return ate_rmse(a, b);
}

void ph_tum_line(double t, const double *p, const double *q, char *out, long cap) {
    FILE *f = fmemopen(out, (size_t)cap, "w");
    write_tum_pose(f, t, p, q);
    std::fclose(f);
}

void ph_csv_line(double t, const double *p, const double *q, char *out, long cap) {
    FILE *f = fmemopen(out, (size_t)cap, "w");
    write_csv_pose(f, t, p, q);
    std::fclose(f);
}

// 0: no known scheme, 1: euroc, 2: tum; the directory goes to `path`
int ph_split_url(const char *url, char *path, long cap) {
    const auto [scheme, dir] = split_dataset_url(url);
    std::snprintf(path, (size_t)cap, "%s", dir.c_str());
    return scheme == "euroc" ? 1 : scheme == "tum" ? 2 : 0;
}

}   // extern "C"
