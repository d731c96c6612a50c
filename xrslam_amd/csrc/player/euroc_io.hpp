// euroc_io.hpp -- data formats either side of the hot path: what the reference's
// player reads and writes around the XRSLAM.h calls.  Host-only, no HIP, no third-party library but zlib.
//
//   EuRoC/ASL directory      cam0/data.csv, imu0/data.csv, state_groundtruth_estimate0/data.csv
//                            (xrslam-pc/player/src/IO/euroc_dataset_reader.h:28-117, .cpp:3-105)
//   event order              gyroscope, then accelerometer, then camera at equal time stamps -- the order the player's
//                            asynchronous reader yields (IO/async_dataset_reader.cpp:16-49)
//   PNG                      8-bit grey / RGB(A) non-interlaced, what cv::imread(IMREAD_UNCHANGED) is fed on EuRoC
//   radial-tangential undistortion   cv::undistort restated (fixed-point bilinear remap), euroc_dataset_reader.cpp:62-69
//   TUM trajectory           "%.18e %.9e %.9e %.9e %.7e %.7e %.7e %.7e\n"  (IO/trajectory_writer.h:54-76)
//   ATE                      RMSE after SE(3) Umeyama alignment (docs/en/tutorials/euroc_evaluation.md: evo_ape tum -a)
#pragma once
#include <zlib.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../host/hla.hpp"
#include "../host/undistort_map.hpp"

namespace xrplayer {

// ----------------------------------------------------------------------------------------------- CSV
struct CameraRow {
    double t;   // seconds
    std::string filename;
};
struct ImuRow {
    double t;
    double w[3], a[3];
};
struct TruthRow {
    double t;
    double p[3], q[4];   // q as stored by the ASL format: w, x, y, z
    double v[3], bg[3], ba[3];
};

inline bool read_line(FILE *f, std::string &line) {
    line.clear();
    int ch;
    while ((ch = std::fgetc(f)) != EOF) {
        if (ch == '\n') return true;
        if (ch != '\r') line.push_back((char)ch);
    }
    return !line.empty();
}

// Time stamps are 19-digit nanosecond integers; like the reference they are read as a double and scaled by 1e-9.
inline std::vector<CameraRow> load_camera_csv(const std::string &path) {
    std::vector<CameraRow> rows;
    FILE *f = std::fopen(path.c_str(), "r");
    if (!f) return rows;
    std::string line;
    while (read_line(f, line)) {
        if (line.empty() || line[0] == '#') continue;
        const size_t comma = line.find(',');
        if (comma == std::string::npos) break;
        CameraRow r;
        r.t = std::strtod(line.substr(0, comma).c_str(), nullptr) * 1e-9;
        r.filename = line.substr(comma + 1);
        rows.push_back(r);
    }
    std::fclose(f);
    return rows;
}

inline std::vector<double> split_numbers(const std::string &line) {
    std::vector<double> v;
    const char *s = line.c_str();
    while (*s) {
        char *end = nullptr;
        const double x = std::strtod(s, &end);
        if (end == s) break;
        v.push_back(x);
        s = end;
        while (*s == ',' || *s == ' ') ++s;
    }
    return v;
}

inline std::vector<ImuRow> load_imu_csv(const std::string &path) {
    std::vector<ImuRow> rows;
    FILE *f = std::fopen(path.c_str(), "r");
    if (!f) return rows;
    std::string line;
    while (read_line(f, line)) {
        if (line.empty() || line[0] == '#') continue;
        const std::vector<double> v = split_numbers(line);
        if (v.size() < 7) break;
        ImuRow r;
        r.t = v[0] * 1e-9;
        for (int i = 0; i < 3; ++i) {
            r.w[i] = v[1 + i];
            r.a[i] = v[4 + i];
        }
        rows.push_back(r);
    }
    std::fclose(f);
    return rows;
}

inline std::vector<TruthRow> load_groundtruth_csv(const std::string &path) {
    std::vector<TruthRow> rows;
    FILE *f = std::fopen(path.c_str(), "r");
    if (!f) return rows;
    std::string line;
    while (read_line(f, line)) {
        if (line.empty() || line[0] == '#') continue;
        const std::vector<double> v = split_numbers(line);
        if (v.size() < 8) break;
        TruthRow r{};
        r.t = v[0] * 1e-9;
        for (int i = 0; i < 3; ++i) r.p[i] = v[1 + i];
        for (int i = 0; i < 4; ++i) r.q[i] = v[4 + i];
        if (v.size() >= 17)
            for (int i = 0; i < 3; ++i) {
                r.v[i] = v[8 + i];
                r.bg[i] = v[11 + i];
                r.ba[i] = v[14 + i];
            }
        rows.push_back(r);
    }
    std::fclose(f);
    return rows;
}

// ----------------------------------------------------------------------------------------- event merge
enum EventType { EV_GYROSCOPE = 0, EV_ACCELEROMETER = 1, EV_CAMERA = 2 };
struct Event {
    double t;
    EventType type;
    size_t index;   // into the imu / camera rows
};

// The order AsyncDatasetReader::next() produces: the gyroscope sample wins ties against everything, the
// accelerometer sample wins ties against the camera.
inline std::vector<Event> merge_events(const std::vector<CameraRow> &cam, const std::vector<ImuRow> &imu,
                                       double camera_time_offset) {
    std::vector<Event> ev;
    ev.reserve(cam.size() + 2 * imu.size());
    size_t ig = 0, ia = 0, ic = 0;
    const double inf = 1.7976931348623157e308;
    while (ig < imu.size() || ia < imu.size() || ic < cam.size()) {
        const double tg = ig < imu.size() ? imu[ig].t : inf;
        const double ta = ia < imu.size() ? imu[ia].t : inf;
        const double tc = ic < cam.size() ? cam[ic].t + camera_time_offset : inf;
        if (tg <= tc && tg <= ta) {
            ev.push_back({tg, EV_GYROSCOPE, ig++});
        } else if (ta < tg && ta <= tc) {
            ev.push_back({ta, EV_ACCELEROMETER, ia++});
        } else {
            ev.push_back({tc, EV_CAMERA, ic++});
        }
    }
    return ev;
}

// ------------------------------------------------------------------------------------------------ PNG
struct GrayImage {
    int w = 0, h = 0;
    std::vector<uint8_t> px;   // row-major, stride w
};

inline uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

// 8-bit, non-interlaced, colour types 0 (grey), 2 (RGB), 4 (grey+alpha), 6 (RGBA).  Colour is reduced with the
// integer weights of cv::cvtColor(BGR2GRAY): (R*4899 + G*9617 + B*1868 + 8192) >> 14.
inline GrayImage decode_png(const std::vector<uint8_t> &file) {
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (file.size() < 8 || std::memcmp(file.data(), sig, 8) != 0) throw std::runtime_error("png: bad signature");
    size_t pos = 8;
    int w = 0, h = 0, depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> zdata;
    while (pos + 12 <= file.size()) {
        const uint32_t len = be32(&file[pos]);
        const char *type = reinterpret_cast<const char *>(&file[pos + 4]);
        const uint8_t *data = &file[pos + 8];
        if (pos + 12 + len > file.size()) throw std::runtime_error("png: truncated chunk");
        if (!std::memcmp(type, "IHDR", 4)) {
            if (len != 13) throw std::runtime_error("png: bad IHDR length");   // the fields below are read at fixed offsets
            w = (int)be32(data);
            h = (int)be32(data + 4);
            depth = data[8];
            ctype = data[9];
            interlace = data[12];
        } else if (!std::memcmp(type, "IDAT", 4)) {
            zdata.insert(zdata.end(), data, data + len);
        } else if (!std::memcmp(type, "IEND", 4)) {
            break;
        }
        pos += 12 + len;
    }
    // 16-bit samples (TUM-VI frames) are reduced to their high byte: cv::imread(IMREAD_GRAYSCALE) asks libpng to strip
    // them to 8 bits (png_set_strip_16) before any colour conversion
    if (w <= 0 || h <= 0 || (depth != 8 && depth != 16) || interlace != 0)
        throw std::runtime_error("png: only 8- or 16-bit non-interlaced images");
    if (w > 16384 || h > 16384) throw std::runtime_error("png: implausible image size");   // bounds the allocation below
    int ch = 0;
    if (ctype == 0) ch = 1;
    else if (ctype == 2) ch = 3;
    else if (ctype == 4) ch = 2;
    else if (ctype == 6) ch = 4;
    else throw std::runtime_error("png: unsupported colour type");
    const int bps = depth / 8;            // bytes per sample (big-endian)
    const int bpp = ch * bps;             // bytes per pixel: the distance the PNG filters look back
    const size_t stride = (size_t)w * bpp;
    std::vector<uint8_t> raw((stride + 1) * (size_t)h);
    uLongf out_len = (uLongf)raw.size();
    if (uncompress(raw.data(), &out_len, zdata.data(), (uLong)zdata.size()) != Z_OK || out_len != raw.size())
        throw std::runtime_error("png: inflate failed");
    std::vector<uint8_t> cur(stride), prev(stride, 0);
    GrayImage img;
    img.w = w;
    img.h = h;
    img.px.resize((size_t)w * h);
    for (int y = 0; y < h; ++y) {
        const uint8_t *line = &raw[(stride + 1) * (size_t)y];
        const int filter = line[0];
        for (size_t i = 0; i < stride; ++i) {
            const int a = i >= (size_t)bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= (size_t)bpp ? prev[i - bpp] : 0;
            int pred = 0;
            if (filter == 1) pred = a;
            else if (filter == 2) pred = b;
            else if (filter == 3) pred = (a + b) >> 1;
            else if (filter == 4) {
                const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
                pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
            } else if (filter != 0) {
                throw std::runtime_error("png: bad filter");
            }
            cur[i] = (uint8_t)(line[1 + i] + pred);
        }
        uint8_t *dst = &img.px[(size_t)y * w];
        for (int x = 0; x < w; ++x) {
            const uint8_t *px = &cur[(size_t)x * bpp];
            if (ch <= 2) dst[x] = px[0];
            else dst[x] = (uint8_t)((px[0] * 4899 + px[bps] * 9617 + px[2 * bps] * 1868 + 8192) >> 14);
        }
        prev.swap(cur);
    }
    return img;
}

inline std::vector<uint8_t> read_file(const std::string &path) {
    std::vector<uint8_t> buf;
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) throw std::runtime_error("cannot open " + path);
    std::fseek(f, 0, SEEK_END);
    const long n = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    buf.resize((size_t)std::max(n, 0L));
    if (n > 0 && std::fread(buf.data(), 1, (size_t)n, f) != (size_t)n) {
        std::fclose(f);
        throw std::runtime_error("short read on " + path);
    }
    std::fclose(f);
    return buf;
}

// -------------------------------------------------------------------------------------- undistortion
// cv::undistort(src, dst, K, D) for the radial-tangential model (k1, k2, p1, p2) and the maps of
// xrslam::extra::ImageUndistorter ("radtan", "equidistant"): the packed 1/32-pixel map of host/undistort_map.hpp (which
// the library's device remap shares) applied on the host -- what the reference's reader thread does with OpenCV.
class Undistorter {
  public:
    Undistorter(int w, int h, const double K4[4], const double D4[4]) : w_(w), h_(h), map_(xrh::build_undistort_map(w, h, K4, D4, 4, "cv_undistort")) {}
    Undistorter(int w, int h, const double K4[4], const std::vector<double> &D, const std::string &model)
        : w_(w), h_(h), map_(xrh::build_undistort_map(w, h, K4, D.data(), (int)D.size(), model)) {}
    void apply(const uint8_t *src, int src_stride, uint8_t *dst, int dst_stride) const {
        xrh::remap_packed(map_.data(), w_, h_, src, src_stride, dst, dst_stride);
    }

  private:
    int w_, h_;
    std::vector<uint32_t> map_;
};

// ------------------------------------------------------------------------------------------------- TUM
inline void write_tum_pose(FILE *f, double t, const double p[3], const double q_xyzw[4]) {
    std::fprintf(f, "%.18e %.9e %.9e %.9e %.7e %.7e %.7e %.7e\n", t, p[0], p[1], p[2], q_xyzw[0], q_xyzw[1], q_xyzw[2], q_xyzw[3]);
}

// CsvTrajectoryWriter::write_pose (IO/trajectory_writer.h:46-52): the same fields, comma separated
inline void write_csv_pose(FILE *f, double t, const double p[3], const double q_xyzw[4]) {
    std::fprintf(f, "%.18e,%.9e,%.9e,%.9e,%.7e,%.7e,%.7e,%.7e\n", t, p[0], p[1], p[2], q_xyzw[0], q_xyzw[1], q_xyzw[2], q_xyzw[3]);
}

// DatasetReader::create_reader (IO/dataset_reader.cpp:17-34): "euroc://<dir>" or "tum://<dir>"; returns the scheme
// ("euroc" / "tum") and the directory, or an empty scheme for anything else (the reference then prints
// 'Cannot open "<input>"' and fails)
inline std::pair<std::string, std::string> split_dataset_url(const std::string &url) {
    for (const char *scheme : {"euroc", "tum"}) {
        const std::string prefix = std::string(scheme) + "://";
        if (url.compare(0, prefix.size(), prefix) == 0) return {scheme, url.substr(prefix.size())};
    }
    return {"", url};
}

// ------------------------------------------------------------------------------------------------- ATE
// RMSE of |R est_i + t - ref_i| with (R, t) the least-squares rigid alignment (Umeyama without scale).
inline double ate_rmse(const std::vector<xrh::V3> &est, const std::vector<xrh::V3> &ref) {
    const size_t n = std::min(est.size(), ref.size());
    if (n < 3) return std::nan("");
    xrh::V3 me{0, 0, 0}, mr{0, 0, 0};
    for (size_t i = 0; i < n; ++i) {
        me = me + est[i];
        mr = mr + ref[i];
    }
    me = me * (1.0 / n);
    mr = mr * (1.0 / n);
    xrh::Dense cov(3, 3);   // sum (ref - mr)(est - me)^T
    for (size_t i = 0; i < n; ++i) {
        const xrh::V3 a = ref[i] - mr, b = est[i] - me;
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) cov(r, c) += a[r] * b[c];
    }
    std::vector<double> s;
    xrh::Dense V, U;
    xrh::jacobi_svd(cov, s, V, &U);   // cov = U diag(s) V^T
    xrh::M3 Um, Vm;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            Um(r, c) = U(r, c);
            Vm(r, c) = V(r, c);
        }
    xrh::M3 D = xrh::M3::identity();
    if (xrh::det(Um) * xrh::det(Vm) < 0) D(2, 2) = -1.0;
    const xrh::M3 R = Um * D * xrh::transpose(Vm);
    double se = 0;
    for (size_t i = 0; i < n; ++i) {
        const xrh::V3 e = R * (est[i] - me) + mr - ref[i];
        se += xrh::dot(e, e);
    }
    return std::sqrt(se / n);
}

}   // namespace xrplayer
