// host_select.hpp -- order-defining host logic of the KLT front-end.
//
// These pieces are inherently sequential (greedy selections whose result
// depends on visiting order) and operate on a few hundred points, so they run
// on the host in the product, exactly where the reference runs them:
//   GreedyMinDistance  <- cv::goodFeaturesToTrack's min-distance grid
//                         (reached from opencv_image.cpp:44 via GFTTDetector)
//   PoissonDisk2       <- xrslam::PoissonDiskFilter<2>
//                         (xrslam/src/xrslam/utility/poisson_disk_filter.h:8-113)
#pragma once
#include <cmath>
#include <cstdint>
#include <unordered_map>
#include <vector>

namespace xrhip {

// PoissonDiskFilter<2>: hash grid with cell = r/sqrt(2); a cell remembers the
// LAST point stored in it.  The scan order quirk of the reference (the scan
// box is entered one cell late and left one cell late) is preserved because it
// decides which cells are consulted.
class PoissonDisk2 {
  public:
    explicit PoissonDisk2(double radius)
        : r2_(radius * radius), cell_(radius / std::sqrt(2.0)), span_((int)std::ceil(std::sqrt(2.0))) {}

    void preset(double x, double y) {
        grid_[key(ix(x), ix(y))] = (int)pts_.size() / 2;
        pts_.push_back(x);
        pts_.push_back(y);
    }

    bool permit(double x, double y) const {
        const int cx = ix(x), cy = ix(y);
        const int bx = cx - span_, by = cy - span_, ex = cx + span_, ey = cy + span_;
        int x_it = bx, y_it = by;
        while (y_it <= ey) {
            ++x_it;
            if (x_it > ex) {
                x_it = bx;
                ++y_it;
            }
            auto it = grid_.find(key(x_it, y_it));
            if (it != grid_.end()) {
                const double dx = x - pts_[2 * it->second], dy = y - pts_[2 * it->second + 1];
                if (dx * dx + dy * dy < r2_) return false;
            }
        }
        return true;
    }

    bool insert(double x, double y) {
        if (!permit(x, y)) return false;
        preset(x, y);
        return true;
    }

  private:
    int ix(double v) const { return (int)std::floor(v / cell_); }
    static int64_t key(int a, int b) { return ((int64_t)a << 32) ^ (uint32_t)b; }
    double r2_, cell_;
    int span_;
    std::vector<double> pts_;
    std::unordered_map<int64_t, int> grid_;
};

// goodFeaturesToTrack's greedy spacing.  `next(idx)` yields candidate linear
// indices in the order (response desc, linear index desc) and returns false when
// exhausted.  Returns accepted linear indices.
template <class NextFn>
inline std::vector<int> greedy_min_distance(NextFn next, int w, int h, double min_distance, int max_corners) {
    std::vector<int> out;
    int idx = 0;
    if (min_distance < 1) {
        while (next(idx)) {
            out.push_back(idx);
            if (max_corners > 0 && (int)out.size() == max_corners) break;
        }
        return out;
    }
    const int cell = (int)std::lrint(min_distance);
    const int gw = (w + cell - 1) / cell, gh = (h + cell - 1) / cell;
    std::vector<std::vector<int>> grid((size_t)gw * gh);
    const double md2 = min_distance * min_distance;
    while (next(idx)) {
        const int y = idx / w, x = idx - y * w;
        const int xc = x / cell, yc = y / cell;
        const int x1 = std::max(0, xc - 1), y1 = std::max(0, yc - 1);
        const int x2 = std::min(gw - 1, xc + 1), y2 = std::min(gh - 1, yc + 1);
        bool good = true;
        for (int yy = y1; yy <= y2 && good; ++yy)
            for (int xx = x1; xx <= x2 && good; ++xx)
                for (int other : grid[(size_t)yy * gw + xx]) {
                    const float dx = (float)(x - other % w), dy = (float)(y - other / w);
                    if ((double)(dx * dx + dy * dy) < md2) {
                        good = false;
                        break;
                    }
                }
        if (good) {
            grid[(size_t)yc * gw + xc].push_back(idx);
            out.push_back(idx);
            if (max_corners > 0 && (int)out.size() == max_corners) break;
        }
    }
    return out;
}

}   // namespace xrhip
