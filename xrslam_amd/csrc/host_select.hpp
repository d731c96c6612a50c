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
    // width/height (pixels) size a dense cell array for the image area plus a margin; points outside it (never
    // produced by the tracker, but legal) fall back to a hash map, so results do not depend on the bounds given.
    explicit PoissonDisk2(double radius, int width = 0, int height = 0)
        : r2_(radius * radius), cell_(radius / std::sqrt(2.0)), span_((int)std::ceil(std::sqrt(2.0))) {
        if (width > 0 && height > 0) {
            gw_ = (int)std::floor(width / cell_) + 1 + 2 * kMargin;
            gh_ = (int)std::floor(height / cell_) + 1 + 2 * kMargin;
            dense_.assign((size_t)gw_ * gh_, -1);
        }
    }

    void preset(double x, double y) {
        slot(ix(x), ix(y)) = (int)pts_.size() / 2;
        pts_.push_back(x);
        pts_.push_back(y);
    }

    bool permit(double x, double y) const {
        const int cx = ix(x), cy = ix(y);
        const int bx = cx - span_, by = cy - span_, ex = cx + span_, ey = cy + span_;
        // The scan below visits (bx+1 .. ex, by), then (bx .. ex, y) for y = by+1 .. ey, then (bx, ey+1) -- the reference's one-cell-late
        // box.  When all of it lies inside the dense array (every point the tracker produces), the same cells are read row by row
        // without the per-cell bounds test and hash fallback; the verdict does not depend on the order the cells are consulted in
        // (any stored point closer than the radius forbids).
        if (inside(bx, by) && inside(ex, ey + 1)) {
            for (int yy = by; yy <= ey + 1; ++yy) {
                const int x0 = yy == by ? bx + 1 : bx, x1 = yy == ey + 1 ? bx : ex;
                const int *row = dense_.data() + (size_t)(yy + kMargin) * gw_ + kMargin;
                for (int xx = x0; xx <= x1; ++xx) {
                    const int at = row[xx];
                    if (at >= 0) {
                        const double dx = x - pts_[2 * at], dy = y - pts_[2 * at + 1];
                        if (dx * dx + dy * dy < r2_) return false;
                    }
                }
            }
            return true;
        }
        int x_it = bx, y_it = by;
        while (y_it <= ey) {
            ++x_it;
            if (x_it > ex) {
                x_it = bx;
                ++y_it;
            }
            const int at = lookup(x_it, y_it);
            if (at >= 0) {
                const double dx = x - pts_[2 * at], dy = y - pts_[2 * at + 1];
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
    static constexpr int kMargin = 4;
    int ix(double v) const { return (int)std::floor(v / cell_); }
    static int64_t key(int a, int b) { return ((int64_t)a << 32) ^ (uint32_t)b; }
    bool inside(int a, int b) const { return a >= -kMargin && b >= -kMargin && a < gw_ - kMargin && b < gh_ - kMargin; }
    int lookup(int a, int b) const {
        if (inside(a, b)) return dense_[(size_t)(b + kMargin) * gw_ + (a + kMargin)];
        auto it = sparse_.find(key(a, b));
        return it == sparse_.end() ? -1 : it->second;
    }
    int &slot(int a, int b) {
        if (inside(a, b)) return dense_[(size_t)(b + kMargin) * gw_ + (a + kMargin)];
        return sparse_[key(a, b)];
    }
    double r2_, cell_;
    int span_;
    int gw_ = 0, gh_ = 0;
    std::vector<double> pts_;
    std::vector<int> dense_;
    std::unordered_map<int64_t, int> sparse_;
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
    // cell -> chain of accepted corners (newest first); the spacing test does not depend on the visiting order
    std::vector<int> head((size_t)gw * gh, -1), next_in_cell;
    next_in_cell.reserve(max_corners > 0 ? (size_t)max_corners : 256);
    const double md2 = min_distance * min_distance;
    while (next(idx)) {
        const int y = idx / w, x = idx - y * w;
        const int xc = x / cell, yc = y / cell;
        const int x1 = std::max(0, xc - 1), y1 = std::max(0, yc - 1);
        const int x2 = std::min(gw - 1, xc + 1), y2 = std::min(gh - 1, yc + 1);
        bool good = true;
        for (int yy = y1; yy <= y2 && good; ++yy)
            for (int xx = x1; xx <= x2 && good; ++xx)
                for (int k = head[(size_t)yy * gw + xx]; k >= 0; k = next_in_cell[k]) {
                    const int other = out[k];
                    const float dx = (float)(x - other % w), dy = (float)(y - other / w);
                    if ((double)(dx * dx + dy * dy) < md2) {
                        good = false;
                        break;
                    }
                }
        if (good) {
            next_in_cell.push_back(head[(size_t)yc * gw + xc]);
            head[(size_t)yc * gw + xc] = (int)out.size();
            out.push_back(idx);
            if (max_corners > 0 && (int)out.size() == max_corners) break;
        }
    }
    return out;
}

}   // namespace xrhip
