// parsac.hpp -- the RD-VIO outlier filters: PARSAC over an image grid of bin
// confidences for the 2D-2D essential-matrix check, and its IMU-aided variant for the 3D-2D pose check.
//
// Host-side mirror of (file:line under /root/reference/xrslam/src/xrslam):
//   Sampler                         utility/parsac.h:9-48        (libc rand(), reseeded with srand(0) per solve)
//   Parsac<>::solve                 utility/parsac.h:50-176, 198-381
//   IMU_Parsac<>::solve             utility/imu_parsac.h:9-415
//   find_essential_matrix_parsac    geometry/stereo.cpp:123-153
//   find_pnp_matrix_parsac_imu      geometry/pnp.h:166-204       (6-point EPnP hypotheses: epnp.hpp)
//   pnp_reproject_error             geometry/pnp.h:76-80
//
// Kept as the reference has them (they decide which tracks are called dynamic):
//   * when more than 20 bins are occupied, the index drawn by bin weight is used directly as a DATA index (the
//     reference's make_sample ignores make_sample_by_prior), i.e. the sample comes from the first few correspondences;
//   * the bin confidences written back by one call are the prior of the next one (function-local statics there,
//     members of ParsacState here: one state per tracker instead of one per process);
//   * scores are accumulated in float.
// PARITY UNPINNED against the reference binary (Eigen / OpenCV are not in this image); behaviour is checked on synthetic
// static + moving point sets in tests/test_parsac.py.
#pragma once
#include <cfloat>
#include <cstdlib>

#include "ba_dump.hpp"
#include "epnp.hpp"

namespace xrh {

// XRSLAM_AMD_DUMP_INIT (ba_dump.hpp): while set, every PARSAC run appends what it looked at and what it decided -- the tracker points
// this at its pipeline's logger around judge_track_status / update_track_status (tests/parsac_model.py re-derives the decisions)
inline InitLogger *&parsac_trace() {
    static thread_local InitLogger *t = nullptr;
    return t;
}

struct ParsacState {
    std::vector<float> essential_bins = std::vector<float>(400, 0.5f);
    std::vector<float> pnp_bins = std::vector<float>(400, 0.5f);
};

namespace parsac_detail {

class Sampler {
  public:
    explicit Sampler(const std::vector<float> &acc) : acc_(acc) { std::srand(0); }
    size_t draw_by_weight() {
        size_t index;
        do {
            const float r = std::rand() / (float)RAND_MAX;
            index = size_t(std::upper_bound(acc_.begin() + 1, acc_.end(), r) - acc_.begin() - 1);
        } while (std::find(sampled_.begin(), sampled_.end(), index) != sampled_.end());
        sampled_.push_back(index);
        return index;
    }
    void refill_all() { sampled_.clear(); }

  private:
    const std::vector<float> &acc_;
    std::vector<size_t> sampled_;
};

// 20 x 20 grid over [-norm_scale, norm_scale]^2 in normalised image coordinates; only occupied ("valid") bins count
struct BinGrid {
    size_t nx = 20, ny = 20, nbins = 400, nvalid = 0;
    float bin_w = 0, bin_h = 0;
    double norm_scale = 1.0;
    std::vector<V2> location;
    std::vector<size_t> bin_to_valid, valid_to_bin, data_to_valid, valid_size;
    std::vector<float> valid_len;   // mean track length per valid bin (IMU variant)

    void build(const std::vector<V2> &pts, const std::vector<size_t> *lens) {
        bin_h = (float)(2 * norm_scale / ny);
        bin_w = (float)(2 * norm_scale / nx);
        location.clear();
        float y = bin_h * 0.5f;
        for (size_t i = 0; i < ny; ++i, y += bin_h) {
            float x = bin_w * 0.5f;
            for (size_t j = 0; j < nx; ++j, x += bin_w) location.push_back({x - norm_scale, y - norm_scale});
        }
        data_to_valid.assign(pts.size(), 0);
        bin_to_valid.assign(nbins, SIZE_MAX);
        valid_to_bin.clear();
        valid_size.clear();
        valid_len.clear();
        for (size_t i = 0; i < pts.size(); ++i) {
            // points outside the grid index past it in the reference (undefined behaviour there); clamped here
            const size_t bx = std::min(nx - 1, size_t(std::max(0.0, (pts[i].x + norm_scale) / bin_w)));
            const size_t by = std::min(ny - 1, size_t(std::max(0.0, (pts[i].y + norm_scale) / bin_h)));
            const size_t bin = bx + nx * by;
            size_t v = bin_to_valid[bin];
            if (v == SIZE_MAX) {
                v = valid_to_bin.size();
                bin_to_valid[bin] = v;
                valid_to_bin.push_back(bin);
                valid_size.push_back(0);
                valid_len.push_back(0.f);
            }
            data_to_valid[i] = v;
            valid_size[v]++;
            if (lens) valid_len[v] += (float)(*lens)[i];
        }
        nvalid = valid_size.size();
        if (lens)
            for (size_t v = 0; v < nvalid; ++v) valid_len[v] /= valid_size[v];
    }
    std::vector<size_t> inliers_per_bin(const std::vector<char> &mask) const {
        std::vector<size_t> cnt(nvalid, 0);
        for (size_t i = 0; i < mask.size(); ++i)
            if (mask[i] == 1) cnt[data_to_valid[i]]++;
        return cnt;
    }
    // covered image area x total confidence: confidence-weighted spatial spread of the bins that hold inliers.
    // dynamic_probability < 0 selects the plain variant (parsac.h), otherwise bins are weighted by 1 - p^(0.1 len)
    float score(const std::vector<size_t> &inl, std::vector<float> &conf, double dynamic_probability) const {
        conf.resize(nvalid);
        float csum = 0, csq = 0;
        V2 sum{0, 0};
        for (size_t v = 0; v < nvalid; ++v) {
            float c = float(inl[v]) / valid_size[v];
            if (dynamic_probability >= 0) {
                const float t = 1 - std::pow(dynamic_probability, 0.10 * valid_len[v]);
                c = t * float(inl[v]) / valid_size[v];
            }
            conf[v] = c;
            const V2 &x = location[valid_to_bin[v]];
            sum.x += x.x * c;
            sum.y += x.y * c;
            csum += c;
            csq += c * c;
        }
        float norm = 1.f / csum;
        const V2 mean{sum.x * norm, sum.y * norm};
        float cxx = 0, cxy = 0, cyy = 0;
        for (size_t v = 0; v < nvalid; ++v) {
            const V2 &x = location[valid_to_bin[v]];
            const double dx = x.x - mean.x, dy = x.y - mean.y;
            cxx += (dx * dx) * conf[v];
            cxy += (dx * dy) * conf[v];
            cyy += (dy * dy) * conf[v];
        }
        norm = csum / (csum * csum - csq);
        const float ratio = norm * std::sqrt(cxx * cyy - cxy * cxy);
        return ratio * csum;
    }
    std::vector<float> accumulated_prior(const std::vector<float> &bin_conf, float floor_conf) const {
        std::vector<float> c(nvalid);
        float sum = 0;
        for (size_t v = 0; v < nvalid; ++v) {
            c[v] = std::max(floor_conf, bin_conf[valid_to_bin[v]]);
            sum += c[v];
        }
        const float norm = 1.0 / sum;
        for (float &v : c) v *= norm;
        std::vector<float> acc(nvalid + 1);
        acc[0] = 0;
        for (size_t v = 0; v < nvalid; ++v) acc[v + 1] = acc[v] + c[v];
        const float n2 = 1.f / acc[nvalid];
        for (size_t v = 0; v < nvalid; ++v) acc[v] *= n2;
        return acc;
    }
    void write_back(const std::vector<float> &conf, std::vector<float> &bin_conf) const {
        bin_conf.assign(nbins, 0.f);
        for (size_t b = 0; b < nbins; ++b)
            if (bin_to_valid[b] != SIZE_MAX) bin_conf[b] = conf[bin_to_valid[b]];
    }
};

// Shared loop of Parsac<>::solve and IMU_Parsac<>::solve.
//   Model / Solver / Evaluator as in ransac_solve; `prior_mask` (IMU variant) restricts the inlier count that drives the
//   termination to correspondences the IMU-predicted model also accepts, and hypotheses with fewer than DoF of
//   those are skipped.  Returns false when the IMU variant gives up (caller then reports "everything is an inlier").
template <size_t DoF, class Model, class S1, class S2, class Solver, class MakeEval>
bool solve(const std::vector<S1> &d1, const std::vector<S2> &d2, const std::vector<V2> &grid_pts, const std::vector<size_t> *lens,
           double dynamic_probability, double norm_scale, const std::vector<char> *prior_mask, double threshold, double confidence,
           size_t max_iteration, int seed, std::vector<float> &bin_conf, Solver solver, MakeEval make_eval, Model &model,
           std::vector<char> &inlier_mask) {
    const size_t size = d1.size();
    LotBox lotbox(size);
    lotbox.seed((unsigned)seed);
    const double K = std::log(std::max(1 - confidence, 1.0e-5));
    size_t inlier_count = 0;
    if (size < DoF) {
        inlier_mask.assign(size, 0);
        return true;
    }
    BinGrid grid;
    grid.norm_scale = norm_scale;
    grid.build(grid_pts, lens);
    const std::vector<float> acc = grid.accumulated_prior(bin_conf, 0.5f);
    Sampler sampler(acc);
    std::vector<size_t> best_bins;
    std::vector<float> conf;
    size_t iter_max = max_iteration;
    float score_max = prior_mask ? -FLT_MAX : 0.f;
    // the run's record (only with the decision log on): inputs now, one entry per scored hypothesis, the outcome at the end
    InitLogger *trace = parsac_trace() && parsac_trace()->enabled() ? parsac_trace() : nullptr;
    const std::vector<float> bins_before = bin_conf;
    std::vector<double> t_iter, t_counted, t_score, t_best, t_itermax, t_masks, t_samples;
    size_t iterations_run = 0;
    for (size_t iter = 0; iter < iter_max; ++iter) {
        iterations_run = iter + 1;
        std::array<S1, DoF> s1;
        std::array<S2, DoF> s2;
        lotbox.refill_all();
        sampler.refill_all();
        for (size_t si = 0; si < DoF; ++si) {
            const size_t idx = grid.nvalid > 20 ? sampler.draw_by_weight() : lotbox.draw_without_replacement();
            s1[si] = d1[idx];
            s2[si] = d2[idx];
            if (trace) t_samples.push_back((double)idx);
        }
        for (const Model &cur : solver(s1, s2)) {
            size_t cur_count = 0;
            std::vector<char> cur_mask(size, 0);
            auto eval = make_eval(cur);
            for (size_t i = 0; i < size; ++i)
                if (eval(d1[i], d2[i]) <= threshold) {
                    cur_count++;
                    cur_mask[i] = 1;
                }
            size_t counted = cur_count;
            if (prior_mask) {
                counted = 0;
                for (size_t i = 0; i < size; ++i)
                    if ((*prior_mask)[i] && cur_mask[i]) counted++;
                if (counted < DoF) continue;
            }
            const std::vector<size_t> bins = grid.inliers_per_bin(cur_mask);
            const float score = grid.score(bins, conf, prior_mask ? dynamic_probability : -1.0);
            const bool takes_over = score > score_max || (score == score_max && counted > inlier_count);
            if (trace) {
                t_iter.push_back((double)iter);
                t_counted.push_back((double)counted);
                t_score.push_back((double)score);
                t_best.push_back(takes_over ? 1.0 : 0.0);
                for (char m : cur_mask) t_masks.push_back((double)m);
            }
            if (takes_over) {
                score_max = score;
                model = cur;
                inlier_count = counted;
                best_bins = bins;
                inlier_mask.swap(cur_mask);
                const double ratio = inlier_count / (double)size;
                const double N = K / std::log(1 - std::pow(ratio, 5));
                if (N < (double)iter_max) iter_max = (size_t)std::ceil(N);
            }
            if (trace) t_itermax.push_back((double)iter_max);
        }
    }
    const bool gave_up = prior_mask && inlier_count < DoF;
    if (!gave_up) {
        if (best_bins.size() != grid.nvalid) best_bins.assign(grid.nvalid, 0);   // no hypothesis scored: all-zero confidences
        grid.score(best_bins, conf, prior_mask ? dynamic_probability : -1.0);
        grid.write_back(conf, bin_conf);
    }
    if (trace) {
        InitLogger::Line ln(*trace, "parsac_run");
        std::vector<double> gp, ln_len, pm, bb(bins_before.begin(), bins_before.end()), ba(bin_conf.begin(), bin_conf.end()),
            ac(acc.begin(), acc.end()), fm;
        for (const V2 &p : grid_pts) {
            gp.push_back(p.x);
            gp.push_back(p.y);
        }
        if (lens)
            for (size_t v : *lens) ln_len.push_back((double)v);
        if (prior_mask)
            for (char m : *prior_mask) pm.push_back((double)m);
        for (char m : inlier_mask) fm.push_back((double)m);
        ln.put("dof", (double)DoF); ln.put("imu", prior_mask ? 1.0 : 0.0); ln.put("size", (double)size);
        ln.put("threshold", threshold); ln.put("confidence", confidence); ln.put("max_iteration", (double)max_iteration);
        ln.put("dynamic_probability", dynamic_probability); ln.put("norm_scale", norm_scale);
        ln.put("grid_pts", gp); ln.put("lens", ln_len); ln.put("prior_mask", pm);
        ln.put("bins_before", bb); ln.put("accumulated_prior", ac); ln.put("nvalid", (double)grid.nvalid);
        ln.put("samples", t_samples); ln.put("cand_iter", t_iter); ln.put("cand_counted", t_counted); ln.put("cand_score", t_score);
        ln.put("cand_takes_over", t_best); ln.put("cand_iter_max_after", t_itermax); ln.put("cand_masks", t_masks);
        ln.put("iterations_run", (double)iterations_run); ln.put("gave_up", gave_up ? 1.0 : 0.0);
        ln.put("inlier_count", (double)inlier_count); ln.put("final_mask", fm); ln.put("bins_after", ba);
    }
    return !gave_up;
}

}   // namespace parsac_detail

inline M3 find_essential_matrix_parsac(ParsacState &st, const std::vector<V2> &p1, const std::vector<V2> &p2, std::vector<char> &mask,
                                       double threshold = 1.0, double confidence = 0.999, size_t max_iteration = 1000, int seed = 0) {
    mask.clear();
    auto solver = [](const std::array<V2, 5> &a, const std::array<V2, 5> &b) { return solve_essential_5pt(a, b); };
    auto make_eval = [](const M3 &E) {
        M3 Et = transpose(E);
        return [E, Et](V2 a, V2 b) { return essential_geometric_error(E, a, b) + essential_geometric_error(Et, b, a); };
    };
    M3 E;
    parsac_detail::solve<5>(p1, p2, p2, nullptr, -1.0, 1.0, nullptr, 2.0 * 3.84 * threshold * threshold, confidence, max_iteration, seed,
                            st.essential_bins, solver, make_eval, E, mask);
    return E;
}

inline double pnp_reproject_error(const Pose34 &T, const V3 &X, const V2 &x) {
    const V3 q = T.R * X + T.t;
    const double dx = x.x - q.x / q.z, dy = x.y - q.y / q.z;
    return dx * dx + dy * dy;
}

// Xs: landmarks (world), xs: normalised image points, lens: track lengths, (R, t): camera-from-world pose predicted by the
// IMU.  mask: 1 = consistent with the static scene.  Returns the identity pose with an all-ones mask when it gives up
// (too few points agree with the prediction, or no hypothesis kept DoF of them), like the reference.
inline Pose34 find_pnp_matrix_parsac_imu(ParsacState &st, const std::vector<V3> &Xs, const std::vector<V2> &xs,
                                         const std::vector<size_t> &lens, const M3 &R, const V3 &t, double dynamic_prob, double scale,
                                         std::vector<char> &mask, double threshold = 1.0, double confidence = 0.999,
                                         size_t max_iteration = 1000, int seed = 0) {
    const double thr = 2.0 * 5.99 * threshold * threshold;
    const size_t size = Xs.size();
    Pose34 prior;
    prior.R = R;
    prior.t = t;
    Pose34 model;
    mask.clear();
    if (size >= 6) {
        std::vector<char> prior_mask(size, 0);
        size_t agree = 0;
        for (size_t i = 0; i < size; ++i)
            if (pnp_reproject_error(prior, Xs[i], xs[i]) <= thr * 2.0) {
                prior_mask[i] = 1;
                agree++;
            }
        if ((double)agree / size < 0.15 || agree < 20) {
            mask.assign(size, 1);
            return Pose34();
        }
        auto solver = [](const std::array<V3, 6> &a, const std::array<V2, 6> &b) {
            return std::vector<Pose34>{solve_pnp_epnp(a.data(), b.data(), 6)};
        };
        auto make_eval = [](const Pose34 &T) { return [T](const V3 &X, const V2 &x) { return pnp_reproject_error(T, X, x); }; };
        if (!parsac_detail::solve<6>(Xs, xs, xs, &lens, dynamic_prob, scale, &prior_mask, thr, confidence, max_iteration, seed, st.pnp_bins,
                                     solver, make_eval, model, mask)) {
            mask.assign(size, 1);
            return Pose34();
        }
        return model;
    }
    mask.assign(size, 0);
    return model;
}

}   // namespace xrh
