// map.hpp -- Frame / Track / Map graph of the host pipeline.
//
// Mirrors the reference's data model (file:line under /root/reference/xrslam/src/xrslam):
//   Frame   map/frame.h:24-82, map/frame.cpp:20-53,176-187
//   Track   map/track.h:23-81, map/track.cpp:6-101
//   Map     map/map.h:14-72,  map/map.cpp:17-124
//   ids     utility/identifiable.h:8-34 (one counter per type; instance-scoped here so that several
//           sequences can live in one process -- SURVEY.md section 8e)
//   tags    map/frame.h:17-22, map/track.h:13-21
// The bookkeeping defines keypoint indices and track ids, which must match the reference bit-exactly.
#pragma once
#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <optional>
#include <vector>

#include "../../../include/xrslam_hip.h"
#include "geometry.hpp"

namespace xrh {

inline constexpr size_t nil() { return size_t(-1); }

struct ImuData {
    double t;
    V3 w, a;
};

enum FrameTag { FT_KEYFRAME = 0, FT_NO_TRANSLATION, FT_FIX_POSE, FT_FIX_MOTION, FT_COUNT };
enum TrackTag { TT_VALID = 0, TT_TRIANGULATED, TT_FIX_INVD, TT_TRASH, TT_STATIC, TT_OUTLIER, TT_TEMP, TT_COUNT };

struct Extrinsic {
    Quat q_cs;
    V3 p_cs;
};
struct PoseState {
    Quat q;
    V3 p;
};
struct MotionState {
    V3 v, bg, ba;
};

class Frame;
class Track;
class Map;
struct Pipeline;

// device image handle (== one OpenCvImage); buffers come from / return to the pipeline's pool
struct HipImage {
    xrhip_image *h = nullptr;
    Pipeline *owner = nullptr;
    double t = 0;
    int w = 0, hgt = 0;
    bool preprocessed = false;   // the CLAHE / pyramid / gradient launches were queued when the frame was handed over (Pipeline::make_image)
    ~HipImage();
    void release_image_buffer();
};

// PreIntegrator (estimation/preintegrator.h:11-48); the arithmetic runs on the GPU (xrhip_ba_preintegrate)
struct PreInt {
    std::vector<ImuData> data;
    double rec[XRHIP_IMU_DIM];   // delta + jacobian + sqrt_inv_cov of the last integrate()
    bool valid = false;
    PreInt() {
        for (double &v : rec) v = 0.0;
        rec[4] = 1.0;
    }
    double dt() const { return rec[0]; }
    Quat dq() const { return {rec[1], rec[2], rec[3], rec[4]}; }
    V3 dp() const { return {rec[5], rec[6], rec[7]}; }
    V3 dv() const { return {rec[8], rec[9], rec[10]}; }
};

struct IdSource {
    size_t frame = 0, track = 0;
};

class Frame {
  public:
    size_t id = 0;
    unsigned long ba_gen = 0;   // problem-assembly scratch (BaBuilder): generation stamp and index in that problem
    int ba_index = -1;
    bool tags[FT_COUNT] = {false, false, false, false};
    Map *map = nullptr;
    Intrinsics K;
    double sqrt_inv_cov[2] = {1, 1};
    std::shared_ptr<HipImage> image;
    PoseState pose;
    MotionState motion;
    Extrinsic camera, imu;
    PreInt preintegration, keyframe_preintegration;
    std::vector<std::unique_ptr<Frame>> subframes;
    std::vector<V3> bearings;
    std::vector<Track *> tracks;

    bool &tag(FrameTag t) { return tags[t]; }
    bool tag(FrameTag t) const { return tags[t]; }
    size_t keypoint_num() const { return bearings.size(); }
    const V3 &get_keypoint(size_t i) const { return bearings[i]; }
    Track *get_track(size_t i) const { return tracks[i]; }
    Track *get_track(size_t i, Map *allocation_map);   // creates the track when absent (frame.cpp:44-53)
    void append_keypoint(const V3 &b) {
        bearings.push_back(b);
        tracks.push_back(nullptr);
    }
    std::unique_ptr<Frame> clone() const {   // frame.cpp:20-36: same id and tags, no track links, no map
        auto f = std::make_unique<Frame>();
        f->id = id;
        for (int i = 0; i < FT_COUNT; ++i) f->tags[i] = tags[i];
        f->K = K;
        f->sqrt_inv_cov[0] = sqrt_inv_cov[0];
        f->sqrt_inv_cov[1] = sqrt_inv_cov[1];
        f->image = image;
        f->pose = pose;
        f->motion = motion;
        f->camera = camera;
        f->imu = imu;
        f->preintegration = preintegration;
        f->bearings = bearings;
        f->tracks.assign(bearings.size(), nullptr);
        return f;
    }
    PoseState get_pose(const Extrinsic &s) const { return {pose.q * s.q_cs, pose.p + pose.q * s.p_cs}; }
    void set_pose(const Extrinsic &s, const PoseState &ps) {
        pose.q = ps.q * s.q_cs.conjugate();
        pose.p = ps.p - pose.q * s.p_cs;
    }
};

struct FrameIdLess {
    bool operator()(const Frame *a, const Frame *b) const { return a->id < b->id; }
};

// The observation list of a track (reference: std::map<Frame*, size_t, compare<Frame*>>, track.h).  A track is seen
// in a dozen frames at most, so the ordered map is a flat vector sorted by frame id: same iteration order and
// interface subset, no node allocation per observation (tracks gain ~150 observations per frame).
class FrameRefs {
  public:
    using value_type = std::pair<Frame *, size_t>;
    using const_iterator = std::vector<value_type>::const_iterator;
    const_iterator begin() const { return v_.begin(); }
    const_iterator end() const { return v_.end(); }
    size_t size() const { return v_.size(); }
    bool empty() const { return v_.empty(); }
    const_iterator find(Frame *f) const {
        auto it = lower(f);
        return (it != v_.end() && !FrameIdLess()(f, it->first)) ? it : v_.end();
    }
    size_t count(Frame *f) const { return find(f) != v_.end() ? 1 : 0; }
    size_t at(Frame *f) const {
        auto it = find(f);
        if (it == v_.end()) throw std::out_of_range("FrameRefs::at");
        return it->second;
    }
    size_t &operator[](Frame *f) {
        auto it = v_.begin() + (lower(f) - v_.begin());
        if (it == v_.end() || FrameIdLess()(f, it->first)) it = v_.insert(it, value_type(f, 0));
        return it->second;
    }
    void erase(Frame *f) {
        auto it = find(f);
        if (it != v_.end()) v_.erase(v_.begin() + (it - v_.begin()));
    }

  private:
    const_iterator lower(Frame *f) const {
        return std::lower_bound(v_.begin(), v_.end(), f, [](const value_type &a, Frame *b) { return FrameIdLess()(a.first, b); });
    }
    std::vector<value_type> v_;
};

struct LandmarkState {
    double inv_depth = 0, reprojection_error = 0;
};

class Track {
  public:
    size_t id = 0;
    unsigned long ba_gen = 0;   // see Frame::ba_gen
    int ba_index = -1;
    unsigned long visit_gen = 0;   // "seen in this sweep" stamp (refine_window walks every track once through its frames' keypoint lists)
    bool tags[TT_COUNT] = {false, false, false, false, true, false, false};   // TT_STATIC set (track.cpp:8)
    size_t map_index = 0;
    Map *map = nullptr;
    LandmarkState landmark;
    size_t m_life = 0;
    FrameRefs keypoint_refs;

    bool &tag(TrackTag t) { return tags[t]; }
    bool tag(TrackTag t) const { return tags[t]; }
    bool all_tagged(std::initializer_list<TrackTag> ts) const {
        for (TrackTag t : ts)
            if (!tags[t]) return false;
        return true;
    }
    size_t keypoint_num() const { return keypoint_refs.size(); }
    std::pair<Frame *, size_t> first_keypoint() const { return *keypoint_refs.begin(); }
    Frame *first_frame() const { return keypoint_refs.begin()->first; }
    bool has_keypoint(Frame *f) const { return keypoint_refs.count(f) > 0; }
    size_t get_keypoint_index(Frame *f) const {
        auto it = keypoint_refs.find(f);
        return it == keypoint_refs.end() ? nil() : it->second;
    }
    void add_keypoint(Frame *frame, size_t keypoint_index) {   // track.cpp:14-23
        keypoint_refs[frame] = keypoint_index;
        frame->tracks[keypoint_index] = this;
        if (tag(TT_TRIANGULATED)) m_life++;
        else m_life = 1;
    }
    void remove_keypoint(Frame *frame, bool suicide_if_empty = true);
    std::optional<V3> triangulate();
    V3 get_landmark_point() const {
        auto [frame, ki] = first_keypoint();
        PoseState cam = frame->get_pose(frame->camera);
        return cam.q * frame->get_keypoint(ki) / landmark.inv_depth + cam.p;
    }
    void set_landmark_point(const V3 &p) {
        auto [frame, ki] = first_keypoint();
        (void)ki;
        PoseState cam = frame->get_pose(frame->camera);
        landmark.inv_depth = 1.0 / norm(cam.q.conjugate() * (p - cam.p));
    }
};

// MarginalizationFactor state (estimation/marginalization_factor.h:10-41)
struct MargPrior {
    std::vector<Frame *> frames;
    std::vector<double> lin;         // [n][16]
    std::vector<double> sqrt_info;   // [15n][15n]
    std::vector<double> infovec;     // [15n]
    // A marginalisation has been queued on the device and its result (lin, sqrt_info, infovec for the frames above)
    // has not been fetched yet: resolve_marginalization() does, before anything reads the three arrays.
    bool pending = false;
};

class Map {
  public:
    explicit Map(IdSource *ids) : ids_(ids) {}
    size_t frame_num() const { return frames.size(); }
    Frame *get_frame(size_t i) const { return frames[i].get(); }
    void attach_frame(std::unique_ptr<Frame> frame, size_t position = nil()) {
        frame->map = this;
        if (position == nil()) frames.emplace_back(std::move(frame));
        else frames.emplace(frames.begin() + position, std::move(frame));
    }
    std::unique_ptr<Frame> detach_frame(size_t index) {
        std::unique_ptr<Frame> f = std::move(frames[index]);
        frames.erase(frames.begin() + index);
        f->map = nullptr;
        return f;
    }
    void untrack_frame(Frame *frame) {
        for (size_t i = 0; i < frame->keypoint_num(); ++i)
            if (Track *t = frame->get_track(i)) t->remove_keypoint(frame);
    }
    void erase_frame(size_t index) {
        untrack_frame(frames[index].get());
        detach_frame(index);
    }
    size_t frame_index_by_id(size_t id) const {
        auto it = std::lower_bound(frames.begin(), frames.end(), id,
                                   [](const std::unique_ptr<Frame> &f, size_t v) { return f->id < v; });
        if (it == frames.end() || id < (*it)->id) return nil();
        return (size_t)std::distance(frames.begin(), it);
    }
    size_t track_num() const { return tracks.size(); }
    Track *get_track(size_t i) const { return tracks[i].get(); }
    Track *create_track() {
        auto t = std::make_unique<Track>();
        t->id = ++ids_->track;
        t->map_index = tracks.size();
        t->map = this;
        tracks.emplace_back(std::move(t));
        return tracks.back().get();
    }
    void recycle_track(Track *track) {   // map.cpp:116-123
        if (track->map_index != tracks.back()->map_index) {
            tracks[track->map_index].swap(tracks.back());
            tracks[track->map_index]->map_index = track->map_index;
        }
        tracks.pop_back();
    }
    void erase_track(Track *track) {
        while (track->keypoint_num() > 0) track->remove_keypoint(track->keypoint_refs.begin()->first, false);
        recycle_track(track);
    }
    void prune_tracks(const std::function<bool(const Track *)> &cond) {
        std::vector<Track *> victims;
        for (size_t i = 0; i < track_num(); ++i)
            if (cond(get_track(i))) victims.push_back(get_track(i));
        for (Track *t : victims) erase_track(t);
    }
    IdSource *ids() const { return ids_; }

    std::unique_ptr<MargPrior> marginalization_factor;
    std::deque<std::unique_ptr<Frame>> frames;
    std::vector<std::unique_ptr<Track>> tracks;

  private:
    IdSource *ids_;
};

inline Track *Frame::get_track(size_t i, Map *allocation_map) {
    if (!allocation_map) allocation_map = map;
    if (tracks[i] == nullptr) {
        Track *t = allocation_map->create_track();
        t->add_keypoint(this, i);
    }
    return tracks[i];
}

inline void Track::remove_keypoint(Frame *frame, bool suicide_if_empty) {   // track.cpp:25-44
    size_t ki = keypoint_refs.at(frame);
    std::optional<V3> lm;
    if (frame == first_frame()) lm = get_landmark_point();
    frame->tracks[ki] = nullptr;
    keypoint_refs.erase(frame);
    if (!keypoint_refs.empty()) {
        if (lm.has_value()) set_landmark_point(lm.value());
    } else {
        tag(TT_VALID) = false;
        if (suicide_if_empty) map->recycle_track(this);
    }
}

inline std::optional<V3> Track::triangulate() {   // track.cpp:46-76
    std::vector<P34> Ps;
    std::vector<V3> zs;
    for (const auto &[frame, ki] : keypoint_refs) {
        PoseState pose = frame->get_pose(frame->camera);
        M3 R = to_matrix(pose.q.conjugate());
        V3 T = -(R * pose.p);
        P34 P;
        for (int r = 0; r < 3; ++r) {
            for (int c = 0; c < 3; ++c) P.m[4 * r + c] = R(r, c);
            P.m[4 * r + 3] = T[r];
        }
        Ps.push_back(P);
        zs.push_back(frame->get_keypoint(ki));
    }
    auto h = triangulate_point(Ps, zs);
    for (size_t i = 0; i < zs.size(); ++i) {
        const double q2 = Ps[i].m[8] * h[0] + Ps[i].m[9] * h[1] + Ps[i].m[10] * h[2] + Ps[i].m[11] * h[3];
        if (!(q2 * h[3] > 0)) return {};
    }
    m_life = 1;
    return V3{h[0] / h[3], h[1] / h[3], h[2] / h[3]};
}

}   // namespace xrh
