// klt_api.hip -- C ABI of the KLT front-end (include/xrslam_hip.h, plug point #1).
// Host side of xrslam::Image for gfx950: buffer management, launches on the
// context's stream, and the order-defining selections (host_select.hpp).
#include "../../include/xrslam_hip.h"
#include "common.hip.h"
#include "group.hip.h"
#include "host_select.hpp"
#include "klt_kernels.hip.h"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <optional>
#if defined(__x86_64__)
#include <immintrin.h>
#endif
#include <utility>
#include <vector>

using namespace xrhip;

namespace {

struct LevelBuf {
    int w = 0, h = 0;
    int istride = 0;   // bytes per padded image row
    int rows = 0;      // padded rows
    uint8_t *img_base = nullptr;
    short2 *der_base = nullptr;
    uint8_t *img = nullptr;   // pixel (0,0)
    short2 *der = nullptr;    // pixel (0,0)
};

// What a request of each kind carries: the argument sets of its kernels, built by the owning context exactly as for a launch of its own
struct PrePayload {
    bool with_upload = false;   // the frame's copy into HBM travels with its preprocessing (grouped contexts: one request per frame)
    UploadArgs up;
    ClaheLutArgs lut;
    PyrAArgs pa;
    PyrBArgs pb;
};
struct DetectPayload {
    HarrisArgs hr;
    HarrisNmsArgs nms;
    HarrisSelectArgs sel;
};
struct TrackPayload {
    LkTrackArgs lk;
    int n_inline = 0;       // > 0: `inl` holds the points as floats (the kernel's own conversion), for a launch that carries this entry alone
    LkInlinePoints inl;
    bool detect = false;   // the Harris pass of the target image rides behind the tracking launch (xrhip_image_prefetch_detect)
    DetectPayload det;
};

}   // namespace

struct xrhip_klt {
    int device = 0;
    int w = 0, h = 0, max_points = 0;
    hipStream_t stream = nullptr;
    // instance group (group.hip.h): when set, the per-frame launches travel as requests -- one outstanding request of a kind per context
    xrhip_group *group = nullptr;
    GroupRequest rq_upload, rq_pre, rq_track, rq_detect;
    UploadArgs a_upload;
    PrePayload a_pre;
    TrackPayload a_track;
    DetectPayload a_detect;
    int uploads_unsynced = 0;   // grouped: uploads since the last point the KLT queue is known to have drained (pinned ring safety)
    bool upload_pending = false;   // grouped: a_upload describes a copy that has not been submitted yet (xrhip_image_preprocess takes it along)
    // scratch shared by the images of this sequence
    uint8_t *lut = nullptr;          // tiles*256
    int lut_tiles = 0;
    // device undistortion (k_undistort): packed 1/32-pixel map [h][w][2] and the frame as the camera recorded it
    uint32_t *undist_map = nullptr;
    uint8_t *undist_src = nullptr;
    bool have_undist = false;
    // Host frames go through a small ring of pinned buffers: the caller's (pageable) buffer is copied into the next slot and
    // the DMA to HBM is queued on the stream -- the upload call returns after the memcpy, the preprocessing kernels are ordered
    // behind the DMA by the stream.  (hipMemcpy2DAsync from pageable memory stages and waits inside the call: 73 us per
    // 752x480 frame against ~25 for the memcpy, measured as the difference to device-resident input, round 3.)
    static constexpr int UP_SLOTS = 3;
    uint8_t *up_buf[UP_SLOTS] = {nullptr, nullptr, nullptr};
    hipEvent_t up_done[UP_SLOTS] = {nullptr, nullptr, nullptr};
    bool up_busy[UP_SLOTS] = {false, false, false};
    int up_next = 0;
    bool fused_pyramid = true;       // xrhip_debug_set_fused_pyramid / XRHIP_NO_FUSED_PYRAMID: the five-launch path (A/B, parity)
    float *resp = nullptr;           // w*h Harris response
    int *max_key = nullptr;          // 1 int (+ candidate counter next to it)
    int *cand_count = nullptr;
    unsigned *sel_hist = nullptr;    // SEL_BINS counters: k_harris_nms fills them, k_harris_select reads and clears them
    HarrisCand *cand = nullptr;
    int cand_cap = 0;
    HarrisCand *h_cand = nullptr;    // pinned
    int *h_count = nullptr;          // pinned (2 ints)
    HarrisCand *h_top = nullptr;     // pinned, device-mapped: the strongest candidates (k_harris_select)
    SelectHeader *h_sel = nullptr;   // pinned, device-mapped
    int top_cap = SEL_SORT;   // everything the selection kernel can hand over in visiting order
    int sel_seq = 0;
    // track scratch
    int pts_cap = 0;
    // zero-copy point block of xrhip_image_track (pinned, device-mapped): curr | next | status, + completion mailbox
    char *h_pts = nullptr;
    int h_pts_cap = 0;
    unsigned *d_done = nullptr;      // device counter of finished wavefronts (monotonic)
    unsigned done_base = 0;
    int *h_trk_seq = nullptr;        // pinned
    int trk_seq = 0;
    double2 *d_curr = nullptr, *d_next = nullptr;
    uint8_t *d_status = nullptr;
    float2 *d_fprev = nullptr, *d_fnext = nullptr;
    LkCounters *d_counters = nullptr;
    LkCounters *h_counters = nullptr;   // pinned
    // profiling: event pairs are recorded on the stream without synchronising and
    // resolved lazily in xrhip_klt_get_stats(), so timing runs inside the timed region
    bool profiling = false;
    struct Pending {
        hipEvent_t e0, e1;
        int cat;
    };
    std::vector<Pending> pending;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> free_events;
    xrhip_klt_stats stats{};
};

struct xrhip_image {
    xrhip_klt *ctx = nullptr;
    uint8_t *raw = nullptr;    // w*h, unpadded upload target (CLAHE input)
    bool have_raw = false;
    bool have_pyramid = false;
    bool want_detect = false;   // xrhip_image_prefetch_detect: queue the Harris pass behind the next tracking launch
    int detect_seq = 0;         // != 0: the Harris pass of this image is in flight / done under that sequence number
    LevelBuf lv[KLT_LEVELS];
};

static PyrView make_view(const xrhip_image *img) {
    PyrView v;
    for (int l = 0; l < KLT_LEVELS; ++l) {
        v.lv[l].img = img->lv[l].img;
        v.lv[l].der = img->lv[l].der;
        v.lv[l].w = img->lv[l].w;
        v.lv[l].h = img->lv[l].h;
        v.lv[l].istride = img->lv[l].istride;
        v.lv[l].pstride = img->lv[l].istride;
    }
    return v;
}

static int ensure_points(xrhip_klt *c, int n) {
    if (n <= c->pts_cap) return XRHIP_OK;
    int cap = std::max(n, std::max(256, c->pts_cap * 2));
    hipFree(c->d_curr);
    hipFree(c->d_next);
    hipFree(c->d_status);
    hipFree(c->d_fprev);
    hipFree(c->d_fnext);
    c->d_curr = c->d_next = nullptr;
    c->d_status = nullptr;
    c->d_fprev = c->d_fnext = nullptr;
    XR_HIP(hipMalloc(&c->d_curr, sizeof(double2) * cap));
    XR_HIP(hipMalloc(&c->d_next, sizeof(double2) * cap));
    XR_HIP(hipMalloc(&c->d_status, cap));
    XR_HIP(hipMalloc(&c->d_fprev, sizeof(float2) * cap));
    XR_HIP(hipMalloc(&c->d_fnext, sizeof(float2) * cap));
    c->pts_cap = cap;
    return XRHIP_OK;
}

enum { CAT_PRE = 0, CAT_TRACK = 1, CAT_DETECT = 2 };

static int resolve_pending(xrhip_klt *c) {
    if (c->pending.empty()) return XRHIP_OK;
    XR_HIP(hipStreamSynchronize(c->stream));
    for (auto &p : c->pending) {
        float ms = 0.f;
        XR_HIP(hipEventElapsedTime(&ms, p.e0, p.e1));
        if (p.cat == CAT_PRE) c->stats.ms_preprocess += ms;
        else if (p.cat == CAT_TRACK) c->stats.ms_track += ms;
        else c->stats.ms_detect += ms;
        c->free_events.emplace_back(p.e0, p.e1);
    }
    c->pending.clear();
    return XRHIP_OK;
}

struct ProfScope {
    xrhip_klt *c;
    int cat;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    ProfScope(xrhip_klt *c_, int cat_) : c(c_), cat(cat_) {
        if (!c->profiling || c->group) return;   // grouped: the launch is the group's (xrhip_group_get_stats times the batches)
        if (c->pending.size() >= 8192) resolve_pending(c);
        if (!c->free_events.empty()) {
            e0 = c->free_events.back().first;
            e1 = c->free_events.back().second;
            c->free_events.pop_back();
        } else {
            hipEventCreate(&e0);
            hipEventCreate(&e1);
        }
        hipEventRecord(e0, c->stream);
    }
    void finish() {
        if (c->profiling && e0 && !c->group) {
            hipEventRecord(e1, c->stream);
            c->pending.push_back({e0, e1, cat});
        }
        if (cat == CAT_PRE) c->stats.n_preprocess++;
        else if (cat == CAT_TRACK) c->stats.n_track++;
        else c->stats.n_detect++;
    }
};

// ---------------------------------------------------------------------------------------------- batched launches
// One function per request kind turns n requests into launches on `s`, XB entries per launch (blockIdx.z = entry).  A context
// that launches for itself calls the same function with its one request and its own stream.
template <class Args, class Fill> static void for_chunks(int n, Fill fill) {
    for (int base = 0; base < n; base += XB) {
        const int m = std::min(XB, n - base);
        Batch<Args> b;
        std::memset(&b, 0, sizeof(b));
        fill(b, base, m);
    }
}

static int launch_upload_batch(GroupRequest **r, int n, hipStream_t s, hipStream_t) {
    for_chunks<UploadArgs>(n, [&](Batch<UploadArgs> &b, int base, int m) {
        size_t most = 0;
        for (int i = 0; i < m; ++i) {
            b.e[i] = *static_cast<const UploadArgs *>(r[base + i]->payload);
            most = std::max(most, (size_t)b.e[i].w * b.e[i].h);
        }
        const int blocks = (int)std::min<size_t>((most / 16 + 255) / 256, 64);   // 64 workgroups per frame saturate the host link
        hipLaunchKernelGGL(k_upload, dim3(std::max(blocks, 1), 1, m), dim3(256), 0, s, b);
    });
    XR_HIP(hipGetLastError());
    return XRHIP_OK;
}

static int launch_preprocess_batch(GroupRequest **r, int n, hipStream_t s, hipStream_t) {
    for (int base = 0; base < n; base += XB) {
        const int m = std::min(XB, n - base);
        Batch<ClaheLutArgs> bl;
        Batch<PyrAArgs> ba;
        Batch<PyrBArgs> bb;
        std::memset(&bl, 0, sizeof(bl));
        std::memset(&ba, 0, sizeof(ba));
        std::memset(&bb, 0, sizeof(bb));
        int gl = 1, ga = 1, gb = 1;
        Batch<UploadArgs> bu;
        std::memset(&bu, 0, sizeof(bu));
        size_t most = 0;
        for (int i = 0; i < m; ++i) {
            const PrePayload &p = *static_cast<const PrePayload *>(r[base + i]->payload);
            if (p.with_upload) {
                bu.e[i] = p.up;
                most = std::max(most, (size_t)p.up.w * p.up.h);
            }
        }
        if (most) hipLaunchKernelGGL(k_upload, dim3((int)std::max<size_t>(1, std::min<size_t>((most / 16 + 255) / 256, 64)), 1, m), dim3(256), 0, s, bu);
        for (int i = 0; i < m; ++i) {
            const PrePayload &p = *static_cast<const PrePayload *>(r[base + i]->payload);
            bl.e[i] = p.lut;
            ba.e[i] = p.pa;
            bb.e[i] = p.pb;
            gl = std::max(gl, p.lut.tiles);
            ga = std::max(ga, p.pa.blocks);
            gb = std::max(gb, p.pb.blocks);
        }
        hipLaunchKernelGGL(k_clahe_lut, dim3(gl, 1, m), dim3(CL_THREADS), 0, s, bl);
        hipLaunchKernelGGL(k_pyr_a, dim3(ga, 1, m), dim3(PF_THREADS), 0, s, ba);
        hipLaunchKernelGGL(k_pyr_b, dim3(gb, 1, m), dim3(PF_THREADS), 0, s, bb);
    }
    XR_HIP(hipGetLastError());
    return XRHIP_OK;
}

static void launch_detect_chunk(const DetectPayload *const *d, int m, hipStream_t s) {
    Batch<HarrisArgs> bh;
    Batch<HarrisNmsArgs> bn;
    Batch<HarrisSelectArgs> bs;
    std::memset(&bh, 0, sizeof(bh));
    std::memset(&bn, 0, sizeof(bn));
    std::memset(&bs, 0, sizeof(bs));
    int gx = 1, gy = 1;
    for (int i = 0; i < m; ++i) {
        bh.e[i] = d[i]->hr;
        bn.e[i] = d[i]->nms;
        bs.e[i] = d[i]->sel;
        gx = std::max(gx, d[i]->hr.gx);
        gy = std::max(gy, d[i]->hr.gy);
    }
    hipLaunchKernelGGL(k_harris, dim3(gx, gy, m), dim3(256), 0, s, bh);
    hipLaunchKernelGGL(k_harris_nms, dim3(gx, gy, m), dim3(256), 0, s, bn);
    hipLaunchKernelGGL(k_harris_select, dim3(1, 1, m), dim3(1024), SEL_LDS_BYTES, s, bs);
}

static int launch_detect_batch(GroupRequest **r, int n, hipStream_t s, hipStream_t) {
    for (int base = 0; base < n; base += XB) {
        const int m = std::min(XB, n - base);
        const DetectPayload *d[XB];
        for (int i = 0; i < m; ++i) d[i] = static_cast<const DetectPayload *>(r[base + i]->payload);
        launch_detect_chunk(d, m, s);
    }
    XR_HIP(hipGetLastError());
    return XRHIP_OK;
}

static int launch_track_batch(GroupRequest **r, int n, hipStream_t s, hipStream_t side) {
    for (int base = 0; base < n; base += XB) {
        const int m = std::min(XB, n - base);
        if (m == 1 && n == 1) {   // a lone entry whose points fit the argument block: no read of the pinned list at kernel start
            const TrackPayload *tp1 = static_cast<const TrackPayload *>(r[base]->payload);
            if (tp1->n_inline > 0) {
                hipLaunchKernelGGL(k_lk_track_inl, dim3(std::max(1, tp1->lk.n), 1, 1), dim3(LK_THREADS), 0, s, tp1->lk, tp1->inl);
                continue;
            }
        }
        Batch<LkTrackArgs> b;
        std::memset(&b, 0, sizeof(b));
        int most = 1;
        for (int i = 0; i < m; ++i) {
            b.e[i] = static_cast<const TrackPayload *>(r[base + i]->payload)->lk;
            most = std::max(most, b.e[i].n);
        }
        hipLaunchKernelGGL(k_lk_track, dim3(most, 1, m), dim3(LK_THREADS), 0, s, b);
    }
    XR_HIP(hipGetLastError());
    // The Harris passes of the target images do not depend on the tracking result: right behind the tracking launch -- on the queue's
    // side stream when there is one (ordered behind this batch by an event: they read the pyramids the stream has built), so that
    // the queue's next batch does not wait for three kernels nobody needs before the tracks have been digested.
    const DetectPayload *d[XB];
    int nd = 0;
    bool any = false;
    for (int i = 0; i < n && !any; ++i) any = static_cast<const TrackPayload *>(r[i]->payload)->detect;
    if (!any) return XRHIP_OK;
    hipStream_t ds = s;
    if (side) {
        static thread_local hipEvent_t fork[8] = {nullptr};
        static thread_local unsigned fork_next = 0;
        hipEvent_t &e = fork[fork_next++ & 7];
        if (!e) XR_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        XR_HIP(hipEventRecord(e, s));
        XR_HIP(hipStreamWaitEvent(side, e, 0));
        ds = side;
    }
    for (int i = 0; i < n; ++i) {
        const TrackPayload *t = static_cast<const TrackPayload *>(r[i]->payload);
        if (!t->detect) continue;
        d[nd++] = &t->det;
        if (nd == XB) {
            launch_detect_chunk(d, nd, ds);
            nd = 0;
        }
    }
    if (nd) launch_detect_chunk(d, nd, ds);
    XR_HIP(hipGetLastError());
    return XRHIP_OK;
}

namespace {
struct RegisterKltLaunchers {
    RegisterKltLaunchers() {
        group_register(GK_UPLOAD, launch_upload_batch);
        group_register(GK_PREPROCESS, launch_preprocess_batch);
        group_register(GK_TRACK, launch_track_batch);
        group_register(GK_DETECT, launch_detect_batch);
    }
} g_register_klt_launchers;
}   // namespace

// a request of `kind` with `payload`: to the group's queue, or launched here and now on the context's own stream
static int klt_issue(xrhip_klt *c, GroupRequest &rq, int kind, void *payload, GroupLaunchFn fn) {
    rq.kind = kind;
    rq.owner = c;
    rq.payload = payload;
    if (c->group) return group_submit(c->group, GQ_KLT, &rq);
    GroupRequest *one = &rq;
    return fn(&one, 1, c->stream, nullptr);
}
static hipStream_t klt_stream(const xrhip_klt *c) { return c->group ? group_stream(c->group, GQ_KLT) : c->stream; }
// a frame copy that was waiting for its preprocessing request goes out on its own (something else is about to read the plane)
static int flush_upload(xrhip_klt *c) {
    if (!c->upload_pending) return XRHIP_OK;
    c->upload_pending = false;
    return klt_issue(c, c->rq_upload, GK_UPLOAD, &c->a_upload, launch_upload_batch);
}
// fn(stream) in the context's launch order (rare, un-batched paths), then wait for everything issued so far
static int klt_run_sync(xrhip_klt *c, std::function<int(hipStream_t)> fn) {
    {
        const int rc = flush_upload(c);
        if (rc) return rc;
    }
    if (!c->group) {
        int rc = fn(c->stream);
        if (rc) return rc;
        XR_HIP(hipStreamSynchronize(c->stream));
        return XRHIP_OK;
    }
    int rc = group_call(c->group, GQ_KLT, c, std::move(fn));
    if (rc) return rc;
    XR_HIP(hipStreamSynchronize(group_stream(c->group, GQ_KLT)));
    c->uploads_unsynced = 0;
    return XRHIP_OK;
}
static int klt_run(xrhip_klt *c, std::function<int(hipStream_t)> fn) {   // same without the wait
    {
        const int rc = flush_upload(c);
        if (rc) return rc;
    }
    if (!c->group) return fn(c->stream);
    return group_call(c->group, GQ_KLT, c, std::move(fn));
}

extern "C" {

int xrhip_klt_join_group(xrhip_klt *c, xrhip_group *g) {
    if (!c) return xr_fail(XRHIP_EINVAL, "xrhip_klt_join_group: null context");
    if (c->group == g) return XRHIP_OK;
    if (g && group_device(g) != c->device)
        return xr_fail(XRHIP_EINVAL, "xrhip_klt_join_group: the context and the group live on different devices");
    // whatever the context has queued so far completes where it was queued
    {
        const int rc = flush_upload(c);
        if (rc) return rc;
    }
    if (c->group) {
        int rc = group_drain(c->group, GQ_KLT, c);
        if (rc) return rc;
        group_gate_unregister(c->group, c);
        group_member_remove(c->group, true);
    } else {
        XR_HIP(hipStreamSynchronize(c->stream));
    }
    c->group = g;
    c->uploads_unsynced = 0;
    for (int i = 0; i < xrhip_klt::UP_SLOTS; ++i) c->up_busy[i] = false;
    if (g) {
        group_member_add(g, true);
        group_gate_register(g, c);
    }
    return XRHIP_OK;
}

int xrhip_klt_frame_gate(xrhip_klt *c) {
    if (!c) return xr_fail(XRHIP_EINVAL, "xrhip_klt_frame_gate: null context");
    if (c->group) group_gate_arrive(c->group, c);
    return XRHIP_OK;
}
int xrhip_klt_group_busy(xrhip_klt *c, int busy) {
    if (!c) return xr_fail(XRHIP_EINVAL, "xrhip_klt_group_busy: null context");
    if (c->group) group_gate_busy(c->group, c, busy != 0);
    return XRHIP_OK;
}

int xrhip_klt_create(int width, int height, int max_points, xrhip_klt **out) {
    if (!out || width < 64 || height < 64 || max_points < 0) return xr_fail(XRHIP_EINVAL, "xrhip_klt_create: bad arguments");
    int rc = xr_require_device();
    if (rc) return rc;
    xrhip_klt *c = new xrhip_klt();
    hipGetDevice(&c->device);
    c->w = width;
    c->h = height;
    c->max_points = max_points;
    c->fused_pyramid = std::getenv("XRHIP_NO_FUSED_PYRAMID") == nullptr;
    XR_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    c->lut_tiles = 0;
    XR_HIP(hipMalloc(&c->resp, sizeof(float) * (size_t)width * height));
    XR_HIP(hipMalloc(&c->max_key, sizeof(int) * 2));
    c->cand_count = c->max_key + 1;
    {   // running maximum (as an order-preserving int key) and candidate counter: k_harris_select leaves them reset for the next pass
        const int init[2] = {(int)0x80000000, 0};
        XR_HIP(hipMemcpy(c->max_key, init, sizeof(init), hipMemcpyHostToDevice));
    }
    c->cand_cap = width * height / 2;
    XR_HIP(hipFuncSetAttribute((const void *)k_harris_select, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SEL_LDS_BYTES));
    XR_HIP(hipMalloc(&c->cand, sizeof(HarrisCand) * (size_t)c->cand_cap));
    XR_HIP(hipMalloc(&c->sel_hist, sizeof(unsigned) * SEL_BINS));
    XR_HIP(hipMemset(c->sel_hist, 0, sizeof(unsigned) * SEL_BINS));
    XR_HIP(hipHostMalloc(&c->h_cand, sizeof(HarrisCand) * (size_t)c->cand_cap, hipHostMallocDefault));
    XR_HIP(hipHostMalloc(&c->h_count, sizeof(int) * 2, hipHostMallocDefault));
    XR_HIP(hipHostMalloc(&c->h_top, sizeof(HarrisCand) * (size_t)c->top_cap, hipHostMallocDefault));
    XR_HIP(hipHostMalloc(&c->h_trk_seq, 64, hipHostMallocDefault));
    *c->h_trk_seq = 0;
    XR_HIP(hipMalloc(&c->d_done, sizeof(unsigned)));
    XR_HIP(hipMemset(c->d_done, 0, sizeof(unsigned)));
    XR_HIP(hipHostMalloc(&c->h_sel, sizeof(SelectHeader), hipHostMallocDefault));
    std::memset(c->h_sel, 0, sizeof(SelectHeader));
    XR_HIP(hipMalloc(&c->d_counters, sizeof(LkCounters)));
    XR_HIP(hipMemset(c->d_counters, 0, sizeof(LkCounters)));
    XR_HIP(hipHostMalloc(&c->h_counters, sizeof(LkCounters), hipHostMallocDefault));
    for (int i = 0; i < xrhip_klt::UP_SLOTS; ++i) {
        XR_HIP(hipHostMalloc(&c->up_buf[i], (size_t)width * height, hipHostMallocDefault));
        XR_HIP(hipEventCreateWithFlags(&c->up_done[i], hipEventDisableTiming));
    }
    rc = ensure_points(c, std::max(256, max_points * 2));
    if (rc) return rc;
    *out = c;
    return XRHIP_OK;
}

void xrhip_klt_destroy(xrhip_klt *c) {
    if (!c) return;
    hostprof_dump();
    if (c->group) xrhip_klt_join_group(c, nullptr);
    hipStreamSynchronize(c->stream);
    hipFree(c->lut);
    hipFree(c->undist_map);
    hipFree(c->undist_src);
    hipFree(c->resp);
    hipFree(c->max_key);
    hipFree(c->cand);
    hipFree(c->sel_hist);
    hipHostFree(c->h_cand);
    hipHostFree(c->h_count);
    hipHostFree(c->h_top);
    hipHostFree(c->h_trk_seq);
    hipHostFree(c->h_pts);
    hipFree(c->d_done);
    hipHostFree(c->h_sel);
    hipFree(c->d_curr);
    hipFree(c->d_next);
    hipFree(c->d_status);
    hipFree(c->d_fprev);
    hipFree(c->d_fnext);
    hipFree(c->d_counters);
    hipHostFree(c->h_counters);
    for (int i = 0; i < xrhip_klt::UP_SLOTS; ++i) {
        hipHostFree(c->up_buf[i]);
        if (c->up_done[i]) hipEventDestroy(c->up_done[i]);
    }
    for (auto &p : c->pending) {
        hipEventDestroy(p.e0);
        hipEventDestroy(p.e1);
    }
    for (auto &p : c->free_events) {
        hipEventDestroy(p.first);
        hipEventDestroy(p.second);
    }
    hipStreamDestroy(c->stream);
    delete c;
}

int xrhip_image_create(xrhip_klt *c, xrhip_image **out) {
    if (!c || !out) return xr_fail(XRHIP_EINVAL, "xrhip_image_create: null argument");
    xrhip_image *im = new xrhip_image();
    im->ctx = c;
    XR_HIP(hipMalloc(&im->raw, (size_t)c->w * c->h));
    int w = c->w, h = c->h;
    for (int l = 0; l < KLT_LEVELS; ++l) {
        LevelBuf &L = im->lv[l];
        L.w = w;
        L.h = h;
        L.istride = (KLT_PADX + w + KLT_PAD + 3 + 63) / 64 * 64;   // + 3: k_lk_track stages rows as whole dwords
        L.rows = h + 2 * KLT_PAD;
        XR_HIP(hipMalloc(&L.img_base, (size_t)L.rows * L.istride));
        XR_HIP(hipMalloc(&L.der_base, sizeof(short2) * (size_t)L.rows * L.istride));
        // derivative borders are BORDER_CONSTANT(0) and never rewritten
        XR_HIP(hipMemsetAsync(L.der_base, 0, sizeof(short2) * (size_t)L.rows * L.istride, c->stream));
        XR_HIP(hipMemsetAsync(L.img_base, 0, (size_t)L.rows * L.istride, c->stream));
        L.img = L.img_base + (size_t)KLT_PAD * L.istride + KLT_PADX;
        L.der = L.der_base + (size_t)KLT_PAD * L.istride + KLT_PADX;
        w = (w + 1) / 2;
        h = (h + 1) / 2;
    }
    XR_HIP(hipStreamSynchronize(c->stream));   // (the planes are cleared before any queue -- the context's or a group's -- can reach them)
    *out = im;
    return XRHIP_OK;
}

void xrhip_image_destroy(xrhip_image *im) {
    if (!im) return;
    // a grouped upload that nobody has submitted yet and that writes THIS plane: dropped, not submitted later into freed memory
    // (the context's teardown flushes whatever is pending)
    if (im->ctx->upload_pending && im->ctx->a_upload.dst == im->raw) im->ctx->upload_pending = false;
    if (im->ctx->group) group_drain(im->ctx->group, GQ_KLT, im->ctx);
    hipStreamSynchronize(im->ctx->stream);
    hipFree(im->raw);
    for (int l = 0; l < KLT_LEVELS; ++l) {
        hipFree(im->lv[l].img_base);
        hipFree(im->lv[l].der_base);
    }
    delete im;
}

// memcpy into the pinned slot with streaming stores where the CPU has AVX2: the slot is written once and read by the DMA engine,
// never by this core -- non-temporal stores skip the read-for-ownership of every destination line (a 752x480 frame: 360 KB).
#if defined(__x86_64__)
__attribute__((target("avx2"))) static void copy_stream_avx2(uint8_t *dst, const uint8_t *src, size_t n) {
    size_t i = 0;
    while (i < n && (reinterpret_cast<uintptr_t>(dst + i) & 31)) {
        dst[i] = src[i];
        ++i;
    }
    for (; i + 128 <= n; i += 128) {
        const __m256i a = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(src + i));
        const __m256i b = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(src + i + 32));
        const __m256i c2 = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(src + i + 64));
        const __m256i d = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(src + i + 96));
        _mm256_stream_si256(reinterpret_cast<__m256i *>(dst + i), a);
        _mm256_stream_si256(reinterpret_cast<__m256i *>(dst + i + 32), b);
        _mm256_stream_si256(reinterpret_cast<__m256i *>(dst + i + 64), c2);
        _mm256_stream_si256(reinterpret_cast<__m256i *>(dst + i + 96), d);
    }
    _mm_sfence();
    for (; i < n; ++i) dst[i] = src[i];
}
#endif
static void copy_to_pinned(uint8_t *dst, const uint8_t *src, size_t n) {
#if defined(__x86_64__)
    static const bool avx2 = __builtin_cpu_supports("avx2");
    if (avx2 && n >= 4096) {
        copy_stream_avx2(dst, src, n);
        return;
    }
#endif
    std::memcpy(dst, src, n);
}

// Copies a host frame into the next pinned slot and queues its DMA into `dst` (w*h, dense); the caller's buffer is free on return.
static int stage_host_frame(xrhip_klt *c, const uint8_t *gray, int stride, uint8_t *dst) {
    const int slot = c->up_next;
    c->up_next = (slot + 1) % xrhip_klt::UP_SLOTS;
    if (c->group) {
        // the slot was read by the upload launched three frames ago; the pipeline has waited for two tracking results since (same
        // queue, in order).  A caller that uploads without ever waiting is held here instead.
        if (c->uploads_unsynced >= xrhip_klt::UP_SLOTS - 1) {
            int rc = group_drain(c->group, GQ_KLT, c);
            if (rc) return rc;
            c->uploads_unsynced = 0;
        }
        int rc = group_wait_launched(&c->rq_upload);   // (its argument block is about to be rewritten)
        if (rc) return rc;
        uint8_t *buf = c->up_buf[slot];
        if (stride == c->w) copy_to_pinned(buf, gray, (size_t)c->w * c->h);
        else
            for (int y = 0; y < c->h; ++y) std::memcpy(buf + (size_t)y * c->w, gray + (size_t)y * stride, (size_t)c->w);
        uint8_t *dbuf = nullptr;
        XR_HIP(hipHostGetDevicePointer((void **)&dbuf, buf, 0));
        rc = flush_upload(c);   // (an earlier frame nobody preprocessed)
        if (rc) return rc;
        c->a_upload = UploadArgs{dbuf, c->w, dst, c->w, c->h};
        c->uploads_unsynced++;
        c->upload_pending = true;   // submitted with the frame's preprocessing (xrhip_image_preprocess), or by whoever reads the plane first
        return XRHIP_OK;
    }
    if (c->up_busy[slot]) XR_HIP(hipEventSynchronize(c->up_done[slot]));   // three uploads ago: long done unless nothing consumed them
    uint8_t *buf = c->up_buf[slot];
    if (stride == c->w) {
        copy_to_pinned(buf, gray, (size_t)c->w * c->h);
    } else {
        for (int y = 0; y < c->h; ++y) std::memcpy(buf + (size_t)y * c->w, gray + (size_t)y * stride, (size_t)c->w);
    }
    XR_HIP(hipMemcpyAsync(dst, buf, (size_t)c->w * c->h, hipMemcpyHostToDevice, c->stream));
    XR_HIP(hipEventRecord(c->up_done[slot], c->stream));
    c->up_busy[slot] = true;
    return XRHIP_OK;
}

int xrhip_image_upload(xrhip_image *im, const uint8_t *gray, int stride) {
    if (!im || !gray || stride < im->ctx->w) return xr_fail(XRHIP_EINVAL, "xrhip_image_upload: bad arguments");
    xrhip_klt *c = im->ctx;
    // the host buffer may be reused by the caller as soon as we return (PushImage deep-copies)
    const int rc = stage_host_frame(c, gray, stride, im->raw);
    if (rc) return rc;
    im->have_raw = true;
    im->have_pyramid = false;
    im->want_detect = false;
    im->detect_seq = 0;
    return XRHIP_OK;
}

int xrhip_klt_set_undistort_map(xrhip_klt *c, const uint32_t *map2) {
    if (!c) return xr_fail(XRHIP_EINVAL, "xrhip_klt_set_undistort_map: null context");
    if (!map2) {   // back to "frames arrive rectified"
        c->have_undist = false;
        return XRHIP_OK;
    }
    const size_t bytes = sizeof(uint32_t) * 2 * (size_t)c->w * c->h;
    if (!c->undist_map) {
        XR_HIP(hipMalloc(&c->undist_map, bytes));
        XR_HIP(hipMalloc(&c->undist_src, (size_t)c->w * c->h));
    }
    int rc = klt_run_sync(c, [&](hipStream_t st) {
        XR_HIP(hipMemcpyAsync(c->undist_map, map2, bytes, hipMemcpyHostToDevice, st));
        return XRHIP_OK;
    });
    if (rc) return rc;
    c->have_undist = true;
    return XRHIP_OK;
}

int xrhip_image_upload_distorted(xrhip_image *im, const void *gray, int stride, int on_device) {
    if (!im || !gray || stride < im->ctx->w) return xr_fail(XRHIP_EINVAL, "xrhip_image_upload_distorted: bad arguments");
    xrhip_klt *c = im->ctx;
    if (!c->have_undist) return xr_fail(XRHIP_ESTATE, "xrhip_image_upload_distorted: no undistortion map (xrhip_klt_set_undistort_map)");
    const uint8_t *src = static_cast<const uint8_t *>(gray);
    int sstride = stride;
    if (!on_device) {   // one upload of the frame as the camera recorded it; the remap below reads it in HBM
        const int rc = stage_host_frame(c, static_cast<const uint8_t *>(gray), stride, c->undist_src);
        if (rc) return rc;
        src = c->undist_src;
        sstride = c->w;
    }
    {   // (not batched: a remap per frame on a path the bench's resident / rectified inputs do not take)
        int rc = klt_run(c, [=](hipStream_t st) {
            hipLaunchKernelGGL(k_undistort, dim3((c->w + 63) / 64, (c->h + 3) / 4), dim3(256), 0, st, src, sstride,
                               (const uint2 *)c->undist_map, im->raw, c->w, c->w, c->h);
            XR_HIP(hipGetLastError());
            return XRHIP_OK;
        });
        if (rc) return rc;
    }
    // (a host buffer may be reused by the caller as soon as we return: stage_host_frame has copied it)
    im->have_raw = true;
    im->have_pyramid = false;
    im->want_detect = false;
    im->detect_seq = 0;
    return XRHIP_OK;
}

/* parity aids: the one-launch pyramid build on / off for this context; a level's plane WITH its border (rows = h + 2 pad, the
   first `cols` = w + 2 pad columns of each padded row, pad = 21) */
int xrhip_debug_set_fused_pyramid(xrhip_klt *c, int on) {
    if (!c) return xr_fail(XRHIP_EINVAL, "xrhip_debug_set_fused_pyramid: null context");
    c->fused_pyramid = on != 0;
    return XRHIP_OK;
}
int xrhip_debug_get_level_padded(const xrhip_image *im, int level, uint8_t *out, int *rows, int *cols) {
    if (!im || level < 0 || level >= KLT_LEVELS || !rows || !cols) return xr_fail(XRHIP_EINVAL, "xrhip_debug_get_level_padded: bad arguments");
    if (!im->have_pyramid) return xr_fail(XRHIP_ESTATE, "xrhip_debug_get_level_padded: preprocess() has not run");
    const LevelBuf &L = im->lv[level];
    *rows = L.h + 2 * KLT_PAD;
    *cols = L.w + 2 * KLT_PAD;
    if (!out) return XRHIP_OK;
    xrhip_klt *c = im->ctx;
    return klt_run_sync(c, [&](hipStream_t st) {
        XR_HIP(hipMemcpy2DAsync(out, (size_t)*cols, L.img - (ptrdiff_t)KLT_PAD * L.istride - KLT_PAD, L.istride, (size_t)*cols, (size_t)*rows,
                                hipMemcpyDeviceToHost, st));
        return XRHIP_OK;
    });
}

/* parity aid: the 8-bit frame preprocess() will read (after an upload / the device undistortion) */
int xrhip_debug_get_raw(xrhip_image *im, uint8_t *out) {
    if (!im || !out) return xr_fail(XRHIP_EINVAL, "xrhip_debug_get_raw: null argument");
    if (!im->have_raw) return xr_fail(XRHIP_ESTATE, "xrhip_debug_get_raw: no image uploaded");
    xrhip_klt *c = im->ctx;
    return klt_run_sync(c, [&](hipStream_t st) {
        XR_HIP(hipMemcpyAsync(out, im->raw, (size_t)c->w * c->h, hipMemcpyDeviceToHost, st));
        return XRHIP_OK;
    });
}

int xrhip_image_upload_device(xrhip_image *im, const void *gray_dev, int stride) {
    if (!im || !gray_dev || stride < im->ctx->w) return xr_fail(XRHIP_EINVAL, "xrhip_image_upload_device: bad arguments");
    xrhip_klt *c = im->ctx;
    if (c->group) {
        int rc = group_wait_launched(&c->rq_upload);
        if (rc) return rc;
        rc = flush_upload(c);
        if (rc) return rc;
        c->a_upload = UploadArgs{static_cast<const uint8_t *>(gray_dev), stride, im->raw, c->w, c->h};
        c->upload_pending = true;
    } else {
        XR_HIP(hipMemcpy2DAsync(im->raw, c->w, gray_dev, stride, c->w, c->h, hipMemcpyDeviceToDevice, c->stream));
    }
    im->have_raw = true;
    im->have_pyramid = false;
    im->want_detect = false;
    im->detect_seq = 0;
    return XRHIP_OK;
}

int xrhip_image_preprocess(xrhip_image *im, double clip_limit, int tiles_x, int tiles_y) {
    if (!im || tiles_x < 1 || tiles_y < 1 || tiles_x * tiles_y > 4096)
        return xr_fail(XRHIP_EINVAL, "xrhip_image_preprocess: bad arguments");
    if (!im->have_raw) return xr_fail(XRHIP_ESTATE, "xrhip_image_preprocess: no image uploaded");
    xrhip_klt *c = im->ctx;
    const int w = c->w, h = c->h;
    const int ew = w + (tiles_x - (w % tiles_x)) % tiles_x, eh = h + (tiles_y - (h % tiles_y)) % tiles_y;
    const int tw = ew / tiles_x, th = eh / tiles_y;
    const int tiles = tiles_x * tiles_y;
    if (tiles > c->lut_tiles) {
        hipFree(c->lut);
        c->lut = nullptr;
        XR_HIP(hipMalloc(&c->lut, (size_t)tiles * 256));
        c->lut_tiles = tiles;
    }
    const int area = tw * th;
    const float lut_scale = 255.0f / area;
    int clip = 0;
    if (clip_limit > 0.0) {
        clip = (int)(clip_limit * area / 256);
        if (clip < 1) clip = 1;
    }
    ProfScope prof(c, CAT_PRE);
    // the pyramid and its derivatives: two launches (k_pyr_a, k_pyr_b) behind the LUT kernel when every level is large enough for its
    // border to be one reflection away, else (or with the development switch) the five launches they replace
    bool fused = c->fused_pyramid;
    for (int l = 0; l < KLT_LEVELS; ++l) fused = fused && im->lv[l].w >= 2 * KLT_PAD + 2 && im->lv[l].h >= 2 * KLT_PAD + 2;
    ClaheLutArgs la{im->raw, w, w, h, tw, th, tiles_x, clip, lut_scale, c->lut, tiles};
    if (fused) {
        if (c->group) {
            int rc = group_wait_launched(&c->rq_pre);   // (its argument block is about to be rewritten)
            if (rc) return rc;
        }
        PrePayload &pp = c->a_pre;
        pp.with_upload = c->upload_pending && c->a_upload.dst == im->raw;
        if (pp.with_upload) {
            pp.up = c->a_upload;
            c->upload_pending = false;
        } else {
            int rc = flush_upload(c);
            if (rc) return rc;
        }
        pp.lut = la;
        PyrAArgs &pa = pp.pa;
        pa.raw = im->raw;
        pa.rstride = w;
        pa.tw = tw;
        pa.th = th;
        pa.tiles_x = tiles_x;
        pa.tiles_y = tiles_y;
        pa.lut = c->lut;
        pa.img0 = im->lv[0].img;
        pa.img1 = im->lv[1].img;
        pa.der0 = im->lv[0].der;
        pa.w0 = im->lv[0].w; pa.h0 = im->lv[0].h; pa.s0 = im->lv[0].istride;
        pa.w1 = im->lv[1].w; pa.h1 = im->lv[1].h; pa.s1 = im->lv[1].istride;
        pa.tiles_across = (w + PA_T - 1) / PA_T;
        pa.blocks = pa.tiles_across * ((h + PA_T - 1) / PA_T);
        PyrBArgs &pb = pp.pb;
        for (int l = 0; l < KLT_LEVELS; ++l) {
            pb.img[l] = im->lv[l].img;
            pb.der[l] = im->lv[l].der;
            pb.w[l] = im->lv[l].w;
            pb.h[l] = im->lv[l].h;
            pb.istride[l] = im->lv[l].istride;
        }
        pb.tiles_across = (pb.w[1] + PB_T - 1) / PB_T;
        pb.blocks = pb.tiles_across * ((pb.h[1] + PB_T - 1) / PB_T);
        int rc = klt_issue(c, c->rq_pre, GK_PREPROCESS, &pp, launch_preprocess_batch);
        if (rc) return rc;
        prof.finish();
        im->have_pyramid = true;
        return XRHIP_OK;
    }
    int rc = klt_run(c, [=](hipStream_t st) {   // the five-launch form (tiny images, A/B reference), un-batched
        Batch<ClaheLutArgs> bl;
        std::memset(&bl, 0, sizeof(bl));
        bl.e[0] = la;
        hipLaunchKernelGGL(k_clahe_lut, dim3(tiles, 1, 1), dim3(CL_THREADS), 0, st, bl);
        {
            const LevelBuf &L = im->lv[0];
            dim3 blk(64, 4);
            dim3 grd((w + 2 * KLT_PAD + 63) / 64, (h + 2 * KLT_PAD + 3) / 4);
            hipLaunchKernelGGL(k_clahe_apply, grd, blk, 0, st, im->raw, w, w, h, tw, th, tiles_x, tiles_y, c->lut, L.img, L.istride);
        }
        PyrView pv = make_view(im);
        for (int l = 1; l < KLT_LEVELS; ++l) {
            const LevelBuf &D = im->lv[l];
            dim3 blk(64, 4);
            dim3 grd((D.w + 2 * KLT_PAD + 63) / 64, (D.h + 2 * KLT_PAD + 3) / 4);
            hipLaunchKernelGGL(k_pyrdown, grd, blk, 0, st, pv.lv[l - 1], D.img, D.w, D.h, D.istride);
        }
        ScharrArgs sa;
        int off = 0;
        for (int l = 0; l < KLT_LEVELS; ++l) {
            sa.lv[l] = pv.lv[l];
            sa.out[l] = im->lv[l].der;
            sa.blk_off[l] = off;
            sa.blk_w[l] = (im->lv[l].w + 63) / 64;
            off += sa.blk_w[l] * ((im->lv[l].h + 3) / 4);
        }
        sa.blk_off[KLT_LEVELS] = off;
        hipLaunchKernelGGL(k_scharr, dim3(off), dim3(256), 0, st, sa);
        XR_HIP(hipGetLastError());
        return XRHIP_OK;
    });
    if (rc) return rc;
    prof.finish();
    im->have_pyramid = true;
    return XRHIP_OK;
}

int xrhip_image_release(xrhip_image *im) {
    if (!im) return xr_fail(XRHIP_EINVAL, "xrhip_image_release: null");
    // buffers are pooled per image object; releasing only invalidates the contents
    im->have_raw = false;
    im->have_pyramid = false;
    return XRHIP_OK;
}

// The argument sets of one detection: Harris response, NMS, strongest-candidate selection; results land in the pinned top
// block / header under a fresh sequence number (one detection in flight per context)
static int fill_detect(xrhip_image *im, DetectPayload &d, int seq) {
    xrhip_klt *c = im->ctx;
    PyrView pv = make_view(im);
    const int w = c->w, h = c->h;
    double scale = (double)(1 << 2) * 3;
    scale *= 255.0;
    scale = 1.0 / scale;
    const float s2 = (float)(scale * scale);
    HarrisCand *d_top = nullptr;
    SelectHeader *d_sel = nullptr;
    XR_HIP(hipHostGetDevicePointer((void **)&d_top, c->h_top, 0));
    XR_HIP(hipHostGetDevicePointer((void **)&d_sel, c->h_sel, 0));
    const int gx = (w + 63) / 64, gy = (h + 15) / 16;
    d.hr = HarrisArgs{pv.lv[0], 0.04, s2, c->resp, c->max_key, gx, gy};
    d.nms = HarrisNmsArgs{c->resp, w, h, c->max_key, 1.0e-3, c->cand, c->cand_count, c->cand_cap, gx, gy, c->sel_hist};
    // the strongest candidates the spacing pass is handed: >= 896 (150 corners visit ~400), 8 per corner asked for beyond that
    const int keep = std::min(c->top_cap - 1024, std::max(SEL_K, c->max_points > 150 ? 8 * c->max_points : SEL_K));
    d.sel = HarrisSelectArgs{c->cand, c->cand_count, c->cand_cap, c->max_key, 1.0e-3, d_top, c->top_cap, d_sel, seq, keep, c->sel_hist};
    return XRHIP_OK;
}

// a detection of its own (no tracking launch to ride behind): asynchronous
static int launch_detect(xrhip_image *im) {
    xrhip_klt *c = im->ctx;
    ProfScope prof(c, CAT_DETECT);
    if (c->group) {
        int rc = group_wait_launched(&c->rq_detect);
        if (rc) return rc;
    }
    const int seq = ++c->sel_seq;
    int rc = fill_detect(im, c->a_detect, seq);
    if (rc) return rc;
    rc = klt_issue(c, c->rq_detect, GK_DETECT, &c->a_detect, launch_detect_batch);
    if (rc) return rc;
    prof.finish();
    im->detect_seq = seq;
    return XRHIP_OK;
}

int xrhip_image_prefetch_detect(xrhip_image *im) {
    if (!im) return xr_fail(XRHIP_EINVAL, "xrhip_image_prefetch_detect: null image");
    im->want_detect = true;
    return XRHIP_OK;
}

int xrhip_image_detect(xrhip_image *im, const double *existing_xy, int n_exist, int max_points, double min_distance,
                       double *out_xy, int *n_out) {
    if (!im || !out_xy || !n_out || n_exist < 0 || (n_exist > 0 && !existing_xy))
        return xr_fail(XRHIP_EINVAL, "xrhip_image_detect: bad arguments");
    if (!im->have_pyramid) return xr_fail(XRHIP_ESTATE, "xrhip_image_detect: preprocess() has not run");
    xrhip_klt *c = im->ctx;
    const int w = c->w, h = c->h;
    HostProfScope hp_all(0, "detect: whole call");
    std::optional<HostProfScope> hp_gpu(std::in_place, 1, "detect: launch+D2H waits");
    bool own_launch = false;
    if (!im->detect_seq) {
        int rc = launch_detect(im);
        if (rc) return rc;
        own_launch = true;
    }
    const int seq = im->detect_seq;
    im->detect_seq = 0;
    im->want_detect = false;
    {   // (a detection that rode behind a tracking launch: that request has been waited for by xrhip_image_track)
        GroupRequest *rq = c->group ? (own_launch ? &c->rq_detect : &c->rq_track) : nullptr;
        int rc = wait_flag(&c->h_sel->seq, seq, klt_stream(c), rq, "xrhip_image_detect", c->group ? group_side_stream(c->group, GQ_KLT) : nullptr);
        if (rc) return rc;
        c->uploads_unsynced = 0;
    }
    const int nc = c->h_sel->n_candidates, n_top = c->h_sel->n_top;
    if (nc > c->cand_cap) return xr_fail(XRHIP_EOVERFLOW, "xrhip_image_detect: corner candidate buffer overflow");
    hp_gpu.reset();
    HostProfScope hp_sel(2, "detect: host selection");
    // total order: response desc, then linear index desc (cv greaterThanPtr).  The greedy spacing pass usually
    // stops after a few hundred candidates (max_points corners), so the order is produced lazily from a heap.
    auto before = [](const HarrisCand &a, const HarrisCand &b) {   // a is visited before b
        if (a.v > b.v) return true;
        if (a.v < b.v) return false;
        return a.idx > b.idx;
    };
    auto heap_less = [&](const HarrisCand &a, const HarrisCand &b) { return before(b, a); };
    // GFTTDetector(max_points, 1e-3, 20, 3, harris) -- minDistance is the literal 20 of opencv_image.cpp:186
    auto select_from = [&](HarrisCand *hb, HarrisCand *he) {
        std::make_heap(hb, he, heap_less);
        return greedy_min_distance(
            [&](int &idx) {
                if (hb == he) return false;
                std::pop_heap(hb, he, heap_less);
                --he;
                idx = he->idx;
                return true;
            },
            w, h, 20.0, max_points);
    };
    std::vector<int> corners;
    bool need_all = n_top > c->top_cap;
    if (!need_all) {
        if (c->h_sel->sorted) {   // already in visiting order: walk it
            const HarrisCand *it = c->h_top, *end = c->h_top + n_top;
            corners = greedy_min_distance(
                [&](int &idx) {
                    if (it == end) return false;
                    idx = (it++)->idx;
                    return true;
                },
                w, h, 20.0, max_points);
        } else {
            corners = select_from(c->h_top, c->h_top + n_top);
        }
        // the pass ran out of strong candidates before it had max_points corners: the weaker ones matter after all
        need_all = n_top < nc && (max_points <= 0 || (int)corners.size() < max_points);
    }
    if (need_all) {
        HostProfScope hp_fb(12, "detect: full candidate list fallback");
        c->stats.detect_full_list += 1;
        int rc = klt_run_sync(c, [&](hipStream_t st) {
            XR_HIP(hipMemcpyAsync(c->h_cand, c->cand, sizeof(HarrisCand) * (size_t)nc, hipMemcpyDeviceToHost, st));
            return XRHIP_OK;
        });
        if (rc) return rc;
        corners = select_from(c->h_cand, c->h_cand + nc);
    }
    int n = 0;
    HostProfScope hp_po(16, "detect: poisson");
    if (!corners.empty()) {
        PoissonDisk2 filter(min_distance, w, h);
        for (int i = 0; i < n_exist; ++i) filter.preset(existing_xy[2 * i], existing_xy[2 * i + 1]);
        for (int idx : corners) {
            const double x = (double)(float)(idx % w), y = (double)(float)(idx / w);
            if (!filter.insert(x, y)) continue;
            if (x < 20 || y < 20 || x >= w - 20 || y >= h - 20) continue;
            out_xy[2 * n] = x;
            out_xy[2 * n + 1] = y;
            ++n;
        }
    }
    *n_out = n;
    return XRHIP_OK;
}

int xrhip_image_track(const xrhip_image *cur, const xrhip_image *next, const double *curr_xy, double *next_xy_inout,
                      int has_guess, uint8_t *status, int n) {
    if (!cur || !next || n < 0 || (n > 0 && (!curr_xy || !next_xy_inout || !status)))
        return xr_fail(XRHIP_EINVAL, "xrhip_image_track: bad arguments");
    if (cur->ctx != next->ctx) return xr_fail(XRHIP_EINVAL, "xrhip_image_track: images belong to different contexts");
    if (!cur->have_pyramid || !next->have_pyramid)
        return xr_fail(XRHIP_ESTATE, "xrhip_image_track: preprocess() has not run on both images");
    if (n == 0) return XRHIP_OK;
    HostProfScope hp_trk(3, "track: whole call");
    xrhip_klt *c = cur->ctx;
    int rc = ensure_points(c, n);
    if (rc) return rc;
    if (n > c->h_pts_cap) {
        if (c->h_pts) hipHostFree(c->h_pts);
        c->h_pts = nullptr;
        const int cap = std::max(n, std::max(512, 2 * c->h_pts_cap));
        XR_HIP(hipHostMalloc(&c->h_pts, (size_t)cap * (2 * sizeof(double2) + 1) + 64, hipHostMallocDefault));
        c->h_pts_cap = cap;
    }
    double2 *h_curr = (double2 *)c->h_pts, *h_next = h_curr + c->h_pts_cap;
    uint8_t *h_status = (uint8_t *)(h_next + c->h_pts_cap);
    std::memcpy(h_curr, curr_xy, sizeof(double2) * n);
    if (has_guess) std::memcpy(h_next, next_xy_inout, sizeof(double2) * n);
    char *d_pts = nullptr;
    int *d_seq = nullptr;
    XR_HIP(hipHostGetDevicePointer((void **)&d_pts, c->h_pts, 0));
    XR_HIP(hipHostGetDevicePointer((void **)&d_seq, c->h_trk_seq, 0));
    double2 *dv_curr = (double2 *)d_pts, *dv_next = dv_curr + c->h_pts_cap;
    uint8_t *dv_status = (uint8_t *)(dv_next + c->h_pts_cap);
    rc = flush_upload(c);
    if (rc) return rc;
    if (c->group) {
        rc = group_wait_launched(&c->rq_track);   // (its argument block is about to be rewritten)
        if (rc) return rc;
    }
    ProfScope prof(c, CAT_TRACK);
    const int seq = ++c->trk_seq;
    c->done_base += (unsigned)n;
    TrackPayload &tp = c->a_track;
    tp.lk = LkTrackArgs{make_view(cur), make_view(next), dv_curr, dv_next, has_guess ? 1 : 0, dv_status, n,
                        c->profiling ? c->d_counters : (LkCounters *)nullptr, c->d_done, c->done_base, d_seq, seq};
    static const bool no_inline = std::getenv("XRHIP_LK_NO_INLINE") != nullptr;   // development switch (A/B)
    tp.n_inline = 0;
    if (!no_inline && n <= LK_INLINE) {
        for (int i = 0; i < n; ++i) {
            const float cx = (float)curr_xy[2 * i], cy = (float)curr_xy[2 * i + 1];   // the kernel's own double -> float conversions
            tp.inl.p[i] = has_guess ? make_float4(cx, cy, (float)next_xy_inout[2 * i], (float)next_xy_inout[2 * i + 1]) : make_float4(cx, cy, cx, cy);
        }
        tp.n_inline = n;
    }
    // the Harris pass of `next` does not depend on the tracking result: it rides right behind the tracking launch so that it runs
    // while the host digests the tracks (the wait below is on the tracking kernel's own mailbox, not on the stream)
    tp.detect = next->want_detect && !next->detect_seq;
    int det_seq = 0;
    if (tp.detect) {
        det_seq = ++c->sel_seq;
        rc = fill_detect(const_cast<xrhip_image *>(next), tp.det, det_seq);
        if (rc) return rc;
    }
    if (c->group) {
        rc = klt_issue(c, c->rq_track, GK_TRACK, &tp, launch_track_batch);
        if (rc) return rc;
        prof.finish();
        if (tp.detect) c->stats.n_detect++;
    } else {   // launched here: the two halves keep their own event pairs (bench.py's roofline_lk is the LK kernel alone)
        const bool with_detect = tp.detect;
        tp.detect = false;
        rc = klt_issue(c, c->rq_track, GK_TRACK, &tp, launch_track_batch);
        if (rc) return rc;
        prof.finish();
        if (with_detect) {
            ProfScope prof_d(c, CAT_DETECT);
            const DetectPayload *d = &tp.det;
            launch_detect_chunk(&d, 1, c->stream);
            XR_HIP(hipGetLastError());
            prof_d.finish();
        }
        tp.detect = with_detect;
    }
    c->stats.lk_points += n;
    if (tp.detect) const_cast<xrhip_image *>(next)->detect_seq = det_seq;
    rc = wait_flag(c->h_trk_seq, seq, klt_stream(c), c->group ? &c->rq_track : nullptr, "xrhip_image_track");
    if (rc) return rc;
    c->uploads_unsynced = 0;
    // results: status first, positions only where status != 0 (reference semantics)
    for (int i = 0; i < n; ++i) {
        status[i] = h_status[i];
        if (status[i]) {
            next_xy_inout[2 * i] = h_next[i].x;
            next_xy_inout[2 * i + 1] = h_next[i].y;
        }
    }
    return XRHIP_OK;
}

int xrhip_image_lk(const xrhip_image *prev, const xrhip_image *next, const float *prev_xy, float *next_xy_inout,
                   uint8_t *status, int n) {
    if (!prev || !next || n < 0 || (n > 0 && (!prev_xy || !next_xy_inout || !status)))
        return xr_fail(XRHIP_EINVAL, "xrhip_image_lk: bad arguments");
    if (!prev->have_pyramid || !next->have_pyramid) return xr_fail(XRHIP_ESTATE, "xrhip_image_lk: preprocess() has not run");
    if (n == 0) return XRHIP_OK;
    xrhip_klt *c = prev->ctx;
    int rc = ensure_points(c, n);
    if (rc) return rc;
    return klt_run_sync(c, [&](hipStream_t st) {
        XR_HIP(hipMemcpyAsync(c->d_fprev, prev_xy, sizeof(float2) * n, hipMemcpyHostToDevice, st));
        XR_HIP(hipMemcpyAsync(c->d_fnext, next_xy_inout, sizeof(float2) * n, hipMemcpyHostToDevice, st));
        PyrView A = make_view(prev), B = make_view(next);
        hipLaunchKernelGGL(k_lk_plain, dim3(n), dim3(LK_THREADS), 0, st, A, B, c->d_fprev, c->d_fnext, c->d_status, n);
        XR_HIP(hipGetLastError());
        XR_HIP(hipMemcpyAsync(status, c->d_status, n, hipMemcpyDeviceToHost, st));
        XR_HIP(hipMemcpyAsync(next_xy_inout, c->d_fnext, sizeof(float2) * n, hipMemcpyDeviceToHost, st));
        return XRHIP_OK;
    });
}

int xrhip_image_level_dims(const xrhip_image *im, int level, int *w, int *h) {
    if (!im || level < 0 || level >= KLT_LEVELS || !w || !h) return xr_fail(XRHIP_EINVAL, "xrhip_image_level_dims: bad arguments");
    *w = im->lv[level].w;
    *h = im->lv[level].h;
    return XRHIP_OK;
}

int xrhip_image_download_level(const xrhip_image *im, int level, uint8_t *img_out, int16_t *deriv_out) {
    if (!im || level < 0 || level >= KLT_LEVELS) return xr_fail(XRHIP_EINVAL, "xrhip_image_download_level: bad arguments");
    if (!im->have_pyramid) return xr_fail(XRHIP_ESTATE, "xrhip_image_download_level: preprocess() has not run");
    const LevelBuf &L = im->lv[level];
    xrhip_klt *c = im->ctx;
    return klt_run_sync(c, [&](hipStream_t st) {
        if (img_out) XR_HIP(hipMemcpy2DAsync(img_out, L.w, L.img, L.istride, L.w, L.h, hipMemcpyDeviceToHost, st));
        if (deriv_out)
            XR_HIP(hipMemcpy2DAsync(deriv_out, (size_t)L.w * 4, L.der, (size_t)L.istride * 4, (size_t)L.w * 4, L.h, hipMemcpyDeviceToHost, st));
        return XRHIP_OK;
    });
}

int xrhip_image_download_harris(xrhip_image *im, float *resp_out) {
    if (!im || !resp_out) return xr_fail(XRHIP_EINVAL, "xrhip_image_download_harris: bad arguments");
    if (!im->have_pyramid) return xr_fail(XRHIP_ESTATE, "xrhip_image_download_harris: preprocess() has not run");
    xrhip_klt *c = im->ctx;
    DetectPayload d;
    int rc = fill_detect(im, d, 0);
    if (rc) return rc;
    return klt_run_sync(c, [&](hipStream_t st) {   // the response plane alone; the running maximum goes back to its reset state
        Batch<HarrisArgs> bh;
        std::memset(&bh, 0, sizeof(bh));
        bh.e[0] = d.hr;
        hipLaunchKernelGGL(k_harris, dim3(d.hr.gx, d.hr.gy, 1), dim3(256), 0, st, bh);
        XR_HIP(hipGetLastError());
        XR_HIP(hipMemcpyAsync(resp_out, c->resp, sizeof(float) * (size_t)c->w * c->h, hipMemcpyDeviceToHost, st));
        static const int init[2] = {(int)0x80000000, 0};
        XR_HIP(hipMemcpyAsync(c->max_key, init, sizeof(init), hipMemcpyHostToDevice, st));
        return XRHIP_OK;
    });
}

int xrhip_klt_set_profiling(xrhip_klt *c, int enable) {
    if (!c) return xr_fail(XRHIP_EINVAL, "xrhip_klt_set_profiling: null");
    c->profiling = enable != 0;
    return XRHIP_OK;
}

int xrhip_klt_get_stats(xrhip_klt *c, xrhip_klt_stats *out, int reset) {
    if (!c || !out) return xr_fail(XRHIP_EINVAL, "xrhip_klt_get_stats: null");
    int rc = resolve_pending(c);
    if (rc) return rc;
    rc = klt_run_sync(c, [&](hipStream_t st) {
        XR_HIP(hipMemcpyAsync(c->h_counters, c->d_counters, sizeof(LkCounters), hipMemcpyDeviceToHost, st));
        return XRHIP_OK;
    });
    if (rc) return rc;
    c->stats.lk_templates = (long long)c->h_counters->templates;
    c->stats.lk_iterations = (long long)c->h_counters->iterations;
    *out = c->stats;
    if (reset) {
        c->stats = xrhip_klt_stats{};
        rc = klt_run_sync(c, [&](hipStream_t st) {
            XR_HIP(hipMemsetAsync(c->d_counters, 0, sizeof(LkCounters), st));
            return XRHIP_OK;
        });
        if (rc) return rc;
    }
    return XRHIP_OK;
}

int xrhip_klt_synchronize(xrhip_klt *c) {
    if (!c) return xr_fail(XRHIP_EINVAL, "xrhip_klt_synchronize: null");
    if (c->group) return group_drain(c->group, GQ_KLT, c);
    XR_HIP(hipStreamSynchronize(c->stream));
    return XRHIP_OK;
}

}   // extern "C"

/* development aid: LK phase timers (shader cycles, summed over all points since the last reset) of -DXRHIP_KPROF builds; zeros otherwise */
extern "C" void xrhip_debug_lkprof(long long *out8, int reset) {
#ifdef XRHIP_KPROF
    unsigned long long h[8] = {0};
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(xrhip::g_lk_prof), sizeof(h)) == hipSuccess)
        for (int i = 0; i < 8; ++i) out8[i] = (long long)h[i];
    if (reset) {
        unsigned long long z[8] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(xrhip::g_lk_prof), z, sizeof(z));
    }
#else
    for (int i = 0; i < 8; ++i) out8[i] = 0;
    (void)reset;
#endif
}
