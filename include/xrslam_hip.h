/*
 * xrslam_hip.h -- C ABI of the MI355X-native (gfx950) XRSLAM hot path.
 *
 * This header is the drop-in boundary for the reference's two inner plug points
 * (SURVEY.md section 8b):
 *
 *   #1 KLT front-end  = xrslam::Image virtuals
 *        xrslam/include/xrslam/xrslam.h:137-161   (interface)
 *        xrslam-extra/src/xrslam/extra/opencv_image.cpp:38-210 (the OpenCV implementation replaced)
 *   #2 Bundle adjustment = xrslam::Solver facade + MarginalizationFactor
 *        xrslam/src/xrslam/estimation/solver.h:16-71, solver.cpp:84-190
 *        xrslam/src/xrslam/estimation/marginalization_factor.h:10-41
 *        xrslam/src/xrslam/estimation/ceres/marginalization_factor.h:74-475
 *
 * Plain C types only: pointers, sizes, doubles.  No torch / Eigen / OpenCV types.
 * All functions return 0 on success and a negative XRHIP_E* code on failure;
 * xrhip_last_error() returns a human readable message for the calling thread.
 * There is NO CPU fallback: every entry point fails with XRHIP_ENODEVICE when
 * no gfx950 device is usable.
 *
 * The outer ABI (XRSLAMCreate ... XRSLAMDestroy) is declared in XRSLAM.h.
 */
#ifndef XRSLAM_HIP_H
#define XRSLAM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define XRHIP_OK 0
#define XRHIP_ENODEVICE (-1)
#define XRHIP_EINVAL (-2)
#define XRHIP_EHIP (-3)
#define XRHIP_ENOMEM (-4)
#define XRHIP_EOVERFLOW (-5)
#define XRHIP_ESTATE (-6)

const char *xrhip_last_error(void);
/* identifies the device code this library was built from (hash of the kernel sources) */
const char *xrhip_kernel_revision(void);
/* number of usable gfx950 devices (0 if none); never fails */
int xrhip_device_count(void);
/* binds the calling thread (and contexts created afterwards) to a device */
int xrhip_set_device(int device);
/* the calling thread's current device / make `device` current for the calling thread without re-validating it
 * (the current device is a per-thread setting: a thread that drives a context created elsewhere binds first) */
int xrhip_get_device(int *device);
int xrhip_bind_device(int device);

/* ------------------------------------------------------------------------
 * Plug point #1: KLT front-end.
 * One xrhip_klt per sequence: owns a HIP stream, the CLAHE/GFTT parameters the
 * reference keeps in function-local statics (opencv_image.cpp:179-188) and
 * scratch buffers.  One xrhip_image per camera frame (== one OpenCvImage).
 * ---------------------------------------------------------------------- */
typedef struct xrhip_klt xrhip_klt;
typedef struct xrhip_image xrhip_image;

#define XRHIP_KLT_LEVELS 4   /* OpenCvImage::level_num()==3 -> maxLevel 3 -> 4 levels (opencv_image.h:20) */
#define XRHIP_KLT_WIN 21     /* Size(21,21)            (opencv_image.cpp:96,122,159) */

/* replaces: OpenCvImage::OpenCvImage + static clahe()/gftt() objects. */
int xrhip_klt_create(int width, int height, int max_points, xrhip_klt **out);
void xrhip_klt_destroy(xrhip_klt *ctx);

/* replaces: XRSLAMManager::PushImage's deep copy into OpenCvImage::image
 * (xrslam-interface/src/XRSLAMManager.cpp:104-136).  `gray` is a host pointer. */
int xrhip_image_create(xrhip_klt *ctx, xrhip_image **out);
int xrhip_image_upload(xrhip_image *img, const uint8_t *gray, int stride_bytes);
/* same, but `gray_dev` already lives in HBM (device pointer; used by bench.py so
 * the timed region starts with inputs resident on the device).  The copy is stream-ordered and, for a member of a
 * group, submitted with the frame's preprocessing: `gray_dev` must stay valid until xrhip_image_preprocess (or the
 * first call that reads the plane) has returned. */
int xrhip_image_upload_device(xrhip_image *img, const void *gray_dev, int stride_bytes);
/* Undistortion on the device.  replaces: cv::undistort in the dataset reader
 * (xrslam-pc/player/src/IO/euroc_dataset_reader.cpp:62-69) / xrslam::extra::ImageUndistorter::undistort_image
 * (IO/tum_dataset_reader.cpp:67-76).  `map2` is the packed 1/32-pixel inverse map [height][width][2]
 * (word 0 = int16 sx | int16 sy << 16, word 1 = ax | ay << 8) built once per camera on the host in the reference's own
 * arithmetic (xrslam_amd/csrc/host/undistort_map.hpp); NULL switches the feature off.  With a map set,
 * xrhip_image_upload_distorted takes the frame as the camera recorded it (host pointer, or an HBM pointer when on_device)
 * and leaves the rectified frame where xrhip_image_preprocess reads it: one upload, no host pass over the pixels. */
int xrhip_klt_set_undistort_map(xrhip_klt *ctx, const uint32_t *map2);
int xrhip_image_upload_distorted(xrhip_image *img, const void *gray, int stride_bytes, int on_device);
/* parity aid: the 8-bit frame xrhip_image_preprocess will read */
int xrhip_debug_get_raw(xrhip_image *img, uint8_t *out);
/* Development / parity aids of the pyramid build (xrhip_image_preprocess): on = 1 (default) builds the CLAHE plane, the three
   pyrDown levels and all Scharr planes in one launch, 0 in the five launches it replaces (same bits: tests/test_klt_gpu.py);
   get_level_padded returns a level's image plane including its 21-pixel reflect-101 border (out may be NULL: sizes only). */
int xrhip_debug_set_fused_pyramid(xrhip_klt *ctx, int on);
int xrhip_debug_get_level_padded(const xrhip_image *img, int level, uint8_t *out, int *rows, int *cols);
void xrhip_image_destroy(xrhip_image *img);

/* replaces: Image::preprocess(clipLimit, width, height)  (xrslam.h:153,
 * opencv_image.cpp:156-161): CLAHE in place + 4-level LK pyramid with Scharr
 * derivatives, all on the context's stream (asynchronous). */
int xrhip_image_preprocess(xrhip_image *img, double clip_limit, int tiles_x, int tiles_y);

/* replaces: Image::release_image_buffer()  (xrslam.h:154, opencv_image.cpp:200-208) */
int xrhip_image_release(xrhip_image *img);

/* replaces: Image::detect_keypoints(keypoints, max_points, keypoint_distance)
 * (xrslam.h:155-157, opencv_image.cpp:38-73).  `existing_xy` = n_exist (x,y)
 * doubles already tracked; new points are written to out_xy (capacity
 * max_points pairs) and *n_out.  Synchronous (returns host data). */
/* Optional hint: xrhip_image_detect will be called on `img`.  Its Harris pass (which does not depend on the
 * tracking result) is then queued right behind the next xrhip_image_track launch that has `img` as its target,
 * so that it runs while the host digests the tracks.  Without the hint detect() launches the pass itself. */
int xrhip_image_prefetch_detect(xrhip_image *img);
int xrhip_image_detect(xrhip_image *img, const double *existing_xy, int n_exist, int max_points,
                       double min_distance, double *out_xy, int *n_out);

/* replaces: Image::track_keypoints(next, curr, next_inout, status)
 * (xrslam.h:158-161, opencv_image.cpp:75-154).  has_guess==0 <=> the reference's
 * "next_keypoints empty" case.  next_xy_inout is only written where status!=0,
 * exactly like the reference.  Synchronous. */
int xrhip_image_track(const xrhip_image *cur, const xrhip_image *next, const double *curr_xy,
                      double *next_xy_inout, int has_guess, uint8_t *status, int n);

/* plain cv::calcOpticalFlowPyrLK equivalent (float points, USE_INITIAL_FLOW) -- parity/testing aid */
int xrhip_image_lk(const xrhip_image *prev, const xrhip_image *next, const float *prev_xy, float *next_xy_inout,
                   uint8_t *status, int n);

/* parity/testing aids: copy a pyramid level (unpadded) back to the host.
 * img_out: w*h bytes (may be NULL); deriv_out: w*h*2 int16 (dx,dy interleaved; may be NULL). */
int xrhip_image_level_dims(const xrhip_image *img, int level, int *w, int *h);
int xrhip_image_download_level(const xrhip_image *img, int level, uint8_t *img_out, int16_t *deriv_out);
/* Harris response map of level 0 (float w*h) exactly as used by detect */
int xrhip_image_download_harris(xrhip_image *img, float *resp_out);

/* measurement aids.  Counters accumulate over calls
 * until reset.  All times are HIP-event milliseconds on the context's stream. */
typedef struct xrhip_klt_stats {
    double ms_preprocess;      /* sum of preprocess kernel time */
    double ms_track;           /* sum of LK kernel time */
    double ms_detect;          /* sum of Harris+NMS kernel time */
    long long n_preprocess, n_track, n_detect;   /* launches counted */
    long long lk_templates;    /* (dir,point,level) templates extracted */
    long long lk_iterations;   /* LK iterations executed */
    long long lk_points;       /* points submitted to track */
    long long detect_full_list;   /* detections that fell back from the strongest-candidate block to the full list */
} xrhip_klt_stats;
int xrhip_klt_set_profiling(xrhip_klt *ctx, int enable);
int xrhip_klt_get_stats(xrhip_klt *ctx, xrhip_klt_stats *out, int reset);
int xrhip_klt_synchronize(xrhip_klt *ctx);


/* ------------------------------------------------------------------------
 * Plug point #2: sliding-window visual-inertial bundle adjustment.
 *
 * The reference's Solver facade receives Frame / Track pointers and registers
 * raw double* blocks with Ceres (solver.cpp:84-173).  Here the same problem is
 * described by flat SoA arrays; states are read and written IN PLACE exactly
 * like the reference (Frame::pose/motion, Track::landmark.inv_depth).
 * ---------------------------------------------------------------------- */
#define XRHIP_STATE_DIM 16 /* q(x,y,z,w) p(3) v(3) bg(3) ba(3): estimation/state.h:12-19, solver.cpp:86 */
#define XRHIP_ES_DIM 15    /* error-state size ES_SIZE */
/* one pre-integrated IMU factor (PreIntegrator::Delta + Jacobian, preintegrator.h:11-48):
 * [0] dt, [1..4] dq(x,y,z,w), [5..7] dp, [8..10] dv,
 * [11..55] dq_dbg, dp_dbg, dp_dba, dv_dbg, dv_dba (3x3 row-major each),
 * [56..280] sqrt_inv_cov (15x15 row-major) */
#define XRHIP_IMU_DIM 281

#define XRHIP_FIX_POSE 1   /* FT_FIX_POSE, or the frame is only referenced as a constant */
#define XRHIP_FIX_MOTION 2 /* FT_FIX_MOTION, or the motion blocks are not part of the problem */

typedef struct xrhip_ba_problem {
    /* frames: add_frame_states (solver.cpp:84-106) */
    int n_frames;
    double *frame_state;          /* [n_frames][16], in/out */
    const uint8_t *frame_fix;     /* [n_frames], XRHIP_FIX_* bits */
    double cam_q_bc[4], cam_p_bc[3]; /* Frame::camera (q_cs xyzw, p_cs), identical for all frames (detail.cpp:110-111) */
    double imu_q_bi[4], imu_p_bi[3]; /* Frame::imu */
    double sqrt_inv_cov[2];       /* Frame::sqrt_inv_cov diagonal = (fx,fy)/sqrt(keypoint noise) (detail.cpp:107-109) */
    /* landmarks: add_track_states (solver.cpp:108-110) */
    int n_landmarks;
    double *inv_depth;            /* [n_landmarks], in/out */
    const uint8_t *landmark_fix;  /* [n_landmarks], 1 = constant (ReprojectionPriorFactor) */
    /* CeresReprojectionErrorFactor / CeresReprojectionPriorFactor (ceres/reprojection_factor.h:25-123), CauchyLoss(1) */
    int n_obs;
    const int *obs_tgt, *obs_ref, *obs_lm; /* frame index of the observing frame, of track->first_frame(), landmark index */
    const double *obs_z_tgt, *obs_z_ref;   /* [n_obs][3] unit bearings */
    /* CeresRotationPriorFactor (ceres/rotation_factor.h:23-59), CauchyLoss(1) */
    int n_rot;
    const int *rot_tgt, *rot_ref;
    const double *rot_z_tgt, *rot_z_ref;   /* [n_rot][3] */
    /* CeresPreIntegrationErrorFactor / ...PriorFactor (ceres/preintegration_factor.h:20-199), no loss */
    int n_imu;
    const int *imu_i, *imu_j;
    const double *imu_data;       /* [n_imu][XRHIP_IMU_DIM] */
    /* CeresMarginalizationFactor (ceres/marginalization_factor.h:27-72), no loss; prior_n == 0 -> absent */
    int prior_n;                  /* number of frames in linearization_frames() */
    const int *prior_frames;      /* [prior_n] indices into the frame arrays */
    const double *prior_sqrt_info; /* [15 prior_n][15 prior_n] row-major sqrt_inv_cov */
    const double *prior_infovec;  /* [15 prior_n] */
    const double *prior_lin;      /* [prior_n][16] linearisation points */
    /* Solver::solve options (solver.cpp:176-190): SPARSE_SCHUR + DOGLEG, Ceres 1.14 defaults otherwise */
    int max_iterations;           /* config solver.iteration_limit */
} xrhip_ba_problem;

#define XRHIP_BA_CONVERGENCE 0
#define XRHIP_BA_NO_CONVERGENCE 1
#define XRHIP_BA_FAILURE 2

typedef struct xrhip_ba_summary {
    int iterations;        /* trust-region iterations executed (successful + unsuccessful) */
    int successful_steps;
    int termination;       /* XRHIP_BA_* */
    int usable;            /* Summary::IsSolutionUsable() */
    double initial_cost, final_cost;
    double ms_solve;       /* host wall-clock of the whole solve: staging, every kernel, result mailbox read */
} xrhip_ba_summary;

/* marginalisation of one frame (CeresMarginalizationFactor::marginalize,
 * ceres/marginalization_factor.h:74-475).  All map frames are given in map order. */
typedef struct xrhip_marg_problem {
    int n_frames;                 /* base_map->frame_num() at call time */
    int victim;                   /* index of the frame to remove (the reference only ever passes 0) */
    const double *frame_state;    /* [n_frames][16] */
    double cam_q_bc[4], cam_p_bc[3], imu_q_bi[4], imu_p_bi[3], sqrt_inv_cov[2];
    /* old prior (factor state before the call) */
    int prior_n;
    const int *prior_frames;
    const double *prior_sqrt_info, *prior_infovec, *prior_lin;
    /* IMU factors touching the victim: (victim-1,victim) and (victim,victim+1) where they exist */
    int n_imu;
    const int *imu_i, *imu_j;
    const double *imu_data;       /* keyframe_preintegration of frame j */
    /* reprojection factors of tracks seen in the victim (marginalization_factor.h:234-380); no robust loss */
    int n_landmarks;
    const double *inv_depth;
    int n_obs;
    const int *obs_tgt, *obs_ref, *obs_lm;
    const double *obs_z_tgt, *obs_z_ref;
} xrhip_marg_problem;

typedef struct xrhip_ba xrhip_ba;
int xrhip_ba_create(int max_frames, int max_landmarks, int max_obs, xrhip_ba **out);
void xrhip_ba_destroy(xrhip_ba *ctx);
/* replaces: Solver::solve() over a problem assembled with add_frame_states/add_track_states/add_factor */
int xrhip_ba_solve(xrhip_ba *ctx, const xrhip_ba_problem *problem, xrhip_ba_summary *summary);
/* xrhip_ba_solve with a piece of the CALLER's host work run beside the device: host_work(arg) is called exactly once, on the
 * calling thread, after the solve's first launches are queued and before the library waits for them (before returning, if
 * the problem needs no launch at all or the solve fails earlier).  It must not touch the problem's arrays or this context; it may
 * use other contexts (the sliding-window tracker runs the corner selection of the next frame's detection -- an xrhip_klt call and
 * host logic -- beside localize_newframe's solve, which does not read what that selection appends).  Same results as
 * xrhip_ba_solve. */
int xrhip_ba_solve_overlapped(xrhip_ba *ctx, const xrhip_ba_problem *problem, xrhip_ba_summary *summary, void (*host_work)(void *), void *arg);

/* Two solves of which the second starts from a state the first produces.  replaces: Solver::solve() of localize_newframe immediately
 * followed by the one of refine_subwindow (sliding_window_tracker.cpp:99 and :114 through :119-143, :370-465): the second problem
 * contains the frame the first one has just localised, and nothing else of it depends on the first solve (manage_keyframe in
 * between reads tags and counts only, :145-223; the sub-window's factors and integrations use the biases of the frames before).
 * Frame `link_second` of `second` takes its whole state (16 doubles) from frame `link_first` of `first` as the first solve leaves it;
 * what second->frame_state holds for that frame on entry is ignored.  Both problems come back solved in place, each with its
 * summary (ms_solve: the pair's wall clock is booked on the first).  When both are single-launch solves the hand-over happens on
 * the device -- one submission and ONE host wait for the pair -- otherwise the two run one after the other with the state handed
 * over on the host: the same results bit for bit.  host_work (may be NULL): as xrhip_ba_solve_overlapped, beside both solves.
 * The contexts must be distinct (each holds one staged problem); a batch staged by xrhip_ba_preintegrate_after_solve on
 * ctx_second starts from the second solve's biases, as after xrhip_ba_solve. */
int xrhip_ba_solve_chained(xrhip_ba *ctx_first, const xrhip_ba_problem *first, xrhip_ba_summary *summary_first, int link_first,
                           xrhip_ba *ctx_second, const xrhip_ba_problem *second, xrhip_ba_summary *summary_second, int link_second,
                           void (*host_work)(void *), void *arg);
/* The same in three steps, so that the caller can ASSEMBLE the second problem while the device works on the first (the keyframe
 * decision and refine_subwindow's problem assembly, ~50 us of host work, run beside localize_newframe's kernel):
 *   xrhip_ba_solve_begin(ctx, first)    queues the solve and returns 1 -- or 0 without having done anything when the problem is not a
 *                                       single-launch solve (free landmarks, a prior, too many free frames): use xrhip_ba_solve then;
 *                                       `first` (its arrays) must stay alive and untouched until xrhip_ba_solve_end
 *   xrhip_ba_solve_linked(ctx2, second, summary2, link_second, ctx, link_first, host_work, arg)
 *                                       solves `second` on ctx2, its frame link_second starting from frame link_first of the solve
 *                                       begun on ctx as that solve leaves it (handed over on the device when `second` is a
 *                                       single-launch solve too, on the host otherwise); returns with `second` solved in place
 *   xrhip_ba_solve_end(ctx, summary)    waits for the begun solve and writes its states into first->frame_state
 * xrhip_ba_solve_chained is begin + linked + end. */
int xrhip_ba_solve_begin(xrhip_ba *ctx, const xrhip_ba_problem *first);
int xrhip_ba_solve_linked(xrhip_ba *ctx_second, const xrhip_ba_problem *second, xrhip_ba_summary *summary_second, int link_second,
                          xrhip_ba *ctx_first, int link_first, void (*host_work)(void *), void *arg);
int xrhip_ba_solve_end(xrhip_ba *ctx, xrhip_ba_summary *summary);
/* the unwind path of an owner that cannot reach _end: waits for a begun solve and forgets it (nothing is written back) */
int xrhip_ba_solve_abort(xrhip_ba *ctx);

/* HIP-event profiling of the dominant BA kernel (kb_solve_try: reduced-system Cholesky + trust-region trials),
 * off by default.  flops = algorithmic work of the launches:
 * na^3/3 + 2 na^2 per launch + (450 M + 3000 NI + 2 np^2) per trial costed. */
typedef struct xrhip_ba_stats {
    long n_solve_try;        /* launches */
    long n_trials;           /* trust-region trials evaluated by them */
    double ms_solve_try;     /* sum of HIP-event durations (only launches made while profiling was on) */
    long n_timed;            /* launches that contributed to ms_solve_try */
    double flops_solve_try;  /* algorithmic flops of the timed launches */
    long n_tiny;             /* solves that ran as ONE launch (kb_chain / kb_tiny: no free landmark, a few free frames); not in the above */
    /* the single-launch solves (kb_chain): HIP-event durations of the launches made while profiling was on, and their
     * ALGORITHMIC bytes (384 B per reprojection factor and linearisation, 280 B per factor and candidate
     * costed, the packed reduced system once per round) -- what bench.py divides by the HBM peak.  A member of an instance group does
     * not time its own launches (xrhip_group_stats has the batches' durations): its solves still count here -- n_chain_timed and
     * bytes_chain grow, ms_chain does not. */
    long n_chain_timed;
    double ms_chain;
    double bytes_chain;
} xrhip_ba_stats;
int xrhip_ba_set_profiling(xrhip_ba *ctx, int enable);
int xrhip_ba_get_stats(xrhip_ba *ctx, xrhip_ba_stats *out, int reset);
/* replaces: MarginalizationFactor::marginalize(index).  Outputs the new prior over the n_frames-1
 * remaining frames (map order): sqrt_info [(15(n-1))^2], infovec [15(n-1)], lin [(n-1)][16]. */
int xrhip_ba_marginalize(xrhip_ba *ctx, const xrhip_marg_problem *problem, double *out_sqrt_info,
                         double *out_infovec, double *out_lin);
/* asynchronous form: _begin stages the problem and queues all of its device work, _end waits and returns the prior.
 * One marginalisation in flight per context, and no other call on that context in between (give the marginalisation a
 * context of its own -- its stream then runs beside the tracker and the solves of the next frame). */
int xrhip_ba_marginalize_begin(xrhip_ba *ctx, const xrhip_marg_problem *problem);
int xrhip_ba_marginalize_end(xrhip_ba *ctx, double *out_sqrt_info, double *out_infovec, double *out_lin);
/* replaces: PreIntegrator::integrate(t, bg, ba, compute_jacobian, compute_covariance)
 * (preintegrator.cpp:78-100).  samples: [n][7] = t, w(3), a(3); noise: cov_w, cov_a, cov_bg, cov_ba as
 * 3x3 row-major (36 doubles).  out: XRHIP_IMU_DIM doubles. */
int xrhip_ba_preintegrate(xrhip_ba *ctx, const double *samples, int n, double t_end, const double *bg,
                          const double *ba, const double *noise_cov36, int compute_jacobian,
                          int compute_covariance, double *out);
/* batched form (one workgroup per IMU segment): job k integrates samples[sample_begin[k] .. +sample_count[k])
 * up to t_end[k] at biases bg[3k..], ba[3k..]; out: [n_jobs][XRHIP_IMU_DIM]. */
int xrhip_ba_preintegrate_batch(xrhip_ba *ctx, const double *samples, const int *sample_begin,
                                const int *sample_count, const double *t_end, const double *bg, const double *ba,
                                int n_jobs, const double *noise_cov36, int compute_jacobian, int compute_covariance,
                                double *out);

/* asynchronous form of the batch: _begin queues the integrations and returns, _end waits for them and copies the
 * records out.  One batch in flight per context; no other call on the context may use the staging block in between
 * (xrhip_ba_marginalize does; xrhip_ba_solve does not). */
int xrhip_ba_preintegrate_begin(xrhip_ba *ctx, const double *samples, const int *sample_begin, const int *sample_count,
                                const double *t_end, const double *bg, const double *ba, int n_jobs,
                                const double *noise_cov36, int compute_jacobian, int compute_covariance);
int xrhip_ba_preintegrate_end(xrhip_ba *ctx, double *out);
/* Between _begin and _end: the DELTA of job `job` -- doubles [0..10] of its record: dt, dq, dp, dv -- as soon as the kernel has it,
 * i.e. before the covariance / Jacobian / sqrt_inv_cov part of the record exists.  The delta does not depend on compute_jacobian /
 * compute_covariance (same expressions): FeatureTracker::work's integrate(t, bg, ba, false, false) of an interval
 * (core/feature_tracker.cpp:75-77) and mirror_frame's integrate(t, bg, ba, true, true) of the same samples at the same biases
 * (sliding_window_tracker.cpp:54-56) are ONE launch here; the tracker takes the delta early, the backend collects the record with
 * _end.  The batch stays in flight. */
int xrhip_ba_preintegrate_early(xrhip_ba *ctx, int job, double *out_delta11);
/* the unwind path of an owner that cannot reach _end (an error between the two calls): waits for the batch's kernel and
 * forgets it, staged-behind-a-solve batches included.  A second _begin without _end or _cancel fails with XRHIP_ESTATE. */
int xrhip_ba_preintegrate_cancel(xrhip_ba *ctx);
/* A batch whose integrations start from biases the NEXT xrhip_ba_solve on this context is about to produce: job k uses
 * bg / ba of frame bias_frame[k] of that problem as the solve leaves them (PreIntegrator::integrate(t, bg, ba, ...) of the
 * reference called right after Solver::solve with frame->motion.bg / .ba, e.g. sliding_window_tracker.cpp:54-56 after :441).
 * The call only stages the batch; the solve launches it behind its last kernel (the single-launch solves: on the device,
 * reading the biases where that kernel leaves them, so the integration needs no host round trip in between), and
 * xrhip_ba_preintegrate_end collects it as usual.  Same values as _begin with the biases read back by the host. */
int xrhip_ba_preintegrate_after_solve(xrhip_ba *ctx, const double *samples, const int *sample_begin, const int *sample_count,
                                      const double *t_end, const int *bias_frame, int n_jobs, const double *noise_cov36,
                                      int compute_jacobian, int compute_covariance);

/* ------------------------------------------------------------------------
 * Instance group: several sequences on one GPU served by shared launches.
 *
 * replaces: nothing the reference has -- its process-global state (XRSLAMManager singleton, xrslam-interface/src/XRSLAMManager.cpp:6-9;
 * static id counters, xrslam/src/xrslam/utility/identifiable.h:23-30) allows ONE sequence per process, and BASELINE config 4 puts
 * eleven sequences on eight GPUs.  Contexts that have joined a group hand the launches of the per-frame path (frame upload, CLAHE /
 * pyramid, LK, Harris, pre-integration, the single-launch solves) to the group's submission thread, which issues ONE launch per
 * kernel for all requests pending at that moment (blockIdx.z = request); since round 5 the rounds of a window-sized solve travel the same
 * way (one request per trust-region round); marginalisations keep the context's own stream.  Results are those of the context running alone, bit for bit (same kernels, same per-request block mapping and
 * summation order); every entry point keeps its meaning, including the blocking ones (they wait on the context's own mailbox).
 * A context may be driven by one thread at a time as before; different contexts of a group from different threads.
 * Join right after creation, before the context's first use; leave (group = NULL) or destroy the context before the group.
 * ---------------------------------------------------------------------- */
typedef struct xrhip_group xrhip_group;
int xrhip_group_create(xrhip_group **out);
int xrhip_group_destroy(xrhip_group *group);   /* XRHIP_ESTATE while contexts are still joined */
int xrhip_klt_join_group(xrhip_klt *ctx, xrhip_group *group);
int xrhip_ba_join_group(xrhip_ba *ctx, xrhip_group *group);
/* Frame gate (round 5): a sequence calls xrhip_klt_frame_gate when it is about to upload a new frame; the call returns when every
 * member of the group that is expected to start a frame has arrived (or after a timeout, default 2.5 ms: members that do not come stop
 * being waited for until they do).  Members whose frames start together issue their uploads, pyramids, tracking launches and solves
 * together, so one launch carries all of them.  xrhip_klt_group_busy(ctx, 1 / 0) brackets a stretch during which the sequence will
 * not start a frame (a keyframe's window solve and marginalisation): the others go on without it.  No-ops outside a group; timing only --
 * no result depends on either call.  (stats slot XRHIP_GK_GATE: gate openings / with everybody present / by timeout.) */
int xrhip_klt_frame_gate(xrhip_klt *ctx);
int xrhip_klt_group_busy(xrhip_klt *ctx, int busy);
/* request kinds of the statistics below */
#define XRHIP_GK_CALL 0        /* un-batched calls run in queue order (rare paths: parity aids, the five-launch pyramid) */
#define XRHIP_GK_UPLOAD 1      /* k_upload */
#define XRHIP_GK_PREPROCESS 2  /* k_clahe_lut + k_pyr_a + k_pyr_b */
#define XRHIP_GK_TRACK 3       /* k_lk_track (+ the prefetched Harris pass of the target image) */
#define XRHIP_GK_DETECT 4      /* k_harris + k_harris_nms + k_harris_select */
#define XRHIP_GK_CHAIN 5       /* kb_stage + kb_chain (+ kp_preintegrate queued behind the solve) */
#define XRHIP_GK_PREINT 6      /* kp_preintegrate */
#define XRHIP_GK_WROUND 7      /* one trust-region round of a window solve: kb_lin_all .. kb_solve_try + the first kb_trials_wide (round 5) */
#define XRHIP_GK_WTRIALS 8     /* kb_trials_wide of a run of rejected trials */
#define XRHIP_GK_GATE 11       /* not a request kind: the frame gate's openings / with everybody present / by timeout */
typedef struct xrhip_group_stats {
    long long batches[12];   /* batches launched, per request kind */
    long long entries[12];   /* requests they served (entries / batches = sequences per launch) */
    double ms[12];           /* HIP-event duration of the timed batches, first to last kernel (xrhip_group_set_profiling) */
    long long timed[12];     /* batches that contributed to ms */
} xrhip_group_stats;
int xrhip_group_set_profiling(xrhip_group *group, int enable);
int xrhip_group_get_stats(xrhip_group *group, xrhip_group_stats *out, int reset);
/* 1: the group's hardware-queue split (two queues for the batches, two for the members' window solves) is in effect -- it needs
 * GPU_MAX_HW_QUEUES=2 in the environment before the process's first HIP call; 0: the slower unprioritised arrangement (same results,
 * ~15 % less grouped throughput); < 0: error.  (Round 5 printed this to stderr; XRHIP_GROUP_VERBOSE=1 still does.) */
int xrhip_group_queue_split(xrhip_group *group);

/* parity/testing aids (not part of the reference interface): the unreduced normal equations of one
 * linearisation, and the MFMA Schur product kernel on arbitrary inputs. */
int xrhip_ba_debug_linearize(xrhip_ba *ctx, const xrhip_ba_problem *problem, double *H, double *g, double *hll,
                             double *gl, double *W, double *cost);
int xrhip_ba_debug_schur(xrhip_ba *ctx, const double *W, const double *w, int L, int P, double *out);
/* parity aid (round 5): the eigenvalue bound the Cholesky fast path of the LAST collected marginalisation on this context computed
 * (lambda_min >= 1 / trace(A^-1), csrc/marg_kernels.hip.h: km_chol) and its eight status words ([1] support size, [2] Cholesky failed /
 * sweeps, [3] guard failed, [4] the eigen path ran) -- tests hold the bound to numpy's on the same matrix. */
int xrhip_ba_debug_marg_guard(xrhip_ba *ctx, double *lambda_bound, int *status8);
/* study aid (BASELINE.json config 5, "fp32 vs bf16 BA solve"): mode 1 / 2 run the Schur contraction of the following solves on this
 * context with f32 / bf16 matrix-core operands; mode 0 (the default, and what the product always uses) is f64. */
int xrhip_ba_debug_set_schur_precision(xrhip_ba *ctx, int mode);
/* BASELINE config 5's precision study on the device (csrc/study_api.hip; not part of the reference interface, never called by
 * the product path): T = W^T diag(w) W for W [L][P] row-major, w [L] >= 0, computed by the same tiling with f64, f32 and
 * bf16 (f32 accumulation) matrix-core operands; out64 / out32 / out16: [P][P]; ms_per_launch[3]: average of `reps` launches of
 * each kernel (HIP events on the stream they run on). */
int xrhip_study_schur_precision(const double *W, const double *w, int L, int P, int reps, double *out64, double *out32,
                                double *out16, float ms_per_launch[3]);
/* development aid: accumulated in-kernel phase timers (100 MHz ticks) of the BA kernels; all zero unless the
 * library was built with -DXRHIP_KPROF (csrc/build.sh, XR_VARIANT=kprof) */
void xrhip_debug_kprof(long long *out32, int reset);
/* LK phase timers of -DXRHIP_KPROF builds (shader cycles summed over all points: 0-5 phases, 6 points, 7 kernel; zeros otherwise) */
void xrhip_debug_lkprof(long long *out8, int reset);

#ifdef __cplusplus
}
#endif
#endif /* XRSLAM_HIP_H */
