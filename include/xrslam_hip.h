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
/* number of usable gfx950 devices (0 if none); never fails */
int xrhip_device_count(void);
/* binds the calling thread (and contexts created afterwards) to a device */
int xrhip_set_device(int device);

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
 * the timed region starts with inputs resident on the device). */
int xrhip_image_upload_device(xrhip_image *img, const void *gray_dev, int stride_bytes);
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
} xrhip_klt_stats;
int xrhip_klt_set_profiling(xrhip_klt *ctx, int enable);
int xrhip_klt_get_stats(xrhip_klt *ctx, xrhip_klt_stats *out, int reset);
int xrhip_klt_synchronize(xrhip_klt *ctx);

#ifdef __cplusplus
}
#endif
#endif /* XRSLAM_HIP_H */
