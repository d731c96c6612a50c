/*
 * XRSLAM.h -- outer C ABI of the MI355X-native XRSLAM hot path.
 *
 * Binary-compatible with the reference's public interface
 * (/root/reference/xrslam-interface/include/XRSLAM.h:19-229): same exported symbols,
 * enum values and POD layouts, so a host application built against the reference
 * (xrslam-pc/player/src/main.cpp:116-169, xrslam-ros/src/xrslam_node.cpp:44) links
 * against libxrslam_hip.so unchanged.  Implementation: xrslam_amd/csrc/host/xrslam_api.cpp.
 *
 * Differences, all additive:
 *   - the XRSLAMAmd* entry points at the bottom (optional externally supplied initial
 *     states, device-resident images, stage timers, initialiser report), and their
 *     XRSLAMAmdInstance* forms for several sequences per process;
 *   - XRSLAMFeatures is declared for C++ only, like the reference (it holds a std::vector).
 */
#ifndef XRSLAM_AMD_XRSLAM_H
#define XRSLAM_AMD_XRSLAM_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
#include <vector>
extern "C" {
#endif

/* ---- sensor inputs ---- */
typedef enum XRSLAMSensorType {
    XRSLAM_SENSOR_CAMERA = 0,
    XRSLAM_SENSOR_DEPTH_CAMERA,
    XRSLAM_SENSOR_ACCELERATION,
    XRSLAM_SENSOR_GYROSCOPE,
    XRSLAM_SENSOR_GRAVITY,
    XRSLAM_SENSOR_ROTATION_VECTOR,
    XRSLAM_SENSOR_UNKNOWN
} XRSLAMSensorType;

typedef struct XRSLAMImageExtension {
    double exposure_time, default_focus_distance, focal_length, focus_distance;
} XRSLAMImageExtension;

typedef struct XRSLAMImage {
    unsigned char *data; /* 8-bit pixels */
    double timeStamp;    /* seconds */
    int stride;          /* bytes per row */
    int camera_id;       /* only 0 is consumed */
    int channel;         /* 1 (gray), 3 (BGR) or 4 (BGRA) */
    XRSLAMImageExtension *ext;
} XRSLAMImage;

typedef struct XRSLAMDepthImage {
    uint16_t *data, *confidence;
    double timeStamp;
} XRSLAMDepthImage;

typedef struct XRSLAMAcceleration {
    double data[3];
    double timestamp;
} XRSLAMAcceleration;
typedef struct XRSLAMGyroscope {
    double data[3];
    double timestamp;
} XRSLAMGyroscope;
typedef struct XRSLAMGravity {
    double data[3];
    double timestamp;
} XRSLAMGravity;
typedef struct XRSLAMRotationVector {
    double data[4];
    double timestamp;
} XRSLAMRotationVector;

/* ---- results ---- */
typedef enum XRSLAMResultType {
    XRSLAM_RESULT_BODY_POSE = 0,
    XRSLAM_RESULT_CAMERA_POSE,
    XRSLAM_RESULT_STATE,
    XRSLAM_RESULT_LANDMARKS,
    XRSLAM_RESULT_FEATURES,
    XRSLAM_RESULT_BIAS,
    XRSLAM_RESULT_DEBUG_LOGS,
    XRSLAM_RESULT_VERSION,
    XRSLAM_RESULT_UNKNOWN,
    XRSLAM_INFO_INTRINSICS
} XRSLAMResultType;

typedef struct XRSLAMPose {
    double quaternion[4];  /* x y z w */
    double translation[3];
    double timestamp;
} XRSLAMPose;

typedef struct XRSLAMIntrinsics {
    double fx, fy, cx, cy;
} XRSLAMIntrinsics;

typedef enum XRSLAMState {
    XRSLAM_STATE_INITIALIZING,
    XRSLAM_STATE_TRACKING_SUCCESS,
    XRSLAM_STATE_TRACKING_FAIL
} XRSLAMState;

typedef struct XRSLAMLandmark {
    double x, y, z;
} XRSLAMLandmark;
typedef struct XRSLAMLandmarks {
    XRSLAMLandmark *landmarks;
    int num_landmarks;
} XRSLAMLandmarks;

typedef struct XRSLAMFeature {
    double x, y;
} XRSLAMFeature;
#ifdef __cplusplus
typedef struct XRSLAMFeatures {
    struct Point {
        double x;
        double y;
    };
    std::vector<Point> pos;
} XRSLAMFeatures;
#endif

typedef struct XRSLAMBias {
    double data[3];
} XRSLAMBias;
typedef struct XRSLAMIMUBias {
    XRSLAMBias acc_bias;
    XRSLAMBias gyr_bias;
} XRSLAMIMUBias;

typedef struct XRSLAMStringOutput {
    int str_length;
    char *data;
} XRSLAMStringOutput;

/* ---- entry points (reference XRSLAM.h:201-229) ---- */
/* returns 1 on success, 0 otherwise; *config receives an opaque configuration handle owned by the library */
int XRSLAMCreate(const char *slam_config_path, const char *device_config_path, const char *license_path,
                 const char *product_name, void **config);
void XRSLAMPushSensorData(XRSLAMSensorType sensor_type, void *sensor_data);
void XRSLAMRunOneFrame();
void XRSLAMSetViewer(void *viewer); /* declared by the reference, defined nowhere there; a no-op here */
void XRSLAMGetResult(XRSLAMResultType result_type, void *result_data);
void XRSLAMDestroy();

/* ---- additive MI355X entry points ---- */
/* Optional: body pose/velocity/biases at image time t (q: x y z w).  While any such state is registered the
 * first window is seeded from them (poses of its 8 keyframes) instead of SfM + IMU alignment
 * (core/initializer.cpp:158-571); with none registered the library initialises itself like the reference. */
void XRSLAMAmdSetInitialState(double t, const double q[4], const double p[3], const double v[3],
                              const double bg[3], const double ba[3]);
/* like XRSLAM_SENSOR_CAMERA but `gray_dev` is an 8-bit single-channel image already resident in HBM */
void XRSLAMAmdPushImageDevice(const void *gray_dev, int stride, double timestamp);
/* What the reference's dataset readers ask the YamlConfig* for (xrslam-pc/player/src/IO/euroc_dataset_reader.cpp:4-7,16,62-66;
 * tum_dataset_reader.cpp:4-6,67-76): camera_time_offset(), camera_distortion_flag(), camera_distortion(),
 * camera_intrinsic(), camera_resolution().  The `config` out-parameter of XRSLAMCreate is an opaque handle here (the
 * reference hands out a C++ object whose virtuals the player calls -- not something a C ABI can promise), so a reader that
 * is built against this library takes those five values from this call instead (INTEGRATION.md section 1 shows the patch). */
typedef struct XRSLAMAmdCameraConfig {
    double time_offset;       /* cam0.time_offset */
    int distortion_flag;      /* cam0.camera_distortion_flag */
    double distortion[4];     /* cam0.distortion: k1 k2 p1 p2 (radtan) or k1..k4 (equidistant) */
    double intrinsics[4];     /* cam0.intrinsics: fx fy cx cy */
    int resolution[2];        /* cam0.resolution: width height */
} XRSLAMAmdCameraConfig;
void XRSLAMAmdGetCameraConfig(XRSLAMAmdCameraConfig *out);
/* Every value the two configuration files resolved to (the accessors of xrslam::Config, xrslam/include/xrslam/xrslam.h:36-116 /
 * yaml_config.cpp:152-362), one "key = value" line each (%.17g), keys as in the yaml files ("cam0.intrinsics", "sliding_window.size",
 * ...): lets a caller -- and tests/test_config.py -- check that two sets of files mean the same configuration.  Writes at most cap - 1
 * characters plus the terminator; returns the length the full text needs. */
int XRSLAMAmdDescribeConfig(char *buf, int cap);
/* Undistortion on the device.  The reference's readers rectify every frame on the host before pushing it
 * (cv::undistort, xrslam-pc/player/src/IO/euroc_dataset_reader.cpp:62-69; xrslam::extra::ImageUndistorter,
 * IO/tum_dataset_reader.cpp:67-76).  After this call the images handed to XRSLAM_SENSOR_CAMERA / XRSLAMAmdPushImageDevice
 * are taken as the camera recorded them and rectified on the GPU with cam0.intrinsics / cam0.distortion of the device
 * configuration -- bit-identical to the host path (the map is built once, on the host, in the reference's arithmetic).
 * model: "cv_undistort" (EuRoC reader), "radtan" or "equidistant" (ImageUndistorter); NULL or "" switches it off. */
void XRSLAMAmdSetDeviceUndistort(const char *model);
typedef struct XRSLAMAmdTimes {
    long frames, solves, solve_iterations, marginalizations, keyframes;
    double ba_device_ms; /* sum of xrhip_ba_summary.ms_solve (staging + kernels + result read-back of every solve) */
    /* host wall-clock seconds inside the inner C-ABI calls (preprocess, track, detect, preintegrate, solve,
     * marginalize) and in the whole per-frame work (FeatureTracker::work incl. the backend) */
    double wall_preprocess, wall_track, wall_detect, wall_preintegrate, wall_solve, wall_marginalize, wall_frame;
    /* host wall-clock seconds of whole pipeline stages (their device waits included): Frame::track_keypoints,
     * of which 5-pt RANSAC, 2-pt RANSAC; Frame::detect_keypoints; mirror_frame; localize_newframe; manage_keyframe;
     * track_landmark; refine_window; slide_window; refine_subwindow; initialiser (SfM + alignment attempts);
     * [12], [13] are COUNTS of the RD-VIO filter (parsac.parsac_flag): frames on which judge_track_status separated a
     * dynamic group, landmark observations it tagged as outliers; [14] COUNTS the constant copies the solver front end made for
     * prior factors whose reference frame / landmark was also a free parameter of the same solve; [15] pipelined mode
     * (XRSLAMAmdSetThreading): seconds the feature tracker waited for the previous frame's backend at the hand-off */
    double wall_scope[16];
} XRSLAMAmdTimes;
void XRSLAMAmdGetTimes(XRSLAMAmdTimes *out);
/* HIP-event profiling of the KLT kernels (off by default) and its accumulated counters; the struct is
 * xrhip_klt_stats from xrslam_hip.h (passed as void* to keep this header free of that include) */
void XRSLAMAmdSetProfiling(int enable);
void XRSLAMAmdGetKltStats(void *xrhip_klt_stats_out, int reset);
/* same for the BA context: xrhip_ba_stats from xrslam_hip.h (XRSLAMAmdSetProfiling switches both) */
void XRSLAMAmdGetBaStats(void *xrhip_ba_stats_out, int reset);
/* What the initialiser (core/initializer.cpp) did so far: attempts that reached SfM, whether the last one
 * succeeded, which of the eight (R, T) hypotheses won its triangulation vote, and the IMU alignment it produced
 * (metric scale of the unit-baseline SfM map, gravity in the SfM frame, gyroscope bias). */
typedef struct XRSLAMAmdInitReport {
    long attempts, successes;
    int sfm_candidate, sfm_triangulated;
    double scale, gravity[3], bg[3];
} XRSLAMAmdInitReport;
void XRSLAMAmdGetInitReport(XRSLAMAmdInitReport *out);
/* last error raised inside the library ("" if none); the reference aborts/throws instead */
const char *XRSLAMAmdLastError(void);
/* Threading.  0 (default) = the reference's PC build: XRSLAMRunOneFrame / the IMU push that completes a frame run feature
 * tracker and sliding-window tracker inline, one after the other (utility/worker.h:37-45 without XRSLAM_ENABLE_THREADING).
 * 1 = its XRSLAM_ENABLE_THREADING build (utility/worker.h:16-60; feature tracker and frontend on threads of their own) with
 * deterministic hand-offs: the sliding-window tracker of frame t runs on a thread of the library beside the feature tracker of
 * frame t+1, which sees the state published for frame t-1 and propagates it by pre-integration exactly as
 * FeatureTracker::work does when the frontend lags (core/feature_tracker.cpp:44-66).  Same arithmetic, same kernels; poses
 * and landmark sets are those of a run whose backend is one frame late, reproducible bit for bit, and NOT those of mode 0.
 * With parsac.parsac_flag the frames stay inline (update_track_status reads the tracking map from inside the backend).
 * May be switched between frames.  XRSLAMAmdFlush waits for the frame in flight (also done by Destroy, by the LANDMARKS / BIAS
 * results and by the statistics getters). */
void XRSLAMAmdSetThreading(int mode);
void XRSLAMAmdFlush(void);

/* ---- instance-scoped entry points (additive) ----
 * The six reference symbols above act on one process-global instance, like the reference's XRSLAMManager singleton
 * (xrslam-interface/src/XRSLAMManager.cpp:6-9).  The reference cannot do otherwise -- solver configuration, CLAHE /
 * GFTT objects, id counters and the RD-VIO bin confidences are function- or class-level statics there --
 * here all of that state lives in the instance, so one process can run several independent sequences on one GPU
 * (each instance owns its HIP streams; kernels of different instances overlap on the device).  Same semantics as the
 * global entry points, with the instance as first argument.  One thread at a time per instance; different instances
 * may be driven from different threads.  An instance belongs to the HIP device that was current when it was created. */
typedef struct XRSLAMAmdInstance XRSLAMAmdInstance;
int XRSLAMAmdInstanceCreate(const char *slam_config_path, const char *device_config_path, XRSLAMAmdInstance **out,
                            void **config);
void XRSLAMAmdInstanceDestroy(XRSLAMAmdInstance *inst);
void XRSLAMAmdInstancePushSensorData(XRSLAMAmdInstance *inst, XRSLAMSensorType sensor_type, void *sensor_data);
void XRSLAMAmdInstanceRunOneFrame(XRSLAMAmdInstance *inst);
void XRSLAMAmdInstanceGetResult(XRSLAMAmdInstance *inst, XRSLAMResultType result_type, void *result_data);
void XRSLAMAmdInstanceSetInitialState(XRSLAMAmdInstance *inst, double t, const double q[4], const double p[3],
                                      const double v[3], const double bg[3], const double ba[3]);
void XRSLAMAmdInstancePushImageDevice(XRSLAMAmdInstance *inst, const void *gray_dev, int stride, double timestamp);
void XRSLAMAmdInstanceGetCameraConfig(XRSLAMAmdInstance *inst, XRSLAMAmdCameraConfig *out);
int XRSLAMAmdInstanceDescribeConfig(XRSLAMAmdInstance *inst, char *buf, int cap);
void XRSLAMAmdInstanceSetDeviceUndistort(XRSLAMAmdInstance *inst, const char *model);
void XRSLAMAmdInstanceGetTimes(XRSLAMAmdInstance *inst, XRSLAMAmdTimes *out);
void XRSLAMAmdInstanceSetProfiling(XRSLAMAmdInstance *inst, int enable);
void XRSLAMAmdInstanceGetBaStats(XRSLAMAmdInstance *inst, void *xrhip_ba_stats_out, int reset);
void XRSLAMAmdInstanceGetKltStats(XRSLAMAmdInstance *inst, void *xrhip_klt_stats_out, int reset);
void XRSLAMAmdInstanceGetInitReport(XRSLAMAmdInstance *inst, XRSLAMAmdInitReport *out);
const char *XRSLAMAmdInstanceLastError(XRSLAMAmdInstance *inst);
void XRSLAMAmdInstanceSetThreading(XRSLAMAmdInstance *inst, int mode);
void XRSLAMAmdInstanceFlush(XRSLAMAmdInstance *inst);
/* ---- instance groups (additive) ----
 * S independent sequences on one GPU leave it waiting on its own front end: every sequence issues ~26 small dependent launches per
 * frame.  Instances that have joined a group keep their state, ids and results to themselves but share launches: the library issues
 * ONE launch per kernel of the per-frame path (frame upload, CLAHE / pyramid, LK, Harris, pre-integration, the localisation and
 * sub-window solves) for all members that are waiting for it at that moment.  The trajectory of a member is the one it has alone,
 * bit for bit.  Usage: create the group on the device, create the instances, join each before its first frame, drive every instance
 * from a thread of its own (XRSLAMAmdInstanceReplay or the per-sample entry points); destroy the instances (or leave: group NULL)
 * before the group.  Returns 1 on success, 0 otherwise (XRSLAMAmdLastError / XRSLAMAmdInstanceLastError).
 * XRSLAMAmdGroupGetStats fills an xrhip_group_stats (xrslam_hip.h): batches and requests per kind -- requests / batches is the
 * number of sequences a launch served. */
typedef struct XRSLAMAmdGroup XRSLAMAmdGroup;
int XRSLAMAmdGroupCreate(XRSLAMAmdGroup **out);
int XRSLAMAmdGroupDestroy(XRSLAMAmdGroup *group);
int XRSLAMAmdInstanceJoinGroup(XRSLAMAmdInstance *inst, XRSLAMAmdGroup *group);
void XRSLAMAmdGroupSetProfiling(XRSLAMAmdGroup *group, int enable);
void XRSLAMAmdGroupGetStats(XRSLAMAmdGroup *group, void *xrhip_group_stats_out, int reset);
/* The player's loop (xrslam-pc/player/src/main.cpp:116-169) over a pre-staged sequence for the next `n_steps` camera frames:
 * IMU samples up to each frame's time (gyroscope before accelerometer, IO/async_dataset_reader.cpp:41-48), the frame,
 * XRSLAMRunOneFrame, state / body-pose query.  imu7: [n_imu][7] = t, gyroscope xyz, accelerometer xyz; cam_t: [n_frames];
 * frames: n_frames 8-bit images of frame_bytes each, host memory or (on_device != 0) HBM; the cursors are advanced.
 * poses_out8 (may be NULL): [n_steps][8] = t, translation xyz, quaternion xyzw of every frame answered in state
 * TRACKING_SUCCESS.  Returns the number of poses written, -1 on bad arguments.  Lets a harness drive several instances from
 * several threads without a host-language call per sensor sample. */
int XRSLAMAmdInstanceReplay(XRSLAMAmdInstance *inst, const double *imu7, int n_imu, const double *cam_t, int n_frames,
                            const void *frames, size_t frame_bytes, int stride, int on_device, int *imu_cursor,
                            int *frame_cursor, int n_steps, double *poses_out8);

#ifdef __cplusplus
}
#endif
#endif /* XRSLAM_AMD_XRSLAM_H */
