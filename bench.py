#!/usr/bin/env python3
"""bench.py -- frames/sec + ms/BA-iteration of the XRSLAM per-frame hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W        (one JSON line on rank 0)

A "step" is one camera frame of one sequence pushed through the whole hot path behind the
reference's C API (include/XRSLAM.h, the player's call sequence of xrslam-pc/player/src/main.cpp:116-169):
~10 gyro + ~10 accel samples, XRSLAMAmdPushImageDevice (frame already resident in HBM), XRSLAMRunOneFrame,
XRSLAMGetResult.  Inside: CLAHE + LK pyramid + Scharr, IMU pre-integration, forward/backward pyramidal LK,
5-pt/2-pt RANSAC gates, Harris re-detection, localize_newframe solve, keyframe policy, landmark
triangulation, refine_window / refine_subwindow dogleg solves, marginalisation.

Workload (config.workload): synthetic S1 "EuRoC MH_01-like" stream: 752x480 (the
real EuRoC cam0 size; BASELINE.json's 640x480 is a known discrepancy, SURVEY.md top table), 20 Hz camera /
200 Hz IMU, 150 features, 10-keyframe window (BASELINE config 2), seeded box-room scene.  The first 36
frames of a stream only seed the window (initial states supplied from the ground truth, so that every run
measures the same steady-state path): they are never timed -- with --warmup W < 40 the missing 40 - W frames
run as an untimed pre-roll in front of the W warmup steps (config.untimed_preroll_frames).
One independent sequence per GPU: no data-path collective, only a barrier and a MAX
reduction of the wall time over RCCL.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
F64_MFMA_PEAK_TFLOPS = 78.6   # MI355X dense FP64 matrix peak (MI355X_MICROARCH.md)
SLAM_YAML = os.path.join(ROOT, "configs", "bench_slam_150.yaml")
SENSOR_YAML = os.path.join(ROOT, "configs", "euroc_sensor.yaml")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=150)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--cpu-frames", type=int, default=240,
                    help="frames of the bounded CPU-reference sample (0 = skip); the first 36 only seed the window")
    ap.add_argument("--no-profile", action="store_true", help="do not record HIP events around the KLT kernels")
    ap.add_argument("--backend", default=None, help="torch.distributed backend (default: nccl == RCCL); 'gloo' lets two "
                    "ranks share one GPU when the multi-process path is exercised on a single-GPU box")
    args = ap.parse_args()
    if args.steps < 1 or args.warmup < 0:
        raise SystemExit("--steps must be >= 1 and --warmup >= 0")
    # the first 36 frames of a stream only seed the sliding window (no pose, no solve): they never fall into the timed
    # region -- a warmup shorter than 40 steps is preceded by the missing frames as an untimed pre-roll
    preroll = max(0, 40 - args.warmup)

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (the product has no CPU fallback)")
    from xrslam_amd import _lib
    from xrslam_amd.harness import runner, scene
    from xrslam_amd.harness.dist import RunGroup
    device_index = int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count()   # == LOCAL_RANK on a full node
    torch.cuda.set_device(device_index)
    group = RunGroup(backend=args.backend)  # one process per GPU; "nccl" == RCCL over xGMI
    rank, local_rank, world = group.rank, group.local_rank, group.world
    _lib.set_device(device_index)

    n_frames = preroll + args.warmup + args.steps
    seq = scene.make_sequence(n_frames=n_frames + 1, seed=1 + rank)
    dev = torch.from_numpy(seq["frames"]).cuda()     # inputs resident in HBM before the timed region
    torch.cuda.synchronize()
    h, w = seq["frames"].shape[1:]
    sess = runner.Session(_lib.LIB_PATH, seq, slam_yaml=SLAM_YAML, sensor_yaml=SENSOR_YAML,
                          device_frames=(dev.data_ptr(), h * w, w))

    def barrier():
        group.barrier()
        torch.cuda.synchronize()

    for _ in range(preroll + args.warmup):
        sess.step()
    if sess.error():
        raise SystemExit("warmup failed: " + sess.error())
    t_w = sess.times()
    sess.klt_stats(reset=True)
    sess.ba_stats(reset=True)
    if not args.no_profile:
        sess.set_profiling(True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        sess.step()
    barrier()
    elapsed = time.perf_counter() - t0
    if sess.error():
        raise SystemExit("timed region failed: " + sess.error())
    t_e = sess.times()
    st = sess.klt_stats(reset=False)
    bst = sess.ba_stats(reset=False)
    sess.set_profiling(False)
    red = group.reduce_metrics(args.steps, elapsed)      # frames: SUM, wall seconds: MAX over ranks
    elapsed, total_frames = red["seconds"], red["frames"]

    if rank == 0:
        poses = list(sess.poses)
        iters = max(1, t_e.solve_iterations - t_w.solve_iterations)
        solves = max(1, t_e.solves - t_w.solves)
        ba_ms = t_e.ba_device_ms - t_w.ba_device_ms
        n_launch = max(1, st.n_track)
        lk_bytes = 2420.0 * st.lk_templates + 484.0 * st.lk_iterations     # SURVEY.md section 8d, B_lk
        lk_ms = st.ms_track / n_launch
        achieved = (lk_bytes / n_launch) / (lk_ms * 1e-3) / 1e9 if lk_ms > 0 else 0.0
        ba_tflops = bst.flops_solve_try / (bst.ms_solve_try * 1e-3) / 1e12 if bst.ms_solve_try > 0 else 0.0
        # HBM-side bytes per launch: PMC counters cannot be read in-process; they come from the committed rocprofv3
        # --pmc passes of this same command (profiles/r*_pmc_traffic.md says how they were collected)
        traffic = {}
        try:
            import glob
            with open(sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))[-1]) as fh:   # newest round
                pmc = json.load(fh)
            traffic = {k: round(1024.0 * (v["fetch_kb"] + v["write_kb"]), 1) for k, v in pmc.items()}
        except (OSError, ValueError, KeyError, IndexError):
            pass
        out = {
            "metric": "frames/sec, full per-frame hot path (KLT tracker + sliding-window VI-BA), one sequence per GPU",
            "value": round(total_frames / elapsed, 3),
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8/i16 images, f32 LK with exact i64 reductions, f64 BA",
            "data": "synthetic",
            "config": {"workload": "S1 EuRoC-MH_01-like synthetic stream, 752x480 @20 Hz + 200 Hz IMU, 150 features, "
                                   "10-keyframe window, 30-iteration dogleg cap (BASELINE config 2)",
                       "features": 150, "window_keyframes": 10, "sequences_per_gpu": 1, "untimed_preroll_frames": preroll},
            "ms_per_ba_iteration": round(ba_ms / iters, 4),
            "ba": {"solves_per_frame": round(solves / args.steps, 3), "iterations_per_solve": round(iters / solves, 2),
                   "device_ms_per_solve": round(ba_ms / solves, 4),
                   "keyframes": int(t_e.keyframes - t_w.keyframes),
                   "marginalizations": int(t_e.marginalizations - t_w.marginalizations)},
            "stage_kernel_ms": {"preprocess": round(st.ms_preprocess / max(1, st.n_preprocess), 4),
                                "lk_track": round(lk_ms, 4),
                                "detect": round(st.ms_detect / max(1, st.n_detect), 4)},
            "host_wall_ms_per_frame": {k[5:]: round(1e3 * (getattr(t_e, k) - getattr(t_w, k)) / args.steps, 4)
                                       for k in ("wall_frame", "wall_preprocess", "wall_track", "wall_detect",
                                                 "wall_preintegrate", "wall_solve", "wall_marginalize")},
            "host_scope_ms_per_frame": dict(zip(
                ("ft_track", "ransac_essential", "ransac_rotation", "ft_detect", "mirror_frame", "localize", "manage_keyframe",
                 "track_landmark", "refine_window", "slide_window", "refine_subwindow"),
                [round(1e3 * (t_e.wall_scope[i] - t_w.wall_scope[i]) / args.steps, 4) for i in range(11)])),
            "ate_rmse_m": (lambda a: round(a, 5) if a == a else None)(runner.ate_rmse(poses, seq)),   # None with < 3 poses
            # dominant kernel by total time (profiles/): kb_solve_try = reduced-system Cholesky (f64 MFMA trailing
            # updates) + trust-region trial costing, one workgroup per launch; flops = algorithmic
            "roofline": {"kernel": "kb_solve_try", "bound": "mfma", "achieved": round(ba_tflops, 6),
                         "peak": F64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ba_tflops / F64_MFMA_PEAK_TFLOPS, 8),
                         "traffic": traffic.get("kb_solve_try"),
                         "algorithmic_flops_per_launch": round(bst.flops_solve_try / max(1, bst.n_timed), 1),
                         "launch_us": round(1e3 * bst.ms_solve_try / max(1, bst.n_timed), 3),
                         "launches": int(bst.n_timed), "trials": int(bst.n_trials)},
            "roofline_lk": {"kernel": "k_lk_track", "bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS,
                            "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic.get("k_lk_track"),
                            "algorithmic_bytes_per_launch": round(lk_bytes / n_launch, 1),
                            "launch_us": round(lk_ms * 1e3, 3)},
        }
        if os.environ.get("XRSLAM_HIP_LIB"):      # instrumented build variant: report its in-kernel phase timers
            import ctypes
            buf = (ctypes.c_longlong * 32)()
            ctypes.CDLL(_lib.LIB_PATH).xrhip_debug_kprof(buf, 0)
            out["kprof_ms"] = [round(v / 1e5, 3) for v in buf]    # 100 MHz ticks -> ms (whole run incl. warmup)
            out["kprof_ms"][19] = int(buf[19])                    # slot 19 counts trust-region trials
        if args.cpu_frames > 40 and world == 1:
            import subprocess
            ref_lib = os.path.join(ROOT, "oracle", "_build", "libxrslam_oracle.so")
            if not os.path.exists(ref_lib):
                subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
            nc = min(args.cpu_frames, n_frames)

            def cpu_leg(threads):
                # XR_ORACLE_THREADS is read when the session creates its KLT context (oracle/xrhip_shim.cpp)
                os.environ["XR_ORACLE_THREADS"] = str(threads)
                cpu = runner.Session(ref_lib, seq, slam_yaml=SLAM_YAML, sensor_yaml=SENSOR_YAML)
                for _ in range(40):
                    cpu.step()
                c0 = time.perf_counter()
                for _ in range(nc - 40):
                    cpu.step()
                ct = time.perf_counter() - c0
                cpu.close()
                return round((nc - 40) / ct, 3)

            sample = ("frames 40..%d of the same stream through the same host pipeline linked against the CPU oracle "
                      "(oracle/_build/libxrslam_oracle.so, gcc -O2)" % nc)
            # the reference-faithful figure: solver num_threads = 1 (estimation/solver.cpp:185), image loops on one core
            out["cpu_baseline"] = {"value": cpu_leg(1), "unit": "frames/s", "cores": 1, "kind": "port",
                                   "sample": sample + ", single thread"}
            # OpenCV spreads the image / LK point loops with parallel_for_: same sample with those loops on the host's
            # cores (capped at 16), solver and marginalisation still single-threaded like the reference
            cores = min(16, len(os.sched_getaffinity(0)))
            if cores > 1:
                out["cpu_baseline_mt"] = {"value": cpu_leg(cores), "unit": "frames/s", "cores": cores, "kind": "port",
                                          "sample": sample + ", image and LK point loops on %d OpenMP threads" % cores}
        print(json.dumps(out))
    sess.close()
    group.close()


if __name__ == "__main__":
    main()
