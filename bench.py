#!/usr/bin/env python3
"""bench.py -- frames/sec + ms/BA-iteration of the XRSLAM per-frame hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W        (one JSON line on rank 0; N > 1 without a launcher: re-executes
                                                        itself as N ranks under torch.distributed.run, one per GPU)
  python bench.py --workload {s1,s2,s3,s4}             (default s1 == BASELINE config 2; the others are extra lines)
  python bench.py --sequences-per-gpu S                (S independent sequences on each GPU, one host thread each, their per-frame
                                                        launches issued together: an instance group; --no-group: S lone instances)

A "step" is one camera frame of one sequence pushed through the whole hot path behind the reference's C API
(include/XRSLAM.h, the player's call sequence of xrslam-pc/player/src/main.cpp:116-169): ~10 gyro + ~10 accel samples,
XRSLAMPushSensorData(XRSLAM_SENSOR_CAMERA) with a HOST image (XRSLAMManager.cpp:104-136: the library copies it -- here: uploads
it -- inside the call), XRSLAMRunOneFrame, XRSLAMGetResult.  Inside: CLAHE + LK pyramid + Scharr, IMU pre-integration,
forward/backward pyramidal LK, 5-pt/2-pt RANSAC gates, Harris re-detection, localize_newframe solve, keyframe policy, landmark
triangulation, refine_window / refine_subwindow dogleg solves, marginalisation.

`value` is the reference-shaped call: host image, threading OFF (the reference's PC build, CMakeLists.txt:13 -- feature tracker
and sliding-window tracker one after the other in the caller, utility/worker.h:35-42).  The same stream is then continued
through the library's faster modes, reported beside it as `variants`: `resident` (frame already in HBM,
XRSLAMAmdPushImageDevice), `pipelined` (XRSLAMAmdSetThreading(1): the reference's XRSLAM_ENABLE_THREADING build with
deterministic hand-offs) and `pipelined_resident`.

Workloads (config.workload):
  s1  synthetic "EuRoC MH_01-like" stream: 752x480 (the real EuRoC cam0 size; BASELINE.json's 640x480 is a known
      discrepancy, SURVEY.md top table), 20 Hz camera / 200 Hz IMU, 150 features, 10-keyframe window (BASELINE config 2)
  s2  "V1_03-like" stress stream: faster motion, 300 features, 15-keyframe window (BASELINE config 3)
  s3  1280x720 stream, 600 features, 20-keyframe window (BASELINE config 5)
  s4  BA micro-bench: the frozen refine_window problems of tests/golden/ba_snapshots/ replayed through xrhip_ba_solve
      (a "step" is one whole solve; value = solves/s, ms_per_ba_iteration is the figure of interest)
The timed region is the steady state the metric is quoted on -- the window at its configured size, one marginalisation per
keyframe.  The first 36 frames of a stream only seed the window (initial states supplied from the ground truth, so that every
run measures the same path) and the window then grows by one keyframe every ~4 frames: it is full, and the first keyframe has
been marginalised (once per sequence through the eigen path: a rank-deficient prior), by frame
4 * window_keyframes + 8.  Frames before 4 * window_keyframes + 16 are therefore never timed: with --warmup W shorter than that
the missing frames run as an untimed pre-roll in front of the W warmup steps (config.untimed_preroll_frames) -- the frames it
skips are cheaper on average (smaller windows), so this never flatters `value` except for that one-off, whose cost the line
states as `one_off_ms` (the longest pre-roll frame against the pre-roll's median frame).
Beside `value` the line carries (round 5): `sustained` -- the headline mode itself continued for --sustained-frames more frames of the
same stream (a --steps 20 line is a 12 ms sample); `variants` -- each leg behind six untimed settle frames, in four timed chunks
(`chunk_ms`), with its keyframe / solve / BA-iteration counts (legs cover different stretches of the stream: comparable at equal counts);
`cpu_baseline.ms_per_ba_iteration` / `.ms_per_marginalization` -- the CPU checker's own solve and marginalisation clocks over the same
frames (the second half of BASELINE.json's metric on the same host).
One independent sequence per GPU: no data-path collective, only a barrier and a MAX
reduction of the wall time over RCCL.  --sequences-per-gpu S puts S sequences on every GPU (instance-scoped entry points,
XRSLAMAmdInstance*): `value` is then the aggregate over all sequences of all GPUs.
"""
import argparse
import gc as pygc
import glob
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

LEG_SETTLE = 6          # untimed frames in front of every variant leg: the first switch to the pipelined mode creates a thread, a BA context
                        # and its device buffers (tens of milliseconds once -- a 100-frame leg that paid it read 810-940 frames/s)
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
F64_MFMA_PEAK_TFLOPS = 78.6   # MI355X dense FP64 matrix peak (MI355X_MICROARCH.md)
CFG = os.path.join(ROOT, "configs")
WORKLOADS = {
    "s1": dict(slam="bench_slam_150.yaml", sensor="euroc_sensor.yaml", w=752, h=480, K=None, traj=None, features=150, window=10,
               text="S1 EuRoC-MH_01-like synthetic stream, 752x480 @20 Hz + 200 Hz IMU, 150 features, 10-keyframe window, "
                    "30-iteration dogleg cap (BASELINE config 2)"),
    "s2": dict(slam="stress_slam_300.yaml", sensor="euroc_sensor.yaml", w=752, h=480, K=None,
               traj=dict(amp=1.5, speed=1.0, rot=0.8), features=300, window=15,
               text="S2 EuRoC-V1_03-like synthetic stress stream (1.5 m/s, 60 deg/s peaks), 752x480 @20 Hz + 200 Hz IMU, "
                    "300 features, 15-keyframe window (BASELINE config 3)"),
    "s3": dict(slam="large_slam_600.yaml", sensor="large_sensor_1280.yaml", w=1280, h=720, K=(780.0, 778.0, 640.0, 360.0), traj=None,
               features=600, window=20,
               text="S3 synthetic 1280x720 stream @20 Hz + 200 Hz IMU, 600 features, 20-keyframe window (BASELINE config 5)"),
}


def newest_profile(pattern):
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
    if not files:
        return None
    try:
        with open(files[-1]) as fh:
            return json.load(fh)
    except (OSError, ValueError):
        return None


def roofline_extras(kernel_rev, workload="s1"):
    """Measured peaks of the MI355X the profiles were taken on (tools/peaks.hip) and the HBM-side traffic of the
    committed rocprofv3 --pmc passes -- the latter only when those passes ran the kernels this library carries ON THIS WORKLOAD
    (a counter figure is per launch of a kernel on one stream's problems: S1's 0.22 MB per kb_chain launch says nothing about S3's)."""
    peaks = newest_profile("r*_peaks.json") or {}
    pmc = newest_profile("r*_pmc_traffic.json") or {}
    traffic, note = {}, "no committed PMC pass"
    if pmc:
        rev, wl = pmc.get("_kernel_rev"), pmc.get("_workload", "s1")
        # A later revision whose kernels are the SAME MACHINE CODE bar a named few (tools/kernel_isa_diff.py compares the instruction
        # streams of the two builds; the listing is committed beside the passes): the figures of the unchanged kernels stand, the
        # rewritten ones are not shown.
        same = pmc.get("_same_machine_code_in", {}).get(kernel_rev) if rev != kernel_rev else None
        if (rev == kernel_rev or same) and wl == workload:
            skip = set(same["except"]) if same else set()
            traffic = {k: round(1024.0 * (v["fetch_kb"] + v["write_kb"]), 1) for k, v in pmc.items()
                       if isinstance(v, dict) and not k.startswith("_") and k not in skip}
            note = "profiles/ PMC passes of kernel revision %s, workload %s" % (rev, wl)
            if same:
                note += "; this library is %s: identical machine code except %s (%s), which are not shown" % (
                    kernel_rev, ", ".join(sorted(skip)), same["evidence"].split(" ")[0])
        elif rev != kernel_rev:
            note = "committed PMC passes are of kernel revision %s, this library is %s: not shown" % (rev, kernel_rev)
        else:
            note = "committed PMC passes are of workload %s, this line is %s: not shown" % (wl, workload)
    return peaks, traffic, note


def kernel_table(kernel_rev, workload="s1"):
    """The committed rocprofv3 kernel-trace summary of THIS kernel revision on this workload (profiles/r*_kernel_stats.json, written by
    tools/make_profile_summary.py from the trace of `bench.py --steps 100 --warmup 40`): kernel time per frame and per kernel.  None
    when the newest committed trace is of other kernels."""
    ks = newest_profile("r*_kernel_stats.json")
    if not ks or ks.get("_kernel_rev") != kernel_rev or ks.get("_workload", "s1") != workload:
        return None
    return ks


def device_power_state(index):
    """Shader / memory clock and power cap of the device as the run found them (rocm-smi; amdsmi is not in the image): boxes of the
    pool differ by up to 30 % on the single-workgroup f64 kernels -- with this in the line a slow box is
    distinguishable from a regression.  Best effort: {} when the tool is missing or prints something else."""
    import re
    import subprocess
    out = {}
    if os.environ.get("XRSLAM_BENCH_NO_SMI"):   # development switch: is the tool's query what stalls a later leg?
        return out
    try:
        txt = subprocess.run(["rocm-smi", "-d", str(index), "--showclocks", "--showmaxpower", "--showpower", "--showperflevel"],
                             capture_output=True, text=True, timeout=20).stdout
    except (OSError, subprocess.SubprocessError):
        return out
    for key, pat in (("sclk_mhz", r"sclk clock level:[^(]*\((\d+)Mhz\)"), ("mclk_mhz", r"mclk clock level:[^(]*\((\d+)Mhz\)"),
                     ("fclk_mhz", r"fclk clock level:[^(]*\((\d+)Mhz\)"), ("power_cap_w", r"Max Graphics Package Power \(W\):\s*([\d.]+)"),
                     ("power_w", r"Graphics Package Power \(W\):\s*([\d.]+)")):
        m = re.search(pat, txt)
        if m:
            out[key] = float(m.group(1))
    m = re.search(r"Performance Level:\s*(\w+)", txt)
    if m:
        out["perf_level"] = m.group(1)
    return out


def precision_study():
    """BASELINE config 5's "fp32 vs bf16 BA solve" on the device: the Schur contraction of the frozen S3 window problem
    (tests/golden/ba_snapshots/s3_window: 21 frames, 1268 landmarks) with f64 / f32 / bf16 matrix-core operands
    (csrc/study_api.hip), the error each implies for the Gauss-Newton step of the reduced system, and the matrix-core rates."""
    from xrslam_amd import ba
    from tests import ba_snapshots
    snap = [x for x in ba_snapshots.load_all() if x[0] == "s3_window"]
    if not snap:
        return None
    _, pd, _ = snap[0]
    ctx = ba.BaContext(max_frames=32, max_landmarks=2048, max_obs=16384)
    lin = ctx.debug_linearize(pd)
    ctx.close()
    F = len(pd.frame_state)
    H, g, hll, gl, W = lin["H"], lin["g"], lin["hll"], lin["gl"], lin["W"]
    free = hll > 0
    W, w, gl = W[free], 1.0 / hll[free], gl[free]
    T, ms = ba.study_schur_precision(W, w, reps=50)
    pose = np.array([15 * f + k for f in range(F) for k in range(6)])
    act = np.where(np.diag(H) > 0)[0]                 # fixed frames contribute empty rows
    rhs = g.copy()
    rhs[pose] -= W.T @ (w * gl)

    def step(Tm):
        S = H.copy()
        S[np.ix_(pose, pose)] -= Tm
        S = S[np.ix_(act, act)]
        d = 1.0 / np.sqrt(np.diag(S).clip(1e-300))   # Jacobi scaling, like the solver
        Ss = S * np.outer(d, d) + 1e-9 * np.eye(len(act))
        return -d * np.linalg.solve(Ss, d * rhs[act])

    d64 = step(T["f64"])
    Ln, PF = len(w), (W.shape[1] + 15) // 16 * 16
    flops = 2.0 * ((Ln + 63) // 64 * 64) * PF * PF
    peak = {"f64": F64_MFMA_PEAK_TFLOPS, "f32": 157.3, "bf16": 2500.0}   # dense matrix-core peaks (MI355X_MICROARCH.md)
    out = {"problem": "s3_window: %d frames, %d free landmarks, contraction [%d x %d]^T diag [%d x %d]" % (F, Ln, Ln, W.shape[1], Ln, W.shape[1]),
           "tolerance": "north_star: 1e-4 relative on the states"}
    for m in ("f64", "f32", "bf16"):
        dm = step(T[m])
        out[m] = {"kernel_us": round(1e3 * ms[m], 3), "tflops": round(flops / (ms[m] * 1e-3) / 1e12, 4),
                  "mfma_frac_of_peak": round(flops / (ms[m] * 1e-3) / 1e12 / peak[m], 6),
                  "product_rel_error": float("%.3e" % (np.abs(T[m] - T["f64"]).max() / np.abs(T["f64"]).max())),
                  "gauss_newton_step_rel_error": float("%.3e" % (np.linalg.norm(dm - d64) / np.linalg.norm(d64)))}
    out["full_solve"] = precision_study_full_solve(pd)
    return out


def precision_study_full_solve(pd):
    """Config 5 as written -- "fp32 vs bf16 BA *solve*": the WHOLE frozen S3 window solve (21 keyframes, 1268 landmarks, 10 466
    observations, 30-iteration cap) with the Schur contraction of every linearisation in f64 (the product), f32 and bf16 matrix-core
    operands (xrhip_ba_debug_set_schur_precision): final states against the f64 solve, iterations / accepted steps / termination, cost."""
    from xrslam_amd import ba
    ctx = ba.BaContext(max_frames=32, max_landmarks=2048, max_obs=16384)
    res, ref = {}, None
    try:
        for mode, name in ((0, "f64"), (1, "f32"), (2, "bf16")):
            ctx.set_schur_precision(mode)
            b = pd.copy()
            sm = ctx.solve(b)
            ms = []
            for _ in range(5):
                bb = pd.copy()
                ms.append(ctx.solve(bb).ms_solve)
            pos, dep = b.frame_state[:, 4:7].copy(), b.inv_depth.copy()
            q = b.frame_state[:, 0:4].copy()
            if ref is None:
                ref = (pos, dep, q, b.frame_state.copy())
            row = {"iterations": int(sm.iterations), "accepted_steps": int(sm.successful_steps), "termination": int(sm.termination),
                   "final_cost": float("%.10g" % sm.final_cost), "ms_per_solve": round(float(np.median(ms)), 4)}
            if mode:
                row["position_rel_error"] = float("%.3e" % (np.linalg.norm(pos - ref[0]) / np.linalg.norm(ref[0])))
                row["position_max_abs_error_m"] = float("%.3e" % np.abs(pos - ref[0]).max())
                row["rotation_max_error_rad"] = float("%.3e" % (2.0 * np.arcsin(np.clip(np.linalg.norm(
                    q[:, :3] * ref[2][:, 3:4] - ref[2][:, :3] * q[:, 3:4] + np.cross(q[:, :3], ref[2][:, :3]), axis=1), 0, 1)).max()))
                row["inv_depth_rel_error"] = float("%.3e" % (np.linalg.norm(dep - ref[1]) / np.linalg.norm(ref[1])))
                row["state_rel_error"] = float("%.3e" % (np.linalg.norm(b.frame_state - ref[3]) / np.linalg.norm(ref[3])))
            res[name] = row
    finally:
        ctx.set_schur_precision(0)
        ctx.close()
    res["tolerance"] = "north_star: 1e-4 relative on the states"
    return res


def bench_s4(args, out_common):
    """S4: frozen refine_window problems replayed through the C ABI of plug point #2."""
    from xrslam_amd import ba
    from tests import ba_snapshots
    snaps = ba_snapshots.load_all()
    if not snaps:
        raise SystemExit("no BA snapshots under tests/golden/ba_snapshots/")
    ctx = ba.BaContext(max_frames=32, max_landmarks=2048, max_obs=16384)
    reps = max(1, args.steps)
    per = []
    for name, pd, exp in snaps:
        for _ in range(max(1, args.warmup // 10)):
            ctx.solve(pd.copy())
        it, ms = 0, 0.0
        for _ in range(reps):
            sm = ctx.solve(pd.copy())     # a solve works in place: every repetition starts from the frozen states
            it += sm.iterations
            ms += sm.ms_solve             # wall clock of the whole xrhip_ba_solve call (staging, kernels, read-back), taken
        dt = 1e-3 * ms                    # inside the library: the interpreter's copy / marshalling is not in it
        per.append(dict(snapshot=name, frames=len(pd.frame_state), landmarks=len(pd.inv_depth), observations=len(pd.obs_tgt),
                        iterations_per_solve=it / reps, ms_per_solve=round(1e3 * dt / reps, 4),
                        ms_per_ba_iteration=round(1e3 * dt / max(1, it), 5)))
    tot_ms = sum(p["ms_per_solve"] for p in per)
    out = dict(out_common)
    out.update({"metric": "BA solves/sec on frozen refine_window snapshots (S4 micro-bench)", "value": round(1e3 * len(per) / tot_ms, 3),
                "unit": "solves/s", "steps": reps, "warmup": args.warmup, "ms_per_step": round(tot_ms / len(per), 4),
                "dtype": "f64 BA", "data": "frozen problems dumped from the synthetic S1/S2/S3 streams (tests/golden/ba_snapshots)",
                "config": {"workload": "S4 frozen refine_window snapshots replayed through xrhip_ba_solve", "snapshots": len(per)},
                "ms_per_ba_iteration": round(sum(p["ms_per_ba_iteration"] * p["iterations_per_solve"] for p in per) /
                                             max(1e-9, sum(p["iterations_per_solve"] for p in per)), 5),
                "snapshots": per})
    print(json.dumps(out))


def relaunch_as_ranks(args):
    """--gpus N > 1 without a launcher's environment: this process becomes the launcher -- the command line the driver itself
    uses (torch.distributed.run, one rank per GPU, rendezvous on 127.0.0.1) around the same bench.py arguments."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: the only mode the host driver supports (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "1")
    raise SystemExit(subprocess.call(cmd, env=env))


def rendezvous_only(args):
    """The multi-process plumbing alone (no device, no measurement): process group, barrier, the SUM / MAX reduction of the
    metrics vector -- what tests/test_dist_cpu.py drives with --backend gloo on a box without a GPU."""
    from xrslam_amd.harness.dist import RunGroup
    group = RunGroup(backend=args.backend or "gloo")
    group.barrier()
    red = group.reduce_metrics(args.steps, 1.0 + 0.25 * group.rank)
    rows = group.gather_rows([group.rank, group.local_rank, args.steps / (1.0 + 0.25 * group.rank)])
    if group.rank == 0:
        print(json.dumps({"rendezvous_only": True, "n_gpus": group.world, "frames": red["frames"], "seconds_max": red["seconds"],
                          "per_rank": [{"rank": int(r[0]), "device": int(r[1]), "frames_per_s": round(r[2], 3)} for r in rows],
                          "note": "no measurement: the launcher / process-group path of bench.py --gpus N exercised without a device"}))
    group.barrier()
    group.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=150)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--workload", default="s1", choices=["s1", "s2", "s3", "s4"])
    ap.add_argument("--euroc", default=None, metavar="MAV0_DIR",
                    help="a real ASL / EuRoC sequence (<...>/MH_01_easy/mav0) instead of the synthetic stream: frames as recorded, "
                         "rectified on the GPU, self-initialising like the reference; reports ATE against its ground truth")
    ap.add_argument("--sequences-per-gpu", type=int, default=1,
                    help="independent sequences per GPU, one host thread each (instance-scoped entry points)")
    ap.add_argument("--no-group", action="store_true",
                    help="with --sequences-per-gpu S: S lone instances (every sequence launches for itself, round 3's form) instead of "
                         "one instance group whose members share launches (XRSLAMAmdGroup)")
    ap.add_argument("--cpu-frames", type=int, default=240,
                    help="most frames of the bounded CPU-reference sample (0 = skip); the first 40 are its warm-up")
    ap.add_argument("--cpu-seconds", type=float, default=20.0,
                    help="time budget of each CPU-reference leg: the sample stops at --cpu-frames or here, whichever comes first")
    ap.add_argument("--variant-frames", type=int, default=100,
                    help="frames of each extra leg on the same stream after the timed region -- resident / pipelined / "
                         "pipelined_resident (0 = skip; one sequence on one GPU only)")
    ap.add_argument("--sustained-frames", type=int, default=150,
                    help="frames of the `sustained` leg: the HEADLINE mode itself continued on the same stream right after the timed "
                         "region (a --steps 20 line is a 12 ms sample; this is the same call over a longer one; 0 = skip; one "
                         "sequence on one GPU only)")
    ap.add_argument("--image", default="host", choices=["host", "resident"],
                    help="host (default): the reference-shaped camera call, a host image uploaded inside XRSLAMPushSensorData; "
                         "resident: frames already in HBM (XRSLAMAmdPushImageDevice)")
    ap.add_argument("--python-loop", action="store_true",
                    help="drive the six reference symbols from the interpreter, one foreign call per sensor sample, instead of the "
                         "native replay loop (XRSLAMAmdInstanceReplay: the player's main loop, same call sequence)")
    ap.add_argument("--threading", default=None, choices=["inline", "pipelined"],
                    help="inline (default): feature tracker and sliding-window tracker one after the other in the caller -- the "
                         "reference's PC build; pipelined: its XRSLAM_ENABLE_THREADING build with deterministic hand-offs -- the "
                         "backend of frame t on a library thread beside the feature tracker of frame t+1 (XRSLAMAmdSetThreading)")
    ap.add_argument("--step-times", action="store_true",
                    help="development aid (with --python-loop): wall time of every timed step -> median and the five longest")
    ap.add_argument("--no-profile", action="store_true", help="do not record HIP events around the KLT kernels")
    ap.add_argument("--backend", default=None, help="torch.distributed backend (default: nccl == RCCL); 'gloo' lets two "
                    "ranks share one GPU when the multi-process path is exercised on a single-GPU box")
    ap.add_argument("--dump-poses", default=None, metavar="DIR",
                    help="every rank writes the poses of each of its sequences (all frames up to the end of the timed region) to "
                         "DIR/poses_rank<R>_seq<I>.npy -- tests compare them with solo runs of the same streams")
    ap.add_argument("--rendezvous-only", action="store_true",
                    help="exercise only the launcher / process-group path (no device needed, prints no measurement)")
    args = ap.parse_args()
    if args.steps < 1 or args.warmup < 0 or args.sequences_per_gpu < 1 or args.gpus < 1:
        raise SystemExit("--gpus and --steps must be >= 1, --warmup >= 0, --sequences-per-gpu >= 1")
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        relaunch_as_ranks(args)          # does not return
    if env_world is not None and args.gpus > 1 and int(env_world) != args.gpus:
        raise SystemExit("--gpus %d contradicts the launcher's WORLD_SIZE=%s" % (args.gpus, env_world))
    if args.rendezvous_only:
        rendezvous_only(args)
        return
    S = args.sequences_per_gpu
    if S > 1:
        # Every instance owns several HIP streams; the runtime multiplexes them over this many hardware queues.  MORE is not better:
        # measured at 8 sequences (profiles/r03_multi_sequence.md) 8 queues 3590 frames/s, 16: 3528, 24: 2557, 32: 2040, 48: 1438 --
        # beyond a handful the scheduler time-slices the queues and every kernel stretches (k_pyrdown 4.7 -> 22 us at 24 queues).
        # With the instance group (round 4) the launches that matter are batched on the group's own streams and the right number is
        # the smallest: 2 queues per priority level, the group's streams at high priority -- two hardware queues for the batches of
        # all members, two for the members' own window solves and marginalisations (xrslam_amd/csrc/group_api.hip;
        # profiles/r04_multi_sequence.md: 10 sequences 5690 frames/s, against 4960 with 4 unprioritised queues).
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8" if args.no_group else "2")
    rank, world = int(os.environ.get("RANK", "0")), int(env_world or "1")

    from xrslam_amd.harness import scene
    from xrslam_amd.harness.trajectory import Trajectory
    wl = dict(WORKLOADS[args.workload]) if args.workload != "s4" else None
    seqs, real, preroll, n_frames, variant_frames, sustained_frames, slam_yaml, sensor_yaml = [], None, 0, 0, 0, 0, None, None
    if wl is not None:
        # steady state (window full, first marginalisation done) from frame 4 * window + 16 on: a shorter warmup is preceded by
        # the missing frames as an untimed pre-roll (module docstring)
        preroll = max(0, 4 * wl["window"] + 16 - args.warmup)
        slam_yaml, sensor_yaml = os.path.join(CFG, wl["slam"]), os.path.join(CFG, wl["sensor"])
        variant_frames = args.variant_frames if (S == 1 and world == 1) else 0
        sustained_frames = args.sustained_frames if (S == 1 and world == 1) else 0
        n_frames = preroll + args.warmup + args.steps + sustained_frames + 3 * (variant_frames + (LEG_SETTLE if variant_frames else 0))
        if args.euroc:
            from xrslam_amd.harness import euroc
            if S != 1 or world != 1:
                raise SystemExit("--euroc runs one sequence on one GPU")
            real = euroc.read_euroc(args.euroc, max_frames=n_frames + 1)
            wl["text"] = ("real EuRoC sequence %s, %dx%d frames as recorded (rectified on the GPU), %d features, %d-keyframe window, "
                          "self-initialising" % (os.path.basename(os.path.dirname(os.path.normpath(args.euroc))),
                                                 real["frames"].shape[2], real["frames"].shape[1], wl["features"], wl["window"]))
            seqs = [real]
        else:
            # rendered before anything in this process touches the GPU runtime (the renderer forks its workers)
            seq_kw = dict(w=wl["w"], h=wl["h"], workers=max(1, min(16, len(os.sched_getaffinity(0)) // max(1, min(world, 8)))))
            if wl["K"]:
                seq_kw["K"] = wl["K"]
            for i in range(S):
                kw = dict(seq_kw)
                if wl["traj"]:
                    kw["traj"] = Trajectory(**wl["traj"])
                seqs.append(scene.make_sequence(n_frames=n_frames + 1, seed=1 + rank * S + i, **kw))

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (the product has no CPU fallback)")
    from xrslam_amd import _lib
    from xrslam_amd.harness import runner
    from xrslam_amd.harness.dist import RunGroup
    local_rank, n_dev = int(os.environ.get("LOCAL_RANK", "0")), torch.cuda.device_count()
    # one rank per GPU.  More ranks than devices used to wrap around silently (LOCAL_RANK % device_count): two
    # ranks then shared a GPU and the line still said n_gpus: N.  RCCL refuses that anyway ("Duplicate GPU detected"); only the
    # gloo backend -- the explicit single-GPU exercise of the multi-process path -- may double up, and the line says so (per_rank).
    if local_rank >= n_dev and (args.backend or "nccl") != "gloo":
        raise SystemExit("rank %d of this node has no GPU of its own: %d device(s) visible (--gpus must not exceed them; "
                         "--backend gloo shares devices on purpose)" % (local_rank, n_dev))
    device_index = local_rank % n_dev
    torch.cuda.set_device(device_index)
    group = RunGroup(backend=args.backend)  # one process per GPU; "nccl" == RCCL over xGMI
    rank, world = group.rank, group.world
    _lib.set_device(device_index)
    kernel_rev = _lib.kernel_revision()
    peaks, traffic, traffic_note = roofline_extras(kernel_rev, args.workload)
    power0 = device_power_state(device_index)
    common = {"n_gpus": world, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "kernel_rev": kernel_rev}

    if args.workload == "s4":
        if rank == 0:
            bench_s4(args, common)
        group.barrier()
        group.close()
        return

    if args.threading is None:
        args.threading = "inline"
    pipelined = args.threading == "pipelined"
    resident = args.image == "resident"
    native = not args.python_loop or S > 1
    sessions, keep = [], []
    igroup = runner.Group(_lib.LIB_PATH) if (S > 1 and not args.no_group) else None
    for seq in seqs:
        dev = torch.from_numpy(seq["frames"]).cuda()     # the `resident` legs read their frames from here
        keep.append(dev)
        h, w = seq["frames"].shape[1:]
        sessions.append(runner.Session(_lib.LIB_PATH, seq, slam_yaml=slam_yaml, sensor_yaml=sensor_yaml,
                                       device_frames=(dev.data_ptr(), h * w, w) if resident else None, instance=native,
                                       init_frames=0 if real is not None else 60, threading=1 if pipelined else 0,
                                       device_undistort="cv_undistort" if real is not None else None, group=igroup))
    torch.cuda.synchronize()
    sess, seq = sessions[0], sessions[0].seq
    dev_frames0 = (keep[0].data_ptr(), seq["frames"].shape[1] * seq["frames"].shape[2], seq["frames"].shape[2])

    def barrier():
        group.barrier()
        torch.cuda.synchronize()

    step_ms = []

    def run_all(n, per_step=None):
        """n frames on every session: inline for one, one host thread per sequence otherwise (the foreign calls release
        the interpreter lock, so the sequences' host work and device waits overlap)."""
        if S == 1:
            if native and per_step is None:
                sess.step_n(n)
            else:
                for _ in range(n):
                    t1 = time.perf_counter()
                    if native:
                        sess.step_n(1)
                    else:
                        sess.step()
                    (step_ms if per_step is None else per_step).append(1e3 * (time.perf_counter() - t1))
            sess.sync()   # pipelined mode: the backend job of the last frame is part of the n frames
            return
        errs = []

        def work(s):
            try:
                s.step_n(n)   # XRSLAMAmdInstanceReplay: the player's loop issued natively, one foreign call per thread
                s.sync()
            except Exception as e:   # noqa: BLE001
                errs.append(repr(e))
        th = [threading.Thread(target=work, args=(s,)) for s in sessions]
        for t in th:
            t.start()
        for t in th:
            t.join()
        if errs:
            raise SystemExit("sequence thread failed: " + errs[0])

    # The interpreter's cyclic collector out of the measurement: with torch imported a full collection walks ~10^6 objects -- tens of
    # milliseconds, once per process, whenever the allocation counters happen to trip (rounds 4-5: "a ~40 ms hiccup in a different leg
    # every run", absent from the torch-free replays of tools/find_stall.py).  Everything that exists now is set-up: collect once, move it
    # to the permanent generation (later collections only see what the measurement itself allocates) and leave the collector on.
    pygc.collect()
    pygc.freeze()
    # untimed frames (pre-roll + warmup), frame by frame when there is one sequence so that the one-off they hide can be stated
    pre_ms = []
    if preroll + args.warmup > 0:
        run_all(preroll + args.warmup, per_step=pre_ms if S == 1 else None)
    for s in sessions:
        if s.error():
            raise SystemExit("warmup failed: " + s.error())
    t_w = sess.times()
    for s in sessions:
        s.klt_stats(reset=True)
        s.ba_stats(reset=True)
        if not args.no_profile:
            s.set_profiling(True)
    if igroup is not None:
        igroup.stats(reset=True)
        igroup.set_profiling(not args.no_profile)
    barrier()
    del step_ms[:]
    t0 = time.perf_counter()
    run_all(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    for s in sessions:
        if s.error():
            raise SystemExit("timed region failed: " + s.error())
    t_e = sess.times()
    st = sess.klt_stats(reset=False)
    bst = sess.ba_stats(reset=False)
    all_klt = [s.klt_stats(reset=False) for s in sessions] if S > 1 else [st]
    gstats = None
    if igroup is not None:
        igroup.set_profiling(False)
        gstats = igroup.stats()
    for s in sessions:
        s.set_profiling(False)
    my_elapsed = elapsed
    if args.dump_poses:
        os.makedirs(args.dump_poses, exist_ok=True)
        for i, s in enumerate(sessions):
            np.save(os.path.join(args.dump_poses, "poses_rank%d_seq%d.npy" % (rank, i)), np.array(s.poses))
    red = group.reduce_metrics(args.steps * S, elapsed)      # frames: SUM, wall seconds: MAX over ranks
    elapsed, total_frames = red["seconds"], red["frames"]
    # per-rank diagnostics into rank 0's line: a slow rank (clocked-down device, a device shared by two ranks) is visible there
    power1 = device_power_state(device_index)
    rows = group.gather_rows([rank, device_index, args.steps * S / my_elapsed, my_elapsed, power1.get("sclk_mhz", -1.0),
                              power1.get("mclk_mhz", -1.0), power1.get("power_cap_w", -1.0), power1.get("power_w", -1.0),
                              float(int(kernel_rev, 16)) if all(ch in "0123456789abcdef" for ch in kernel_rev) else -1.0])
    # every rank must have run the same kernels (a stale in-tree .so on one rank would silently mix two revisions into one figure):
    # the revisions travel as numbers (ten hexadecimal digits are exact in a double) and a mismatch fails the run on every rank
    revs = sorted({r[8] for r in rows})
    if len(revs) != 1:
        raise SystemExit("bench.py: the ranks ran different kernel revisions (%s): rebuild xrslam_amd/lib on every rank" %
                         ", ".join("%010x" % int(v) if v >= 0 else "?" for v in revs))

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
        # kb_chain / k_lk_track of a GROUP: the members do not time their own launches, the group times the batch they travel in
        # (first to last kernel: kb_stage + kb_chain for all entries; LK + the Harris passes that ride along).  bytes = all members'
        # algorithmic bytes (every member accounts its own), time = the batches': the rate of the batched launch.
        chain_bytes_, chain_ms_, chain_n_, chain_note = bst.bytes_chain, bst.ms_chain, bst.n_chain_timed, None
        lk_note = None
        if gstats is not None and not args.no_profile:
            allb = [s_.ba_stats(reset=False) for s_ in sessions]
            gc, gt = gstats.get("chain", {}), gstats.get("track", {})
            if gc.get("timed"):
                chain_bytes_, chain_ms_, chain_n_ = sum(b.bytes_chain for b in allb), gc["ms"], gc["timed"]
                chain_note = ("group: %d batches served %d solves of all members; launch_us = kb_stage + kb_chain of a batch, "
                              "bytes = the batch's entries'" % (gc["timed"], gc["requests"]))
            if gt.get("timed"):
                lk_all = sum(2420.0 * k.lk_templates + 484.0 * k.lk_iterations for k in all_klt)
                lk_bytes, n_launch, lk_ms = lk_all, gt["timed"], gt["ms"] / gt["timed"]
                achieved = (lk_bytes / n_launch) / (lk_ms * 1e-3) / 1e9
                lk_note = ("group: %d batches served %d tracking requests; launch_us = k_lk_track AND the Harris passes of the batch "
                           "(one stream): a lower bound of the LK kernel's rate" % (gt["timed"], gt["requests"]))
        hbm_meas, mfma_meas = peaks.get("stream_read_gbs"), peaks.get("mfma_f64_16x16x4_tflops")
        out = dict(common)
        out.update({
            "metric": "frames/sec, full per-frame hot path (KLT tracker + sliding-window VI-BA), %s per GPU"
                      % ("one sequence" if S == 1 else "%d sequences" % S),
            "value": round(total_frames / elapsed, 3),
            "unit": "frames/s",
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4),
            "dtype": "u8/i16 images, f32 LK with exact i64 reductions, f64 BA",
            "data": "synthetic" if real is None else "real (EuRoC)",
            "config": {"workload": wl["text"], "features": wl["features"], "window_keyframes": wl["window"],
                       "sequences_per_gpu": S, "untimed_preroll_frames": preroll,
                       "image": ("resident: frames already in HBM (XRSLAMAmdPushImageDevice)" if resident else
                                 "host: XRSLAMPushSensorData(XRSLAM_SENSOR_CAMERA) with a host image, uploaded inside the timed call "
                                 "(the reference's call, XRSLAMManager.cpp:104-136)"),
                       "driver": ("native replay loop (XRSLAMAmdInstanceReplay: the reference player's main loop, "
                                  "xrslam-pc/player/src/main.cpp:116-169, over the instance-scoped entry points)") if native else
                                 "the six reference symbols called from the interpreter, one foreign call per sensor sample",
                       "threading": ("pipelined: sliding-window tracker of frame t on a library thread beside the feature tracker of "
                                     "frame t+1 (the reference's XRSLAM_ENABLE_THREADING build, deterministic hand-offs; timed region "
                                     "starts and ends with the pipeline drained)") if pipelined else
                                    "inline: feature tracker then sliding-window tracker in the caller (the reference's PC build)"},
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
            # pipelined mode: what the feature tracker's thread spent waiting for the previous frame's backend at the hand-off
            "backend_wait_ms_per_frame": round(1e3 * (t_e.wall_scope[15] - t_w.wall_scope[15]) / args.steps, 4),
            "ate_rmse_m": (lambda a: round(a, 5) if a == a else None)(runner.ate_rmse(poses, seq)),   # None with < 3 poses
            # dominant kernel by total time (profiles/r0x_full_*_kernel_stats.md): kb_chain, the LDS-resident single-launch
            # solve of localize_newframe / refine_subwindow (one workgroup; factor linearisation + trial costing stream the
            # observation records: HBM-bound per SURVEY.md 8d).  achieved = algorithmic bytes per launch / HIP-event
            # duration of that kernel on the BA stream.
            "roofline": {"kernel": "kb_chain", "bound": "hbm",
                         "achieved": round(chain_bytes_ / max(1e-9, chain_ms_ * 1e-3) / 1e9, 4) if chain_ms_ > 0 else 0.0,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(chain_bytes_ / max(1e-9, chain_ms_ * 1e-3) / 1e9 / HBM_PEAK_GBS, 8) if chain_ms_ > 0 else 0.0,
                         "traffic": traffic.get("kb_chain") if chain_note is None else None,
                         "peak_measured": hbm_meas,
                         "frac_of_measured": round(chain_bytes_ / max(1e-9, chain_ms_ * 1e-3) / 1e9 / hbm_meas, 8) if hbm_meas and chain_ms_ > 0 else None,
                         "algorithmic_bytes_per_launch": round(chain_bytes_ / max(1, chain_n_), 1),
                         "launch_us": round(1e3 * chain_ms_ / max(1, chain_n_), 3), "launches": int(chain_n_),
                         **({"batched": chain_note} if chain_note else {})},
            # the window solve's factorisation kernel: reduced-system Cholesky on the f64 matrix cores (+ substitutions); flops =
            # algorithmic; launch_us = HIP events around that kernel alone
            "roofline_solve": {"kernel": "kb_solve_try", "bound": "mfma", "achieved": round(ba_tflops, 6),
                               "peak": F64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ba_tflops / F64_MFMA_PEAK_TFLOPS, 8),
                               "traffic": traffic.get("kb_solve_try"),
                               "peak_measured": mfma_meas, "frac_of_measured": round(ba_tflops / mfma_meas, 8) if mfma_meas else None,
                               "algorithmic_flops_per_launch": round(bst.flops_solve_try / max(1, bst.n_timed), 1),
                               "launch_us": round(1e3 * bst.ms_solve_try / max(1, bst.n_timed), 3),
                               "launches": int(bst.n_timed), "trials": int(bst.n_trials)},
            "roofline_lk": {"kernel": "k_lk_track", "bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS,
                            "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic.get("k_lk_track"),
                            "peak_measured": hbm_meas, "frac_of_measured": round(achieved / hbm_meas, 6) if hbm_meas else None,
                            "algorithmic_bytes_per_launch": round(lk_bytes / n_launch, 1),
                            "launch_us": round(lk_ms * 1e3, 3), **({"batched": lk_note} if lk_note else {})},
            "traffic_source": traffic_note,
            # how much of a frame the device is busy: kernel time per frame of the committed kernel trace of this revision (rocprofv3;
            # kernels on the auxiliary streams overlap the main chain, so this is an upper bound of the busy fraction) over this run's
            # ms_per_step -- the frame IS its kernels' latency -- and where that time goes, kernel by kernel
            **((lambda ks: {"device_busy_frac": round(ks["kernel_ms_per_frame"] / (1e3 * elapsed / args.steps), 4) if S == 1 else None,
                            "kernel_ms_per_frame": ks["kernel_ms_per_frame"],
                            "kernel_us_per_frame": {k: v["us_per_frame"] for k, v in list(ks["kernels"].items())[:16]},
                            "kernel_table_source": "profiles/%s (rocprofv3 --kernel-trace of this kernel revision, %d frames)" % (
                                ks["_source"], ks["_frames"])})(kernel_table(kernel_rev, args.workload))
               if kernel_table(kernel_rev, args.workload) else {"device_busy_frac": None,
                                                                "kernel_table_source": "no committed kernel trace of this kernel revision"}),
            # the whole GPU against the HBM roofline: ALGORITHMIC bytes of the tracker stage (B_trk = CLAHE + pyramid +
            # Scharr + Harris passes of every frame + the LK templates and iterations the kernels counted) of ALL sequences of this
            # rank over the timed wall clock.  (The BA's ~0.1-0.6 MB per iteration are not in it: a lower bound.)
            "hbm_whole_gpu": (lambda px, pyr: (lambda b: {"algorithmic_bytes": round(b, 1), "achieved": round(b / my_elapsed / 1e9, 4),
                                                          "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(b / my_elapsed / 1e9 / HBM_PEAK_GBS, 8),
                                                          "scope": "tracker stage of %d sequence(s) on this GPU; timed wall clock" % S})(
                sum(args.steps * (3.0 * px + (pyr - px / 64.0) + (pyr - px) + 5.0 * pyr + 5.0 * px) + 2420.0 * k.lk_templates + 484.0 * k.lk_iterations
                    for k in all_klt)))(float(wl["w"] * wl["h"]), float(sum(((wl["w"] + (1 << l) - 1) >> l) * ((wl["h"] + (1 << l) - 1) >> l) for l in range(4)))),
            # what every rank measured on its own clock, and the clock / power state of its device right after the timed region
            "per_rank": [{"rank": int(r[0]), "device": int(r[1]), "frames_per_s": round(r[2], 3), "seconds": round(r[3], 4),
                          "sclk_mhz": r[4] if r[4] >= 0 else None, "mclk_mhz": r[5] if r[5] >= 0 else None,
                          "power_cap_w": r[6] if r[6] >= 0 else None, "power_w": r[7] if r[7] >= 0 else None} for r in rows],
            "device_state": {"before": power0, "after_timed_region": power1},
        })
        if gstats is not None:
            # the instance group: requests per batch = sequences one launch served; ms = HIP events around a batch's kernels
            out["group"] = {k: dict(v, requests_per_batch=round(v["requests"] / max(1, v["batches"]), 2),
                                    us_per_batch=round(1e3 * v["ms"] / max(1, v["timed"]), 2)) for k, v in gstats.items()}
        elif S > 1:
            out["group"] = "off (--no-group): every sequence launches for itself"
        if pre_ms:
            # what the untimed pre-roll hides: the first marginalisation of a sequence goes through the eigen path (km_jacobi,
            # DESIGN.md section 6) -- once per sequence; stated as the longest pre-roll frame against the pre-roll's median frame
            worst = max(range(len(pre_ms)), key=lambda i: pre_ms[i])
            tracked = [v for v in pre_ms[36:]] or pre_ms      # the first 36 frames only seed the window (no solve)
            med = sorted(tracked)[len(tracked) // 2]
            out["one_off_ms"] = {"frame": worst, "ms": round(pre_ms[worst], 3), "untimed_median_ms": round(med, 3),
                                 "extra_ms": round(pre_ms[worst] - med, 3),
                                 "note": "longest of the untimed frames (pre-roll + warmup): the first marginalisation's eigen path, once "
                                         "per sequence; median over the untimed frames after the 36 that only seed the window"}
        if args.step_times and step_ms:
            ts = sorted(step_ms[:args.steps])
            out["step_ms"] = {"median": round(ts[len(ts) // 2], 4), "longest": [round(v, 3) for v in ts[-5:]],
                              "at": [int(i) for i in sorted(range(len(step_ms[:args.steps])), key=lambda i: step_ms[i])[-5:]]}
        if args.workload == "s3":
            try:
                out["precision_study"] = precision_study()
            except Exception as e:   # noqa: BLE001  (a study, not the measurement)
                out["precision_study"] = {"error": repr(e)}
        if os.environ.get("XRSLAM_HIP_LIB"):      # instrumented build variant: report its in-kernel phase timers
            import ctypes
            buf = (ctypes.c_longlong * 32)()
            ctypes.CDLL(_lib.LIB_PATH).xrhip_debug_kprof(buf, 0)
            out["kprof_ms"] = [round(v / 1e5, 3) for v in buf]    # 100 MHz ticks -> ms (whole run incl. warmup)
            for slot in (19, 27, 31):                             # counters: trust-region trials, single-launch rounds, solves
                out["kprof_ms"][slot] = int(buf[slot])
            lkb = (ctypes.c_longlong * 8)()
            try:
                ctypes.CDLL(_lib.LIB_PATH).xrhip_debug_lkprof(lkb, 0)
                if lkb[6] > 0:    # shader cycles per point: level set-up | A exchange + tile | taps | scans | exchange | update | total
                    out["lk_prof_cycles_per_point"] = {k: round(lkb[i] / lkb[6], 1) for i, k in enumerate(
                        ("level_setup_templates", "a_exchange_tile_to_lds", "iter_weights_taps", "iter_dpp_scans", "iter_exchange",
                         "iter_update_convergence"))}
                    out["lk_prof_cycles_per_point"]["kernel"] = round(lkb[7] / lkb[6], 1)
                    out["lk_prof_cycles_per_point"]["points"] = int(lkb[6])
            except AttributeError:
                pass
        if sustained_frames > 0:
            # the headline mode itself over a longer sample of the same stream (same call, same threading, same image path)
            sess.sync()
            torch.cuda.synchronize()
            k0, m0 = sess.times(), time.perf_counter()
            if native:
                sess.step_n(sustained_frames)
            else:
                for _ in range(sustained_frames):
                    sess.step()
            sess.sync()
            torch.cuda.synchronize()
            dt_s, k1 = time.perf_counter() - m0, sess.times()
            out["sustained"] = {"value": round(sustained_frames / dt_s, 3), "unit": "frames/s", "frames": sustained_frames,
                                "ms_per_step": round(1e3 * dt_s / sustained_frames, 4),
                                "keyframes": int(k1.keyframes - k0.keyframes), "solves": int(k1.solves - k0.solves),
                                "ba_iterations": int(k1.solve_iterations - k0.solve_iterations),
                                "ms_per_ba_iteration": round((k1.ba_device_ms - k0.ba_device_ms) /
                                                             max(1, k1.solve_iterations - k0.solve_iterations), 4),
                                "note": "the timed region's mode continued for %d more frames of the same stream" % sustained_frames}
        if variant_frames > 0:
            # the same stream continued through the other three combinations of {host, resident} x {inline, pipelined}
            def leg(thr, res):
                sess.api.set_threading(1 if thr else 0)
                sess.device_frames = dev_frames0 if res else None
                if native:
                    sess.step_n(LEG_SETTLE)      # the mode's one-off costs (thread, contexts, buffers) stay out of the leg
                else:
                    for _ in range(LEG_SETTLE):
                        sess.step()
                sess.sync()
                torch.cuda.synchronize()
                tt0 = sess.times()
                kf0, it0, sv0 = tt0.keyframes, tt0.solve_iterations, tt0.solves
                # in four chunks, each timed: a leg is 55-60 ms of wall clock, and one hiccup of the host (observed: ~45 ms once per
                # bench run, in a different leg every time, with identical solve / iteration counts) halves its rate -- the chunk
                # times show it for what it is
                v0 = time.perf_counter()
                chunk_ms, left = [], variant_frames
                while left > 0:
                    nchunk = min(left, max(1, (variant_frames + 3) // 4))
                    c0 = time.perf_counter()
                    if native:
                        sess.step_n(nchunk)
                    else:
                        for _ in range(nchunk):
                            sess.step()
                    sess.sync()
                    chunk_ms.append(round(1e3 * (time.perf_counter() - c0), 2))
                    left -= nchunk
                torch.cuda.synchronize()
                dt_leg = time.perf_counter() - v0
                tt1 = sess.times()
                return round(variant_frames / dt_leg, 3), {"keyframes": int(tt1.keyframes - kf0), "solves": int(tt1.solves - sv0),
                                                           "ba_iterations": int(tt1.solve_iterations - it0), "chunk_ms": chunk_ms}

            var = {}
            for name, thr, res in (("resident", False, True), ("pipelined", True, False), ("pipelined_resident", True, True)):
                if (thr, res) == (pipelined, resident):
                    thr, res, name = False, False, "inline_host"   # the timed region was that combination: this leg is the default one
                v_leg, kf_leg = leg(thr, res)
                # every leg runs over ITS OWN stretch of the stream: a keyframe costs several ordinary frames (S3: ~9 ms against ~1 ms),
                # and a window solve that runs into the iteration cap costs four ordinary ones: legs are comparable only at equal
                # keyframe AND iteration counts -- both stated per leg
                var[name] = dict({"value": v_leg, "unit": "frames/s", "frames": variant_frames}, **kf_leg)
            var["note"] = ("same stream continued after the timed region; resident = frame already in HBM (XRSLAMAmdPushImageDevice), "
                           "pipelined = XRSLAMAmdSetThreading(1), the reference's XRSLAM_ENABLE_THREADING build with deterministic "
                           "hand-offs (a different, reproducible trajectory: the backend is one frame late)")
            out["variants"] = var
            sess.api.set_threading(1 if pipelined else 0)
        if args.cpu_frames > 40 and world == 1 and S == 1 and real is None:
            import subprocess
            ref_lib = os.path.join(ROOT, "oracle", "_build", "libxrslam_oracle.so")
            if not os.path.exists(ref_lib):
                subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
            nc = min(args.cpu_frames, n_frames)

            def cpu_leg(threads, threading=0):
                # XR_ORACLE_THREADS is read when the session creates its KLT context (oracle/xrhip_shim.cpp)
                os.environ["XR_ORACLE_THREADS"] = str(threads)
                cpu = runner.Session(ref_lib, seq, slam_yaml=slam_yaml, sensor_yaml=sensor_yaml, threading=threading)
                for _ in range(40):
                    cpu.step()
                cpu.sync()
                import ctypes
                clk = (ctypes.c_double * 5)()
                has_clk = hasattr(cpu.lib, "orc_shim_clocks")
                if has_clk:
                    cpu.lib.orc_shim_clocks(clk, 1)     # the checker's own solve / marginalisation clocks, reset behind the warm-up
                c0, done = time.perf_counter(), 0
                while done < nc - 40 and time.perf_counter() - c0 < args.cpu_seconds:
                    cpu.step()
                    done += 1
                cpu.sync()
                ct = time.perf_counter() - c0
                if has_clk:
                    cpu.lib.orc_shim_clocks(clk, 0)
                cpu.close()
                ba = None
                if has_clk and clk[2] > 0:
                    ba = {"ms_per_ba_iteration": round(clk[0] / clk[2], 4), "solves": int(clk[1]), "iterations": int(clk[2]),
                          "ms_per_solve": round(clk[0] / max(1.0, clk[1]), 4),
                          "ms_per_marginalization": round(clk[3] / clk[4], 4) if clk[4] > 0 else None, "marginalizations": int(clk[4])}
                return round(done / ct, 3), done, ba

            sample = ("frames 40..%d of the same stream (host images, the six reference symbols) through the same host pipeline linked "
                      "against the CPU oracle (oracle/_build/libxrslam_oracle.so, gcc -O2; our restatement, not the XRSLAM binary)")
            # the reference-faithful figure: solver num_threads = 1 (estimation/solver.cpp:185), image loops on one core
            v1, d1, ba1 = cpu_leg(1)
            out["cpu_baseline"] = {"value": v1, "unit": "frames/s", "cores": 1, "kind": "port",
                                   "sample": sample % (40 + d1) + ", single thread, inline"}
            if ba1:
                # the second half of BASELINE.json's metric on the same host and the same frames: wall clock of the checker's own
                # Solver::solve / marginalize restatements (estimation/solver.cpp:176-190), all solves of the sample
                out["cpu_baseline"].update({"ms_per_ba_iteration": ba1["ms_per_ba_iteration"],
                                            "ms_per_marginalization": ba1["ms_per_marginalization"], "ba": ba1})
            # OpenCV spreads the image / LK point loops with parallel_for_: same sample with those loops on the host's
            # cores (capped at 16), solver and marginalisation still single-threaded like the reference, and the backend on a
            # thread of its own (the strongest CPU configuration this port has)
            cores = min(16, len(os.sched_getaffinity(0)))
            if cores > 1:
                vm, dm, bam = cpu_leg(cores, 1)
                out["cpu_baseline_mt"] = {"value": vm, "unit": "frames/s", "cores": cores, "kind": "port",
                                          "sample": sample % (40 + dm) + ", image and LK point loops on %d OpenMP threads, backend "
                                                    "thread beside the feature tracker (pipelined mode)" % cores}
        print(json.dumps(out))
    group.barrier()
    for s in sessions:
        s.close()
    if igroup is not None:
        igroup.close()
    group.close()


if __name__ == "__main__":
    main()
