#!/usr/bin/env python3
"""bench.py -- frames/sec of the XRSLAM per-frame hot path on MI355X.

Contract (one JSON line on rank 0):
  python bench.py --gpus N --steps K --warmup W
A "step" is one camera frame pushed through the hot path of one sequence
(FeatureTracker::work, reference core/feature_tracker.cpp:24-153): device-resident
raw frame -> CLAHE + 4-level LK pyramid + Scharr -> forward/backward pyramidal LK
of the previous frame's keypoints -> Harris/GFTT re-detection.  One independent
sequence per GPU; no data-path collective, only a barrier
and a max-reduce of the wall time.

Workload (config.workload): synthetic "S1" EuRoC-MH_01-like stream, 752x480
(the real EuRoC cam0 size; BASELINE.json says 640x480 -- see SURVEY.md top table),
150 features (BASELINE config 2), seeded texture under a smooth camera-like image
motion.  Inputs are resident in HBM before the timed region starts.
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

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def make_stream(w, h, n_frames, seed):
    """Seeded S1-like stream: band-limited texture viewed through a slowly rotating /
    translating / zooming similarity (a few px per frame, like 20 Hz EuRoC MH_01)."""
    from tests.util import noise_image, warp_affine
    big = noise_image(w + 256, h + 256, seed=seed)
    frames = []
    for k in range(n_frames):
        t = k / 20.0
        ang = 0.06 * np.sin(0.9 * t)
        s = 1.0 + 0.04 * np.sin(0.5 * t)
        M = s * np.array([[np.cos(ang), -np.sin(ang)], [np.sin(ang), np.cos(ang)]])
        c = np.array([(w + 256) / 2.0, (h + 256) / 2.0])
        off = np.array([40.0 * np.sin(0.7 * t), 30.0 * np.sin(1.1 * t + 0.5)])
        # sample big at M (x - c_img) + c + off
        tvec = c + off - M @ np.array([w / 2.0, h / 2.0])
        full = warp_affine(big, M, tvec)
        frames.append(full[:h, :w].copy())
    return np.stack(frames)


def boomerang(n_frames, total):
    """frame index sequence 0,1,..,n-1,n-2,..,0,1.. so consecutive frames stay consecutive."""
    idx = []
    k, d = 0, 1
    for _ in range(total):
        idx.append(k)
        if k + d < 0 or k + d >= n_frames:
            d = -d
        k += d
    return idx


class GpuTracker:
    """Minimal FeatureTracker::work loop over the C ABI (python is plumbing only)."""

    def __init__(self, w, h, max_pts, dev_frames, stride):
        from xrslam_amd import klt
        self.klt = klt
        self.ctx = klt.KltContext(w, h, max_pts)
        self.imgs = [self.ctx.image(), self.ctx.image()]
        self.cur = 0
        self.kp = np.zeros((0, 2))
        self.max_pts = max_pts
        self.dev_frames = dev_frames
        self.stride = stride
        self.first = True
        self.tracked = 0

    def step(self, frame_index):
        nxt = self.imgs[1 - self.cur]
        nxt.upload_device(self.dev_frames + frame_index * self.stride, self.ctx.w)
        nxt.preprocess(6.0, 8, 8)
        if not self.first and len(self.kp):
            nx, st = self.imgs[self.cur].track_keypoints(nxt, self.kp, self.kp.copy())
            self.kp = nx[st > 0]
            self.tracked += int(st.sum())
        else:
            self.kp = np.zeros((0, 2))
        self.kp = nxt.detect_keypoints(self.kp, self.max_pts, 20.0)[:self.max_pts]
        self.cur = 1 - self.cur
        self.first = False


class CpuTracker:
    def __init__(self, w, h, max_pts, frames):
        from oracle import klt_oracle as ko
        self.ko = ko
        self.frames = frames
        self.prev = None
        self.kp = np.zeros((0, 2))
        self.max_pts = max_pts

    def step(self, frame_index):
        nxt = self.ko.OracleImage(self.frames[frame_index])
        nxt.preprocess(6.0, 8, 8)
        if self.prev is not None and len(self.kp):
            nx, st = self.prev.track_keypoints(nxt, self.kp, self.kp.copy())
            self.kp = nx[st > 0]
        else:
            self.kp = np.zeros((0, 2))
        self.kp = nxt.detect_keypoints(self.kp, self.max_pts, 20.0)[:self.max_pts]
        self.prev = nxt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--width", type=int, default=752)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--features", type=int, default=150)
    ap.add_argument("--unique-frames", type=int, default=48)
    ap.add_argument("--cpu-frames", type=int, default=60, help="frames of the bounded CPU-oracle sample (0 = skip)")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl" if torch.cuda.is_available() else "gloo")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    torch.cuda.set_device(local_rank)
    from xrslam_amd import _lib
    _lib.set_device(local_rank)

    w, h = args.width, args.height
    frames = make_stream(w, h, args.unique_frames, seed=1 + rank)
    dev = torch.from_numpy(frames).cuda()          # inputs resident in HBM before timing
    torch.cuda.synchronize()
    order = boomerang(args.unique_frames, args.warmup + args.steps)
    trk = GpuTracker(w, h, args.features, dev.data_ptr(), w * h)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for k in range(args.warmup):
        trk.step(order[k])
    trk.ctx.stats(reset=True)
    trk.ctx.set_profiling(True)
    trk.tracked = 0
    barrier()
    t0 = time.perf_counter()
    for k in range(args.warmup, args.warmup + args.steps):
        trk.step(order[k])
    barrier()
    elapsed = time.perf_counter() - t0
    st = trk.ctx.stats(reset=False)
    trk.ctx.set_profiling(False)
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        fps = world * args.steps / elapsed
        # roofline of the dominant kernel (k_lk_track): algorithmic bytes per launch
        # = sum over (dir, point, level) of 22*22*(1+4) template bytes + 22*22 bytes per
        # LK iteration / mean launch duration (HIP events, same stream)
        n_launch = max(1, st.n_track)
        lk_bytes = 2420.0 * st.lk_templates + 484.0 * st.lk_iterations
        lk_ms = st.ms_track / n_launch
        achieved = (lk_bytes / n_launch) / (lk_ms * 1e-3) / 1e9 if lk_ms > 0 else 0.0
        out = {
            "metric": "frames/sec (per-frame hot path, one sequence per GPU)",
            "value": round(fps, 3),
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8/i16 images, f32 LK, i64 reductions",
            "data": "synthetic",
            "config": {"workload": "S1 EuRoC-MH_01-like synthetic stream %dx%d, %d features, tracker stage "
                                   "(CLAHE+pyramid+fwd/bwd LK+Harris redetect)" % (w, h, args.features),
                       "features": args.features, "sequences_per_gpu": 1},
            "stage_ms": {"preprocess": round(st.ms_preprocess / max(1, st.n_preprocess), 4),
                         "lk_track": round(lk_ms, 4),
                         "detect": round(st.ms_detect / max(1, st.n_detect), 4)},
            "mean_tracked": round(trk.tracked / max(1, st.n_track), 2),
            "roofline": {"kernel": "k_lk_track", "bound": "hbm", "achieved": round(achieved, 3),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6),
                         "traffic": None,
                         "algorithmic_bytes_per_launch": round(lk_bytes / n_launch, 1),
                         "launch_us": round(lk_ms * 1e3, 3)},
        }
        if args.cpu_frames > 0 and world == 1:
            cpu = CpuTracker(w, h, args.features, frames)
            corder = boomerang(args.unique_frames, args.cpu_frames + 5)
            for k in range(5):
                cpu.step(corder[k])
            c0 = time.perf_counter()
            for k in range(5, 5 + args.cpu_frames):
                cpu.step(corder[k])
            ct = time.perf_counter() - c0
            out["cpu_baseline"] = {"value": round(args.cpu_frames / ct, 3), "unit": "frames/s", "cores": 1,
                                   "kind": "port",
                                   "sample": "%d frames of the same stream through oracle/klt_oracle.c "
                                             "(single thread, gcc -O2)" % args.cpu_frames}
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
