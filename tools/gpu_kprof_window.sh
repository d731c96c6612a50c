#!/bin/bash
# kprof phase sums of the window solves only (na >= 100): per-launch microseconds of kb_solve_try's phases
cd "$(dirname "$0")/.."; R=$PWD; TAG="${1:-kw}"; mkdir -p gpurun_out
XRSLAM_HIP_LIB=$R/xrslam_amd/lib/libxrslam_hip_kprof.so XRHIP_KPROF_MIN_NA=100 timeout 200 python bench.py --steps 300 --warmup 50 --cpu-frames 0 --variant-frames 0 --threading inline 2>/dev/null | grep '^{' > gpurun_out/kprof_${TAG}_window.json
python - gpurun_out/kprof_${TAG}_window.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
k = d["kprof_ms"]; n = d["roofline_solve"]["launches"]
print("fps", d["value"], "solve_try launches (timed part)", n, "us", d["roofline_solve"]["launch_us"])
print("slots ms:", {i: v for i, v in enumerate(k) if v})
PY
