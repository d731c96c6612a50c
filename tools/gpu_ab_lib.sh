#!/bin/bash
# One GPU call: parity suite on the default library, then default vs a build variant (lib/libxrslam_hip_<name>.so), alternating:
# the S1 bench line and the frozen-problem replay (S4: ms per solve straight from the solver).
#   gpurun --timeout 600 -- tools/gpu_ab_lib.sh TAG prev
set -uo pipefail
cd "$(dirname "$0")/.."
TAG="${1:?tag}"; VAR="${2:?variant}"
lib="$PWD/xrslam_amd/lib/libxrslam_hip_$VAR.so"
mkdir -p gpurun_out
timeout 400 python -m pytest tests -m gpu -x -q > "gpurun_out/gpu_tests_$TAG.log" 2>&1; tail -3 "gpurun_out/gpu_tests_$TAG.log"
s1() { env "$@" timeout 120 python bench.py --steps 300 --warmup 50 --cpu-frames 0 --variant-frames 0 2>/dev/null | python -c "
import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); sc=d['host_scope_ms_per_frame']
print('%.1f f/s  ba-it %.4f  chain %.1f us  solve_try %.1f us  localize %.3f window %.3f sub %.3f' % (d['value'], d['ms_per_ba_iteration'], d['roofline']['launch_us'], d['roofline_solve']['launch_us'], sc['localize'], sc['refine_window'], sc['refine_subwindow']))"; }
s4() { env "$@" timeout 120 python bench.py --workload s4 --steps 100 2>/dev/null | python -c "
import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0])
print('  '.join('%s %.4f' % (s['snapshot'], s['ms_per_solve']) for s in d['snapshots']))"; }
for rep in 1 2; do
  echo "default rep$rep: $(s1 XR_DUMMY=0)"
  echo "$VAR    rep$rep: $(s1 XRSLAM_HIP_LIB="$lib")"
done
echo "default S4: $(s4 XR_DUMMY=0)"
echo "$VAR    S4: $(s4 XRSLAM_HIP_LIB="$lib")"
echo "default S4: $(s4 XR_DUMMY=0)"
echo "$VAR    S4: $(s4 XRSLAM_HIP_LIB="$lib")"
