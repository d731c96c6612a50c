#!/bin/bash
# A/B of a build variant against the default library on a GPU box:
#   tools/ab_variant.sh gptr -DXRHIP_GLOBAL_PTRS
#   tools/ab_variant.sh dpp -DXRHIP_DPP_SUM
#   tools/ab_variant.sh both -DXRHIP_GLOBAL_PTRS -DXRHIP_DPP_SUM
# builds lib/libxrslam_hip_<name>.so (if missing), runs the BA / pipeline / KLT parity tests against it and then
# bench.py three times per library, alternating (run-to-run spread on one box is about 1 %).
set -euo pipefail
cd "$(dirname "$0")/.."
name="$1"; shift
lib="$PWD/xrslam_amd/lib/libxrslam_hip_$name.so"
[ -f "$lib" ] || XR_VARIANT="$name" bash xrslam_amd/csrc/build.sh "$@"
[ -n "${AB_NO_TESTS:-}" ] || XRSLAM_HIP_LIB="$lib" python -m pytest tests/test_ba_gpu.py tests/test_klt_gpu.py tests/test_pipeline.py -m gpu -x -q
one() { env "$@" python bench.py --steps 300 --warmup 50 --cpu-frames 0 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print(d['value'], d['ms_per_step'], d['ms_per_ba_iteration'], d['host_scope_ms_per_frame']['localize'], d['host_scope_ms_per_frame']['refine_window'])"; }
for rep in $(seq 1 ${AB_REPS:-3}); do
  echo "default rep$rep: $(one XR_DUMMY=0)"
  echo "$name rep$rep: $(one XRSLAM_HIP_LIB="$lib")"
done
