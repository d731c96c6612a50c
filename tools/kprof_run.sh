#!/bin/bash
# In-kernel phase timers of the BA kernels (build variant -DXRHIP_KPROF), split by problem size:
#   tiny (na <= 16: localize_newframe, kb_tiny), mid (17..99: refine_subwindow), window (>= 100: refine_window)
# usage (GPU box): tools/kprof_run.sh TAG   -> gpurun_out/kprof_TAG_{tiny,mid,window}.json
set -uo pipefail
cd "$(dirname "$0")/.."
TAG="${1:?tag}"
lib="$PWD/xrslam_amd/lib/libxrslam_hip_kprof.so"
[ -f "$lib" ] || XR_VARIANT=kprof bash xrslam_amd/csrc/build.sh -DXRHIP_KPROF
run() { env XRSLAM_HIP_LIB="$lib" "$@" python bench.py --steps 300 --warmup 50 --cpu-frames 0 --variant-frames 0 --threading inline 2>/dev/null | grep '^{' ; }
run XRHIP_KPROF_MAX_NA=16 > gpurun_out/kprof_${TAG}_tiny.json
run XRHIP_KPROF_MIN_NA=17 XRHIP_KPROF_MAX_NA=99 > gpurun_out/kprof_${TAG}_mid.json
run XRHIP_KPROF_MIN_NA=100 > gpurun_out/kprof_${TAG}_window.json
for k in tiny mid window; do python - "$k" "gpurun_out/kprof_${TAG}_$k.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
print(sys.argv[1], "fps", d["value"], "kprof_ms", d.get("kprof_ms"))
PY
done
