#!/bin/bash
# In-kernel phase timers of the BA kernels (build variant -DXRHIP_KPROF), split by problem size:
#   tiny (na <= 16: localize_newframe), mid (17..99: refine_subwindow), window (>= 100: refine_window)
# usage (GPU box): tools/kprof_run.sh TAG   -> gpurun_out/kprof_TAG_{tiny,mid,window}.json (+ a kernel trace of the same library)
set -uo pipefail
cd "$(dirname "$0")/.."
R="$PWD"
TAG="${1:?tag}"
lib="$PWD/xrslam_amd/lib/libxrslam_hip_kprof.so"
[ -f "$lib" ] || XR_VARIANT=kprof bash xrslam_amd/csrc/build.sh -DXRHIP_KPROF
mkdir -p gpurun_out
run() { env XRSLAM_HIP_LIB="$lib" "$@" python bench.py --steps 300 --warmup 50 --cpu-frames 0 --variant-frames 0 --sustained-frames 0 --threading inline 2>/dev/null | grep '^{' ; }
run XRHIP_KPROF_MAX_NA=16 > gpurun_out/kprof_${TAG}_tiny.json
run XRHIP_KPROF_MIN_NA=17 XRHIP_KPROF_MAX_NA=99 > gpurun_out/kprof_${TAG}_mid.json
run XRHIP_KPROF_MIN_NA=100 > gpurun_out/kprof_${TAG}_window.json
for k in tiny mid window; do python - "$k" "gpurun_out/kprof_${TAG}_$k.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
print(sys.argv[1], "fps", d["value"], "kprof_ms", d.get("kprof_ms"))
PY
done
# the instrumented library's own kernel durations (what the phase sums must be compared with)
cd /tmp && export TMPDIR=/tmp
XRSLAM_HIP_LIB="$lib" timeout 200 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof_kprof_$TAG" -o full -- python "$R/bench.py" --steps 100 --warmup 40 --cpu-frames 0 --variant-frames 0 --sustained-frames 0 --no-profile > "$R/gpurun_out/prof_kprof_$TAG.log" 2>&1
python - "$R/gpurun_out/prof_kprof_$TAG/full_results.db" <<'PY'
import sqlite3, sys, collections
c = sqlite3.connect(sys.argv[1])
tot = collections.defaultdict(lambda: [0, 0])
for n, s, e in c.execute("select name,start,end from kernels"):
    k = n.split("(")[0]
    tot[k][0] += e - s
    tot[k][1] += 1
for k, (t, n) in sorted(tot.items(), key=lambda x: -x[1][0])[:8]:
    print("%-40s calls %4d avg %.2f us" % (k, n, t / 1e3 / n))
PY
