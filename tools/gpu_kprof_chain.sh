#!/bin/bash
# kprof phase sums of kb_chain: tiny (localize_newframe) and mid (refine_subwindow) problems separately; per-launch microseconds
cd "$(dirname "$0")/.."; R=$PWD; TAG="${1:-kc}"; mkdir -p gpurun_out
run() { env XRSLAM_HIP_LIB=$R/xrslam_amd/lib/libxrslam_hip_kprof.so "$@" timeout 200 python bench.py --steps 300 --warmup 50 --cpu-frames 0 --variant-frames 0 --threading inline 2>/dev/null | grep '^{'; }
run XRHIP_KPROF_MAX_NA=16 > gpurun_out/kprof_${TAG}_tiny.json
run XRHIP_KPROF_MIN_NA=17 XRHIP_KPROF_MAX_NA=99 > gpurun_out/kprof_${TAG}_mid.json
for k in tiny mid; do python - $k gpurun_out/kprof_${TAG}_$k.json <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
k = d["kprof_ms"]
rounds = k[27] if len(k) > 27 else 0
print(sys.argv[1], "fps", d["value"], "chain_us", d["roofline"]["launch_us"], "launches(timed)", d["roofline"]["launches"], "rounds(all)", rounds)
print("   ms by slot:", {i: v for i, v in enumerate(k) if v})
PY
done
