#!/bin/bash
# Development iteration (GPU box):  gpurun --timeout 900 -- bash tools/gpu_kprof_print.sh TAG [tests|notests] [PATTERN]
#   1. the kprint variant (XR_VARIANT=kprint build.sh -DXRHIP_KPROF -DXRHIP_KPROF_PRINT): in-kernel printf timers on the default S1 stream -> gpurun_out/blocks_TAG.txt (lines matching PATTERN shown)
#   2. the default library: BA / pipeline parity tests (unless "notests"), the bench line, a kernel trace (average durations)
cd "$(dirname "$0")/.."; R=$PWD; TAG="${1:-blk}"; mkdir -p gpurun_out
[ "${3:-}" = NOPRINT ] || XRSLAM_HIP_LIB=$R/xrslam_amd/lib/libxrslam_hip_kprint.so timeout 120 python bench.py --steps 60 --warmup 40 --cpu-frames 0 --variant-frames 0 --threading inline 2>/dev/null | grep -v '^{' > gpurun_out/blocks_$TAG.txt
[ "${3:-}" = NOPRINT ] || grep "${3:-kb_trials_wide}" gpurun_out/blocks_$TAG.txt | grep -v '{' | tail -10
if [ "${2:-}" != notests ]; then
  timeout 500 python -m pytest tests/test_ba_gpu.py tests/test_zz_golden_pinned_gpu.py tests/test_pipeline.py tests/test_bench_stream_parity.py -m gpu -x -q > gpurun_out/tests_$TAG.log 2>&1
  tail -3 gpurun_out/tests_$TAG.log
fi
timeout 240 python bench.py --cpu-frames 0 > "gpurun_out/bench_$TAG.json" 2> "gpurun_out/bench_$TAG.err"
python - "gpurun_out/bench_$TAG.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], "variants", {k: v["value"] for k, v in d.get("variants", {}).items() if isinstance(v, dict)})
print("chain_us", d["roofline"]["launch_us"], "solve_try_us", d["roofline_solve"]["launch_us"], "lk_us", d["roofline_lk"]["launch_us"], "ate", d["ate_rmse_m"])
print(d["host_scope_ms_per_frame"])
PY
cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$TAG -o full -- python $R/bench.py --steps 100 --warmup 40 --cpu-frames 0 --variant-frames 0 --no-profile > $R/gpurun_out/prof_$TAG.log 2>&1
python - $R/gpurun_out/prof_$TAG/full_results.db <<'PY'
import sqlite3, sys, collections
c = sqlite3.connect(sys.argv[1])
tot = collections.defaultdict(lambda: [0, 0])
for n, s, e in c.execute("select name,start,end from kernels"):
    k = n.split("(")[0].replace("void ", "").replace("xrhip::", ""); tot[k][0]+=1; tot[k][1]+=e-s
print("  ".join("%s %d x %.1f" % (k[:22], n, t/n/1e3) for k,(n,t) in sorted(tot.items(), key=lambda x:-x[1][1])[:26]))
PY
