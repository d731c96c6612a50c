#!/bin/bash
# Development iteration on the GPU box: BA / pipeline parity tests, in-kernel phase timers (kprof variant), the default bench line.
#   gpurun --timeout 900 -- tools/gpu_iter.sh TAG [quick]
set -uo pipefail
cd "$(dirname "$0")/.."
TAG="${1:?tag}"
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_ba_gpu.py tests/test_zz_golden_pinned_gpu.py tests/test_pipeline.py tests/test_pipelined.py tests/test_bench_stream_parity.py -m gpu -x -q > "gpurun_out/gpu_tests_$TAG.log" 2>&1; tail -6 "gpurun_out/gpu_tests_$TAG.log"
bash tools/kprof_run.sh "$TAG" 2>&1 | tail -14
timeout 240 python bench.py --cpu-frames 0 > "gpurun_out/bench_$TAG.json" 2> "gpurun_out/bench_$TAG.err"
python - "gpurun_out/bench_$TAG.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], "variants", {k: v["value"] for k, v in d.get("variants", {}).items() if isinstance(v, dict)})
print("chain_us", d["roofline"]["launch_us"], "solve_try_us", d["roofline_solve"]["launch_us"], "lk_us", d["roofline_lk"]["launch_us"], "ate", d["ate_rmse_m"])
print(d["host_scope_ms_per_frame"])
PY
