#!/bin/bash
# One GPU call: parity suite, then the bench line in both threading modes, alternating.
#   gpurun --timeout 700 -- tools/gpu_pipelined.sh TAG
set -uo pipefail
cd "$(dirname "$0")/.."
TAG="${1:?tag}"
mkdir -p gpurun_out
timeout 200 python -m pytest tests/test_pipelined.py -m gpu -x -q > "gpurun_out/gpu_tests_pipelined_$TAG.log" 2>&1; tail -3 "gpurun_out/gpu_tests_pipelined_$TAG.log"
if [ -z "${PIPE_ONLY:-}" ]; then timeout 400 python -m pytest tests -m gpu -x -q > "gpurun_out/gpu_tests_$TAG.log" 2>&1; tail -3 "gpurun_out/gpu_tests_$TAG.log"; fi
for rep in 1 2; do
  for M in pipelined inline; do
    XRHIP_HOSTPROF=1 timeout 120 python bench.py --threading $M --steps 300 --warmup 50 --cpu-frames 0 --host-frames 0 --inline-frames 0 \
      > "gpurun_out/bench_${TAG}_${M}_$rep.json" 2> "gpurun_out/bench_${TAG}_${M}_$rep.err"
    python - "$M" "gpurun_out/bench_${TAG}_${M}_$rep.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print(sys.argv[1], d["value"], "frames/s", d["ms_per_step"], "ms/step  ba-it", d["ms_per_ba_iteration"], "ate", d["ate_rmse_m"],
          "wall", d["host_wall_ms_per_frame"], "scopes", d["host_scope_ms_per_frame"])
except Exception as e:
    print(sys.argv[1], "failed:", repr(e))
PY
    grep "mirror_frame:" "gpurun_out/bench_${TAG}_${M}_$rep.err" | tail -1
  done
done
for M in pipelined inline; do
  timeout 200 python bench.py --threading $M --sequences-per-gpu 4 --cpu-frames 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('S=4 $M', d['value'])"
done
timeout 120 python bench.py > "gpurun_out/bench_$TAG.json" 2> "gpurun_out/bench_$TAG.err"; cut -c1-400 "gpurun_out/bench_$TAG.json"
