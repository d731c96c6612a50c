#!/bin/bash
# One GPU call: parity suite, default bench line, sequences-per-GPU sweep, the other workloads.
#   gpurun --timeout 900 -- tools/gpu_multi.sh TAG
set -uo pipefail
cd "$(dirname "$0")/.."
TAG="${1:?tag}"
mkdir -p gpurun_out
timeout 300 python -m pytest tests -m gpu -x -q > "gpurun_out/gpu_tests_$TAG.log" 2>&1; tail -3 "gpurun_out/gpu_tests_$TAG.log"
timeout 120 python bench.py > "gpurun_out/bench_$TAG.json" 2> "gpurun_out/bench_$TAG.err"; cut -c1-260 "gpurun_out/bench_$TAG.json"
for S in 2 4 8; do
  timeout 200 python bench.py --sequences-per-gpu $S --cpu-frames 0 > "gpurun_out/bench_${TAG}_seq$S.json" 2> "gpurun_out/bench_${TAG}_seq$S.err"
  python - "$S" "gpurun_out/bench_${TAG}_seq$S.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print("S=%s" % sys.argv[1], d["value"], "frames/s", d["ms_per_step"], "ms/step")
except Exception as e:
    print("S=%s failed: %r" % (sys.argv[1], e))
PY
done
GPU_MAX_HW_QUEUES=4 timeout 200 python bench.py --sequences-per-gpu 4 --cpu-frames 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('S=4 hwq=4', d['value'])"
for W in s2 s3 s4; do
  timeout 300 python bench.py --workload $W --cpu-frames 0 --steps 100 > "gpurun_out/bench_${TAG}_$W.json" 2> "gpurun_out/bench_${TAG}_$W.err"; cut -c1-200 "gpurun_out/bench_${TAG}_$W.json"; tail -2 "gpurun_out/bench_${TAG}_$W.err"
done
