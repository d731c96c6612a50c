#!/bin/bash
# Host-side development iteration (GPU box): the parity suites that exercise the host pipeline and the upload path, then the bench line twice
cd "$(dirname "$0")/.."; TAG="${1:-host}"; mkdir -p gpurun_out
timeout 700 python -m pytest tests/test_klt_gpu.py tests/test_pipeline.py tests/test_pipelined.py tests/test_bench_stream_parity.py tests/test_player_gpu.py -m gpu -x -q > gpurun_out/tests_$TAG.log 2>&1; tail -3 gpurun_out/tests_$TAG.log
for r in 1 2; do
  timeout 240 python bench.py --cpu-frames 0 > "gpurun_out/bench_${TAG}_$r.json" 2> "gpurun_out/bench_${TAG}_$r.err"
  python - "gpurun_out/bench_${TAG}_$r.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], "variants", {k: v["value"] for k, v in d.get("variants", {}).items() if isinstance(v, dict)})
print(d["host_scope_ms_per_frame"]); print(d["host_wall_ms_per_frame"])
PY
done
