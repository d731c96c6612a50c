#!/bin/bash
# Quick GPU confirmation of a host-side change: the pipeline parity tests and one bench line
cd "$(dirname "$0")/.."; TAG="${1:-q}"; mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_pipeline.py tests/test_player_gpu.py -m gpu -x -q > gpurun_out/tests_$TAG.log 2>&1; tail -2 gpurun_out/tests_$TAG.log
timeout 200 python bench.py --cpu-frames 0 > "gpurun_out/bench_$TAG.json" 2> "gpurun_out/bench_$TAG.err"
python - "gpurun_out/bench_$TAG.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], "ms/step", d["ms_per_step"], "variants", {k: v["value"] for k, v in d.get("variants", {}).items() if isinstance(v, dict)})
print(d["host_wall_ms_per_frame"])
PY
