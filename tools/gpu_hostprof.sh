#!/bin/bash
# Host-side scope accumulators (XRHIP_HOSTPROF) of the default S1 bench stream, inline mode -> gpurun_out/hostprof_TAG.txt; then the plain bench line
cd "$(dirname "$0")/.."; TAG="${1:-hp}"; mkdir -p gpurun_out
XRHIP_HOSTPROF=1 timeout 200 python bench.py --steps 150 --warmup 50 --cpu-frames 0 --variant-frames 0 --threading inline > gpurun_out/hostprof_$TAG.json 2> gpurun_out/hostprof_$TAG.txt
grep "hostprof" gpurun_out/hostprof_$TAG.txt | cut -c1-130 | tail -40
timeout 240 python bench.py --cpu-frames 0 > "gpurun_out/bench_$TAG.json" 2> "gpurun_out/bench_$TAG.err"
python - "gpurun_out/bench_$TAG.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], "variants", {k: v["value"] for k, v in d.get("variants", {}).items() if isinstance(v, dict)})
print(d["host_scope_ms_per_frame"])
print(d["host_wall_ms_per_frame"])
PY
