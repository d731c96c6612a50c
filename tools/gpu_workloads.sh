#!/bin/bash
# The other workloads' bench lines (BASELINE configs 3 and 5, frozen BA problems), each with its own cpu_baseline on this host.
#   gpurun --timeout 1500 -- tools/gpu_workloads.sh TAG
set -uo pipefail
cd "$(dirname "$0")/.."
TAG="${1:?tag}"
mkdir -p gpurun_out
for W in s2 s3; do
  timeout 600 python bench.py --workload $W --steps 100 --warmup 50 --variant-frames 60 > "gpurun_out/bench_${TAG}_$W.json" 2> "gpurun_out/bench_${TAG}_$W.err"
  python - "$W" "gpurun_out/bench_${TAG}_$W.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print(sys.argv[1], d["value"], "frames/s", {k: v["value"] for k, v in d.get("variants", {}).items() if isinstance(v, dict)}, "cpu", d.get("cpu_baseline", {}).get("value"), d.get("cpu_baseline_mt", {}).get("value"), "ms/it", d["ms_per_ba_iteration"], "ate", d["ate_rmse_m"])
except Exception as e:
    print(sys.argv[1], "failed", repr(e))
PY
  tail -2 "gpurun_out/bench_${TAG}_$W.err"
done
timeout 300 python bench.py --workload s4 --steps 100 > "gpurun_out/bench_${TAG}_s4.json" 2> "gpurun_out/bench_${TAG}_s4.err"
python -c "
import json; d=json.loads(open('gpurun_out/bench_${TAG}_s4.json').read().strip().splitlines()[-1])
print('s4', d['value'], 'solves/s', 'ms/it', d['ms_per_ba_iteration'], [(s['snapshot'], s['ms_per_solve']) for s in d['snapshots']])"
