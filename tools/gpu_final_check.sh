#!/bin/bash
# The round's last GPU call: smoke(), the whole GPU parity suite, the default bench line, the driver-shaped short bench line.
#   gpurun --timeout 900 -- bash tools/gpu_final_check.sh TAG
set -uo pipefail
cd "$(dirname "$0")/.."; TAG="${1:?tag}"; mkdir -p gpurun_out
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 600 python -m pytest tests -m gpu -x -q > "gpurun_out/gpu_tests_$TAG.log" 2>&1; tail -3 "gpurun_out/gpu_tests_$TAG.log"
timeout 240 python bench.py > "gpurun_out/bench_$TAG.json" 2> "gpurun_out/bench_$TAG.err"; cut -c1-420 "gpurun_out/bench_$TAG.json"; tail -2 "gpurun_out/bench_$TAG.err"
timeout 120 python bench.py --steps 20 --warmup 5 > "gpurun_out/bench_${TAG}_driver.json" 2> "gpurun_out/bench_${TAG}_driver.err"; cut -c1-420 "gpurun_out/bench_${TAG}_driver.json"
