#!/bin/bash
# One GPU call: parity suite, the default bench line, the driver's short bench line, the multi-rank launcher on one GPU, kernel trace.
#   gpurun --timeout 1500 -- tools/gpu_round_check.sh TAG
set -uo pipefail
R="$(cd "$(dirname "$0")/.." && pwd)"
TAG="${1:?tag}"
mkdir -p "$R/gpurun_out"
cd "$R"
timeout 900 python -m pytest tests -m gpu -x -q > "gpurun_out/gpu_tests_$TAG.log" 2>&1; tail -4 "gpurun_out/gpu_tests_$TAG.log"
timeout 240 python bench.py > "gpurun_out/bench_$TAG.json" 2> "gpurun_out/bench_$TAG.err"; cut -c1-300 "gpurun_out/bench_$TAG.json"; tail -2 "gpurun_out/bench_$TAG.err"
timeout 240 python bench.py --steps 20 --warmup 5 --cpu-frames 0 --variant-frames 0 > "gpurun_out/bench_${TAG}_driver.json" 2> "gpurun_out/bench_${TAG}_driver.err"; cut -c1-200 "gpurun_out/bench_${TAG}_driver.json"
# --gpus 2 on a one-GPU box: the launcher path.  gloo lets two ranks share the GPU; RCCL refuses two ranks on one device (recorded)
timeout 240 python bench.py --gpus 2 --backend gloo --cpu-frames 0 > "gpurun_out/bench_${TAG}_gpus2_gloo.json" 2> "gpurun_out/bench_${TAG}_gpus2_gloo.err"; cut -c1-200 "gpurun_out/bench_${TAG}_gpus2_gloo.json"; tail -2 "gpurun_out/bench_${TAG}_gpus2_gloo.err"
timeout 120 python bench.py --gpus 2 --steps 20 --warmup 5 --cpu-frames 0 > "gpurun_out/bench_${TAG}_gpus2_nccl.json" 2> "gpurun_out/bench_${TAG}_gpus2_nccl.err"; echo "nccl 2 ranks on 1 GPU rc=$?"; cut -c1-200 "gpurun_out/bench_${TAG}_gpus2_nccl.json"; grep -i "duplicate\|error" "gpurun_out/bench_${TAG}_gpus2_nccl.err" | head -3
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof_$TAG" -o full -- python "$R/bench.py" --steps 100 --warmup 40 --cpu-frames 0 --variant-frames 0 --no-profile > "$R/gpurun_out/prof_$TAG.log" 2>&1
ls "$R/gpurun_out/prof_$TAG"/* | head
