#!/bin/bash
# Overlapped detection (inline mode): parity tests, then an interleaved A/B of the development switch on the bench line
cd "$(dirname "$0")/.."; TAG="${1:-ov}"; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_ba_gpu.py tests/test_pipeline.py tests/test_bench_stream_parity.py tests/test_player_gpu.py tests/test_instances.py -m gpu -x -q > gpurun_out/tests_$TAG.log 2>&1; tail -3 gpurun_out/tests_$TAG.log
bash tools/gpu_ab_env.sh $TAG XRSLAM_AMD_NO_DETECT_OVERLAP 3
