#!/bin/bash
# Front-end development iteration (GPU box): KLT parity tests (+ pinned golden pair, pipeline), the bench line, a kernel trace
cd "$(dirname "$0")/.."; R=$PWD; TAG="${1:-klt}"; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_klt_gpu.py tests/test_zz_golden_pinned_gpu.py tests/test_pipeline.py -m gpu -x -q > gpurun_out/tests_$TAG.log 2>&1; tail -5 gpurun_out/tests_$TAG.log
bash tools/gpu_kprof_print.sh $TAG notests NOPRINT 2>&1 | tail -6
