#!/bin/bash
# One profiling call on the GPU box (run through gpurun): parity tests, default bench line, rocprofv3 kernel trace and
# the two PMC passes (each counter in its own run, no other trace domain), everything under gpurun_out/.
#   gpurun --timeout 600 -- tools/gpu_profile.sh v18
# afterwards, here:  XR_ROUND=r02 python tools/make_profile_summary.py v18 v18 "what changed"
set -uo pipefail
R="$(cd "$(dirname "$0")/.." && pwd)"
TAG="${1:?tag}"
mkdir -p "$R/gpurun_out"
cd "$R"
timeout 900 python -m pytest tests -m gpu -x -q > "gpurun_out/gpu_tests_$TAG.log" 2>&1; tail -2 "gpurun_out/gpu_tests_$TAG.log"
timeout 300 python bench.py > "gpurun_out/bench_$TAG.json" 2> "gpurun_out/bench_$TAG.err"; cut -c1-300 "gpurun_out/bench_$TAG.json"
cd /tmp && export TMPDIR=/tmp
timeout 120 rocprofv3 --kernel-trace --memory-copy-trace --stats -d "$R/gpurun_out/prof_$TAG" -o full -- python "$R/bench.py" --steps 100 --warmup 40 --cpu-frames 0 --variant-frames 0 --no-profile > "$R/gpurun_out/prof_$TAG.log" 2>&1
"$R/xrslam_amd/bin/xr-peaks" > "$R/gpurun_out/peaks_$TAG.json" 2> "$R/gpurun_out/peaks_$TAG.err"; cat "$R/gpurun_out/peaks_$TAG.json"
rocprofv3 -L > "$R/gpurun_out/counters_$TAG.txt" 2>&1
# MFMA utilisation pass (SQ block only; own run like the TCC passes).  MFMA_ALT is the fallback counter set.
timeout 120 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d "$R/gpurun_out/pmc_$TAG/MFMA" -o pmc -- python "$R/bench.py" --steps 60 --warmup 40 --cpu-frames 0 --variant-frames 0 --no-profile > "$R/gpurun_out/pmc_${TAG}_MFMA.log" 2>&1 || \
  timeout 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d "$R/gpurun_out/pmc_$TAG/MFMA_ALT" -o pmc -- python "$R/bench.py" --steps 60 --warmup 40 --cpu-frames 0 --variant-frames 0 --no-profile > "$R/gpurun_out/pmc_${TAG}_MFMA_ALT.log" 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 120 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$R/gpurun_out/pmc_$TAG/$C" -o pmc -- python "$R/bench.py" --steps 60 --warmup 40 --cpu-frames 0 --variant-frames 0 --no-profile > "$R/gpurun_out/pmc_${TAG}_$C.log" 2>&1
done
ls "$R/gpurun_out/prof_$TAG" "$R/gpurun_out/pmc_$TAG"/* 2>/dev/null | head
