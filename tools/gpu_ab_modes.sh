#!/bin/bash
# One GPU call: the parity tests of the pipelined mode and of the queued pre-integration, then bench.py in several
# configurations, alternating (boxes differ by up to 30 % on the single-workgroup f64 kernels: only compare within a call).
#   gpurun --timeout 600 -- tools/gpu_ab_modes.sh TAG
set -uo pipefail
cd "$(dirname "$0")/.."
TAG="${1:?tag}"
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_ba_gpu.py tests/test_pipelined.py -m gpu -x -q -k "queued or preintegration or pipelined" > "gpurun_out/gpu_tests_ab_$TAG.log" 2>&1; tail -15 "gpurun_out/gpu_tests_ab_$TAG.log"
one() {   # label, env assignments / bench flags
  local label="$1"; shift
  local envs=() flags=()
  for a in "$@"; do if [[ "$a" == *=* && "$a" != --* ]]; then envs+=("$a"); else flags+=("$a"); fi; done
  env "${envs[@]}" XRHIP_HOSTPROF=1 timeout 120 python bench.py --steps 300 --warmup 50 --cpu-frames 0 --variant-frames 0 "${flags[@]}" \
      > "gpurun_out/bench_${TAG}_$label.json" 2> "gpurun_out/bench_${TAG}_$label.err"
  python - "$label" "gpurun_out/bench_${TAG}_$label.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    sc = d["host_scope_ms_per_frame"]
    print("%-22s %8.1f f/s  %.4f ms  chain %.1f us  wait %.3f  mirror %.3f localize %.3f window %.3f slide %.3f sub %.3f  ft_track %.3f ate %s"
          % (sys.argv[1], d["value"], d["ms_per_step"], d["roofline"]["launch_us"], d.get("backend_wait_ms_per_frame", 0), sc["mirror_frame"],
             sc["localize"], sc["refine_window"], sc["slide_window"], sc["refine_subwindow"], sc["ft_track"], d["ate_rmse_m"]))
except Exception as e:
    print(sys.argv[1], "failed:", repr(e))
PY
}
for rep in 1 2; do
  one "pipelined_native_$rep" --threading pipelined

  one "inline_native_$rep" --threading inline


done
grep "mirror_frame:" "gpurun_out/bench_${TAG}_pipelined_native_1.err" | tail -1
timeout 120 python bench.py > "gpurun_out/bench_$TAG.json" 2> "gpurun_out/bench_$TAG.err"; cut -c1-300 "gpurun_out/bench_$TAG.json"; tail -3 "gpurun_out/bench_$TAG.err"
