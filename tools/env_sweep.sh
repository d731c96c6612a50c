#!/bin/bash
# Runtime-knob sweep: bench.py (GPU leg only) under a few HIP / ROCr environment settings, two runs each.
# Usage (GPU box): tools/env_sweep.sh > gpurun_out/env_sweep.txt
cd "$(dirname "$0")/.."
run() {
  name="$1"; shift
  for rep in 1 2; do
    v=$(env "$@" timeout 60 python bench.py --steps 300 --warmup 50 --cpu-frames 0 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print(d['value'], d['ms_per_step'], d['ms_per_ba_iteration'], d['host_wall_ms_per_frame']['solve'])")
    echo "$name rep$rep: $v"
  done
}
run base XR_DUMMY=0
run dev_kernarg HIP_FORCE_DEV_KERNARG=1
run no_interrupt HSA_ENABLE_INTERRUPT=0
run both HIP_FORCE_DEV_KERNARG=1 HSA_ENABLE_INTERRUPT=0
run hwq8 GPU_MAX_HW_QUEUES=8
