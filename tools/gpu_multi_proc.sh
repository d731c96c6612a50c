#!/bin/bash
# Several PROCESSES on one GPU, one sequence each (gloo rendezvous; every rank lands on device 0 of a 1-GPU box):
# does a process per sequence get past the ~3.5 k frames/s plateau of S host threads in one process?
#   gpurun --timeout 600 -- bash tools/gpu_multi_proc.sh TAG
cd "$(dirname "$0")/.."; TAG="${1:-mp}"; mkdir -p gpurun_out
for cfg in "4 2" "8 1" "8 2" "8 4" "12 2" "16 1"; do
  set -- $cfg; N=$1; Q=$2
  GPU_MAX_HW_QUEUES=$Q OMP_NUM_THREADS=1 timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29500 + N + Q)) bench.py --gpus $N --backend gloo --steps 200 --warmup 50 --cpu-frames 0 --variant-frames 0 > gpurun_out/mp_${TAG}_${N}_${Q}.json 2> gpurun_out/mp_${TAG}_${N}_${Q}.err
  python - $N $Q gpurun_out/mp_${TAG}_${N}_${Q}.json <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[3]) if l.startswith("{")][-1])
    print("procs", sys.argv[1], "queues/proc", sys.argv[2], "frames/s", d["value"], "ms/step", d["ms_per_step"], "chain_us", d["roofline"]["launch_us"], "lk_us", d["roofline_lk"]["launch_us"])
except Exception as e:
    print("procs", sys.argv[1], "queues", sys.argv[2], "failed", repr(e))
PY
done
