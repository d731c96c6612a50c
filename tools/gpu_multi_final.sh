#!/bin/bash
# Several sequences on one GPU with the round's final library: 8 threads in one process (8 hardware queues), 8 processes x 2 queues
cd "$(dirname "$0")/.."; TAG="${1:-msf}"; mkdir -p gpurun_out
timeout 200 python bench.py --sequences-per-gpu 8 --cpu-frames 0 --variant-frames 0 --steps 200 --warmup 50 > gpurun_out/ms_${TAG}_threads8.json 2> gpurun_out/ms_${TAG}_threads8.err
python -c "
import json; d=json.loads([l for l in open('gpurun_out/ms_${TAG}_threads8.json') if l.startswith('{')][-1]); print('threads 8:', d['value'], 'frames/s', d['ms_per_step'], 'chain_us', d['roofline']['launch_us'])"
GPU_MAX_HW_QUEUES=2 OMP_NUM_THREADS=1 timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 8 --backend gloo --steps 200 --warmup 50 --cpu-frames 0 --variant-frames 0 > gpurun_out/ms_${TAG}_procs8.json 2> gpurun_out/ms_${TAG}_procs8.err
python -c "
import json; d=json.loads([l for l in open('gpurun_out/ms_${TAG}_procs8.json') if l.startswith('{')][-1]); print('procs 8 x 2 queues:', d['value'], 'frames/s', d['ms_per_step'], 'chain_us', d['roofline']['launch_us'])"
