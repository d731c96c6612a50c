#!/bin/bash
# In-kernel block timers (kprof variant prints the slow blocks of kb_landmark_vision / kb_schur_aux) + a kernel trace of the default library.
#   gpurun --timeout 900 -- bash tools/gpu_block_timers.sh TAG [notests]
cd "$(dirname "$0")/.."; R=$PWD; TAG="${1:-blk}"; mkdir -p gpurun_out
[ "${2:-}" = notests ] || timeout 500 python -m pytest tests/test_ba_gpu.py tests/test_zz_golden_pinned_gpu.py tests/test_pipeline.py tests/test_bench_stream_parity.py -m gpu -x -q 2>&1 | tail -4
XRSLAM_HIP_LIB=$R/xrslam_amd/lib/libxrslam_hip_kprof.so timeout 120 python bench.py --steps 60 --warmup 40 --cpu-frames 0 --variant-frames 0 --threading inline 2>/dev/null | grep -v '^{' > gpurun_out/blocks_$TAG.txt
sort gpurun_out/blocks_$TAG.txt | cut -d: -f1 | uniq -c | sort -rn | head -20
cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$TAG -o full -- python $R/bench.py --steps 100 --warmup 40 --cpu-frames 0 --variant-frames 0 --no-profile > $R/gpurun_out/prof_$TAG.log 2>&1
grep '^{' $R/gpurun_out/prof_$TAG.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('value', d['value'])"
python - $R/gpurun_out/prof_$TAG/full_results.db <<'PY'
import sqlite3, sys, collections
c = sqlite3.connect(sys.argv[1])
tot = collections.defaultdict(lambda: [0, 0])
for n, s, e in c.execute("select name,start,end from kernels"):
    k = n.split("(")[0]; tot[k][0]+=1; tot[k][1]+=e-s
for k,(n,t) in sorted(tot.items(), key=lambda x:-x[1][1])[:22]: print("%-28s n=%5d avg=%.2f us"%(k[:28],n,t/n/1e3))
PY
