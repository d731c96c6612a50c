#!/bin/bash
# Cholesky development iteration (GPU box): the standalone check/timing, then the BA / pipeline parity tests, the bench line, a kernel trace
cd "$(dirname "$0")/.."; R=$PWD; TAG="${1:-chol}"; mkdir -p gpurun_out
./xrslam_amd/bin/xr-chol-test > gpurun_out/chol_test_$TAG.jsonl; python - gpurun_out/chol_test_$TAG.jsonl <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l)
    if d["threads"] == 512 and d["n"] in (16, 45, 90, 150, 175):
        print(d["n"], "fail", d["packed_fail"], d["tiled_fail"], "err", d["x_err_packed"], d["x_err_tiled"], "t_vs_p", d["tiled_vs_packed"], "packed_us", d["packed_us"], "tiled_us", d["tiled_us"])
PY
bash tools/gpu_kprof_print.sh $TAG tests NOPRINT 2>&1 | tail -8
