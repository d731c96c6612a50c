#!/bin/bash
# Several sequences per GPU: throughput at S = 1, 2, 4, 8 (+16) and the kernel trace at S = 8 (do kernels stretch, or do they wait?).
#   gpurun --timeout 900 -- tools/gpu_multi_seq.sh TAG
set -uo pipefail
R="$(cd "$(dirname "$0")/.." && pwd)"
TAG="${1:?tag}"
cd "$R"; mkdir -p gpurun_out
for S in 1 2 4 8 16; do
  timeout 300 python bench.py --sequences-per-gpu $S --cpu-frames 0 --variant-frames 0 --steps 150 --warmup 50 > "gpurun_out/bench_${TAG}_seq$S.json" 2> "gpurun_out/bench_${TAG}_seq$S.err"
  python - "$S" "gpurun_out/bench_${TAG}_seq$S.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print("S=%s" % sys.argv[1], d["value"], "frames/s", d["ms_per_step"], "ms/step", "chain_us", d["roofline"]["launch_us"], "solve_us", d["roofline_solve"]["launch_us"], "lk_us", d["roofline_lk"]["launch_us"])
except Exception as e:
    print("S=%s failed: %r" % (sys.argv[1], e))
PY
done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof_${TAG}_seq8" -o full -- python "$R/bench.py" --sequences-per-gpu 8 --steps 100 --warmup 40 --cpu-frames 0 --variant-frames 0 --no-profile > "$R/gpurun_out/prof_${TAG}_seq8.log" 2>&1
python - "$R/gpurun_out/prof_${TAG}_seq8/full_results.db" <<'PY'
import sqlite3, sys, collections
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name,start,end from kernels order by start").fetchall()
tot = collections.defaultdict(lambda: [0, 0])
for n, s, e in rows:
    k = n.split("(")[0]; tot[k][0] += e - s; tot[k][1] += 1
T = sum(t for t, _ in tot.values())
span = rows[-1][2] - rows[0][1]
print("kernel time %.1f ms over a span of %.1f ms: %.2f kernels in flight on average" % (T / 1e6, span / 1e6, T / span))
for k, (t, n) in sorted(tot.items(), key=lambda x: -x[1][0])[:10]:
    print("%-40s calls %5d avg %.2f us" % (k[:40], n, t / 1e3 / n))
PY
