#!/bin/bash
# same-call A/B of kernel traces: default library vs a variant library, alternating
R="$(cd "$(dirname "$0")/.." && pwd)"; O="$R/gpurun_out"; V="$1"; REPS="${2:-4}"
for rep in $(seq "$REPS"); do for v in default "$V"; do
  e="XR_DUMMY=0"; [ "$v" != default ] && e="XRSLAM_HIP_LIB=$R/xrslam_amd/lib/libxrslam_hip_$v.so"
  (cd /tmp && export TMPDIR=/tmp && env $e timeout 300 rocprofv3 --kernel-trace --stats -d "$O/abtrace_${v}_$rep" -o full -- python "$R/bench.py" --steps 100 --warmup 40 --cpu-frames 0 --variant-frames 0 --sustained-frames 0 --no-profile > "$O/abtrace_${v}_$rep.log" 2>&1)
  python - "$O/abtrace_${v}_$rep" "$v rep$rep" <<'PY'
import sqlite3, sys, collections, glob
db = glob.glob(sys.argv[1] + "/*results.db")
c = sqlite3.connect(db[0]); tot = collections.defaultdict(lambda: [0, 0])
for n, s, e in c.execute("select name,start,end from kernels"):
    k = n.split("(")[0].replace("void ", "").replace("xrhip::", ""); tot[k][0] += 1; tot[k][1] += e - s
allt = sum(t for n, t in tot.values())
print(sys.argv[2], "total %.2f ms" % (allt / 1e6), "  ".join("%s %.1f" % (k, tot[k][1] / tot[k][0] / 1e3) for k in ("k_lk_track", "k_harris", "k_harris_nms", "k_harris_select", "kp_preintegrate", "kb_chain")))
PY
done; done
