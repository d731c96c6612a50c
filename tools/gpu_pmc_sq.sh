#!/bin/bash
# SQ / instruction-cache counter passes (each in its own run, kernel trace only) of the default workload.
#   gpurun --timeout 600 -- tools/gpu_pmc_sq.sh TAG
set -uo pipefail
R="$(cd "$(dirname "$0")/.." && pwd)"
TAG="${1:?tag}"
mkdir -p "$R/gpurun_out"
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 60 --warmup 40 --cpu-frames 0 --variant-frames 0 --no-profile"
timeout 200 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH --kernel-trace --output-format csv -d "$R/gpurun_out/pmc_$TAG/ICACHE" -o pmc -- $B > "$R/gpurun_out/pmc_${TAG}_ICACHE.log" 2>&1
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES --kernel-trace --output-format csv -d "$R/gpurun_out/pmc_$TAG/SQ" -o pmc -- $B > "$R/gpurun_out/pmc_${TAG}_SQ.log" 2>&1
python - "$R/gpurun_out/pmc_$TAG" <<'PY'
import csv, sys, collections, os
for sub in ("ICACHE", "SQ"):
    f = os.path.join(sys.argv[1], sub, "pmc_counter_collection.csv")
    if not os.path.exists(f):
        print(sub, "missing"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    seen = set()
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].split("(")[0].replace("xrhip::", "").replace("void ", "")
        agg[n][r["Counter_Name"]] += float(r["Counter_Value"])
        k = (r["Dispatch_Id"], n)
        if k not in seen:
            seen.add(k); agg[n]["_n"] += 1; agg[n]["_ns"] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    for n, a in sorted(agg.items(), key=lambda x: -x[1]["_ns"])[:12]:
        print(sub, "%-28s" % n[:28], "n=%d avg_us=%.1f" % (a["_n"], a["_ns"] / a["_n"] / 1e3),
              " ".join("%s=%.0f" % (k.replace("SQC_ICACHE_", "IC_").replace("SQ_", ""), v / a["_n"]) for k, v in sorted(a.items()) if not k.startswith("_")))
PY
