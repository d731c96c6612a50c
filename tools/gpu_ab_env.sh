#!/bin/bash
# A/B of an environment switch on the default bench line, interleaved repetitions:  gpu_ab_env.sh TAG VAR [reps]
cd "$(dirname "$0")/.."; TAG="${1:-ab}"; VAR="${2:?env var}"; REPS="${3:-3}"; mkdir -p gpurun_out
for r in $(seq $REPS); do
  for mode in off on; do
    if [ $mode = on ]; then export $VAR=1; else unset $VAR; fi
    timeout 200 python bench.py --steps 300 --warmup 50 --cpu-frames 0 --variant-frames 0 2>/dev/null | grep '^{' > gpurun_out/ab_${TAG}_${mode}_$r.json
    python - $mode $r gpurun_out/ab_${TAG}_${mode}_$r.json "$VAR" <<'PY'
import json, sys
d = json.load(open(sys.argv[3]))
print("%s=%s rep %s: %.1f frames/s  ft_track %.4f track_wall %.4f preprocess_wall %.4f" % (sys.argv[4], "1" if sys.argv[1] == "on" else "-", sys.argv[2], d["value"],
      d["host_scope_ms_per_frame"]["ft_track"], d["host_wall_ms_per_frame"]["track"], d["host_wall_ms_per_frame"]["preprocess"]))
PY
  done
done
