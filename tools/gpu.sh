#!/bin/bash
# One parametrised GPU call (run through gpurun); everything lands under gpurun_out/.  Replaces the one-off gpu_*.sh scripts of
# rounds 1-3 (git history has them): every measurement of a round is a list of steps of this script.
#   gpurun --timeout 1500 -- tools/gpu.sh TAG step [step ...]
# steps (NAME or NAME:ARGS; ARGS are passed on verbatim):
#   tests[:pytest args]     the GPU suite (default: tests)              smoke              __graft_entry__.smoke()
#   bench[:bench.py args]   one bench.py line + a one-line digest       driver             the driver's --steps 20 --warmup 5 line
#   multi:S [args]          bench.py --sequences-per-gpu S              procs:N Q [S]      N processes on this GPU (gloo), Q HW queues each, S grouped sequences in each
#   rccl1                   RunGroup's collectives over nccl (= RCCL) with one rank (tools/check_rccl.py)
#   trace[:bench.py args]   rocprofv3 --kernel-trace --stats + per-kernel averages         pmc[:args]   MFMA / FETCH_SIZE / WRITE_SIZE passes
#   pmcg[:S]                FETCH_SIZE / WRITE_SIZE passes of S grouped sequences (default 8)        pmcsq   SQ wave-cycle / instruction counters (tools/sq_summary.py)
#   ab:VARIANT              default library vs lib/libxrslam_hip_VARIANT.so, alternating (S1 line and S4 replay)
#   abenv:VAR [reps]        the default bench line with VAR unset / =1, interleaved
#   hostprof                XRHIP_HOSTPROF scope accumulators of the S1 stream             kprint[:PATTERN]   in-kernel printf timers (kprint variant)
#   multiq[:Q ...]          tools/multiq.hip: small dependent kernels from 1..16 host threads, per GPU_MAX_HW_QUEUES
#   peaks                   tools/peaks.hip micro-benchmarks            clocks / host      rocm-smi clock / power state; CPU cores
set -uo pipefail
R="$(cd "$(dirname "$0")/.." && pwd)"
TAG="${1:?tag}"; shift
O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
BENCH_PROF="--steps 100 --warmup 40 --cpu-frames 0 --variant-frames 0 --sustained-frames 0 --no-profile"
digest() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
except Exception as e:
    print("no bench line:", repr(e)); sys.exit(0)
print("value", d.get("value"), d.get("unit"), "ms/step", d.get("ms_per_step"), "variants", {k: v["value"] for k, v in d.get("variants", {}).items() if isinstance(v, dict)})
if "roofline" in d:
    print("chain_us", d["roofline"]["launch_us"], "solve_try_us", d["roofline_solve"]["launch_us"], "lk_us", d["roofline_lk"]["launch_us"], "ba_it_ms", d.get("ms_per_ba_iteration"), "ate", d.get("ate_rmse_m"))
    print(d.get("host_scope_ms_per_frame")); print(d.get("host_wall_ms_per_frame"))
for k in ("group", "per_rank", "cpu_baseline"):
    if k in d: print(k, d[k])
PY
}
kernel_avgs() { python - "$1" <<'PY'
import sqlite3, sys, collections, glob
db = glob.glob(sys.argv[1] + "/*results.db")
if not db: print("no results db under", sys.argv[1]); sys.exit(0)
c = sqlite3.connect(db[0]); tot = collections.defaultdict(lambda: [0, 0])
for n, s, e in c.execute("select name,start,end from kernels"):
    k = n.split("(")[0].replace("void ", "").replace("xrhip::", ""); tot[k][0] += 1; tot[k][1] += e - s
print("  ".join("%s %d x %.1f" % (k[:24], n, t / n / 1e3) for k, (n, t) in sorted(tot.items(), key=lambda x: -x[1][1])[:30]))
PY
}
n=0
for step in "$@"; do
  n=$((n+1)); name="${step%%:*}"; arg=""; [ "$step" != "$name" ] && arg="${step#*:}"
  echo "--- [$n] $step"
  case "$name" in
    tests)  timeout 1500 python -m pytest ${arg:-tests} -m gpu -x -q > "$O/tests_${TAG}_$n.log" 2>&1; tail -3 "$O/tests_${TAG}_$n.log" ;;
    smoke)  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$O/smoke_$TAG.log" 2>&1; tail -2 "$O/smoke_$TAG.log" ;;
    bench)  timeout 500 python bench.py $arg > "$O/bench_${TAG}_$n.json" 2> "$O/bench_${TAG}_$n.err"; digest "$O/bench_${TAG}_$n.json"; tail -2 "$O/bench_${TAG}_$n.err" ;;
    driver) timeout 240 python bench.py --steps 20 --warmup 5 > "$O/bench_${TAG}_driver.json" 2> "$O/bench_${TAG}_driver.err"; digest "$O/bench_${TAG}_driver.json" ;;
    multi)  set -- $arg; S="$1"; shift; ENVS=""; ARGS=""   # multi:S [VAR=VALUE ...] [bench.py args]
            for t in "$@"; do case "$t" in -*) ARGS="$ARGS $t" ;; *=*) ENVS="$ENVS $t" ;; *) ARGS="$ARGS $t" ;; esac; done
            env $ENVS timeout 500 python bench.py --sequences-per-gpu "$S" --steps 200 --warmup 50 --cpu-frames 0 $ARGS > "$O/bench_${TAG}_multi${S}_$n.json" 2> "$O/bench_${TAG}_multi${S}_$n.err"
            digest "$O/bench_${TAG}_multi${S}_$n.json"; tail -2 "$O/bench_${TAG}_multi${S}_$n.err" ;;
    procs)  set -- $arg; N="$1"; Q="${2:-2}"; S="${3:-1}"   # procs:N Q [S]: N processes on this GPU (gloo), Q hardware queues each, S grouped sequences in each
            GPU_MAX_HW_QUEUES=$Q OMP_NUM_THREADS=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port $((29500 + N + Q + S)) bench.py --gpus "$N" --backend gloo --sequences-per-gpu "$S" --steps 200 --warmup 50 --cpu-frames 0 --variant-frames 0 --sustained-frames 0 > "$O/procs_${TAG}_${N}_${Q}_$S.json" 2> "$O/procs_${TAG}_${N}_${Q}_$S.err"
            digest "$O/procs_${TAG}_${N}_${Q}_$S.json"; tail -2 "$O/procs_${TAG}_${N}_${Q}_$S.err" ;;
    rccl1)  timeout 120 python tools/check_rccl.py > "$O/rccl1_$TAG.txt" 2>&1; tail -2 "$O/rccl1_$TAG.txt" ;;
    trace)  (cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$O/prof_${TAG}_$n" -o full -- python "$R/bench.py" ${arg:-$BENCH_PROF} > "$O/prof_${TAG}_$n.log" 2>&1); kernel_avgs "$O/prof_${TAG}_$n" ;;
    pmc)    (cd /tmp && export TMPDIR=/tmp
             timeout 150 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d "$O/pmc_$TAG/MFMA" -o pmc -- python "$R/bench.py" --steps 60 --warmup 40 --cpu-frames 0 --variant-frames 0 --sustained-frames 0 --no-profile $arg > "$O/pmc_${TAG}_MFMA.log" 2>&1
             for C in FETCH_SIZE WRITE_SIZE; do timeout 150 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$O/pmc_$TAG/$C" -o pmc -- python "$R/bench.py" --steps 60 --warmup 40 --cpu-frames 0 --variant-frames 0 --sustained-frames 0 --no-profile $arg > "$O/pmc_${TAG}_$C.log" 2>&1; done); ls "$O/pmc_$TAG" ;;
    pmcsq)  # where the wavefronts' cycles go (issue-bound or parked?): SQ counters of the solo run, two passes of four counters (own runs)
            (cd /tmp && export TMPDIR=/tmp
             timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d "$O/pmc_$TAG/SQ/a" -o pmc -- python "$R/bench.py" --steps 60 --warmup 40 --cpu-frames 0 --variant-frames 0 --sustained-frames 0 --no-profile $arg > "$O/pmc_${TAG}_SQa.log" 2>&1
             timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d "$O/pmc_$TAG/SQ/b" -o pmc -- python "$R/bench.py" --steps 60 --warmup 40 --cpu-frames 0 --variant-frames 0 --sustained-frames 0 --no-profile $arg > "$O/pmc_${TAG}_SQb.log" 2>&1)
            python "$R/tools/sq_summary.py" "$O/pmc_$TAG/SQ" ;;
    pmcg)   # the grouped run's counters (S = ${arg:-8} members): a batched launch should move ~S times a solo launch's bytes in about the time of one
            S="${arg:-8}"
            (cd /tmp && export TMPDIR=/tmp
             for C in FETCH_SIZE WRITE_SIZE; do timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$O/pmc_${TAG}_group/$C" -o pmc -- python "$R/bench.py" --sequences-per-gpu "$S" --steps 40 --warmup 20 --cpu-frames 0 --no-profile > "$O/pmc_${TAG}_group_$C.log" 2>&1; done); ls "$O/pmc_${TAG}_group" ;;
    ab)     lib="$R/xrslam_amd/lib/libxrslam_hip_$arg.so"
            for rep in 1 2; do for v in default "$arg"; do
              e="XR_DUMMY=0"; [ "$v" != default ] && e="XRSLAM_HIP_LIB=$lib"
              env $e timeout 200 python bench.py --steps 300 --warmup 50 --cpu-frames 0 --variant-frames 0 --sustained-frames 0 2>/dev/null > "$O/ab_${TAG}_${v}_$rep.json"; echo "$v rep$rep:"; digest "$O/ab_${TAG}_${v}_$rep.json" | head -2
            done; done
            for v in default "$arg"; do e="XR_DUMMY=0"; [ "$v" != default ] && e="XRSLAM_HIP_LIB=$lib"
              env $e timeout 200 python bench.py --workload s4 --steps 100 2>/dev/null | python -c "
import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$v S4:', '  '.join('%s %.4f' % (s['snapshot'], s['ms_per_solve']) for s in d['snapshots']))"; done ;;
    abenv)  set -- $arg; VAR="$1"; REPS="${2:-3}"
            for r in $(seq "$REPS"); do for mode in off on; do
              if [ $mode = on ]; then export "$VAR"=1; else unset "$VAR"; fi
              timeout 200 python bench.py --steps 300 --warmup 50 --cpu-frames 0 --variant-frames 0 --sustained-frames 0 2>/dev/null > "$O/abenv_${TAG}_${mode}_$r.json"; echo "$VAR $mode rep $r:"; digest "$O/abenv_${TAG}_${mode}_$r.json" | head -1
            done; done; unset "$VAR" ;;
    hostprof) XRHIP_HOSTPROF=1 timeout 300 python bench.py --steps 150 --warmup 50 --cpu-frames 0 --variant-frames 0 --sustained-frames 0 --threading inline > "$O/hostprof_$TAG.json" 2> "$O/hostprof_$TAG.txt"; grep hostprof "$O/hostprof_$TAG.txt" | cut -c1-130 | tail -40 ;;
    kprint) XRSLAM_HIP_LIB="$R/xrslam_amd/lib/libxrslam_hip_kprint.so" timeout 200 python bench.py --steps 60 --warmup 40 --cpu-frames 0 --variant-frames 0 --sustained-frames 0 --threading inline 2>/dev/null | grep -v '^{' > "$O/blocks_$TAG.txt"; grep "${arg:-kb_}" "$O/blocks_$TAG.txt" | tail -12 ;;
    multiq) for Q in ${arg:-4 8}; do echo "GPU_MAX_HW_QUEUES=$Q"; GPU_MAX_HW_QUEUES=$Q timeout 300 "$R/xrslam_amd/bin/xr-multiq" 1500 > "$O/multiq_${TAG}_q$Q.jsonl" 2> "$O/multiq_${TAG}_q$Q.err"; python - "$O/multiq_${TAG}_q$Q.jsonl" <<'PY'
import json, sys
rows = [json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")]
for m in sorted({r["mode"] for r in rows}):
    for c in sorted({r["chain"] for r in rows}):
        rr = [r for r in rows if r["mode"] == m and r["chain"] == c]
        print("mode %d chain %d: " % (m, c) + "  ".join("T%d %.1fus %.0fk/s" % (r["threads"], r["round_us"], r["kernels_per_s_all"] / 1e3) for r in rr))
PY
            done ;;
    peaks)  "$R/xrslam_amd/bin/xr-peaks" > "$O/peaks_$TAG.json" 2> "$O/peaks_$TAG.err"; cat "$O/peaks_$TAG.json" ;;
    host)   echo "nproc $(nproc)  affinity $(python -c 'import os; print(len(os.sched_getaffinity(0)))')"; lscpu | grep -E "Model name|Socket|Thread|Core" | head -5 ;;
    clocks) rocm-smi --showclocks --showpower --showmaxpower --showperflevel > "$O/clocks_$TAG.txt" 2>&1; grep -i "clock level\|power\|level" "$O/clocks_$TAG.txt" | head -12 ;;
    *)      echo "unknown step $step" ;;
  esac
done
