#!/bin/bash
# ThreadSanitizer run of the host pipeline's threading (no GPU): the product's host sources over the CPU oracle, built with
# -fsanitize=thread, driven through ctypes in four scenarios -- pipelined mode on the process-global instance, inline mode with
# two mode switches, and two pipelined instances from two threads (each with its backend and marginalisation-launcher threads).
# Expected output: four "done" lines and no "WARNING: ThreadSanitizer".      tools/tsan_check.sh
set -euo pipefail
cd "$(dirname "$0")/../oracle"
B=_build/tsan; mkdir -p $B
H=../xrslam_amd/csrc/host
F="-O1 -g -fPIC -fsanitize=thread -pthread -ffp-contract=off -fno-fast-math -Wno-unused-function"
gcc $F -std=gnu99 -c klt_oracle.c -o $B/klt.o
g++ $F -std=c++17 -c ba_oracle.cpp -o $B/ba.o
g++ $F -std=c++17 -c xrhip_shim.cpp -o $B/shim.o
g++ $F -std=c++17 -c $H/xrslam_api.cpp -o $B/api.o
g++ -shared -fsanitize=thread -pthread -Wl,-Bsymbolic -o $B/libxrslam_oracle_tsan.so $B/klt.o $B/ba.o $B/shim.o $B/api.o -lm
cd ..
cat > /tmp/xr_tsan_run.py <<'PY'
import sys, threading
sys.path.insert(0, sys.argv[1])
from xrslam_amd.harness import runner, scene
lib = sys.argv[1] + "/oracle/_build/tsan/libxrslam_oracle_tsan.so"
seq = scene.make_sequence(n_frames=66, seed=1)
def run(mode, instance):
    s = runner.Session(lib, seq, threading=mode, instance=instance)
    k = 0
    while s.step():
        k += 1
        if k == 50: s.api.set_threading(1 - mode)
        if k == 58: s.api.set_threading(mode)
    s.flush(); s.sync()
    t = s.times()
    print("done", mode, instance, t.frames, t.solves, t.marginalizations, round(runner.ate_rmse(s.poses, seq), 5))
    s.close()
run(1, False)
run(0, False)
th = [threading.Thread(target=run, args=(1, True)) for _ in range(2)]
[t.start() for t in th]; [t.join() for t in th]
PY
TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0 history_size=4" LD_PRELOAD="$(gcc -print-file-name=libtsan.so)" \
  python /tmp/xr_tsan_run.py "$PWD" 2>&1 | tee /tmp/xr_tsan.log | grep "^done\|ThreadSanitizer: reported" || true
n=$(grep -c "WARNING: ThreadSanitizer" /tmp/xr_tsan.log || true)
echo "ThreadSanitizer warnings: $n"
test "$n" = "0"
