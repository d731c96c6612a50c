#!/bin/bash
# Builds libxrslam_hip.so (gfx950 only) in-tree.  Usage: build.sh [extra hipcc flags]
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/../lib"
# XR_VARIANT=name builds lib/libxrslam_hip_name.so from _obj_name/ (e.g. XR_VARIANT=kprof build.sh -DXRHIP_KPROF)
VAR="${XR_VARIANT:+_$XR_VARIANT}"
OBJ="$HERE/_obj$VAR"
mkdir -p "$OUT" "$OBJ"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -pthread -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-result -I$HERE/../../include"
# kernel revision = hash of the device sources; bench.py prints PMC traffic only from profiles taken at the same revision
REV="$(cat "$HERE"/*.hip.h | sha1sum | cut -c1-10)"   # the kernels live in the *.hip.h headers; the *.hip files hold the C ABI around them
if [ "$(cat "$HERE/kernel_rev.gen.h" 2>/dev/null)" != "#define XRHIP_KERNEL_REV \"$REV\"" ]; then
  echo "#define XRHIP_KERNEL_REV \"$REV\"" > "$HERE/kernel_rev.gen.h"
fi
SRCS="$(ls "$HERE"/*.hip) $(ls "$HERE"/host/*.cpp)"
OBJS=""
pids=()
for s in $SRCS; do
  b="$(basename "$s")"; o="$OBJ/${b%.*}.o"
  OBJS="$OBJS $o"
  if [ ! -f "$o" ] || [ -n "$(find "$HERE" "$HERE/../../include" "$HERE/host" -maxdepth 1 \( -name '*.h' -o -name '*.hpp' -o -name "$(basename "$s")" \) -newer "$o" | head -1)" ]; then
    $HIPCC $FLAGS "$@" -c "$s" -o "$o" &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -pthread -o "$OUT/libxrslam_hip$VAR.so" $OBJS
echo "built $OUT/libxrslam_hip$VAR.so"
# headless EuRoC player (host only: XRSLAM.h + zlib), links the library just built
if [ -z "$VAR" ]; then
  BIN="$HERE/../bin"; mkdir -p "$BIN"
  P="$HERE/player/xrslam_player.cpp"
  if [ ! -f "$BIN/xrslam-player" ] || [ -n "$(find "$HERE/player" "$HERE/host" "$HERE/../../include" -maxdepth 1 -newer "$BIN/xrslam-player" | head -1)" ] || [ "$OUT/libxrslam_hip.so" -nt "$BIN/xrslam-player" ]; then
    g++ -O2 -std=c++17 -Wall -I"$HERE/../../include" -I"$HERE/player" "$P" -o "$BIN/xrslam-player" -L"$OUT" -lxrslam_hip -lz -Wl,-rpath,'$ORIGIN/../lib'
    echo "built $BIN/xrslam-player"
  fi
fi
