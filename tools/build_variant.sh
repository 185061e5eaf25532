#!/bin/bash
# Build a variant of the library for experiments: tools/build_variant.sh NAME "EXTRA FLAGS" file.hip [file.hip ...]
# recompiles the named sources with the extra flags, links them with the default build's other objects into
# tools/_abl/lib_NAME.so (git-ignored; travels to the GPU box).  Use with POLYBLUR_HIP_LIB=tools/_abl/lib_NAME.so.
set -e
NAME=$1; FLAGS=$2; shift 2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OBJ=$ROOT/polyblur_amd/lib/obj; OUT=$ROOT/tools/_abl; mkdir -p "$OUT/obj_$NAME"
OBJS=""
for o in "$OBJ"/*.o; do
  b=$(basename "$o" .o); use=$o
  for s in "$@"; do
    if [ "$(basename "$s" .hip)" = "$b" ]; then
      use=$OUT/obj_$NAME/$b.o
      /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $FLAGS -c "$ROOT/polyblur_amd/csrc/$b.hip" -o "$use" &
    fi
  done
  OBJS="$OBJS $use"
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o "$OUT/lib_$NAME.so" $OBJS -ldl
echo "$OUT/lib_$NAME.so"
