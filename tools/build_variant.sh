#!/bin/bash
# Build a variant of libsininn.so with extra compiler flags for ONE source file, next to the default build:
#   tools/build_variant.sh <name> <source.hip> "<flags>"      ->  build/variants/libsininn_<name>.so
# A/B on one box:  SININN_LIB=build/variants/libsininn_<name>.so python tools/bench_pair.py   (build/ is git-ignored but travels
# to the GPU box with the snapshot).
set -e
NAME=$1; SRC=$2; FLAGS=$3
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CS=$ROOT/sin-inn_amd/csrc
OUT=$ROOT/build/variants
mkdir -p "$OUT"
make -C "$CS" -j8 > /dev/null
OBJ=$OUT/${NAME}_$(basename "${SRC%.*}").o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $FLAGS -c "$CS/$SRC" -o "$OBJ"
OBJS=""
for o in $(sed -n 's/^OBJS = //p' "$CS/Makefile"); do
  if [ "$o" = "$(basename "${SRC%.*}").o" ]; then OBJS="$OBJS $OBJ"; else OBJS="$OBJS $CS/$o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o "$OUT/libsininn_$NAME.so"
echo "$OUT/libsininn_$NAME.so"
