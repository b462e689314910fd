#!/bin/bash
# Build a variant of libgml_hip.so for interleaved A/B runs: scripts/build_variant.sh TAG "-DFLAG ..." [file.hip|file.cpp ...]
# Recompiles the listed source files (default: gml_kernels_i8w.hip) with the extra flags and links them with the objects of
# the current build (the Makefile's OBJS) into gpurun_ab/libgml_TAG.so (git-ignored, travels to the GPU box).
set -e
TAG=$1; FLAGS=$2; shift 2 || true
FILES=${@:-gml_kernels_i8w.hip}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CS=$ROOT/graphicalmodellearning.jl_amd/csrc
mkdir -p $ROOT/gpurun_ab /tmp/variant_$TAG
ALL=$(sed -n 's/^OBJS = //p' $CS/Makefile)
[ -n "$ALL" ] || { echo "no OBJS line in $CS/Makefile"; exit 1; }
OBJS=""
for o in $ALL; do
  src=${o%.o}.hip; x=""
  [ "$o" = gml_solver_host.o ] && src=gml_solver.cpp
  [ -f $CS/$src ] || src=${o%.o}.cpp
  [[ $src == *.cpp ]] && x="-x hip"
  if [[ " $FILES " == *" $src "* ]]; then
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -fno-slp-vectorize $FLAGS $x -c $CS/$src -o /tmp/variant_$TAG/$o
    OBJS="$OBJS /tmp/variant_$TAG/$o"
  else
    OBJS="$OBJS $CS/$o"
  fi
done
# (an object missing from the list fails here, not at the first call on the GPU box)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -Wl,--no-undefined -o $ROOT/gpurun_ab/libgml_$TAG.so $OBJS -lpthread -ldl
echo built gpurun_ab/libgml_$TAG.so
