#!/bin/bash
# Build a variant of libgml_hip.so for interleaved A/B runs: scripts/build_variant.sh TAG "-DFLAG ..." [file.hip ...]
# Recompiles the listed kernel files (default: gml_kernels_i8w.hip) with the extra flags and links them with the objects of
# the current build into gpurun_ab/libgml_TAG.so (git-ignored, travels to the GPU box).
set -e
TAG=$1; FLAGS=$2; shift 2 || true
FILES=${@:-gml_kernels_i8w.hip}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CS=$ROOT/graphicalmodellearning.jl_amd/csrc
mkdir -p $ROOT/gpurun_ab /tmp/variant_$TAG
OBJS=""
for o in gml_pack.o gml_alloc.o gml_host.o gml_ingest.o gml_sampled.o gml_operator.o gml_testhooks.o gml_solver_host.o gml_multi.o gml_kernels_f64.o gml_kernels_f64gemm.o gml_i8_pack.o gml_i8_fwd.o gml_i8_bwd.o gml_i8_hess.o gml_i8_pass.o gml_kernels_i8w.o gml_solver.o gml_sampler.o gml_dedupe.o; do
  src=${o%.o}.hip
  if [[ " $FILES " == *" $src "* ]]; then
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -fno-slp-vectorize $FLAGS -c $CS/$src -o /tmp/variant_$TAG/$o
    OBJS="$OBJS /tmp/variant_$TAG/$o"
  else
    OBJS="$OBJS $CS/$o"
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $ROOT/gpurun_ab/libgml_$TAG.so $OBJS -lpthread -ldl
echo built gpurun_ab/libgml_$TAG.so
