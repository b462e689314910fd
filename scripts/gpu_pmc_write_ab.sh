#!/bin/bash
# WRITE_SIZE / FETCH_SIZE of the i8w forward kernel for the tree's build and a variant: gpu_pmc_write_ab.sh TAG
export TMPDIR=/tmp
tag=$1; o=gpurun_out/pmc_ab_$tag; rm -rf $o; mkdir -p $o
for which in tree $tag; do
  [ $which = tree ] && unset GML_LIB_OVERRIDE || export GML_LIB_OVERRIDE=gpurun_ab/libgml_$tag.so
  for c in WRITE_SIZE FETCH_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $o/${which}_$c -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-learn --no-host-learn --no-weighted --no-f64 --no-i8x > $o/${which}_$c.log 2>&1
  done
  python3 scripts/pmc_summarize.py $o/$which.json $o/${which}_WRITE_SIZE $o/${which}_FETCH_SIZE | grep -A3 "fwd_i8w"
done
find $o -name "*.csv" -delete; find $o -name "*.db" -delete
