#!/bin/bash
# Round profile: bench line, rocprofv3 kernel stats, PMC HBM traffic (separate passes).  Run on the GPU box
# from the repo root: bash scripts/gpu_profile.sh r1   (writes under gpurun_out/prof_r1/)
set -u
tag=${1:-r4}
o=gpurun_out/prof_$tag
mkdir -p $o
export TMPDIR=/tmp
python bench.py > $o/bench.json 2> $o/bench.err
tail -c 600 $o/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats -- python3 bench.py --no-cpu --no-host-learn --no-weighted > $o/stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats_pass -- python3 bench.py --no-cpu --no-learn --no-host-learn --no-weighted --no-f64 --no-i8x --no-sparse-theta > $o/stats_pass.log 2>&1
f=$(find $o/stats_pass -name "*kernel_stats.csv" | head -1); cp "$f" $o/kernel_stats_pass_only.csv
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $o/pmc_fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-learn --no-host-learn --no-weighted --no-sparse-theta > $o/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $o/pmc_write -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-learn --no-host-learn --no-weighted --no-sparse-theta > $o/pmc_write.log 2>&1
python scripts/pmc_summarize.py $o/pmc_traffic.json $o/pmc_fetch $o/pmc_write
f=$(find $o/stats -name "*kernel_stats.csv" | head -1); cp "$f" $o/kernel_stats.csv; head -12 $o/kernel_stats.csv
find $o -name "*.csv" -size +2M -delete; find $o -name "*.db" -delete
