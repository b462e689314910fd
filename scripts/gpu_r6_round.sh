#!/bin/bash
# Round-6 evidence: bench line, per-config records with the CPU learn() of C3 / C4-shard run in full and the config-5 front door.
# Run on the GPU box from the repo root: bash scripts/gpu_r6_round.sh   (writes under gpurun_out/)
set -u
mkdir -p gpurun_out/configs
python bench.py > gpurun_out/r6_bench.json 2> gpurun_out/r6_bench.err
tail -c 400 gpurun_out/r6_bench.json
python scripts/gpu_configs.py c1 c2 c3 c4 c5 c5d --round r6 --cpu-full --front-door > gpurun_out/r6_configs.log 2>&1
tail -12 gpurun_out/r6_configs.log
