#!/bin/bash
# usage: tools/ab_env.sh "<bench args>" "VAR=a VAR2=b" "VAR=c" ...   -- interleaved A/B of environment settings, 2 rounds
args="$1"; shift
for round in 1 2; do
  for cfg in "$@"; do
    env $cfg timeout -k 10 200 python bench.py $args --no-cpu-baseline 2>/dev/null > gpurun_out/ab.json && python -c "
import json; d=json.load(open('gpurun_out/ab.json')); print('%-40s %8.3f ms/step  spmv %7.1f us  frac %.3f' % ('$cfg', d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['frac']))"
  done
done
