#!/bin/bash
# Interleaved comparison of environment settings on one box:  bash tools/ab_bench.sh ROUNDS "A=1" "MMSA_X=0" ...
# prints the median-block ms/step of every run (bench.py --blocks 3, no instrumented steps, no CPU baseline).
set -e
N=$1; shift
for i in $(seq $N); do
  for cfg in "$@"; do
    env $cfg timeout -k 10 120 python bench.py --steps 50 --warmup 10 --blocks 3 --profile-steps 0 --no-cpu-baseline > /tmp/ab.json 2> /tmp/ab.err
    python3 -c "import json;d=json.load(open('/tmp/ab.json'));print('$cfg', d['ms_per_step'], d['block_ms_per_step'])"
  done
done
