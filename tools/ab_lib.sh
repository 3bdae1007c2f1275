#!/bin/bash
# A/B of two builds of the library inside one box:  bash tools/ab_lib.sh [rounds]   (tools/ab/lib_old.so, lib_new.so)
set -e
R=${1:-2}
L=multimodal-sentiment-analysis_amd/mmsa_amd/libmmsa_hip.so
for i in $(seq $R); do
  for v in old new; do
    cp tools/ab/lib_$v.so $L
    timeout -k 10 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --profile-steps 0 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'], d['final_loss'])"
  done
done
cp tools/ab/lib_new.so $L
