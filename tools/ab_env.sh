#!/bin/bash
# A/B of one environment switch inside one box:  bash tools/ab_env.sh NAME VALUE_A VALUE_B [rounds]
set -e
N=$1; A=$2; B=$3; R=${4:-2}
for i in $(seq $R); do
  for v in $A $B; do
    env $N=$v timeout -k 10 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --profile-steps 0 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$N=$v', d['value'], d['ms_per_step'], d['final_loss'])"
  done
done
