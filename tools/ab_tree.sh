#!/bin/bash
# Interleaved comparison of this tree with a second copy of it (another revision, built) on one box:
#   bash tools/ab_tree.sh ROUNDS OTHER_DIR    ->  median-block ms/step of every run
N=$1; OTHER=$2
run() { (cd "$1" && timeout -k 10 120 python bench.py --steps 50 --warmup 10 --blocks 3 --profile-steps 0 --no-cpu-baseline 2>/tmp/ab.err | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print('$2', d['ms_per_step'], d['block_ms_per_step'])") || exit 1; }
for i in $(seq $N); do
  run "$OTHER" other
  run . this
done
