#!/bin/bash
# Upper bound on what fewer multiplications could buy the 3x3 convolutions at batch 32: the strip-sharing kernel as
# shipped against a build that leaves out the matrix instructions of two of every three taps (tools/igemm_lab_third:
# hipcc -DMMSA_LAB_SKIP_TAPS=1) while issuing every load, fragment read and barrier.  Winograd F(2x2, 3x3) removes
# 5/9 of the multiplications (less than the 2/3 removed here), needs 16/9 of the weight bytes and adds two transforms.
# tools/igemm_lab_none (-DMMSA_LAB_SKIP_TAPS=2) issues no matrix instruction at all: the time of operand delivery alone.
B=${1:-32}
for geo in "$B 32 65 64 64 9 stats" "$B 16 32 128 128 9 stats" "$B 8 16 256 256 9 stats"; do
  for bin in tools/igemm_lab tools/igemm_lab_third tools/igemm_lab_none; do
    echo -n "$bin $geo: "; timeout -k 10 60 $bin $geo | grep -E "us|TF" | head -2 | tr '\n' ' '; echo
  done
done
