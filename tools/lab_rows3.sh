#!/bin/bash
# A/B of the strip-sharing implicit GEMM (MMSA_IGEMM3=2: every eligible launch) against the general kernel (=0) per layer geometry
# (developer aid; needs tools/igemm_lab built).  Usage: tools/lab_rows3.sh [batch]
L=tools/igemm_lab
B=${1:-32}
for geo in "$B 32 65 32 64 9 stats" "$B 32 65 64 64 9 stats" "$B 16 32 64 128 9 stats" "$B 16 32 128 128 9 stats" \
           "$B 8 16 128 256 9 stats" "$B 8 16 256 256 9 stats" "$B 32 65 64 64 9" "$B 32 65 64 32 9" \
           "$B 16 32 128 128 9" "$B 16 32 128 64 9" "$B 8 16 256 256 9" "$B 8 16 256 128 9"; do
  for v in 0 1; do
    echo "== MMSA_IGEMM3=$((2*v))  $geo"
    MMSA_IGEMM3=$((2*v)) timeout -k 10 60 $L $geo || exit 1
  done
done
