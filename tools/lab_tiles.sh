#!/bin/bash
# Implicit-GEMM tile / ring A/B at inference batch sizes (developer aid; needs tools/igemm_lab built).
L=tools/igemm_lab
for geo in "256 32 65 32 64 9" "256 32 65 64 64 9" "256 16 32 64 128 9" "256 16 32 128 128 9" "256 8 16 128 256 9" "256 8 16 256 256 9"; do
  for cfg in "MMSA_CONV_BM=64" "MMSA_CONV_BM=64 MMSA_IGEMM_RING3=0" "MMSA_CONV_BM=128 MMSA_IGEMM_RING128=5" "MMSA_CONV_BM=128 MMSA_IGEMM_RING128=3"; do
    echo "== $cfg"
    env $cfg timeout -k 10 60 $L $geo stats || exit 1
  done
done
