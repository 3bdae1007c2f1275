#!/bin/bash
# Prices the pipeline stages of the implicit GEMM: builds of tools/igemm_lab.hip with -DMMSA_LAB_ABLATE=bits
# (1 no global->LDS requests in the main loop, 4 no barriers, 8 no LDS fragment reads, 16 requests issue but fetch nothing)
for geo in "32 8 16 256 256 9" "32 16 32 128 128 9" "32 32 65 64 64 9"; do
  echo "== $geo"
  for a in "" _a1 _a16 _a8 _a4 _a5 _a17; do
    echo -n "ablate${a:-_none}: "; timeout -k 10 60 tools/igemm_lab$a $geo stats | grep -E "^workgroups" || exit 1
  done
done
