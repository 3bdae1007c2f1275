#!/bin/bash
# A second build of the library with extra -D flags on one source, for in-box A/B and timing aids:
#   bash tools/build_lab_lib.sh head.hip tools/ab/lib_stamps.so -DMMSA_LAB_HEAD_STAMPS
# (the other objects come from the last regular build under csrc/build)
set -e
SRC=$1; OUT=$2; shift 2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
C=$ROOT/multimodal-sentiment-analysis_amd/csrc
mkdir -p "$(dirname "$OUT")" /tmp/mmsa_lab
hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -fvisibility=hidden "$@" -c $C/$SRC -o /tmp/mmsa_lab/lab.o
OBJS=$(ls $C/build/*.o | grep -v "/${SRC%.hip}.o")
(cd /tmp/mmsa_lab && hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/mmsa_lab/lab.o -o /tmp/mmsa_lab/lib.so)
cp /tmp/mmsa_lab/lib.so "$OUT"
