#!/bin/bash
# Weight-gradient slab kernel alone, per layer geometry at batch B (default 32), without and with the shortcut.
#   tools/lab_wgrad2.sh [batch]      (needs tools/wgrad2_lab built)
B=${1:-32}
for geo in "$B 32 65 64 64 0" "$B 32 65 32 64 1" "$B 32 65 32 64 0" "$B 16 32 128 128 0" "$B 16 32 64 128 1" "$B 16 32 64 128 0" \
           "$B 8 16 256 256 0" "$B 8 16 128 256 1" "$B 8 16 128 256 0"; do
  timeout -k 10 60 tools/wgrad2_lab $geo || exit 1
done
