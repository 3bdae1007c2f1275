#!/bin/bash
# Runs GPU test files one pytest process at a time; continues after ordinary test failures (exit 1) but stops at
# the first crash, abort, timeout or GPU fault (any other non-zero status).
mkdir -p gpurun_out
for f in "$@"; do
  name=$(basename "$f" .py)
  timeout -k 10 900 python -u -m pytest "$f" -v -m gpu --timeout 300 --timeout-method thread -p no:cacheprovider \
    > "gpurun_out/$name.log" 2>&1
  rc=$?
  echo "$f -> exit $rc"
  tail -n 25 "gpurun_out/$name.log"
  if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then echo "stopping: $f ended with status $rc"; exit $rc; fi
done
exit 0
