#!/bin/bash
# Collects, on the GPU box, what profiles/ holds for a round: the kernel-trace statistics of the bench command as it is
# timed (HIP-graph replay, no instrumented steps), the timeline of one replayed step, and the two TCC passes (reads,
# writes) for the HBM traffic per launch.  Each rocprofv3 pass runs on its own.
#   bash tools/collect_profiles.sh TAG  -> gpurun_out/TAG_kernel_stats.{csv,txt}, TAG_step_timeline.txt, TAG_pmc_traffic.json
set -e
TAG=${1:-r02}
ROOT=$(pwd)
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/prof_s /tmp/prof_t /tmp/prof_f /tmp/prof_w
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_s -o bench -- python3 $ROOT/bench.py --steps 50 --warmup 10 --blocks 2 --profile-steps 0 --no-cpu-baseline > $ROOT/gpurun_out/${TAG}_bench_under_profiler.json 2> /dev/null
DB=$(find /tmp/prof_s -name "bench_results.db" | head -1)
python3 $ROOT/tools/rocpd_stats.py $DB $ROOT/gpurun_out/${TAG}_kernel_stats.csv > $ROOT/gpurun_out/${TAG}_kernel_stats.txt
timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_t -o b -- python3 $ROOT/bench.py --steps 8 --warmup 4 --blocks 1 --profile-steps 0 --no-cpu-baseline > /dev/null 2>&1
T=$(find /tmp/prof_t -name "b_kernel_trace.csv" | head -1)
python3 $ROOT/tools/step_timeline.py $T > $ROOT/gpurun_out/${TAG}_step_timeline.txt
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/prof_f -o b -- python3 $ROOT/bench.py --steps 6 --warmup 3 --blocks 1 --no-cpu-baseline --no-graph --profile-steps 0 > /dev/null 2>&1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/prof_w -o b -- python3 $ROOT/bench.py --steps 6 --warmup 3 --blocks 1 --no-cpu-baseline --no-graph --profile-steps 0 > /dev/null 2>&1
F=$(find /tmp/prof_f -name "b_counter_collection.csv" | head -1)
W=$(find /tmp/prof_w -name "b_counter_collection.csv" | head -1)
python3 $ROOT/tools/pmc_traffic.py $F $W $ROOT/gpurun_out/${TAG}_pmc_traffic.json > /dev/null
head -14 $ROOT/gpurun_out/${TAG}_kernel_stats.txt
