#!/bin/bash
# Run ON THE GPU BOX (through gpurun): rocprofv3 kernel trace + the two HBM counter passes of the
# default bench workload, plus the calibration copy; writes summaries under gpurun_out/prof_<tag>/.
#   tools/profile_round.sh <tag>
set -u
TAG=${1:-r1_final}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-e2e --no-live-traffic --no-second-layout"
cd /tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_trace -- $BENCH > "$OUT/bench_under_trace.json" 2> "$OUT/trace.log"
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/p_fetch -- $BENCH > /dev/null 2> "$OUT/fetch.log"
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/p_write -- $BENCH > /dev/null 2> "$OUT/write.log"
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/c_fetch -- python $ROOT/tools/pmc_calib.py > /dev/null 2>> "$OUT/fetch.log"
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/c_write -- python $ROOT/tools/pmc_calib.py > /dev/null 2>> "$OUT/write.log"
cd "$ROOT"
python tools/prof_summary.py /tmp/p_trace /tmp/p_fetch /tmp/p_write /tmp/c_fetch /tmp/c_write > "$OUT/rocprofv3_summary.txt"
python tools/prof_summary.py --traffic-json "$OUT/traffic.json" --bench /tmp/p_fetch /tmp/p_write --calib /tmp/c_fetch /tmp/c_write
find /tmp/p_trace -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \;
python bench.py > "$OUT/bench.json" 2> "$OUT/bench.log"
tail -c 1500 "$OUT/bench.json"
cat "$OUT/traffic.json"
