#!/bin/bash
# Run ON THE GPU BOX: kernel trace of the bench's steps -> tools/exp_gaps.py (idle time between the kernels of a job, turnaround between jobs)
ROOT=$(pwd)
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/gt
timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/gt -- python $ROOT/bench.py --steps 40 --no-e2e --no-cpu-baseline --no-live-traffic --no-second-layout "$@" > /tmp/gt_bench.json 2> /tmp/gt_err.log
cd "$ROOT"
tail -c 300 /tmp/gt_bench.json; echo
find /tmp/gt -name "*kernel_trace.csv" | head -3
python tools/exp_gaps.py /tmp/gt
