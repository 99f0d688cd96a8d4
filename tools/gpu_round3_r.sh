#!/bin/bash
# Round 3, GPU call R: per-kernel times of configs[4] and [3] (where does the bucketing's time go).
set -u
cd "$(dirname "$0")/.."
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
for c in 4 3; do
( cd /tmp && timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_c$c -- python $OLDPWD/bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline --no-e2e --no-live-traffic --no-second-layout > /dev/null 2> $OLDPWD/$O/r3r_trace_c$c.log )
python tools/prof_summary.py /tmp/p_c$c > $O/r3r_kernels_c$c.txt 2>&1
grep -E "^k_|kernel " $O/r3r_kernels_c$c.txt | head -24
done
