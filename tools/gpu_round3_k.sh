#!/bin/bash
# Round 3, GPU call K: k_exact with earlier loads -- suite, kernel trace, benches.
set -u
cd "$(dirname "$0")/.."
O=gpurun_out
mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q -x > $O/r3k_tests.log 2>&1; echo "tests rc=$?" >> $O/r3k_tests.log
tail -25 $O/r3k_tests.log
export TMPDIR=/tmp
( cd /tmp && timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_c1 -- python $OLDPWD/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-e2e --no-live-traffic --no-second-layout > /dev/null 2> $OLDPWD/$O/r3k_trace.log )
python tools/prof_summary.py /tmp/p_c1 2>/dev/null | grep -E "^k_|kernel " | head -16
B="--steps 20 --warmup 5 --no-e2e --no-cpu-baseline --no-live-traffic --no-second-layout"
for c in 1 3 4; do timeout 300 python bench.py --config $c $B > $O/r3k_c$c.json 2> $O/r3k_c$c.err; done
for f in $O/r3k_c*.json; do python - "$f" <<'P'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['kernel_ms_per_step'], 'rec', d['planted_errors_recovered'])
except Exception as e: print(sys.argv[1], 'ERR', e)
P
done
