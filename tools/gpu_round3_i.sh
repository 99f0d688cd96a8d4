#!/bin/bash
# Round 3, GPU call I: two-pass vote (clean positions skip the vote proper) -- suite + benches.
set -u
cd "$(dirname "$0")/.."
O=gpurun_out
mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q -x > $O/r3i_tests.log 2>&1; echo "tests rc=$?" >> $O/r3i_tests.log
tail -25 $O/r3i_tests.log
B="--steps 20 --warmup 5 --no-e2e --no-cpu-baseline --no-live-traffic"
for c in 1 2 3 4; do timeout 300 python bench.py --config $c $B > $O/r3i_c$c.json 2> $O/r3i_c$c.err; done
timeout 300 python bench.py $B --recipe subs --no-second-layout > $O/r3i_subs.json 2> $O/r3i_subs.err
timeout 300 python bench.py $B --nd-frac 0.01 --no-second-layout > $O/r3i_nd.json 2> $O/r3i_nd.err
for f in $O/r3i_c*.json $O/r3i_subs.json $O/r3i_nd.json; do python - "$f" <<'P'
import json,sys
try:
    d=json.load(open(sys.argv[1])); w=d.get('roofline_window_grouped_seq') or {}
    print(sys.argv[1].split('/')[-1], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['kernel_ms_per_step'], 'rec', d['planted_errors_recovered'], 'win', w.get('kernel_ms'), w.get('frac'), w.get('ms_per_step'))
except Exception as e: print(sys.argv[1], 'ERR', e)
P
done
