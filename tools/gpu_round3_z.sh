#!/bin/bash
# Round 3, GPU call Z: the whole suite with the 4-bit mirror / one-lane-per-read path in, then the layouts side by side.
set -u
cd "$(dirname "$0")/.."
O=gpurun_out
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x > $O/r3z_tests.log 2>&1; echo "tests rc=$?" >> $O/r3z_tests.log
tail -6 $O/r3z_tests.log
B="--steps 20 --warmup 5 --no-e2e --no-cpu-baseline --no-live-traffic --no-second-layout"
for c in 1 3 4; do
  timeout 300 python bench.py --config $c $B > $O/r3z_c${c}_file.json 2> $O/r3z_c${c}_file.err
  timeout 300 python bench.py --config $c --seq-layout window $B > $O/r3z_c${c}_win.json 2> $O/r3z_c${c}_win.err
  timeout 300 python bench.py --config $c --seq-layout window --seq4 on $B > $O/r3z_c${c}_win4.json 2> $O/r3z_c${c}_win4.err
done
timeout 300 python bench.py --config 2 --seq-layout window --seq4 on $B > $O/r3z_c2_win4.json 2> $O/r3z_c2_win4.err
for f in $O/r3z_c*.json; do python - "$f" <<'P'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['kernel_ms_per_step'], 'rec', d['planted_errors_recovered'])
except Exception as e: print(sys.argv[1], 'ERR', e); print(open(sys.argv[1].replace('.json','.err')).read()[-600:])
P
done
