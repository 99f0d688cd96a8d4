#!/bin/bash
# Round 3, GPU call L: k_tile with its first-trip loads issued together (A/B against the build before), the one-indel edge
# cases, k_exact's early loads.
set -u
cd "$(dirname "$0")/.."
O=gpurun_out
mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q -x > $O/r3l_tests.log 2>&1; echo "tests rc=$?" >> $O/r3l_tests.log
tail -8 $O/r3l_tests.log
B="--steps 20 --warmup 5 --no-e2e --no-cpu-baseline --no-live-traffic --no-second-layout"
for c in 1 4 3 2; do
  for v in base new; do
    if [ $v = base ]; then export PP_LIB_PATH=$PWD/polypolish_amd/_build/var_base/libpolypolish_hip.so; else unset PP_LIB_PATH; fi
    timeout 300 python bench.py --config $c $B > $O/r3l_c${c}_$v.json 2> $O/r3l_c${c}_$v.err
  done
done
unset PP_LIB_PATH
for f in $O/r3l_c*.json; do python - "$f" <<'P'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['kernel_ms_per_step'], 'rec', d['planted_errors_recovered'])
except Exception as e: print(sys.argv[1], 'ERR', e)
P
done
timeout 300 python tools/exp_rank_share.py 4 8 > $O/r3l_share_c4.txt 2>&1; tail -1 $O/r3l_share_c4.txt
timeout 300 python tools/exp_rank_share.py 3 8 > $O/r3l_share_c3.txt 2>&1; tail -1 $O/r3l_share_c3.txt
