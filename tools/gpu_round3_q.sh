#!/bin/bash
# Round 3, GPU call Q: k_tile's plain passes with the next pass's loads in flight (A/B: before | with MachineLICM | without).
set -u
cd "$(dirname "$0")/.."
O=gpurun_out
mkdir -p $O
B="--steps 20 --warmup 5 --no-e2e --no-cpu-baseline --no-live-traffic --no-second-layout"
for c in 1 4 3; do
  for v in base licm new; do
    if [ $v = new ]; then unset PP_LIB_PATH; else export PP_LIB_PATH=$PWD/polypolish_amd/_build/var_$v/libpolypolish_hip.so; fi
    timeout 300 python bench.py --config $c $B > $O/r3q_c${c}_$v.json 2> $O/r3q_c${c}_$v.err
  done
done
unset PP_LIB_PATH
timeout 300 python bench.py --seq-layout window $B > $O/r3q_c1w_new.json 2> $O/r3q_c1w_new.err
PP_LIB_PATH=$PWD/polypolish_amd/_build/var_base/libpolypolish_hip.so timeout 300 python bench.py --seq-layout window $B > $O/r3q_c1w_base.json 2> $O/r3q_c1w_base.err
for f in $O/r3q_c*.json; do python - "$f" <<'P'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['kernel_ms_per_step'], 'rec', d['planted_errors_recovered'])
except Exception as e: print(sys.argv[1], 'ERR', e)
P
done
timeout 1200 python -m pytest tests -m gpu -q -x > $O/r3q_tests.log 2>&1; echo "tests rc=$?" >> $O/r3q_tests.log
tail -3 $O/r3q_tests.log
