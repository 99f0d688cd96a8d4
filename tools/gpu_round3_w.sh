#!/bin/bash
# Round 3, GPU call W: the plain class over the 4-bit mirror -- parity first, then what it buys.
set -u
cd "$(dirname "$0")/.."
O=gpurun_out
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "mirror or one_indel or record_cases or tokenizer_equals or window_grouped or fuzz" > $O/r3w_tests1.log 2>&1; echo "tests rc=$?" >> $O/r3w_tests1.log
tail -12 $O/r3w_tests1.log
B="--steps 20 --warmup 5 --no-e2e --no-cpu-baseline --no-live-traffic --no-second-layout"
for c in 1 4; do
  for v in off on; do
    timeout 300 python bench.py --config $c --seq4 $v $B > $O/r3w_c${c}_$v.json 2> $O/r3w_c${c}_$v.err
  done
done
timeout 300 python bench.py --seq4 on --seq-layout window $B > $O/r3w_c1w_on.json 2> $O/r3w_c1w_on.err
for f in $O/r3w_c*.json; do python - "$f" <<'P'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['kernel_ms_per_step'], 'rec', d['planted_errors_recovered'])
except Exception as e: print(sys.argv[1], 'ERR', e); print(open(sys.argv[1].replace('.json','.err')).read()[-600:])
P
done
