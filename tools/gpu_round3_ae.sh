#!/bin/bash
# Round 3, GPU call AE: wide4_pass with the assembly's words carried from chunk to chunk -- parity subset, then entry 3.
set -u
cd "$(dirname "$0")/.."
O=gpurun_out
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "mirror or one_indel or record_cases or window_grouped or fuzz" > $O/r3ae_tests.log 2>&1; echo "tests rc=$?" >> $O/r3ae_tests.log
tail -3 $O/r3ae_tests.log
B="--steps 30 --warmup 5 --no-e2e --no-cpu-baseline --no-live-traffic --no-second-layout --seq-layout window --seq4 on"
for c in 1 4; do timeout 300 python bench.py --config $c $B > $O/r3ae_c${c}.json 2> $O/r3ae_c${c}.err; done
for f in $O/r3ae_c*.json; do python - "$f" <<'P'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], 'rec', d['planted_errors_recovered'])
except Exception as e: print(sys.argv[1], 'ERR', e); print(open(sys.argv[1].replace('.json','.err')).read()[-600:])
P
done
