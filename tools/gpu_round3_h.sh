#!/bin/bash
# Round 3, GPU call H: the tokenizer's window-grouped SEQ layout -- suite; the full default bench line (second roofline entry).
set -u
cd "$(dirname "$0")/.."
O=gpurun_out
mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q -x > $O/r3h_tests.log 2>&1; echo "tests rc=$?" >> $O/r3h_tests.log
tail -25 $O/r3h_tests.log
SECONDS=0
timeout 600 python bench.py --steps 20 --warmup 5 > $O/r3h_bench.json 2> $O/r3h_bench.err; echo "bench rc=$? in $SECONDS s"
python - <<'P'
import json
d=json.load(open('gpurun_out/r3h_bench.json'))
print({k:v for k,v in d.items() if k not in ('e2e','config')})
for k,v in d['e2e'].items(): print('e2e', k, v)
P
