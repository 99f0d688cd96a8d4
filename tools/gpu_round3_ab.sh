#!/bin/bash
# Round 3, GPU call AB: the window layout's counters in LDS -- tokenizer tests, then what the layout costs end to end.
set -u
cd "$(dirname "$0")/.."
O=gpurun_out
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "window_grouped or tokenizer" > $O/r3ab_tests.log 2>&1; echo "tests rc=$?" >> $O/r3ab_tests.log
tail -4 $O/r3ab_tests.log
timeout 600 python bench.py --e2e-only > $O/r3ab_e2e.json 2> $O/r3ab_e2e.err
python - <<'P'
import json
d=json.load(open('gpurun_out/r3ab_e2e.json'))
e=d.get('e2e', d)
print({k:v for k,v in e.items() if k in ('polish','polish_window_grouped_seq','parity')})
P
