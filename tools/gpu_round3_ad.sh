#!/bin/bash
# Round 3, GPU call AD: the batch's arrays sized once for all files (pp_dev_ingest_expect) -- tokenizer / CLI tests, stage timers of both layouts.
set -u
cd "$(dirname "$0")/.."
O=gpurun_out
mkdir -p $O /tmp/e2e
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "window_grouped or tokenizer or cli or configs0 or file_cases or files" > $O/r3ad_tests.log 2>&1; echo "tests rc=$?" >> $O/r3ad_tests.log
tail -3 $O/r3ad_tests.log
timeout 600 python bench.py --e2e-only --e2e-dir /tmp/e2e > $O/r3ad_e2e.json 2> $O/r3ad_e2e.err
FA=/tmp/e2e/asm.fasta; S1=/tmp/e2e/reads_1.sam; S2=/tmp/e2e/reads_2.sam
for lay in file window; do
  for rep in 1 2 3; do
    PP_TIMING=1 PP_SEQ_LAYOUT=$lay bin/polypolish polish $FA $S1 $S2 2> $O/r3ad_timing_${lay}_$rep.txt > /dev/null
  done
  echo "== $lay"; grep -E "tokenizer|uploaded \+" $O/r3ad_timing_${lay}_3.txt | head -30
done
python - <<'P'
import json
d=json.load(open('gpurun_out/r3ad_e2e.json')); e=d.get('e2e', d)
print({k:(v.get('wall_s'), v.get('parity'), v.get('tokenizer_extra_ms'), v.get('tokenizer_mirror_ms')) for k,v in e.items() if isinstance(v,dict) and 'wall_s' in v})
P
