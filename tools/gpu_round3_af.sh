#!/bin/bash
# Round 3, GPU call AF: do 32-byte aligned reads (pitch 160) buy k_tile anything in file order?
set -u
cd "$(dirname "$0")/.."
O=gpurun_out
mkdir -p $O
B="--steps 30 --warmup 5 --no-e2e --no-cpu-baseline --no-live-traffic --no-second-layout"
for p in 0 160 192; do timeout 300 python bench.py --seq-pitch $p $B > $O/r3af_p$p.json 2> $O/r3af_p$p.err; done
timeout 300 python bench.py --config 4 --seq-pitch 160 $B > $O/r3af_c4_p160.json 2> $O/r3af_c4_p160.err
for f in $O/r3af_*.json; do python - "$f" <<'P'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], 'rec', d['planted_errors_recovered'])
except Exception as e: print(sys.argv[1], 'ERR', e); print(open(sys.argv[1].replace('.json','.err')).read()[-600:])
P
done
