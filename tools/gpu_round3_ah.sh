#!/bin/bash
# Round 3, GPU call AH (the budget's last minutes): the continuity lines and the per-rank cost with SEQ records on 32-byte boundaries.
set -u
cd "$(dirname "$0")/.."
O=gpurun_out
mkdir -p $O
B="--steps 20 --warmup 5 --no-e2e --no-cpu-baseline --no-live-traffic --no-second-layout"
timeout 100 python bench.py $B --recipe subs > $O/r3ah_subs.json 2> $O/r3ah_subs.err
timeout 100 python bench.py $B --nd-frac 0.01 > $O/r3ah_nd.json 2> $O/r3ah_nd.err
timeout 100 python bench.py $B --seq-pitch 0 > $O/r3ah_packed.json 2> $O/r3ah_packed.err
for f in subs nd packed; do python - $O/r3ah_$f.json <<'P'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['kernel_ms_per_step'])
except Exception as e: print(sys.argv[1], 'ERR', e)
P
done
timeout 120 python tools/exp_rank_share.py 4 8 > $O/r3ah_rank_share_c4.txt 2>&1; tail -1 $O/r3ah_rank_share_c4.txt
timeout 100 python tools/exp_rank_share.py 3 8 > $O/r3ah_rank_share_c3.txt 2>&1; tail -1 $O/r3ah_rank_share_c3.txt
