#!/bin/bash
# Round 3, GPU call V: the 4-bit plain class -- bytes moved (PMC passes) and times, configs[1] and [4].
set -u
cd "$(dirname "$0")/.."
O=gpurun_out
mkdir -p $O
B="--steps 20 --warmup 5 --no-e2e --no-cpu-baseline --no-second-layout"
for v in on off; do
  timeout 400 python bench.py --seq4 $v $B > $O/r3v_c1_$v.json 2> $O/r3v_c1_$v.err
done
timeout 300 python bench.py --config 4 --seq4 on $B --no-live-traffic > $O/r3v_c4_on.json 2> $O/r3v_c4_on.err
timeout 300 python bench.py --config 4 --seq4 off $B --no-live-traffic > $O/r3v_c4_off.json 2> $O/r3v_c4_off.err
for f in $O/r3v_c*.json; do python - "$f" <<'P'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d['roofline']; print(sys.argv[1].split('/')[-1], d['ms_per_step'], r['kernel_ms'], r['frac'], 'traffic', r['traffic'], r.get('hbm_actual'), d['kernel_ms_per_step'])
except Exception as e: print(sys.argv[1], 'ERR', e); print(open(sys.argv[1].replace('.json','.err')).read()[-600:])
P
done
