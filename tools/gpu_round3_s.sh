#!/bin/bash
# Round 3, GPU call S: the kernels' translation unit with and without MachineLICM (same source), all kernel groups.
set -u
cd "$(dirname "$0")/.."
O=gpurun_out
mkdir -p $O
B="--steps 30 --warmup 5 --no-e2e --no-cpu-baseline --no-live-traffic --no-second-layout"
for rep in 1 2; do
for c in 1 4 3 2; do
  for v in licm new; do
    if [ $v = new ]; then unset PP_LIB_PATH; else export PP_LIB_PATH=$PWD/polypolish_amd/_build/var_$v/libpolypolish_hip.so; fi
    timeout 300 python bench.py --config $c $B > $O/r3s_c${c}_${v}_$rep.json 2> $O/r3s_c${c}_${v}_$rep.err
  done
done
done
unset PP_LIB_PATH
for v in licm new; do
  if [ $v = new ]; then unset PP_LIB_PATH; else export PP_LIB_PATH=$PWD/polypolish_amd/_build/var_$v/libpolypolish_hip.so; fi
  timeout 300 python bench.py --nd-frac 0.01 $B > $O/r3s_nd_${v}_1.json 2> $O/r3s_nd_${v}.err
done
unset PP_LIB_PATH
for f in $O/r3s_*.json; do python - "$f" <<'P'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['kernel_ms_per_step'])
except Exception as e: print(sys.argv[1], 'ERR', e)
P
done
