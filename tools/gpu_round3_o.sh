#!/bin/bash
# Round 3, GPU call O: persistent k_tile, arguments copied per phase, claim sent at the vote (A/B: before | with MachineLICM | without).
set -u
cd "$(dirname "$0")/.."
O=gpurun_out
mkdir -p $O
B="--steps 20 --warmup 5 --no-e2e --no-cpu-baseline --no-live-traffic --no-second-layout"
for c in 1 4 3; do
  for v in base licm new; do
    if [ $v = new ]; then unset PP_LIB_PATH; else export PP_LIB_PATH=$PWD/polypolish_amd/_build/var_$v/libpolypolish_hip.so; fi
    timeout 300 python bench.py --config $c $B > $O/r3o_c${c}_$v.json 2> $O/r3o_c${c}_$v.err
  done
done
unset PP_LIB_PATH
for f in $O/r3o_c*.json; do python - "$f" <<'P'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['kernel_ms_per_step'], 'rec', d['planted_errors_recovered'])
except Exception as e: print(sys.argv[1], 'ERR', e)
P
done
export PP_LIB_PATH=$PWD/polypolish_amd/_build/var_stamps/libpolypolish_hip.so
PP_TILE_STAMPS_FILE=/tmp/st_3.bin timeout 300 python bench.py --config 3 --no-e2e --no-cpu-baseline --no-live-traffic --no-second-layout --steps 1 --warmup 1 > $O/r3o_st.json 2> $O/r3o_st.err
python tools/exp_tile_stamps.py /tmp/st_3.bin > $O/r3o_stamps_c3.txt 2>&1
grep -E "ordinary blocks: p|end of a block|kernel span" $O/r3o_stamps_c3.txt
unset PP_LIB_PATH
timeout 1200 python -m pytest tests -m gpu -q -x > $O/r3o_tests.log 2>&1; echo "tests rc=$?" >> $O/r3o_tests.log
tail -3 $O/r3o_tests.log
