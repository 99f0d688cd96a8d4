#!/bin/bash
# Round 3, GPU call P: persistent k_tile with NO claims (static round robin) against the build before: what is there to gain?
set -u
cd "$(dirname "$0")/.."
O=gpurun_out
mkdir -p $O
B="--steps 20 --warmup 5 --no-e2e --no-cpu-baseline --no-live-traffic --no-second-layout"
for c in 1 4 3 2; do
  for v in base static; do
    export PP_LIB_PATH=$PWD/polypolish_amd/_build/var_$v/libpolypolish_hip.so
    timeout 300 python bench.py --config $c $B > $O/r3p_c${c}_$v.json 2> $O/r3p_c${c}_$v.err
  done
done
for f in $O/r3p_c*.json; do python - "$f" <<'P'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['kernel_ms_per_step'], 'rec', d['planted_errors_recovered'])
except Exception as e: print(sys.argv[1], 'ERR', e)
P
done
export PP_LIB_PATH=$PWD/polypolish_amd/_build/var_ststamps/libpolypolish_hip.so
PP_TILE_STAMPS_FILE=/tmp/st_3.bin timeout 300 python bench.py --config 3 --no-e2e --no-cpu-baseline --no-live-traffic --no-second-layout --steps 1 --warmup 1 > $O/r3p_st.json 2> $O/r3p_st.err
python tools/exp_tile_stamps.py /tmp/st_3.bin > $O/r3p_stamps_c3.txt 2>&1
grep -E "ordinary blocks: p|end of a block|kernel span" $O/r3p_stamps_c3.txt
