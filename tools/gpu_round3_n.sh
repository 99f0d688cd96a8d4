#!/bin/bash
# Round 3, GPU call N: k_tile as persistent workgroups (A/B: the build before | persistent with MachineLICM | persistent without).
set -u
cd "$(dirname "$0")/.."
O=gpurun_out
mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q -x > $O/r3n_tests.log 2>&1; echo "tests rc=$?" >> $O/r3n_tests.log
tail -8 $O/r3n_tests.log
B="--steps 20 --warmup 5 --no-e2e --no-cpu-baseline --no-live-traffic --no-second-layout"
for c in 1 4 3 2; do
  for v in base licm new; do
    if [ $v = new ]; then unset PP_LIB_PATH; else export PP_LIB_PATH=$PWD/polypolish_amd/_build/var_$v/libpolypolish_hip.so; fi
    timeout 300 python bench.py --config $c $B > $O/r3n_c${c}_$v.json 2> $O/r3n_c${c}_$v.err
  done
done
unset PP_LIB_PATH
for f in $O/r3n_c*.json; do python - "$f" <<'P'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['kernel_ms_per_step'], 'rec', d['planted_errors_recovered'])
except Exception as e: print(sys.argv[1], 'ERR', e)
P
done
timeout 300 python tools/exp_rank_share.py 4 8 > $O/r3n_share_c4.txt 2>&1; tail -1 $O/r3n_share_c4.txt
timeout 300 python tools/exp_rank_share.py 3 8 > $O/r3n_share_c3.txt 2>&1; tail -1 $O/r3n_share_c3.txt
export PP_LIB_PATH=$PWD/polypolish_amd/_build/var_stamps/libpolypolish_hip.so
PP_TILE_STAMPS_FILE=/tmp/st_3.bin timeout 300 python bench.py --config 3 --no-e2e --no-cpu-baseline --no-live-traffic --no-second-layout --steps 1 --warmup 1 > $O/r3n_st.json 2> $O/r3n_st.err
python tools/exp_tile_stamps.py /tmp/st_3.bin > $O/r3n_stamps_c3.txt 2>&1
grep -E "ordinary blocks: p|end of a block|kernel span" $O/r3n_stamps_c3.txt
