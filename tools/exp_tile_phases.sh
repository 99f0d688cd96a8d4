#!/bin/bash
# Run ON THE GPU BOX: what every phase of k_tile's workgroup costs -- the kernel truncated behind phase K (builds var_stopK:
# `for k in 1 2 3 4 5; do make variant NAME=stop$k DEFS=-DPP_TILE_STOP=$k; done`; "0" = the library as built), its duration
# (kernel trace) and its instruction counters (one --pmc pass), per launch.  Differences between neighbours = the phases.
#   tools/exp_tile_phases.sh TAG [CONFIG]
set -u
TAG=${1:-phases}; CONFIG=${2:-1}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
for k in 1 2 3 4 5 0; do
  if [ "$k" = 0 ]; then unset PP_LIB_PATH; else export PP_LIB_PATH=$ROOT/polypolish_amd/_build/var_stop$k/libpolypolish_hip.so; fi
  rm -rf /tmp/ph_t /tmp/ph_c
  timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/ph_t -- python $ROOT/tools/exp_tile_phases.py $CONFIG 8 > "$OUT/run_$k.log" 2>&1
  timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d /tmp/ph_c -- python $ROOT/tools/exp_tile_phases.py $CONFIG 4 >> "$OUT/run_$k.log" 2>&1
  echo "==== stop after phase $k (0 = whole kernel)" >> "$OUT/phases_config$CONFIG.txt"
  python $ROOT/tools/prof_summary.py /tmp/ph_t /tmp/ph_c | grep -A9 "^k_tile\|^kernel \|k_tile" >> "$OUT/phases_config$CONFIG.txt"
done
cat "$OUT/phases_config$CONFIG.txt"
