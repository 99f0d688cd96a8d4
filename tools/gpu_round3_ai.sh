#!/bin/bash
# Round 3, GPU call AI: where a k_tile block's time goes at 50x in the final build -- file order and window-grouped + mirror.
set -u
cd "$(dirname "$0")/.."
O=gpurun_out
mkdir -p $O
export PP_LIB_PATH=$PWD/polypolish_amd/_build/var_stamps/libpolypolish_hip.so
for v in file win4; do
  if [ $v = file ]; then X=""; else X="--seq-layout window --seq4 on"; fi
  PP_TILE_STAMPS_FILE=/tmp/st_$v.bin timeout 100 python bench.py --config 3 --coverage 50 $X --no-e2e --no-cpu-baseline --no-live-traffic --no-second-layout --steps 1 --warmup 1 > $O/r3ai_$v.json 2> $O/r3ai_$v.err
  python tools/exp_tile_stamps.py /tmp/st_$v.bin > $O/r3ai_stamps_$v.txt 2>&1
  echo "== $v"; grep -E "ordinary blocks: p|end of a block|kernel span" $O/r3ai_stamps_$v.txt | cut -c1-200
done
