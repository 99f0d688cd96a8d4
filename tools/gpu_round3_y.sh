#!/bin/bash
# Round 3, GPU call Y: block timeline of k_tile with a whole read per lane, file order against window order.
set -u
cd "$(dirname "$0")/.."
O=gpurun_out
mkdir -p $O
export PP_LIB_PATH=$PWD/polypolish_amd/_build/var_stamps/libpolypolish_hip.so
for lay in file window; do
  PP_TILE_STAMPS_FILE=/tmp/st_$lay.bin timeout 300 python bench.py --seq4 on --seq-layout $lay --no-e2e --no-cpu-baseline --no-live-traffic --no-second-layout --steps 1 --warmup 1 > $O/r3y_$lay.json 2> $O/r3y_$lay.err
  python tools/exp_tile_stamps.py /tmp/st_$lay.bin > $O/r3y_stamps_$lay.txt 2>&1
  echo "== $lay"; grep -E "ordinary blocks|end of a block|kernel span" $O/r3y_stamps_$lay.txt | cut -c1-400
done
