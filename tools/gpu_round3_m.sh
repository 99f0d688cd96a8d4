#!/bin/bash
# Round 3, GPU call M: where a k_tile block's time goes at 50x and at 200x (stamps build).
set -u
cd "$(dirname "$0")/.."
O=gpurun_out
mkdir -p $O
export PP_LIB_PATH=$PWD/polypolish_amd/_build/var_stamps/libpolypolish_hip.so
for c in 3 1; do
  PP_TILE_STAMPS_FILE=/tmp/st_$c.bin timeout 300 python bench.py --config $c --no-e2e --no-cpu-baseline --no-live-traffic --no-second-layout --steps 1 --warmup 1 > $O/r3m_c$c.json 2> $O/r3m_c$c.err
  python tools/exp_tile_stamps.py /tmp/st_$c.bin > $O/r3m_stamps_c$c.txt 2>&1
  grep -E "ordinary blocks|end of a block|kernel span|in flight" $O/r3m_stamps_c$c.txt
done
