#!/bin/bash
# Round 3, GPU call AC: the tokenizer's stage timers with and without the window layout (configs[1] files).
set -u
cd "$(dirname "$0")/.."
O=gpurun_out
mkdir -p $O /tmp/e2e
timeout 600 python bench.py --e2e-only --e2e-dir /tmp/e2e > $O/r3ac_e2e.json 2> $O/r3ac_e2e.err
ls -la /tmp/e2e | head
FA=$(ls /tmp/e2e/*.fasta | head -1); S1=$(ls /tmp/e2e/*_1.sam | head -1); S2=$(ls /tmp/e2e/*_2.sam | head -1)
for lay in file window; do
  for rep in 1 2; do
    PP_TIMING=1 PP_SEQ_LAYOUT=$lay bin/polypolish polish $FA $S1 $S2 2> $O/r3ac_timing_${lay}_$rep.txt > /dev/null
  done
  echo "== $lay"; grep -E "tokenizer|total|polish" $O/r3ac_timing_${lay}_2.txt | head -30
done
