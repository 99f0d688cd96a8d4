#!/bin/bash
# Round 3, GPU call G: compact runs of tiled contigs, fewer stream operations -- suite, per-rank cost; k_fill pass width;
# the window-grouped SEQ layout as a second roofline entry.
set -u
cd "$(dirname "$0")/.."
O=gpurun_out
mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q -x > $O/r3g_tests.log 2>&1; echo "tests rc=$?" >> $O/r3g_tests.log
tail -25 $O/r3g_tests.log
B="--steps 20 --warmup 5 --no-e2e --no-cpu-baseline --no-live-traffic"
timeout 300 python bench.py $B > $O/r3g_c1_base.json 2> $O/r3g_c1_base.err
for fr in 1024 512 256 128; do PP_FILL_RANGE=$fr timeout 300 python bench.py $B > $O/r3g_c1_fr$fr.json 2> $O/r3g_c1_fr$fr.err; done
timeout 300 python bench.py --config 4 $B > $O/r3g_c4_base.json 2> $O/r3g_c4_base.err
for fr in 512 256; do PP_FILL_RANGE=$fr timeout 300 python bench.py --config 4 $B > $O/r3g_c4_fr$fr.json 2> $O/r3g_c4_fr$fr.err; done
timeout 400 python bench.py --steps 20 --warmup 5 --no-e2e --no-cpu-baseline --seq-layout window > $O/r3g_c1_window_layout.json 2> $O/r3g_c1_window_layout.err
timeout 400 python tools/exp_rank_share.py 3 8 > $O/r3g_rank_share_c3.txt 2>&1
timeout 400 python tools/exp_rank_share.py 4 8 > $O/r3g_rank_share_c4.txt 2>&1
tail -1 $O/r3g_rank_share_c3.txt; tail -1 $O/r3g_rank_share_c4.txt
for f in $O/r3g_c1_*.json $O/r3g_c4_*.json; do python - "$f" <<'P'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline'].get('traffic'), d['kernel_ms_per_step'])
except Exception as e: print(sys.argv[1], 'ERR', e)
P
done
