#!/bin/bash
# Round 3, GPU call D: one-indel reads as three work items -- suite + benches.
set -u
cd "$(dirname "$0")/.."
O=gpurun_out
mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q -x > $O/r3d_tests.log 2>&1; echo "tests rc=$?" >> $O/r3d_tests.log
tail -30 $O/r3d_tests.log
for c in 1 2 3 4; do
  timeout 300 python bench.py --config $c --steps 20 --warmup 5 --no-e2e --no-live-traffic --no-cpu-baseline > $O/r3d_bench_c$c.json 2> $O/r3d_bench_c$c.err; echo "c$c rc=$?"
done
timeout 300 python bench.py --steps 20 --warmup 5 --no-e2e --no-cpu-baseline --no-live-traffic --indel-frac 0.01 > $O/r3d_bench_indel1pct.json 2> $O/r3d_bench_indel1pct.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-e2e --no-cpu-baseline --no-live-traffic --recipe subs > $O/r3d_bench_subs.json 2> $O/r3d_bench_subs.err
timeout 400 python tools/exp_rank_share.py 3 8 > $O/r3d_rank_share_c3.txt 2>&1
