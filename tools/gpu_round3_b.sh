#!/bin/bash
# Round 3, GPU call B: the GPU suite, kernel benches of every config with the survey recipe, the per-rank cost with
# partitioned records, the multi-context CLI at configs[3] size.
set -u
cd "$(dirname "$0")/.."
O=gpurun_out
mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q > $O/r3b_tests.log 2>&1; echo "tests rc=$?" >> $O/r3b_tests.log
for c in 1 2 3 4; do
  timeout 300 python bench.py --config $c --steps 20 --warmup 5 --no-e2e > $O/r3b_bench_c$c.json 2> $O/r3b_bench_c$c.err; echo "c$c rc=$?"
done
timeout 300 python bench.py --steps 20 --warmup 5 --no-e2e --no-cpu-baseline --nd-frac 0.01 > $O/r3b_bench_nd.json 2> $O/r3b_bench_nd.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-e2e --no-cpu-baseline --recipe subs > $O/r3b_bench_subs.json 2> $O/r3b_bench_subs.err
timeout 400 python tools/exp_rank_share.py 4 8 > $O/r3b_rank_share_c4.txt 2>&1
timeout 400 python tools/exp_rank_share.py 3 8 > $O/r3b_rank_share_c3.txt 2>&1
timeout 600 python tools/exp_multi_ctx.py 3 8 > $O/r3b_multi_c3.json 2> $O/r3b_multi_c3.err
tail -5 $O/r3b_tests.log
