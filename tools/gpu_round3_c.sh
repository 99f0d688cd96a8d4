#!/bin/bash
# Round 3, GPU call C: suite, profile of the default bench (kernel trace + HBM counter passes + live-traffic bench line),
# kernel benches of the other configs, the layout probe, per-rank cost, multi-context CLI at configs[4] size.
set -u
cd "$(dirname "$0")/.."
O=gpurun_out
mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q -x > $O/r3c_tests.log 2>&1; echo "tests rc=$?" >> $O/r3c_tests.log
tail -3 $O/r3c_tests.log
timeout 900 bash tools/profile_round.sh r3_mid > $O/r3c_profile.log 2>&1
for c in 2 3 4; do
  timeout 400 python bench.py --config $c --steps 20 --warmup 5 --no-e2e > $O/r3c_bench_c$c.json 2> $O/r3c_bench_c$c.err; echo "c$c rc=$?"
done
timeout 300 python bench.py --steps 20 --warmup 5 --no-e2e --no-cpu-baseline --no-live-traffic --indel-frac 0.01 > $O/r3c_bench_indel1pct.json 2> $O/r3c_bench_indel1pct.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-e2e --no-cpu-baseline --no-live-traffic --nd-frac 0.01 > $O/r3c_bench_nd.json 2> $O/r3c_bench_nd.err
timeout 120 tools/microbench/gather5 > $O/r3c_gather5.txt 2>&1
timeout 400 python tools/exp_rank_share.py 3 8 > $O/r3c_rank_share_c3.txt 2>&1
timeout 400 python tools/exp_rank_share.py 4 8 > $O/r3c_rank_share_c4.txt 2>&1
D=/dev/shm/pp_multi; mkdir -p $D
timeout 900 python tools/exp_multi_ctx.py 4 8 $D > $O/r3c_multi_c4.json 2> $O/r3c_multi_c4.err
rm -rf $D
cat $O/r3c_gather5.txt
