#!/bin/bash
# Round 3, GPU call J: the round's reference run (after the two-pass vote) -- suite, profile of the default bench (kernel trace + HBM counter
# passes + the full bench line), the bench lines of configs[2], [3], [4] with their end-to-end legs, per-rank cost.
set -u
cd "$(dirname "$0")/.."
O=gpurun_out
mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q > $O/r3j_tests.log 2>&1; echo "tests rc=$?" >> $O/r3j_tests.log
tail -4 $O/r3j_tests.log
timeout 900 bash tools/profile_round.sh r3_final > $O/r3j_profile.log 2>&1
timeout 400 python bench.py --config 2 --steps 20 --warmup 5 > $O/r3j_bench_c2.json 2> $O/r3j_bench_c2.err; echo "c2 rc=$?"
D=/tmp/pp_e2e_big; mkdir -p $D
timeout 600 python bench.py --config 3 --steps 20 --warmup 5 --e2e-dir $D > $O/r3j_bench_c3.json 2> $O/r3j_bench_c3.err; echo "c3 rc=$?"
rm -rf $D; mkdir -p $D
timeout 300 python bench.py --config 4 --steps 20 --warmup 5 --no-e2e > $O/r3j_bench_c4.json 2> $O/r3j_bench_c4.err; echo "c4 rc=$?"
timeout 900 python bench.py --config 4 --e2e-only --e2e-dir $D > $O/r3j_e2e_c4.json 2> $O/r3j_e2e_c4.err; echo "c4 e2e rc=$?"
rm -rf $D
timeout 300 python bench.py --steps 20 --warmup 5 --no-e2e --no-cpu-baseline --no-live-traffic --nd-frac 0.01 > $O/r3j_bench_nd.json 2> $O/r3j_bench_nd.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-e2e --no-cpu-baseline --no-live-traffic --recipe subs > $O/r3j_bench_subs.json 2> $O/r3j_bench_subs.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-e2e --no-cpu-baseline --no-live-traffic --indel-frac 0.01 > $O/r3j_bench_indel1pct.json 2> $O/r3j_bench_indel1pct.err
timeout 400 python tools/exp_rank_share.py 3 8 > $O/r3j_rank_share_c3.txt 2>&1
timeout 400 python tools/exp_rank_share.py 4 8 > $O/r3j_rank_share_c4.txt 2>&1
tail -1 $O/r3j_rank_share_c3.txt; tail -1 $O/r3j_rank_share_c4.txt
