#!/bin/bash
# Round 3, GPU call E: compact runs of sharded jobs -- suite; kernel trace of configs[2]; per-rank cost.
set -u
cd "$(dirname "$0")/.."
O=gpurun_out
mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q -x > $O/r3e_tests.log 2>&1; echo "tests rc=$?" >> $O/r3e_tests.log
tail -30 $O/r3e_tests.log
export TMPDIR=/tmp
( cd /tmp && timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_c2 -- python $OLDPWD/bench.py --config 2 --steps 6 --warmup 2 --no-cpu-baseline --no-e2e --no-live-traffic > $OLDPWD/$O/r3e_c2_under_trace.json 2> $OLDPWD/$O/r3e_c2_trace.log )
python tools/prof_summary.py /tmp/p_c2 2>/dev/null | grep -E "^k_|kernel " | head -30 > $O/r3e_c2_kernels.txt
cat $O/r3e_c2_kernels.txt
timeout 300 python bench.py --config 2 --steps 20 --warmup 5 --no-e2e --no-live-traffic > $O/r3e_bench_c2.json 2> $O/r3e_bench_c2.err
timeout 300 python bench.py --config 2 --steps 20 --warmup 5 --no-e2e --no-live-traffic --no-cpu-baseline --indel-frac 0 > $O/r3e_bench_c2_noindel.json 2> $O/r3e_bench_c2_noindel.err
timeout 400 python tools/exp_rank_share.py 3 8 > $O/r3e_rank_share_c3.txt 2>&1
timeout 400 python tools/exp_rank_share.py 4 8 > $O/r3e_rank_share_c4.txt 2>&1
tail -1 $O/r3e_rank_share_c3.txt; tail -1 $O/r3e_rank_share_c4.txt
