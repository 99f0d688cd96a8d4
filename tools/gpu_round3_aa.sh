#!/bin/bash
# Round 3, GPU call AA: the round's reference run after the 4-bit mirror -- suite, profile of the default bench (kernel trace + HBM
# counter passes + the full bench line with its three roofline entries and the end-to-end leg), the other configurations, per-rank cost.
set -u
cd "$(dirname "$0")/.."
O=gpurun_out
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/r3aa_tests.log 2>&1; echo "tests rc=$?" >> $O/r3aa_tests.log
tail -4 $O/r3aa_tests.log
timeout 1200 bash tools/profile_round.sh r3_final2 > $O/r3aa_profile.log 2>&1; echo "profile rc=$?"
timeout 400 python bench.py --config 2 --steps 20 --warmup 5 --no-e2e > $O/r3aa_bench_c2.json 2> $O/r3aa_bench_c2.err; echo "c2 rc=$?"
timeout 500 python bench.py --config 3 --steps 20 --warmup 5 --no-e2e > $O/r3aa_bench_c3.json 2> $O/r3aa_bench_c3.err; echo "c3 rc=$?"
timeout 700 python bench.py --config 4 --steps 20 --warmup 5 --no-e2e --no-live-traffic > $O/r3aa_bench_c4.json 2> $O/r3aa_bench_c4.err; echo "c4 rc=$?"
timeout 300 python bench.py --steps 20 --warmup 5 --no-e2e --no-cpu-baseline --no-live-traffic --no-second-layout --nd-frac 0.01 > $O/r3aa_bench_nd.json 2> $O/r3aa_bench_nd.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-e2e --no-cpu-baseline --no-live-traffic --no-second-layout --recipe subs > $O/r3aa_bench_subs.json 2> $O/r3aa_bench_subs.err
timeout 400 python tools/exp_rank_share.py 3 8 > $O/r3aa_rank_share_c3.txt 2>&1
timeout 400 python tools/exp_rank_share.py 4 8 > $O/r3aa_rank_share_c4.txt 2>&1
tail -1 $O/r3aa_rank_share_c3.txt; tail -1 $O/r3aa_rank_share_c4.txt
for f in $O/prof_r3_final2/bench.json $O/r3aa_bench_*.json; do python - "$f" <<'P'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d['roofline']
    print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], r['kernel_ms'], r['frac'], r.get('hbm_actual') and r['hbm_actual']['frac_of_practical_copy_rate'], d['kernel_ms_per_step'])
    for k in ('roofline_window_grouped_seq','roofline_window_grouped_seq4'):
        e=d.get(k)
        if e: print('   ', k, e['kernel_ms'], e['frac'], e['ms_per_step'], e['traffic'], e['same_polished_bytes'])
    if d.get('e2e'): print('    e2e', {k:(v.get('wall_s'), v.get('parity')) for k,v in d['e2e'].items() if isinstance(v,dict) and 'wall_s' in v})
except Exception as e: print(sys.argv[1], 'ERR', e)
P
done
