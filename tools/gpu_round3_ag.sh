#!/bin/bash
# Round 3, GPU call AG: SEQ records on 32-byte boundaries (both ingests, the shard split, the bench's resident records) -- the whole
# suite, the default bench line (three roofline entries, traffic passes, end-to-end leg), its kernel trace, configs[4] and [3].
set -u
cd "$(dirname "$0")/.."
O=gpurun_out
mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q -x > $O/r3ag_tests.log 2>&1; echo "tests rc=$?" >> $O/r3ag_tests.log
tail -4 $O/r3ag_tests.log
timeout 400 python bench.py > $O/r3ag_bench.json 2> $O/r3ag_bench.err; echo "bench rc=$?"
export TMPDIR=/tmp
( cd /tmp && timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_tr -- python $OLDPWD/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-e2e --no-live-traffic --no-second-layout > /dev/null 2> $OLDPWD/$O/r3ag_trace.log )
python tools/prof_summary.py /tmp/p_tr > $O/r3ag_kernels.txt 2>&1
find /tmp/p_tr -name "*kernel_stats.csv" -exec cp {} $O/r3ag_kernel_stats.csv \;
B="--steps 20 --warmup 5 --no-e2e --no-cpu-baseline --no-live-traffic --no-second-layout"
timeout 300 python bench.py --config 4 $B > $O/r3ag_c4.json 2> $O/r3ag_c4.err
timeout 300 python bench.py --config 3 $B > $O/r3ag_c3.json 2> $O/r3ag_c3.err
timeout 300 python bench.py --config 2 $B > $O/r3ag_c2.json 2> $O/r3ag_c2.err
for f in $O/r3ag_bench.json $O/r3ag_c4.json $O/r3ag_c3.json $O/r3ag_c2.json; do python - "$f" <<'P'
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
grep -E "^k_tile|^k_fill|^k_prep" $O/r3ag_kernels.txt | head -4
