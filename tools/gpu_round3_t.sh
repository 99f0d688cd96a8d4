#!/bin/bash
# Round 3, GPU call T: gather6 -- cache-policy bits, pitches and 4-bit reads in file order: times, then fetched bytes.
set -u
cd "$(dirname "$0")/.."
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
./tools/microbench/gather6 > $O/r3t_gather6.txt 2>&1
cat $O/r3t_gather6.txt
( cd /tmp && timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/g6 -- $OLDPWD/tools/microbench/gather6 > /dev/null 2> $OLDPWD/$O/r3t_pmc.log )
python - <<'P' > gpurun_out/r3t_gather6_fetch.txt 2>&1
import csv, glob, collections
agg = collections.OrderedDict()
for path in glob.glob('/tmp/g6/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(path)):
        if row['Counter_Name'] != 'FETCH_SIZE': continue
        agg.setdefault((row['Kernel_Name'][:60], row.get('Dispatch_Id')), 0.0)
        agg[(row['Kernel_Name'][:60], row.get('Dispatch_Id'))] += float(row['Counter_Value'])
# dispatches come in groups of 6 per variant, in launch order
rows = sorted(agg.items(), key=lambda kv: int(kv[0][1]))
for i in range(0, len(rows), 6):
    grp = rows[i:i + 6]
    vals = [v for _, v in grp]
    print(grp[0][0][0], 'dispatches', grp[0][0][1], '..', grp[-1][0][1], 'FETCH_SIZE raw mean %.0f  (x2 x 32B? see profiles/traffic notes)' % (sum(vals) / len(vals)))
P
cat $O/r3t_gather6_fetch.txt
