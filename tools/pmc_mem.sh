#!/bin/bash
# Run ON THE GPU BOX: memory-side counters (L2 / fabric / TA) of k_tile and of the gather microbenchmark.
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_mem
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-e2e --no-live-traffic --no-second-layout"
MB="$ROOT/tools/microbench/gather"
cd /tmp
i=0
for set in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum" \
           "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum" \
           "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout -s KILL 150 rocprofv3 --pmc $set --output-format csv -d /tmp/m_b$i -- $BENCH > /dev/null 2>> "$OUT/log.txt" || echo "bench pass $i failed"
  timeout -s KILL 60 rocprofv3 --pmc $set --output-format csv -d /tmp/m_g$i -- $MB > /dev/null 2>> "$OUT/log.txt" || echo "gather pass $i failed"
done
cd "$ROOT"
python tools/prof_summary.py /tmp/m_b1 /tmp/m_b2 /tmp/m_b3 /tmp/m_b4 > "$OUT/bench_mem.txt"
python tools/prof_summary.py /tmp/m_g1 /tmp/m_g2 /tmp/m_g3 /tmp/m_g4 > "$OUT/gather_mem.txt"
grep -A8 "^k_tile" "$OUT/bench_mem.txt"
grep -A8 "k_gather<8, true>\|k_gather<5, false>" "$OUT/gather_mem.txt" | head -80
tail -3 "$OUT/log.txt"
