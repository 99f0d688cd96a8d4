#!/bin/bash
# Run ON THE GPU BOX: SQ issue/stall counters of the bench workload (two passes, 8 SQ slots each).
set -u
TAG=${1:-sq}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-e2e --no-live-traffic --no-second-layout"
cd /tmp
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d /tmp/sq1 -- $BENCH > /dev/null 2> "$OUT/sq1.log"
timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM --output-format csv -d /tmp/sq2 -- $BENCH > /dev/null 2> "$OUT/sq2.log"
timeout -s KILL 300 rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_WAVES_EQ_64 SQ_INSTS_VALU_MFMA_I8 --output-format csv -d /tmp/sq3 -- $BENCH > /dev/null 2> "$OUT/sq3.log"
cd "$ROOT"
python tools/prof_summary.py /tmp/sq1 /tmp/sq2 /tmp/sq3 > "$OUT/sq_summary.txt"
grep -A26 "^k_stream\|^k_regroup\|^k_tile" "$OUT/sq_summary.txt"
tail -3 "$OUT"/sq*.log
