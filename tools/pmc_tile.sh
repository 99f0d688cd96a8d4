#!/bin/bash
# Run ON THE GPU BOX: utilisation counters of k_tile (vector / LDS / memory pipelines) on bench.py's configuration CONFIG,
# one rocprofv3 --pmc pass per set (counters only), tools/exp_tile_phases.py as the driver.
#   tools/pmc_tile.sh TAG [CONFIG]
set -u
TAG=${1:-pmc_tile}; CONFIG=${2:-1}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
i=0
for set in "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" \
           "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TA_BUFFER_WAVEFRONTS_sum TCP_GATE_EN1_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  rm -rf /tmp/pt_$i
  timeout -s KILL 200 rocprofv3 --pmc $set --output-format csv -d /tmp/pt_$i -- python $ROOT/tools/exp_tile_phases.py $CONFIG 5 > "$OUT/run_$i.log" 2>&1 || echo "pass $i failed: $set" | tee -a "$OUT/failed.txt"
done
cd "$ROOT"
python tools/prof_summary.py /tmp/pt_1 /tmp/pt_2 /tmp/pt_3 /tmp/pt_4 /tmp/pt_5 /tmp/pt_6 | grep -A10 "^k_tile" > "$OUT/pmc_tile_config$CONFIG.txt"
cat "$OUT/pmc_tile_config$CONFIG.txt"
