#!/bin/bash
# Run ON THE GPU BOX: kernel trace of the two text front ends on the 5 Mbp / 200x files
# (the device tokenizer inside polish, the device load inside filter; through the Python binding, because the CLI
# leaves with _exit and rocprofv3 would never get to write its trace).
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_tokenizers
mkdir -p "$OUT"
export TMPDIR=/tmp
python tools/e2e_filter.py 5000000 200 > "$OUT/e2e_filter.log" 2>&1     # leaves /tmp/flt_1.sam, flt_2.sam, flt.fasta
cd /tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tk_polish -- python -c "import sys; sys.path.insert(0, '$ROOT'); import torch, polypolish_amd as pp; pp.polish('/tmp/flt.fasta', ['/tmp/flt_1.sam', '/tmp/flt_2.sam'])" > /dev/null 2> "$OUT/polish.log"
PP_DEVICE_FILTER=1 timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tk_filter -- python -c "import sys; sys.path.insert(0, '$ROOT'); import torch, polypolish_amd as pp; pp.filter('/tmp/flt_1.sam', '/tmp/flt_2.sam', '/tmp/o1.sam', '/tmp/o2.sam')" > /dev/null 2> "$OUT/filter.log"
cd "$ROOT"
python tools/prof_summary.py /tmp/tk_polish /tmp/tk_filter > "$OUT/tokenizers_kernel_trace.txt"
cat "$OUT/tokenizers_kernel_trace.txt" | head -100
