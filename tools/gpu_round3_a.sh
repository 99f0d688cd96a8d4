#!/bin/bash
# Round 3, GPU call A: box probe, the GPU test suite, the bench line of configs[1] and the end-to-end legs of
# configs[2], [3], [4] from SAM text (results under gpurun_out/r3a_*).
set -u
cd "$(dirname "$0")/.."
O=gpurun_out
mkdir -p $O
{ nproc; free -g; df -h /tmp /dev/shm .; rocm-smi --showmeminfo vram 2>/dev/null | head -8; } > $O/r3a_probe.txt 2>&1
# where the 33 GB of configs[4] text can live
pick_dir() {
  for d in /tmp /dev/shm "$PWD/$O"; do
    avail=$(df -k --output=avail "$d" 2>/dev/null | tail -1)
    if [ -n "$avail" ] && [ "$avail" -gt $((70*1024*1024)) ]; then echo "$d"; return; fi
  done
  echo /tmp
}
BIG=$(pick_dir)/pp_e2e_big
echo "big files in $BIG" >> $O/r3a_probe.txt
timeout 900 python -m pytest tests -m gpu -x -q > $O/r3a_tests.log 2>&1; echo "tests rc=$?" >> $O/r3a_tests.log
timeout 300 python bench.py --steps 20 --warmup 5 > $O/r3a_bench_c1.json 2> $O/r3a_bench_c1.err; echo "c1 rc=$?" >> $O/r3a_probe.txt
timeout 400 python bench.py --config 2 --steps 20 --warmup 5 > $O/r3a_bench_c2.json 2> $O/r3a_bench_c2.err; echo "c2 rc=$?" >> $O/r3a_probe.txt
mkdir -p $BIG
timeout 600 python bench.py --config 3 --e2e-only --e2e-dir $BIG > $O/r3a_e2e_c3.json 2> $O/r3a_e2e_c3.err; echo "c3 rc=$?" >> $O/r3a_probe.txt
rm -rf $BIG; mkdir -p $BIG
{ free -g; df -h $BIG; } >> $O/r3a_probe.txt 2>&1
timeout 1100 python bench.py --config 4 --e2e-only --e2e-dir $BIG > $O/r3a_e2e_c4.json 2> $O/r3a_e2e_c4.err; echo "c4 rc=$?" >> $O/r3a_probe.txt
rm -rf $BIG
tail -3 $O/r3a_tests.log
cat $O/r3a_probe.txt
