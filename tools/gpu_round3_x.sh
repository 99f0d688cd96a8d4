#!/bin/bash
# Round 3, GPU call X: gather6 with the one-lane-per-read access pattern.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
./tools/microbench/gather6 > gpurun_out/r3x_gather6.txt 2>&1
cat gpurun_out/r3x_gather6.txt
