#!/bin/bash
# Run ON THE GPU BOX: bench.py (no CPU legs, no counter passes) against kernel-experiment builds of the library
# (make variant NAME=x DEFS=... -> polypolish_amd/_build/var_x/; "default" = the library as built) -- REPS rounds over all
# of them, interleaved, one line per run: step time and the per-kernel milliseconds.
#   tools/exp_variants.sh "<variant names>" [REPS] [extra bench args]
ROOT=$(pwd)
REPS=${2:-3}
EXTRA=${3:-}
for r in $(seq 1 "$REPS"); do
  for v in $1; do
    if [ "$v" = "default" ]; then unset PP_LIB_PATH; else export PP_LIB_PATH=$ROOT/polypolish_amd/_build/var_$v/libpolypolish_hip.so; fi
    python bench.py --no-e2e --no-cpu-baseline --no-live-traffic --no-second-layout --steps 30 --warmup 5 $EXTRA 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('variant=$v rep=$r', 'ms/step', d['ms_per_step'], d['kernel_ms_per_step'], 'recovered', d['planted_errors_recovered'])"
  done
done
