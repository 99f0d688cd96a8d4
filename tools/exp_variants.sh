#!/bin/bash
# Run ON THE GPU BOX: bench.py (config 1, no CPU legs) against kernel-experiment builds of the library
# (make variant NAME=x DEFS=...) and against PP_NBK_TARGET settings; one line per run.
#   tools/exp_variants.sh "<variant names>" "<nbk targets>" [extra bench args]
ROOT=$(pwd)
EXTRA=${3:-}
run() {
  python bench.py --no-e2e --no-cpu-baseline --steps 10 --warmup 3 $EXTRA 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', 'ms/step', d['ms_per_step'], d['kernel_ms_per_step'], 'recovered', d['planted_errors_recovered'])"
}
for v in $1; do
  if [ "$v" = "base" ]; then unset PP_LIB_PATH; else export PP_LIB_PATH=$ROOT/polypolish_amd/_build/var_$v/libpolypolish_hip.so; fi
  for t in $2; do
    export PP_NBK_TARGET=$t
    run "variant=$v nbk_target=$t"
  done
done
