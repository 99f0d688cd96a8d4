#!/bin/bash
# Run ON THE GPU BOX: k_prepd's geometry with the noted records worked off inside it (variant tail: make variant NAME=tail DEFS=-DPP_PREPD_TAIL=1);
# the build for reference.   tools/exp_prepd_tail_geometry.sh CONFIG "THREADS BLOCKS" ...
c=$1; shift
for rep in 1 2; do
  echo "build:"; tools/exp_variants_quick.sh default $c 2>&1 | grep variant
  for cfg in "$@"; do
    set -- $cfg
    echo "tail, $1 threads x $2 workgroups:"; PP_PREPD_THREADS=$1 PP_PREPD_BLOCKS=$2 tools/exp_variants_quick.sh tail $c 2>&1 | grep variant
  done
done
