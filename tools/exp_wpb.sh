#!/bin/bash
# Run ON THE GPU BOX: windows per workgroup of k_tile (PP_TILE_WPB) against the build in var_head, interleaved.
#   tools/exp_wpb.sh "<configs>" [reps]
for rep in $(seq 1 ${2:-2}); do
  for c in $1; do
    tools/exp_variants_quick.sh head $c 2>&1 | grep variant
    for w in 1 2 4; do
      echo "wpb $w"; PP_TILE_WPB=$w tools/exp_variants_quick.sh default $c 2>&1 | grep variant
    done
  done
done
