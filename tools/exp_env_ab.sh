#!/bin/bash
# Run ON THE GPU BOX: kernel groups and step time of the library as built with an environment switch off / on, interleaved.
#   tools/exp_env_ab.sh "PP_EMIT_FUSE=0" "<configs>" [reps]
for rep in $(seq 1 ${3:-2}); do
  for c in $2; do
    echo "with $1:"; env $1 tools/exp_variants_quick.sh default $c 2>&1 | grep variant
    echo "default:"; tools/exp_variants_quick.sh default $c 2>&1 | grep variant
  done
done
