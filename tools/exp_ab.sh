#!/bin/bash
# Run ON THE GPU BOX: kernel groups of experiment builds against each other, interleaved rounds.
#   tools/exp_ab.sh "<variants>" "<configs>" [reps]      ("default" = the library as built; others: polypolish_amd/_build/var_<name>)
for rep in $(seq 1 ${3:-2}); do
  for c in $2; do
    for v in $1; do tools/exp_variants_quick.sh $v $c 2>&1 | grep variant; done
  done
done
