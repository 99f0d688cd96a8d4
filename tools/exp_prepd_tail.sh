#!/bin/bash
# Run ON THE GPU BOX: the noted records worked off inside k_prepd (the build) against k_prepd + k_prepg (variant notail: make variant NAME=notail DEFS=-DPP_PREPD_TAIL=0), interleaved
for rep in 1 2; do for c in ${1:-1 2 3 4}; do tools/exp_variants_quick.sh "notail default" $c 2>&1 | grep variant; done; done
