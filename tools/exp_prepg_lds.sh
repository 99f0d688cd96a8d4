#!/bin/bash
# Run ON THE GPU BOX: k_prepg with 13 KB of LDS a workgroup (the build) against 29 KB (variant pg4: make variant NAME=pg4 DEFS="-DPP_PREPG_XSTAGE=4 -DPP_PREPG_CTG=1024"),
# records of the job per workgroup swept (PP_PREPG_DIV), interleaved
for rep in 1 2; do for d in 4096 3072 2048; do for v in pg4 default; do echo "div $d"; PP_PREPG_DIV=$d tools/exp_variants_quick.sh $v ${1:-1} 2>&1 | grep variant; done; done; done
