#!/bin/bash
for rep in 1 2; do for d in 8192 4096 2048 1024 512; do echo "div $d"; PP_PREPG_DIV=$d tools/exp_variants_quick.sh default 1 2>&1 | grep variant; done; done
