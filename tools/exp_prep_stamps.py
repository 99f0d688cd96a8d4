"""Profiling build only (make variant NAME=pstamps DEFS=-DPP_PREP_STAMPS): where the blocks of k_prep and k_fill spend
their time, from the ticks thread 0 of every block leaves in PP_PREP_STAMPS_FILE (100 MHz).
    PP_LIB_PATH=polypolish_amd/_build/var_pstamps/libpolypolish_hip.so PP_PREP_STAMPS_FILE=/tmp/ps.bin \
        python bench.py --no-e2e --no-cpu-baseline --no-live-traffic --no-second-layout --steps 1 --warmup 1
    python tools/exp_prep_stamps.py /tmp/ps.bin"""
import sys
import numpy as np

a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(2, 16384, 8).astype(np.int64)
names = {0: ("k_prep", ["start", "counters cleared", "stream loop done (thread 0)", "... whole block", "noted records done", "longest read noted, barrier", "histogram row written"]),
         1: ("k_fill", ["start", "cursors set", "items written (thread 0)"])}
if "--direct" in sys.argv:  # the direct path's one pass over the mirror (pp_k_direct.h)
    names = {0: ("k_prepd", ["start", "stage cleared, first window known", "stream loop done (thread 0)", "... whole block", "noted records done (thread 0)",
                             "slots taken for the staged extras", "extras written (thread 0)"]),
             1: ("k_prepg", ["start", "stage cleared, first window known", "noted records done (thread 0)", "... whole block", "extras written (thread 0)"])}
t0 = a[0][:, 0][a[0][:, 0] > 0].min()
order = {}
if "--direct" in sys.argv and "--prepg-trips" in sys.argv:  # k_prepg's record of thread 0, trip by trip (slots 5-7 sit between slots 1 and 2)
    names[1] = ("k_prepg", ["start", "stage cleared, first window known", "run count / CIGAR offset / last 8 bases there", "the runs there", "runs walked, trim known",
                            "pieces staged (thread 0's record done)", "... whole block", "extras written (thread 0)"])
    order[1] = [0, 1, 5, 6, 7, 2, 3, 4]
for k in sorted(names):
    nm, pts = names[k]
    b = a[k]
    if k in order:
        b = b[:, order[k]]
    ran = b[:, 0] > 0
    us = (b[ran][:, :len(pts)] - t0) / 100.0
    print(f"{nm}: {ran.sum()} blocks; first start {us[:, 0].min():.1f} us, last start {us[:, 0].max():.1f}, last end {us[:, len(pts) - 1].max():.1f}")
    for i in range(1, len(pts)):
        d = us[:, i] - us[:, i - 1]
        print(f"   {pts[i - 1]:32s} -> {pts[i]:32s} mean {d.mean():7.2f} us   p10 {np.percentile(d, 10):7.2f}   p90 {np.percentile(d, 90):7.2f}   max {d.max():7.2f}")
