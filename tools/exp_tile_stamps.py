"""Profiling build only (make variant NAME=stamps DEFS=-DPP_TILE_STAMPS): timeline of k_tile's blocks from the
per-block stamps the kernel leaves in PP_TILE_STAMPS_FILE (start, items done, end; 100 MHz ticks).
    PP_LIB_PATH=polypolish_amd/_build/var_stamps/libpolypolish_hip.so PP_TILE_STAMPS_FILE=/tmp/st.bin \
        python bench.py --config 2 --no-e2e --no-cpu-baseline --steps 1 --warmup 1; python tools/exp_tile_stamps.py /tmp/st.bin"""
import sys
import numpy as np

a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 8)
ran = a[:, 0] > 0
t0 = a[ran, 0].min()
us = lambda x: (x.astype(np.int64) - int(t0)) / 100.0
start, mid, end = us(a[:, 0]), us(a[:, 1]), us(a[:, 2])
win = (a[:, 3] & 0xFFFFFFFF).astype(np.int64)
part = ((a[:, 3] >> 32) & 0xFF).astype(np.int64)
heavy = ((a[:, 3] >> 40) & 1).astype(bool)
idx = np.nonzero(ran)[0]
hw = a[:, 4].astype(np.int64)
cu = ((a[:, 5].astype(np.int64) & 15) << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15)   # xcc | se | sh | cu
pro = us(a[:, 6])
print("blocks that ran", len(idx), "of", len(a), "| kernel span %.1f us" % end[ran].max())
print("ordinary blocks: items %.1f us mean (%.1f max), vote+epilogue %.1f us mean" %
      ((mid - start)[ran & ~heavy].mean(), (mid - start)[ran & ~heavy].max(), (end - mid)[ran & ~heavy].mean()))
h = idx[heavy[idx]]
if len(h):
    print("helper blocks: %d, start %.1f..%.1f, items done %.1f..%.1f us, end %.1f..%.1f us" %
          (len(h), start[h].min(), start[h].max(), mid[h].min(), mid[h].max(), end[h].min(), end[h].max()))
    for w in np.unique(win[h]):
        b = h[win[h] == w]
        print("  window %5d: parts %d, items done by %.1f, last end %.1f" % (w, len(b), mid[b].max(), end[b].max()))
    print("helpers: prologue %.1f us mean, items %.1f mean %.1f max" % ((pro - start)[h].mean(), (mid - pro)[h].mean(), (mid - pro)[h].max()))
    print("CUs with helpers:", len(np.unique(cu[h])), "helpers per CU (max)", np.bincount(np.unique(cu[h], return_inverse=True)[1]).max())
    # blocks in flight per CU in the first 50 us
    early = idx[(start[idx] < 50)]
    per_cu = np.bincount(np.unique(cu[early], return_inverse=True)[1])
    print("first 50 us: CUs seen", len(per_cu), "blocks started per CU: min %d mean %.2f max %d" % (per_cu.min(), per_cu.mean(), per_cu.max()))
    hc = set(cu[h].tolist())
    on_h = np.array([c in hc for c in cu[early]])
    print("  started in the first 50 us on CUs with a helper: %d, elsewhere: %d" % (on_h.sum(), (~on_h).sum()))
print("ordinary blocks: prologue %.1f us mean, items %.1f" % ((pro - start)[ran & ~heavy].mean(), (mid - pro)[ran & ~heavy].mean()))
p1 = us(a[:, 7])
oo = ran & ~heavy & (a[:, 7] > 0)
if oo.any():
    print("ordinary blocks: prefix sum + vote pass 1 %.1f us mean, pass 2 + epilogue %.1f, whole block %.1f" %
          ((p1 - mid)[oo].mean(), (end - p1)[oo].mean(), (end - start)[oo].mean()))
    # the gap between a block's end and the start of the next block on the same CU slot: dispatch cost
    order = np.argsort(start[oo]); cus = cu[oo][order]; st = start[oo][order]; en = end[oo][order]
    gaps = []
    for c in np.unique(cus)[:64]:
        m = cus == c
        s_c, e_c = st[m], np.sort(en[m])
        # k-th start after the first two on this CU follows the (k-2)-th end
        for k in range(2, min(len(s_c), len(e_c) + 2)):
            gaps.append(s_c[k] - e_c[k - 2])
    if gaps:
        g = np.array(gaps)
        print("end of a block -> start of the next one on its CU: median %.1f us, mean %.1f, p90 %.1f" % (np.median(g), g.mean(), np.percentile(g, 90)))
o = idx[~heavy[idx]]
bins = (start[o] // 50).astype(int)
print("ordinary blocks by start time (50 us bins): count, mean items us:",
      [(int((bins == b).sum()), round(float((mid - pro)[o][bins == b].mean()), 1)) for b in range(bins.max() + 1)])
print("distinct CUs seen over the whole kernel:", len(np.unique(cu[idx])))
late = idx[np.argsort(-end[idx])[:12]]
print("latest blocks:")
for b in late:
    print("  block %5d window %5d heavy %d part %d: start %.1f items %.1f end %.1f" % (b, win[b], heavy[b], part[b], start[b], mid[b], end[b]))
# how busy is the chip over time: blocks in flight per 25 us
edges = np.arange(0, end[ran].max() + 25, 25)
busy = [(int(((start[idx] < e + 25) & (end[idx] > e)).sum())) for e in edges]
print("blocks in flight per 25 us:", busy)
# the longest blocks, phase by phase
dur = (end - start)
longest = idx[np.argsort(-dur[idx])[:8]]
print("longest blocks (window: start | prologue, items, prefix sums + pass 1, pass 2 + epilogue):")
for b in longest:
    print("  block %5d window %5d heavy %d: start %.1f | %.1f %.1f %.1f %.1f = %.1f us" %
          (b, win[b], heavy[b], start[b], pro[b] - start[b], mid[b] - pro[b], p1[b] - mid[b], end[b] - p1[b], dur[b]))
