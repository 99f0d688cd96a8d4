#!/usr/bin/env python3
"""CPU experiment for DESIGN section 9 (lead 2): how often does the vote really need the order-dependent f64 depth?

The vote consumes depth only through `depth < min_depth` and bankers_rounding(depth * fraction) for two fractions
(pileup.rs:70-72,113), all monotone in depth.  With every share 1/k replaced by floor(2^20/k)/2^20 the sum is an
integer count (exact in any order) and   lo <= depth_f64 <= lo + c * 2^-20 (+ f64 summation error),
c = number of covering reads whose share is not a power of two.  This script builds reads without indels, derives
lo / hi per position with numpy, and checks against the oracle's ordered f64 depth and thresholds:
  * soundness: the oracle's depth lies inside [lo, hi] and its thresholds inside [T(lo), T(hi)] everywhere;
  * yield: the fraction of positions where T(lo) == T(hi) for all three quantities (no replay needed).
Usage: exp_interval_vote.py [genome_bp] [coverage] [nondyadic_fraction] [k3]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth
from oracle import orc

G = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000
cov = int(sys.argv[2]) if len(sys.argv) > 2 else 200
nd = float(sys.argv[3]) if len(sys.argv) > 3 else 0.03
BITS = 20
MIN_DEPTH, FV, FI = 5, 0.5, 0.2

only3 = len(sys.argv) > 4 and sys.argv[4] == "k3"     # every shared read has exactly three alignments
ks = (1, 2, 3) if only3 else (1, 2, 3, 5, 6, 7)
probs = (1 - nd - 0.01, 0.01, nd) if only3 else (1 - nd - 0.01, 0.01, nd / 4, nd / 4, nd / 4, nd / 4)
o, b, r = synth.fast_records(seed=71, contig_lens=(G,), coverage=cov, k_choices=ks, k_probs=probs, indel_read_frac=0.0)
t = time.time()
want = orc.polish_records(o, b, r, min_depth=MIN_DEPTH, fraction_valid=FV, fraction_invalid=FI, positions=True)
P = want["positions"]
print(f"{len(r['k'])} reads, oracle {time.time() - t:.1f} s")

# kept extent of every read (alignment.rs:364-378 for reads without indels): entries [0, start - 1), start = first
# base of the trailing homopolymer
L = int(r["seq_len"][0])
seq = r["seq"].reshape(-1, L)
differs = seq != seq[:, -1:]
last_diff = np.where(differs.any(axis=1), L - 1 - np.argmax(differs[:, ::-1], axis=1), -1)   # index of last differing base
nkeep = np.maximum(last_diff, 0)      # start = last_diff + 1; kept = start - 1 = last_diff (>= 0)
start = r["ref_start"].astype(np.int64)
scaled = np.zeros(G + 1, dtype=np.int64)     # difference array of sum floor(2^BITS / k)
cnt_nd = np.zeros(G + 1, dtype=np.int64)     # ... of the number of non-dyadic covering reads
k = r["k"].astype(np.int64)
w = (1 << BITS) // k
is_nd = (k & (k - 1)) != 0
np.add.at(scaled, start, w); np.add.at(scaled, start + nkeep, -w)
np.add.at(cnt_nd, start, is_nd.astype(np.int64)); np.add.at(cnt_nd, start + nkeep, -is_nd.astype(np.int64))
scaled = np.cumsum(scaled)[:G]
cnt_nd = np.cumsum(cnt_nd)[:G]
lo = scaled / float(1 << BITS)               # exact: scaled < 2^53
slack = 1e-9                                 # >> the f64 summation error of a few hundred additions
hi = lo + cnt_nd / float(1 << BITS) + slack
lo = lo - slack
depth = P["depth"]
assert np.all(lo <= depth) and np.all(depth <= hi), "bounds do not contain the ordered f64 depth"


def bankers(x):   # misc.rs:208-215, vectorised
    fl = np.floor(x)
    frac = x - fl
    up = (frac > 0.5) | ((frac == 0.5) & (fl % 2 == 1))
    return (fl + up).astype(np.int64)


amb = np.zeros(G, dtype=bool)
for f, name, floor_ in ((FV, "valid_thr", MIN_DEPTH), (FI, "invalid_thr", 0)):
    tl, th = np.maximum(bankers(lo * f), floor_), np.maximum(bankers(hi * f), floor_)   # pileup.rs:70-72
    truth = P[name].astype(np.int64)
    assert np.all(tl <= truth) and np.all(truth <= th), name
    amb |= tl != th
amb |= (lo < MIN_DEPTH) != (hi < MIN_DEPTH)
touched = cnt_nd > 0
print(f"positions touched by a non-dyadic share: {touched.mean() * 100:.2f} %  (all of them are replayed today)")
print(f"positions whose vote is NOT decided by the bounds: {amb.sum()} of {G} = {amb.mean() * 100:.4f} %")
print(f"windows of 2048 bp with at least one such position: {len(np.unique(np.nonzero(amb)[0] // 2048))} of {(G + 2047) // 2048}")
