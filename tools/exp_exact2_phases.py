#!/usr/bin/env python3
"""Experiment (needs an instrumented build of the library, PP_LIB_PATH): per-phase wall-clock stamps of k_exact2 on a
job whose every window has order-dependent depths (3 % of the reads with share 1/3, scattered).

The instrumented build adds, in a scratch copy of pp_k_exact.h, `__syncthreads(); if (tid == 0) T[i] = wall_clock64();`
at the phase boundaries of k_exact2 and one device printf of the differences per 97th window.  Round-1 result
(MI355X, n = 2.7-3.0 K items per window, all 2048 positions flagged; units of 10 ns):
    sort 9-11 us | step 2 (extent + share per item) 12-17 us | ordered pass 60-63 us | vote 2.5-3.3 us
The ordered pass is issue-bound: n x 2.17 wave-visits x 20 instructions on the four SIMDs of one CU."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401  (before the library: torch ships its own HIP runtime)
import synth
import polypolish_amd as pp

t = time.time()
o, b, r = synth.fast_records(seed=61, contig_lens=(1_000_000,), coverage=200, k_choices=(1, 3), k_probs=(0.97, 0.03),
                             indel_read_frac=0.01)
print(f"generated {len(r['ref_start'])} records in {time.time() - t:.1f} s", flush=True)
ctx = pp.Context(0)
ctx.set_profiling(1)
for rep in range(2):
    print(f"--- run {rep}", flush=True)
    ctx.polish_records(o, b, r)
    ctx.sync() if hasattr(ctx, "sync") else None
    print(ctx.kernel_times(), flush=True)
