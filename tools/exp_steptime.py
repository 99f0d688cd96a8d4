#!/usr/bin/env python3
"""Experiment: wall time per step with and without the per-kernel event timers."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import polypolish_amd as pp
dev = torch.device("cuda", 0)
job = bench.make_job(dev)
ctx = pp.Context(0)
for prof in (0, 2, 1, 0, 2):  # 0 no events, 2 one pair around k_tile (what bench.py's timed steps use), 1 every group
    ctx.set_profiling(prof)
    for _ in range(3):
        bench.run_job(ctx, pp, job)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        bench.run_job(ctx, pp, job)
        if prof:
            ctx.kernel_times()
    ctx.sync()
    dt = (time.perf_counter() - t0) / 20
    print(f"profiling={prof}: {dt * 1e3:.4f} ms/step", {k: round(v, 4) for k, v in ctx.kernel_times()["ms"].items()} if prof else "")
