#!/usr/bin/env python3
"""Experiment: does k_tile run at a different speed when the records live in the library's own hipMalloc'd
buffers (PP_MEM_HOST upload) rather than in torch-allocated tensors (PP_MEM_DEVICE)?"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import polypolish_amd as pp
dev = torch.device("cuda", 0)
job = bench.make_job(dev)
ctx = pp.Context(0)
ctx.set_profiling(True)
for i in range(3):
    bench.run_job(ctx, pp, job)
    ctx.sync()
    print("device-resident torch tensors:", {k: round(v, 4) for k, v in ctx.kernel_times()["ms"].items()})
host = bench.to_host_records(job)
hb = job["bases"].cpu().numpy()
off = np.array([0, job["G"]], dtype=np.uint64)
for i in range(3):
    ctx.polish_records(off, hb, host)
    print("library-owned buffers (uploaded):", {k: round(v, 4) for k, v in ctx.kernel_times()["ms"].items()})
