#!/usr/bin/env python3
"""Experiment: which positions does a job list for the exact replay (k_exact), with the direct path and with the bucketing
path, and what do the per-position records say about them?  (PP_TRACE_FLAGGED=1 makes the library print the list.)"""
import os, sys
os.environ["PP_TRACE_FLAGGED"] = "1"
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import polypolish_amd as pp
from bench import synthjob
dev = torch.device("cuda", 0)
lens, cov, repeat, _ = bench.config_shape(int(os.environ.get("CONFIG", "1")))
job = synthjob.make_job(dev, contig_lens=lens, coverage=cov, seed=42 + int(os.environ.get("CONFIG", "1")) + 1 + 1000 * 0, repeat=repeat)
job = synthjob.with_wo(synthjob.with_seq4(job))
ctx = pp.Context(0)
ctx.set_profiling(1)
for label, env in (("direct", None), ("bucketing", "1")):
    if env:
        os.environ["PP_BENCH_NO_RUNS"] = env
    job.pop("_prepared", None)
    print(f"== {label}", flush=True)
    bench.run_job(ctx, pp, job)
    ctx.sync()
    print(label, ctx.kernel_times(), flush=True)
# the per-position records (--debug planes) of the same job
pp.lib().pp_polish_set_debug(ctx._h, 1)
job.pop("_prepared", None)
bench.run_job(ctx, pp, job)
pos = ctx.positions()
pp.lib().pp_polish_set_debug(ctx._h, 0)
cand = np.nonzero((pos["count_other"] > 0) & (pos["count_other"] >= pos["invalid_thr"]) & (pos["status"] != 2))[0]
print("positions whose string-keyed tally reaches the invalid threshold:", len(cand))
show = [int(x) for x in os.environ.get("POS", "").split(",") if x] or [int(x) for x in cand[:12]]
for gp in show:
    print(gp, "window", gp // 2048, "at", gp % 2048, {k: (float(v[gp]) if k == "depth" else int(v[gp])) for k, v in pos.items()},
          "bases", bytes(job["bases"][gp - 3:gp + 4].cpu().numpy()))
