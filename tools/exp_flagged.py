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
torch.cuda.synchronize()   # (the library runs on a stream of its own: the job's arrays have to be there)
ctx = pp.Context(0)
ctx.set_profiling(int(os.environ.get("PROFILING", "1")))
r = job["recs"]
wo = job["wo"].cpu().numpy().view(np.uint8).reshape(-1).view(pp.WO_DTYPE)
i0 = int(np.nonzero(wo["file_idx"] == 0)[0][0])
print("record 0:", {k: int(r[k][0]) for k in ("contig", "ref_start", "k", "seq_off", "seq_len", "cig_off", "n_cig")},
      "its mirror entry", i0, wo[i0], "runs", job["wo_runs"], flush=True)
for label in os.environ.get("ORDER", "direct,bucketing,direct").split(","):
    os.environ.pop("PP_BENCH_NO_RUNS", None)
    if label == "bucketing":
        os.environ["PP_BENCH_NO_RUNS"] = "1"
    job.pop("_prepared", None)
    print(f"== {label}", flush=True)
    try:
        bench.run_job(ctx, pp, job)
        ctx.sync()
        print(label, ctx.kernel_times(), flush=True)
    except pp.PolypolishError as e:
        print(label, "FAILED:", e, flush=True)
os.environ.pop("PP_BENCH_NO_RUNS", None)
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
