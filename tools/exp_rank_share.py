"""What ONE rank of an N-way sharded job costs (strong scaling, measured on one GPU): the job of bench.py --config C as
rank r of `world` sees it -- the records that reach its units (pp_shard_split) and its emit ranges.  `full` as a third
argument gives every rank ALL records instead (the scheme of round 2: k_prep / k_fill stream and drop).
    python tools/exp_rank_share.py 4 8        # config 4, world 8: per-rank step time and kernel groups for every rank"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
import polypolish_amd as pp

config, world = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device("cuda:0")
lens, cov, repeat, label = bench.config_shape(config)
job = bench.make_job(dev, contig_lens=lens, coverage=cov, repeat=repeat)
job = bench.synthjob.with_wo(bench.synthjob.with_seq4(job))   # the two mirrors, as bench.py's default job carries them
ctx = pp.Context(0)
ctx.trust_mirrors(True)   # (as bench.py: the job's torch-made mirror is what an ingest would hand over)
ctx.set_profiling(1)
counts = np.bincount(job["recs"]["contig"].cpu().numpy().astype(np.int64), minlength=len(lens))
plan = pp.Plan(job["contig_off"], counts, world, 0)
def timed(j, reps=5):
    bench.run_job(ctx, pp, j)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps):
        bench.run_job(ctx, pp, j)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e3, ctx.kernel_times()
ms, kt = timed(job)
print(label); print("whole job: %.3f ms" % ms, kt)
tot = 0
full = len(sys.argv) > 3 and sys.argv[3] == "full"
for r in range(world):
    if full:
        j = dict(job); j.pop("_prepared", None); j["emit"] = plan.emit_ranges(r)
    else:
        j = bench.shard_of(ctx, pp, job, plan, r)
    ms_r, kt = timed(j)
    if not full:
        print("   records of rank %d: %d of %d" % (r, j["part"].n_aln, job["n_aln"]))
        j["part"].close()
    tot = max(tot, ms_r)
    print("rank %d of %d: %.3f ms%s" % (r, world, ms_r, " (direct path)" if ctx.took_direct_path() else ""), kt)
print("slowest rank %.3f ms -> speedup %.2f of %d" % (tot, ms / tot, world))
