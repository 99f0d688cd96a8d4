#!/bin/bash
# Run ON THE GPU BOX: step time and kernel groups of experiment builds whose results are wrong by design (no verification):
#   tools/exp_variants_quick.sh "<variant names>" [CONFIG]   ("default" = the library as built)
ROOT=$(pwd); CONFIG=${2:-1}
for v in $1; do
  if [ "$v" = "default" ]; then unset PP_LIB_PATH; else export PP_LIB_PATH=$ROOT/polypolish_amd/_build/var_$v/libpolypolish_hip.so; fi
  python - "$CONFIG" "$v" <<'PY'
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
import bench
from bench import synthjob
import polypolish_amd as pp
config, name = int(sys.argv[1]), sys.argv[2]
dev = torch.device("cuda", 0)
lens, coverage, repeat, label = bench.config_shape(config, None, None)
job = synthjob.make_job(dev, contig_lens=lens, coverage=coverage, seed=42 + config + 1, indel_read_frac=synthjob.SURVEY_INDEL_READ_FRAC, repeat=repeat, recipe="survey")
job = synthjob.with_wo(synthjob.with_seq4(job))
torch.cuda.synchronize()
ctx = pp.Context(0)
ctx.trust_mirrors(True)
ctx.set_profiling(1)
acc = {}
for i in range(12):
    bench.run_job(ctx, pp, job)
    if i >= 4:
        for k, v in ctx.kernel_times()["ms"].items():
            acc[k] = acc.get(k, 0.0) + v / 8
wall = {}
for prof in (2, 0):  # wall time per step as bench.py's timed steps run (events around the dominant kernel only) / without any events
    ctx.set_profiling(prof)
    for _ in range(3):
        bench.run_job(ctx, pp, job)
    best = 1e9
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            bench.run_job(ctx, pp, job)
            if prof:
                ctx.kernel_times()
        ctx.sync()
        best = min(best, (time.perf_counter() - t0) / 20)
    wall["step_ms_prof%d" % prof] = round(best * 1e3, 4)
print("variant", name, "config", config, {k: round(v, 4) for k, v in acc.items()}, wall)
PY
done
