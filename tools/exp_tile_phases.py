#!/usr/bin/env python3
"""Experiment driver for the phase builds of k_tile (-DPP_TILE_STOP=K, see pp_k_tile.h): STEPS jobs of bench.py's
configuration CONFIG (the resident layout of the default bench line: window-grouped SEQ, seq4, mirror + run table) with
nothing verified -- a truncated kernel polishes nothing.  Run it under rocprofv3 (--kernel-trace for the truncated kernels'
durations, --pmc SQ_INSTS_VALU ... for their instruction counts): tools/exp_tile_phases.sh does.
    PP_LIB_PATH=polypolish_amd/_build/var_stop2/libpolypolish_hip.so python tools/exp_tile_phases.py [CONFIG] [STEPS]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from bench import synthjob
import polypolish_amd as pp
config = int(sys.argv[1]) if len(sys.argv) > 1 else 1
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
dev = torch.device("cuda", 0)
lens, coverage, repeat, label = bench.config_shape(config, None, None)
job = synthjob.make_job(dev, contig_lens=lens, coverage=coverage, seed=42 + config + 1, indel_read_frac=synthjob.SURVEY_INDEL_READ_FRAC,
                        repeat=repeat, recipe="survey")
job = synthjob.with_wo(synthjob.with_seq4(job))
torch.cuda.synchronize()
ctx = pp.Context(0)
for _ in range(steps):
    bench.run_job(ctx, pp, job)
ctx.sync()
print("ran", steps, "jobs of", label, "| direct path:", ctx.took_direct_path())
