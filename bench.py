#!/usr/bin/env python3
"""bench.py -- throughput of the polish hot path on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N>1 it is launched with
torch.distributed.run, one rank per GPU.  One *step* = one full pass of the device hot path over one
synthetic job whose assembly bases and parsed alignment records are ALREADY RESIDENT IN HBM when the
timed region starts.  `--config` picks the workload (BASELINE.json `configs` index):

  1 (default)  configs[1]: 5 Mbp single contig, 200x of 2x150 bp -- the configuration the metric is quoted on
  2            configs[2]: the same genome with a 5 kbp segment present in 5 copies; every read inside a copy
               has 5 alignment records (all-hits), depth share 1/5 each (order-dependent f64 depth)
  3            configs[3]: 100 contigs, log-uniform 100 kbp..2 Mbp rescaled to 50 Mbp, 100x
  4            configs[4]: one 250 Mbp contig, 50x

With N ranks and config 1 every rank polishes its own 5 Mbp contig (contigs shard across GPUs with no
data-path collective; "weak" scaling); with configs 3 / 4 the ONE job is sharded across the ranks (whole
contigs by longest-processing-time; the single contig in 2048-aligned windows with a read-length halo;
"strong" scaling).  The polished bytes are gathered to rank 0 inside the timed step.

Rank 0 prints ONE JSON line with metric/value/unit, plus
  roofline     achieved HBM GB/s of the dominant kernel = algorithmic bytes per launch / mean launch
               duration measured with HIP events on the library's stream
  cpu_baseline the single-threaded C oracle (kind "port": the Rust reference cannot be built here) timed on
               a bounded sample of the same workload, with a live parity check
  e2e          (N=1) the drop-in CLI from SAM TEXT to FASTA on files of the configuration's shape generated on the
               box: wall seconds of `polypolish polish` (device tokenizer and host ingest), `filter`, `filter-polish`
               (configs[2]: real all-hits files, and `filter` followed by `polish`; configs[3], [4]: polish on 13 / 33 GB
               of text), the oracle's CLI on one core on the same files, and whether the output bytes are identical.

Synthetic data: SURVEY.md section 8d's recipe (tools/synthjob.py) -- assembly errors 1/3 substitutions, 1/3 1-bp
deletions, 1/3 1-bp insertions (half of the indels in homopolymers), reads aligned to the ASSEMBLY with the resulting
I / D CIGARs; `--recipe subs` is the substitution-only assembly of rounds 1 and 2.
"""
from __future__ import annotations

import argparse
import ctypes
import hashlib
import json
import os
import re
import subprocess
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

OP_M, OP_I, OP_D = 0, 1, 2

METRIC = "assembly Mbp polished/sec at 200x coverage; bit-identical FASTA vs reference"


def config_shape(config, genome=None, coverage=None):
    """(contig_lens, coverage, repeat, label) of BASELINE.json configs[config]."""
    if config == 0:
        G, cov = genome or 50_000, coverage or 30
        return [G], cov, None, f"configs[0]: {G / 1e3:g} kbp single contig, {G * cov // 150} x 150 bp paired reads"
    if config == 1:
        G, cov = genome or 5_000_000, coverage or 200
        return [G], cov, None, f"configs[1]: {G / 1e6:g} Mbp single-contig assembly per GPU, {cov}x 2x150 bp"
    if config == 2:
        G, cov = genome or 5_000_000, coverage or 200
        return [G], cov, (5000, 5), (f"configs[2]: {G / 1e6:g} Mbp single contig with a 5 kbp segment in 5 copies, {cov}x 2x150 bp, "
                                     "all-hits: 5 records of share 1/5 per read inside a copy")
    if config == 3:
        total, cov = genome or 50_000_000, coverage or 100
        rng = np.random.default_rng(42 + 3)
        lens = np.exp(rng.uniform(np.log(100e3), np.log(2e6), 100))
        floor = 0 if total >= 100 * 50_000 else 2048  # reduced sizes (tests): keep every contig longer than a window
        lens = (floor + np.rint(lens * ((total - 100 * floor) / lens.sum()))).astype(np.int64)
        lens[-1] += total - int(lens.sum())
        return [int(x) for x in lens], cov, None, (f"configs[3]: 100-contig metagenome, log-uniform contig lengths rescaled to "
                                                   f"{total / 1e6:g} Mbp, {cov}x 2x150 bp")
    if config == 4:
        G, cov = genome or 250_000_000, coverage or 50
        return [G], cov, None, f"configs[4]: {G / 1e6:g} Mbp single contig (chr1 scale), {cov}x 2x150 bp"
    raise SystemExit(f"unknown --config {config}")


sys.path.insert(0, os.path.join(ROOT, "tools"))
import synthjob  # noqa: E402  (tools/synthjob.py: the synthetic workloads of SURVEY.md section 8d)
from synthjob import make_job, subset_job, algorithmic_bytes, to_host_records, write_sam_pair, recovered  # noqa: E402,F401


def run_job(ctx, pp, job, params=(5, 0.5, 0.2)):
    """One step: begin + add (device-resident, borrowed) + finish.  The arguments of the three C calls are marshalled
    once per job (ctx.prepared_job): a step is then the calls themselves, not tens of microseconds of Python between
    them while the GPU waits.  job["part"] (a pp.ShardPart: the records one rank of a sharded job needs) replaces the
    job's own record arrays."""
    prepared(ctx, pp, job, params)()


def prepared(ctx, pp, job, params=(5, 0.5, 0.2)):
    """The job's step as one callable (see run_job).  The timed loop looks it up ONCE: building the lookup key -- four
    data_ptr() calls, the contig table's bytes -- is a few microseconds of Python per step that belong to this harness,
    not to the path."""
    part = job.get("part")
    seq4 = job.get("seq4")  # the 4-bit mirror of the seq array (pp_aln_batch.seq4), when the job has one
    wo = job.get("wo")      # the window-order mirror of the records (pp_aln_batch.wo), when the job has one
    key = (id(ctx), params, None if job.get("emit") is None else id(job["emit"]), job["bases"].data_ptr(),
           job["recs"]["seq"].data_ptr(), None if seq4 is None else seq4.data_ptr(), None if wo is None else wo.data_ptr(), job["n_aln"], id(part),
           job["contig_off"].tobytes() if len(job["contig_off"]) < 64 else id(job["contig_off"]))
    run = job.setdefault("_prepared", {}).get(key)
    if run is None:
        r = job["recs"]
        if part is not None:
            run = ctx.prepared_job(job["contig_off"], job["bases"].data_ptr(), pp.MEM_DEVICE, part.n_aln, part.ptrs, part.seq_bytes,
                                   part.n_cig_total, pp.MEM_DEVICE, *params, emit=job.get("emit"))
        else:
            ptrs = {k: v.data_ptr() for k, v in r.items()}
            if seq4 is not None:
                ptrs["seq4"] = seq4.data_ptr()
            if wo is not None:
                ptrs["wo"] = wo.data_ptr()
                if job.get("wo_runs") and not os.environ.get("PP_BENCH_NO_RUNS"):  # the mirror's runs: the direct path
                    ptrs["wo_runs"] = job["wo_runs"]
            run = ctx.prepared_job(job["contig_off"], job["bases"].data_ptr(), pp.MEM_DEVICE, job["n_aln"],
                                   ptrs, r["seq"].numel(), r["cigar"].numel(), pp.MEM_DEVICE,
                                   *params, emit=job.get("emit"))
        job["_prepared"][key] = run
    return run


def shard_of(ctx, pp, job, plan, rank):
    """The job as rank `rank` of the plan sees it: the records that reach its units (pp_shard_split, on the device) and
    its emit ranges; everything else is shared with `job`."""
    r = job["recs"]
    ptrs = {k: v.data_ptr() for k, v in r.items()}
    if job.get("wo") is not None:   # the window-order mirror goes along into the part (restricted to its records), and its runs
        ptrs["wo"] = job["wo"].data_ptr()
        if job.get("wo_runs") and not os.environ.get("PP_BENCH_NO_RUNS"):
            ptrs["wo_runs"] = job["wo_runs"]
    part = pp.ShardPart(ctx, plan, rank, job["n_aln"], ptrs, r["seq"].numel(), r["cigar"].numel(), pp.MEM_DEVICE)
    mine = dict(job)
    mine.pop("_prepared", None)
    mine["part"] = part
    mine["emit"] = plan.emit_ranges(rank)
    return mine


# ---- the end-to-end leg: SAM text -> FASTA through the drop-in CLI ----------------------------------------
def _timed(cmd, env=None, repeat=1, stdout_to=None):
    best, out = None, None
    for _ in range(repeat):
        t = time.perf_counter()
        r = subprocess.run(cmd, capture_output=True, env=env)
        dt = time.perf_counter() - t
        if r.returncode != 0:
            return None, r
        if best is None or dt < best:
            best, out = dt, r
    return best, out


def live_traffic(argv_tail, kernel):
    """HBM bytes per launch of `kernel`, measured in THIS run: two child passes of this very command under
    rocprofv3 --pmc (FETCH_SIZE, then WRITE_SIZE: separate passes, counters only, as MI355X_MICROARCH.md's HBM section
    prescribes), a few steps each; mean per dispatch, KiB -> bytes, FETCH_SIZE doubled (gfx950 counts a 128-byte request
    as 64 bytes), WRITE_SIZE as it is (both factors re-measured on a 2 GiB copy by tools/profile_round.sh:
    profiles/*traffic.json "calibration").  None when rocprofv3 is not there or a pass fails."""
    import csv
    import glob
    import shutil
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    got = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="pp_pmc_", dir="/tmp")
        try:
            cmd = [exe, "--pmc", counter, "--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__),
                   "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-e2e", "--no-live-traffic", "--no-second-layout", "--no-other-configs"] + argv_tail
            r = subprocess.run(cmd, capture_output=True, timeout=240, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"))
            if r.returncode != 0:
                return None
            vals = []
            for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                with open(path) as f:
                    for row in csv.DictReader(f):
                        kn = row["Kernel_Name"].replace("pp::", "")  # (k_tile's instances are templates: "k_tile_direct<5, true>(...)")
                        if row["Counter_Name"] == counter and (kernel + "(" in kn or kernel + "<" in kn):
                            vals.append(float(row["Counter_Value"]))
            if not vals:
                return None
            got[counter] = sum(vals) / len(vals) * 1024.0
        except (subprocess.TimeoutExpired, OSError, KeyError, ValueError):
            return None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return {"hbm_bytes": int(round(2.0 * got["FETCH_SIZE"] + 1.0 * got["WRITE_SIZE"])), "fetch_raw_bytes": int(got["FETCH_SIZE"]),
            "write_raw_bytes": int(got["WRITE_SIZE"])}


def filter_roofline(ctx, pp, device, n_pairs, G, read_len=150, reps=5):
    """The filter kernels (seam A: k_filter_reads -- one pass over the reads: ends from the runs, samples, verdicts -- and
    k_filter_listed for the reads whose verdicts need the thresholds; src/alignment.rs:138-149, src/filter.rs:155-167,189-218,
    352-377) on a resident input of BASELINE.json configs[1]'s shape: n_pairs read pairs, one alignment per mate and file
    (2 n_pairs alignments), forward/reverse at ~350 bp.  SURVEY 8d's algorithmic bytes: 17 B per alignment (ref_id, ref_start,
    ref_end, flags in, one pass byte out) + 8 B per pair (the two group offsets).  Kernel times: HIP events on the library's
    stream (pp_filter_kernel_times), mean of `reps` runs; the copies of the verdicts to the host are outside them."""
    import ctypes as C
    g = torch.Generator(device=device)
    g.manual_seed(4711)
    n = n_pairs
    i32, i64 = torch.int32, torch.int64
    keep = []

    def ffile(start, flags, with_ref_end):
        t = dict(ref_id=torch.zeros(n, dtype=i32, device=device), ref_start=start.to(i32), flags=flags.to(i32),
                 cig_off=torch.arange(n, dtype=i64, device=device), n_cig=torch.ones(n, dtype=i32, device=device),
                 cigar=torch.full((n,), (read_len << 4) | 0, dtype=i32, device=device), read=torch.arange(n, dtype=i32, device=device),
                 grp_off=torch.arange(n + 1, dtype=i32, device=device), grp_idx=torch.arange(n, dtype=i32, device=device),
                 ref_end=(start + read_len).to(i64))
        keep.append(t)
        return pp.FilterFile(n, t["ref_id"].data_ptr(), t["ref_start"].data_ptr(), t["flags"].data_ptr(), t["cig_off"].data_ptr(),
                             t["n_cig"].data_ptr(), t["cigar"].data_ptr(), n, t["read"].data_ptr(), t["grp_off"].data_ptr(),
                             t["grp_idx"].data_ptr(), t["ref_end"].data_ptr() if with_ref_end else None)
    s1 = torch.randint(0, max(G - 1000, 1), (n,), device=device, generator=g)
    ins = torch.clamp((350 + 35 * torch.randn(n, device=device, generator=g)).round().long(), 160, 700)
    s2 = s1 + ins - read_len
    fwd_first = torch.rand(n, device=device, generator=g) < 0.5           # which mate is the forward one
    orient, insert = np.zeros(n, np.uint8), np.zeros(n, np.uint32)
    p1, p2 = np.zeros(n, np.uint8), np.zeros(n, np.uint8)
    L = pp.lib()

    def run(with_ref_end):
        f1 = ffile(torch.where(fwd_first, s1, s2), torch.where(fwd_first, 0, 16), with_ref_end)
        f2 = ffile(torch.where(fwd_first, s2, s1), torch.where(fwd_first, 16, 0), with_ref_end)
        inp = pp.FilterInput(n, (pp.FilterFile * 2)(f1, f2))
        torch.cuda.synchronize()
        ctx.set_profiling(1)
        tot = {}
        try:
            for rep in range(reps + 1):
                ctx._chk(L.pp_filter_begin(ctx._h, C.byref(inp), pp.MEM_DEVICE))
                ctx._chk(L.pp_filter_samples(ctx._h, orient.ctypes.data, insert.ctypes.data))
                lo, hi = int(np.percentile(insert, 0.1)), int(np.percentile(insert, 99.9))
                ctx._chk(L.pp_filter_pairs(ctx._h, lo, hi, 0, p1.ctypes.data, p2.ctypes.data))
                kt = pp.KernelTimes()
                L.pp_filter_kernel_times(ctx._h, C.byref(kt))
                if rep:  # (the first run allocates)
                    for k, v in kt.as_dict()["ms"].items():
                        tot[k] = tot.get(k, 0.0) + v
        finally:
            ctx.set_profiling(0)
        return tot
    # the input as the ABI's general form hands it over (CIGAR runs: the kernel works the ends out), and with the ends
    # precomputed (pp_filter_file.ref_end: what the device loader, pp_filter_dev_input, hands over -- no CIGAR array is read)
    tot_cigar = run(False)
    tot = run(True)
    ms_cigar = sum(tot_cigar.values()) / reps
    ms = {k: v / reps for k, v in tot.items()}
    total_ms = sum(ms.values())
    b_alg = 17 * 2 * n + 8 * n
    # what the pass reads and writes as the ABI hands the input over (include/polypolish_hip.h, pp_filter_file: 32-bit fields, a
    # group index per alignment): per read 8 B of group offsets, per alignment grp_idx, ref_id, ref_start, flags (4 B each) and
    # ref_end (8 B) in, one verdict byte out; the sample (1 + 4 B) out per read
    b_abi = n * 8 + 2 * n * (4 * 4 + 8 + 1) + n * 5  # (with ref_end: grp_idx, ref_id, ref_start, flags (4 B each) and ref_end (8 B) per alignment)
    return {"bound": "hbm", "kernels": "k_filter_reads ('samples') + k_filter_listed ('pairs': the reads with several alignments; none here)", "kernel_ms": {k: round(v, 4) for k, v in sorted(ms.items())},
            "input": "ends precomputed (pp_filter_file.ref_end, as pp_filter_dev_input hands the input over): no CIGAR array is read",
            "with_cigar_runs_instead_of_ref_end": {"total_ms": round(ms_cigar, 4), "frac": round(b_alg / (ms_cigar * 1e-3) / 1e9 / 8000.0, 4) if ms_cigar else 0.0},
            "total_ms": round(total_ms, 4), "algorithmic_bytes": b_alg, "achieved": round(b_alg / (total_ms * 1e-3) / 1e9, 1) if total_ms else 0.0,
            "peak": 8000.0, "unit": "GB/s", "frac": round(b_alg / (total_ms * 1e-3) / 1e9 / 8000.0, 4) if total_ms else 0.0,
            "bytes_the_abi_makes_them_move": b_abi, "frac_of_peak_on_those": round(b_abi / (total_ms * 1e-3) / 1e9 / 8000.0, 4) if total_ms else 0.0,
            "workload": f"{n} read pairs, one {read_len}M alignment per mate and file ({2 * n} alignments), all pairs unique: configs[1]'s shape",
            "all_pass": bool(p1.all() and p2.all()), "orientation_fr": int((orient == 0).sum())}


def _file_sha(path):
    h = hashlib.sha256()
    with open(path, "rb") as fh:
        for blk in iter(lambda: fh.read(1 << 24), b""):
            h.update(blk)
    return h.hexdigest()


def end_to_end(device, config, lens, coverage, repeat, seed, keep_dir=None, recipe="survey"):
    """SAM text in, FASTA out (reference contract: src/main.rs:112-126 -> src/polish.rs:26-38, 196-203; src/filter.rs:
    26-37), on files of BASELINE.json configs[config]'s shape generated here: bin/polypolish against the oracle's CLI on
    one core, sha256 of every output compared.  Page-cache-warm files.
      configs[0], [1]  polish (device tokenizer, host ingest), filter, fused filter-polish
      configs[2]       the same on real all-hits files (primary + secondary records with SEQ "*", both strands), plus the
                       two-command flow the configuration is about: `filter`, then `polish` on its tagged outputs
      configs[3], [4]  polish (13 GB / 33 GB of text) -- the filter is not part of these configurations"""
    exe = os.path.join(ROOT, "bin", "polypolish")
    orc_exe = os.path.join(ROOT, "oracle", "_build", "pp_oracle")
    if not (os.path.exists(exe) and os.path.exists(orc_exe) and synthjob.samgen_lib() is not None):
        return {"skipped": "bin/polypolish, oracle/_build/pp_oracle or tools/_build/libsamgen.so is missing"}
    tmp = keep_dir or tempfile.mkdtemp(prefix="pp_e2e_", dir=os.environ.get("TMPDIR", "/tmp"))
    os.makedirs(tmp, exist_ok=True)
    big = config in (3, 4)
    t0 = time.perf_counter()
    job = make_job(device, contig_lens=lens, coverage=coverage, seed=seed, pairs=True, unaligned_frac=1e-3, repeat=repeat,
                   recipe=recipe)
    if device.type == "cuda":
        torch.cuda.synchronize()
    fa, sams = write_sam_pair(job, tmp, qual=True)
    n_rec, genome, planted = job["sam"]["n"], job["G"], job["planted"]
    n_secondary = int((job["sam"]["seq_len"] == 0).sum().item())
    del job
    if device.type == "cuda":
        torch.cuda.empty_cache()
    gen_s = time.perf_counter() - t0
    text_bytes = sum(os.path.getsize(p) for p in sams)
    sha = lambda b: hashlib.sha256(b).hexdigest()
    out = {"files": f"configs[{config}]: {genome / 1e6:g} Mbp FASTA ({len(lens)} contig{'s' if len(lens) > 1 else ''}) + 2 SAM files, "
                    f"{n_rec} records ({n_secondary} secondary with SEQ '*'), {text_bytes / 1e9:.2f} GB of text "
                    f"(QUAL strings included), recipe {recipe} {planted}, generated in {gen_s:.1f} s",
           "host_cores": os.cpu_count(), "text_bytes": text_bytes}
    rep = 2
    try:
        env = dict(os.environ)
        env["PP_DEVICE_INGEST"] = "1"
        t_dev, r_dev = _timed([exe, "polish", fa] + sams, env, repeat=rep)
        env["PP_DEVICE_INGEST"] = "0"
        t_host, r_host = (None, None) if config == 4 else _timed([exe, "polish", fa] + sams, env, repeat=rep)
        t_cpu, r_cpu = _timed([orc_exe, "polish", fa] + sams)
        bad = [r for t, r in ((t_dev, r_dev), (t_host, r_host), (t_cpu, r_cpu)) if t is None and r is not None]
        if bad:
            out["error"] = bad[0].stderr.decode(errors="replace")[-400:]
            return out
        want = sha(r_cpu.stdout)
        out["polish"] = {"wall_s": round(t_dev, 3), "mbp_per_s": round(genome / 1e6 / t_dev, 2), "ingest": "device tokenizer (default)",
                         "parity": sha(r_dev.stdout) == want}
        if t_host is not None:
            out["polish_host_ingest"] = {"wall_s": round(t_host, 3), "mbp_per_s": round(genome / 1e6 / t_host, 2),
                                         "parity": sha(r_host.stdout) == want}
        out["oracle_polish"] = {"wall_s": round(t_cpu, 2), "mbp_per_s": round(genome / 1e6 / t_cpu, 4), "cores": 1,
                                "sha256": want[:16]}
        # where the default command's wall time goes: its own stage lines (PP_TIMING=1: seconds since the process started;
        # the tokenizer's stages end in a stream synchronisation each, so this run is a little slower than the timed one)
        t_st, r_st = _timed([exe, "polish", fa] + sams, dict(env, PP_DEVICE_INGEST="1", PP_TIMING="1"))
        if t_st is not None:
            out["polish_stages"] = {"wall_s": round(t_st, 3),
                                    "lines": [l for l in r_st.stderr.decode(errors="replace").splitlines() if l.startswith("[timing]")]}
        out["speedup_polish"] = round(t_cpu / min(t for t in (t_dev, t_host) if t is not None), 1)
        if not big:
            # What the default SEQ layout costs the tokenizer, read off its own stage timers (PP_TIMING=1: a synchronisation
            # after every stage): "window layout" = the multisplit of the rooms into the windows, against "file-order layout" =
            # the scan of the rooms it replaces; the 4-bit mirror is written by the pass that writes the bytes.  And the same
            # command in the layout of rounds 1-3 (SEQ bytes in file order, no mirror): same bytes out.
            env["PP_DEVICE_INGEST"] = "1"
            def stages(r):
                lines = [l for l in r.stderr.decode(errors="replace").splitlines() if l.startswith("[timing]   tokenizer:")]
                tot = {}
                for l in lines:
                    m = re.match(r"\[timing\]\s+tokenizer:\s+(.*?)\s+([0-9.]+) s\b", l)
                    if m:
                        tot[m.group(1)] = tot.get(m.group(1), 0.0) + float(m.group(2))
                return {k: round(1e3 * v, 3) for k, v in tot.items()}
            t_w, r_w = _timed([exe, "polish", fa] + sams, dict(env, PP_TIMING="1"), repeat=rep)
            t_f, r_f = _timed([exe, "polish", fa] + sams, dict(env, PP_SEQ_LAYOUT="file", PP_SEQ4="0", PP_WO="0", PP_TIMING="1"), repeat=rep)
            t_file, r_file = _timed([exe, "polish", fa] + sams, dict(env, PP_SEQ_LAYOUT="file", PP_SEQ4="0", PP_WO="0"), repeat=rep)
            if t_w is not None and t_f is not None and t_file is not None:
                sw, sf = stages(r_w), stages(r_f)
                out["tokenizer_stages_ms"] = {"default (window-grouped + mirror)": sw, "PP_SEQ_LAYOUT=file PP_SEQ4=0 PP_WO=0": sf,
                                              "layout_extra_ms": round(sw.get("window layout", 0.0) - sf.get("file-order layout", 0.0), 3),
                                              "seq_and_mirror_extra_ms": round(sw.get("seq bytes + mirror", 0.0) - sf.get("seq bytes + mirror", 0.0), 3),
                                              "note": "sums over the two SAM files; every stage ends in a stream synchronisation under PP_TIMING"}
                out["polish_file_order_seq"] = {"wall_s": round(t_file, 3), "parity": sha(r_file.stdout) == want,
                                                "batch": "PP_SEQ_LAYOUT=file PP_SEQ4=0 PP_WO=0: SEQ bytes in the order of the records, no mirrors (rounds 1-3)"}
            del r_w, r_f, r_file
        ok = out["polish"]["parity"] and out.get("polish_host_ingest", {}).get("parity", True)
        del r_dev, r_host, r_cpu
        if not big:
            # the chain filter -> polish: fused in one process against the oracle's two commands
            del env["PP_DEVICE_INGEST"]
            t_fp, r_fp = _timed([exe, "filter-polish", "--in1", sams[0], "--in2", sams[1], fa], env, repeat=rep)
            f1, f2 = os.path.join(tmp, "f_1.sam"), os.path.join(tmp, "f_2.sam")
            g1, g2 = os.path.join(tmp, "g_1.sam"), os.path.join(tmp, "g_2.sam")
            t = time.perf_counter()
            ra = subprocess.run([orc_exe, "filter", "--in1", sams[0], "--in2", sams[1], "--out1", f1, "--out2", f2], capture_output=True)
            t_orc_filter = time.perf_counter() - t
            rb = subprocess.run([orc_exe, "polish", fa, f1, f2], capture_output=True)
            t_chain = time.perf_counter() - t
            # `filter` on its own: two tagged SAM files out (reference contract: src/filter.rs:26-37, 309-349)
            t_f, r_f = _timed([exe, "filter", "--in1", sams[0], "--in2", sams[1], "--out1", g1, "--out2", g2], env, repeat=rep)
            if t_f is not None and ra.returncode == 0:
                with open(g1, "rb") as fh:
                    n_fail = sum(blk.count(b"ZP:Z:fail") for blk in iter(lambda: fh.read(1 << 24), b""))
                out["filter"] = {"wall_s": round(t_f, 3), "parity": _file_sha(g1) == _file_sha(f1) and _file_sha(g2) == _file_sha(f2),
                                 "oracle_wall_s": round(t_orc_filter, 2), "speedup": round(t_orc_filter / t_f, 1),
                                 "records_failed_in_file_1": n_fail}
            else:
                out["filter"] = {"error": (r_f.stderr if t_f is None else ra.stderr).decode(errors="replace")[-400:], "parity": False}
            if t_fp is None or ra.returncode or rb.returncode:
                out["filter_polish"] = {"error": (r_fp.stderr if t_fp is None else (ra.stderr + rb.stderr)).decode(errors="replace")[-400:],
                                        "parity": False}
            else:
                out["filter_polish"] = {"wall_s": round(t_fp, 3), "mbp_per_s": round(genome / 1e6 / t_fp, 2),
                                        "parity": sha(r_fp.stdout) == sha(rb.stdout),
                                        "oracle_chain_wall_s": round(t_chain, 2), "speedup": round(t_chain / t_fp, 1)}
            ok = ok and out["filter"]["parity"] and out["filter_polish"]["parity"]
            if config == 2 and t_f is not None and rb.returncode == 0:
                # the configuration's own flow as two commands: polish the product's tagged files
                t_p2, r_p2 = _timed([exe, "polish", fa, g1, g2], env, repeat=rep)
                out["filter_then_polish"] = ({"error": r_p2.stderr.decode(errors="replace")[-400:], "parity": False} if t_p2 is None else
                                             {"wall_s": round(t_f + t_p2, 3), "parity": sha(r_p2.stdout) == sha(rb.stdout),
                                              "differs_from_unfiltered_polish": sha(rb.stdout) != want,
                                              "oracle_chain_wall_s": round(t_chain, 2), "speedup": round(t_chain / (t_f + t_p2), 1)})
                ok = ok and out["filter_then_polish"]["parity"]
        out["parity"] = bool(ok)
    finally:
        if keep_dir is None:
            for p in os.listdir(tmp):
                os.unlink(os.path.join(tmp, p))
            os.rmdir(tmp)
    return out


def _error_line(world, args, msg):
    """The contract's JSON line when the run cannot produce a value (a communicator that does not come up, a collective that
    hangs): the driver gets a parseable record with an "error" field instead of a silent hang or a traceback."""
    return json.dumps({"metric": METRIC, "value": None, "unit": "Mbp/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                       "ms_per_step": None, "higher_is_better": True, "scaling": "strong" if args.config in (3, 4) and world > 1 else "weak",
                       "vs_baseline": None, "dtype": "u8/u32 counts + f64 depth", "data": "synthetic",
                       "config": {"workload": f"configs[{args.config}]"}, "error": msg})


class _Watchdog:
    """A collective that hangs (one rank gone, a fabric problem) cannot be caught: a timer thread prints the error line on
    rank 0 and ends the process.  arm(seconds, what) before the risky stretch, disarm() after it."""

    def __init__(self, world, rank, args):
        import threading
        self._t, self._threading, self.world, self.rank, self.args = None, threading, world, rank, args

    def arm(self, seconds, what):
        self.disarm()
        def fire():
            if self.rank == 0:
                print(_error_line(self.world, self.args, f"{what} did not finish within {seconds} s (rank {self.rank} gave up)"), flush=True)
            os._exit(3)
        self._t = self._threading.Timer(seconds, fire)
        self._t.daemon = True
        self._t.start()

    def disarm(self):
        if self._t is not None:
            self._t.cancel()
            self._t = None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=1, choices=[0, 1, 2, 3, 4],
                    help="BASELINE.json configs index: 1 (default, the metric's configuration), 2 repeats / all-hits, "
                         "3 100-contig metagenome, 4 one 250 Mbp contig")
    ap.add_argument("--genome", type=int, default=None, help="override the configuration's total assembly length (bp)")
    ap.add_argument("--coverage", type=int, default=None, help="override the configuration's coverage")
    ap.add_argument("--cpu-sample", type=int, default=5_000_000,
                    help="bp of one contig given to the CPU oracle (default: the whole 5 Mbp job of config 1, ~12 s)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--seq-pitch", type=int, default=-1,
                    help="bytes of the seq array per record: -1 (default) = as the product's ingests lay it out, every record's SEQ on a "
                         "32-byte boundary (PP_SEQ_ALIGN); 0 = packed back to back (the bench lines of rounds 1-3)")
    ap.add_argument("--seq4", default="on", choices=["on", "off"],
                    help="hand the 4-bit mirror of the seq array over with the batch (pp_aln_batch.seq4), as the device tokenizer does "
                         "(default; 'off' = PP_SEQ4=0)")
    ap.add_argument("--wo", default="on", choices=["on", "off"],
                    help="hand the window-order mirror of the records over with the batch (pp_aln_batch.wo), as both ingests do "
                         "(default; 'off' = PP_WO=0: the bucketing kernels read the records in file order)")
    ap.add_argument("--seq-layout", default="window", choices=["file", "window"],
                    help="'window' (default): the SEQ bytes of a SAM file window-grouped, as both ingests lay them out since round 4 "
                         "(PP_SEQ_WINDOW_GROUPED; tools/synthjob.py window_grouped); 'file' = in the order of the records "
                         "(PP_SEQ_LAYOUT=file: the layout of rounds 1-3)")
    ap.add_argument("--no-second-layout", action="store_true", help="skip the second roofline entry (SEQ bytes in file order, no mirror: rounds 1-3)")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end CLI leg (one GPU)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the compact entries of configs[2], [3], [4] and the no-speculation step (child runs of this command)")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="do not measure roofline.traffic with rocprofv3 child passes (the committed figure is used if the workload matches)")
    ap.add_argument("--e2e-only", action="store_true", help="only the end-to-end CLI leg of --config (no kernel bench)")
    ap.add_argument("--recipe", default="survey", choices=["survey", "subs"],
                    help="assembly errors: 'survey' = SURVEY.md 8d (1/3 substitutions, 1/3 1-bp deletions, 1/3 1-bp insertions, "
                         "half of the indels in homopolymers; reads aligned to the assembly with the resulting I/D CIGARs), "
                         "'subs' = substitutions only (the recipe of rounds 1 and 2)")
    ap.add_argument("--e2e-dir", default=None, help="keep the end-to-end files in this directory")
    ap.add_argument("--indel-frac", type=float, default=None,
                    help="fraction of reads with a 1-bp SEQUENCING indel (default: the recipe's -- survey 0.0015 = 1e-5 per base, subs 0.01)")
    ap.add_argument("--sub-rate", type=float, default=0.002, help="experiments only: per-base substitution rate")
    ap.add_argument("--n-rate", type=float, default=1e-4, help="experiments only: per-base N rate")
    ap.add_argument("--read-len", type=int, default=150, help="experiments only: read length (coverage is kept)")
    ap.add_argument("--repeat-bp", type=int, default=0,
                    help="experiments only: reads starting in a region of this many bp get depth share 1/5 "
                         "(order-dependent f64 depth -> exact replay kernel) without the extra records of config 2")
    ap.add_argument("--collective-timeout", type=float, default=float(os.environ.get("PP_BENCH_COLLECTIVE_TIMEOUT", "180")),
                    help="N > 1: seconds a communicator set-up or one step's gather may take before the run is given up with an "
                         "\"error\" line (a hung collective cannot be caught any other way)")
    ap.add_argument("--nd-frac", type=float, default=0.0,
                    help="experiments only: this fraction of the reads gets depth share 1/3 (every window then has "
                         "order-dependent depths: the worst case of the exact replay)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    # PP_BENCH_SHARE_GPU=1 (testing on a one-GPU box only): every rank uses GPU 0 and the gather goes
    # through gloo on host copies, so that the N>1 control flow can be exercised without N GPUs
    share = os.environ.get("PP_BENCH_SHARE_GPU") == "1"
    dev_index = 0 if share else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    dist = None
    dog = _Watchdog(world, rank, args)
    if world > 1:
        import datetime
        import torch.distributed as dist
        os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "1")
        os.environ.setdefault("NCCL_ASYNC_ERROR_HANDLING", "1")
        dog.arm(args.collective_timeout, "setting up the process group")
        try:
            if share:
                dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=args.collective_timeout))
            else:
                dist.init_process_group("nccl", device_id=device, timeout=datetime.timedelta(seconds=args.collective_timeout))
        except Exception as e:  # noqa: BLE001 -- whatever it is, the driver gets a line
            if rank == 0:
                print(_error_line(world, args, f"init_process_group failed: {e}"), flush=True)
            raise SystemExit(3)
        dog.disarm()

    import polypolish_amd as pp
    ctx = pp.Context(dev_index)
    # The resident job is laid out exactly as the library's ingests lay a batch out (tests/test_synthjob_cpu.py), but its
    # window-order mirror is built here, not by them: told so, the library takes it as one of its own instead of comparing
    # it with the arrays on every step (what that comparison costs a foreign caller: "foreign_mirror_check_ms" below).
    ctx.trust_mirrors(True)

    lens, coverage, repeat, label = config_shape(args.config, args.genome, args.coverage)
    G_total = int(sum(lens))
    strong = world > 1 and args.config in (3, 4)
    if args.indel_frac is None:
        args.indel_frac = synthjob.SURVEY_INDEL_READ_FRAC if args.recipe == "survey" else 0.01
    default_shape = (args.seq_layout == "window" and args.seq_pitch == -1 and args.seq4 == "on" and args.wo == "on" and args.recipe == "survey" and args.indel_frac == synthjob.SURVEY_INDEL_READ_FRAC and args.sub_rate == 0.002 and args.n_rate == 1e-4 and args.read_len == 150 and
                     args.repeat_bp == 0 and args.nd_frac == 0.0 and args.genome is None and args.coverage is None)
    # config 1: this rank's own 5 Mbp contig (seed differs per rank); configs 3 / 4 with N > 1: every rank builds
    # the same job and keeps its shard
    if args.e2e_only:
        if world > 1:
            raise SystemExit("--e2e-only is a one-GPU leg")
        out = {"metric": METRIC, "config": {"workload": label + " -- end-to-end leg only (SAM text -> FASTA through bin/polypolish)"},
               "n_gpus": 1, "data": "synthetic",
               "e2e": end_to_end(device, args.config, lens, coverage, repeat, seed=4242 + args.config, keep_dir=args.e2e_dir,
                                 recipe=args.recipe)}
        print(json.dumps(out))
        return
    job = make_job(device, contig_lens=lens, coverage=coverage, read_len=args.read_len,
                   seed=42 + args.config + 1 + (0 if strong else 1000 * rank),
                   indel_read_frac=args.indel_frac, sub_rate=args.sub_rate, n_rate=args.n_rate, repeat=repeat,
                   repeat_bp=args.repeat_bp, recipe=args.recipe, seq_pitch=None if args.seq_pitch < 0 else args.seq_pitch,
                   seq_layout=args.seq_layout)
    G_total = job["G"]  # the assembly's length (the truth's +- the planted indels)
    if args.nd_frac > 0:
        gg = torch.Generator(device=device)
        gg.manual_seed(7)
        nd = torch.rand(job["n_aln"], device=device, generator=gg) < args.nd_frac
        job["recs"]["k"] = torch.where(nd, 3, job["recs"]["k"]).int().contiguous()
    if args.seq4 == "on":   # the mirror the device tokenizer hands over with its batch (and pp_polish_add packs for any other)
        job = synthjob.with_seq4(job)
    if args.wo == "on":     # the window-order mirror of the records both ingests hand over with their batch (pp_aln_batch.wo)
        job = synthjob.with_wo(job)
    torch.cuda.synchronize()   # (the library runs on a stream of its own: the job's arrays have to be written by now)
    plan = None
    if strong:
        # ONE job; a rank keeps the records that reach its units (pp_shard_plan_create: whole contigs by longest-
        # processing-time, the single contig in one window per rank; pp_shard_split picks the records, on the device)
        # and polishes them with the ranges of its units (pp_polish_set_emit)
        counts = torch.bincount(job["recs"]["contig"].long(), minlength=len(lens)).cpu().numpy()
        plan = pp.Plan(job["contig_off"], counts, world)
        whole_job = job
        job = shard_of(ctx, pp, whole_job, plan, rank)
    torch.cuda.synchronize()
    nc_job = len(job["contig_off"]) - 1
    gdev = "cpu" if share else device
    g_max = job["G"]
    if world > 1:  # weak scaling: the ranks' contigs differ in length by their planted indels -- one buffer size for all
        t = torch.tensor([g_max], dtype=torch.int64, device=gdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        g_max = int(t.item())
    cap = g_max + g_max // 16 + (1 << 16)          # polished bytes of one rank, at most
    total_cap = cap if strong else world * cap
    rank_lens = rank_offs = None
    gbuf = None
    if world > 1 and not share:
        # the exchange of the path, inside the library: pp_polish_gather (RCCL over xGMI)
        dog.arm(args.collective_timeout, "pp_comm_init (ncclCommInitRank)")
        try:
            ident = [pp.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(ident, src=0, device=device)
            ctx.comm_init(rank, world, ident[0])
        except Exception as e:  # noqa: BLE001
            if rank == 0:
                print(_error_line(world, args, f"pp_comm_init failed: {e}"), flush=True)
            os._exit(3)
        dog.disarm()
        gbuf = torch.zeros(total_cap, dtype=torch.uint8, device=device) if rank == 0 else None
    elif world > 1:
        # one GPU shared by all ranks (tests): RCCL refuses that, the bytes travel over gloo
        sbuf = torch.zeros(cap, dtype=torch.uint8, device=device)
        gathered = [torch.empty(cap, dtype=torch.uint8) for _ in range(world)] if rank == 0 else None
    last = {}
    split_s = {"compute": 0.0, "gather": 0.0, "n": 0}   # host clock around the two halves of a step (both end in a stream synchronisation)

    run_step = prepared(ctx, pp, job)

    def step():
        ta = time.perf_counter()
        run_step()
        tb = time.perf_counter()
        if world > 1 and not share:
            try:
                last["lens"], last["offs"] = ctx.gather(gbuf.data_ptr() if rank == 0 else None, total_cap if rank == 0 else 0)
            except Exception as e:  # noqa: BLE001 -- (a rank that backed out alone would leave the others in ncclSend: everybody leaves)
                if rank == 0:
                    print(_error_line(world, args, f"pp_polish_gather failed: {e}"), flush=True)
                os._exit(3)
        elif world > 1:
            pp.lib().pp_polish_result(ctx._h, sbuf.data_ptr(), pp.MEM_DEVICE, None, None)
            src = sbuf.cpu()
            if rank == 0:
                gathered[0].copy_(src)
                ops = [dist.P2POp(dist.irecv, gathered[r], r) for r in range(1, world)]
            else:
                ops = [dist.P2POp(dist.isend, src, 0)]
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        split_s["compute"] += tb - ta
        split_s["gather"] += time.perf_counter() - tb
        split_s["n"] += 1

    # Timed region: only the dominant kernel carries an event pair (on the library's stream), so that the
    # timers do not perturb what `value` measures.  The per-group breakdown (kernel_ms_per_step) comes from
    # a few extra, untimed steps afterwards with every kernel group under HIP events.
    ctx.set_profiling(0)
    # (one watchdog over all the steps, not one per step: a timer thread per step would sit inside the timed region)
    if world > 1:
        dog.arm(args.collective_timeout + 2.0 * (args.warmup + args.steps + 8), "the steps' gathers (pp_polish_gather)")
    for _ in range(args.warmup):
        step()
    ctx.set_profiling(2)
    dom_ms = []
    dom_name = None
    split_s.update(compute=0.0, gather=0.0, n=0)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        kt = ctx.first_kernel_ms()   # (the one event pair of profiling level 2: name and milliseconds, no dictionaries)
        if kt is not None:
            dom_name = kt[0]
            dom_ms.append(kt[1])
    if isinstance(dom_name, bytes):
        dom_name = dom_name.decode()
    ctx.sync()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=gdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    timed_split = dict(split_s)   # the timed steps' compute / gather halves on this rank
    # the kernel the timed steps ran: k_tile_direct when the job took its bulk straight from the window-order mirror (round 5)
    direct_path = ctx.took_direct_path()
    dom_kernel = None if dom_name is None else ("k_" + dom_name + ("_direct" if direct_path and dom_name == "tile" else ""))
    dog.disarm()
    if world > 1:
        dog.arm(args.collective_timeout + 60.0, "the untimed steps and the verification of the gathered bytes")
    ctx.set_profiling(1)
    all_ms, n_break, work = {}, 5, {}
    for _ in range(n_break):
        step()
        kt = ctx.kernel_times()
        work = {"work_items": kt["n_entries"], "positions_replayed_exactly": kt["n_flagged"], "pipeline_passes": kt["n_passes"]}
        for k, v in kt["ms"].items():
            all_ms[k] = all_ms.get(k, 0.0) + v
    ctx.set_profiling(0)

    gather_ok = None
    per_rank = None
    one_gpu_ms = None
    if world > 1:
        # what every rank did, so that a multi-GPU record can be read: its kernels, and its steps split into compute (begin /
        # add / finish: ends in a stream synchronisation) and the gather
        mine_diag = {"rank": rank, "records": job["part"].n_aln if strong else job["n_aln"],
                     "kernel_ms_per_step": {k: round(v / max(n_break, 1), 4) for k, v in sorted(all_ms.items())},
                     "compute_ms_per_step": round(1e3 * timed_split["compute"] / max(timed_split["n"], 1), 4),
                     "gather_ms_per_step": round(1e3 * timed_split["gather"] / max(timed_split["n"], 1), 4)}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine_diag)
        # every rank's polished bytes must have arrived on rank 0 unchanged, and -- strong scaling -- put back in
        # assembly order they must be the bytes ONE GPU produces for the whole job
        mine, my_offs, _ = ctx.result()
        meta = [None] * world
        dist.all_gather_object(meta, (hashlib.sha256(mine).hexdigest(), len(mine), [int(x) for x in my_offs]))
        if rank == 0:
            if share:
                rank_bytes = [bytes(gathered[r][:meta[r][1]].numpy()) for r in range(world)]
            else:
                host = gbuf.cpu().numpy()
                starts = np.concatenate([[0], np.cumsum(last["lens"].astype(np.int64))])
                rank_bytes = [host[int(starts[r]):int(starts[r + 1])].tobytes() for r in range(world)]
            gather_ok = all(len(rank_bytes[r]) == meta[r][1] and hashlib.sha256(rank_bytes[r]).hexdigest() == meta[r][0]
                            for r in range(world))
            if strong:
                whole, _ = plan.assemble(rank_bytes, [np.array(m[2], dtype=np.uint64) for m in meta])
                full = dict(whole_job)
                full["emit"] = None
                run_job(ctx, pp, full)
                ref, _, _ = ctx.result()
                gather_ok = bool(gather_ok and whole == ref)
                # ... and what the whole job takes on this ONE GPU: the compute side of the scaling curve
                run_job(ctx, pp, full)
                t1 = time.perf_counter()
                for _ in range(3):
                    run_job(ctx, pp, full)
                one_gpu_ms = 1e3 * (time.perf_counter() - t1) / 3
                run_job(ctx, pp, job)  # leave the rank's own result in the context for the report below
    G_all_ranks = G_total
    if world > 1 and not strong:  # weak scaling: every rank's own contig (their lengths differ by the planted indels)
        t = torch.tensor([G_total], dtype=torch.int64, device=gdev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        G_all_ranks = int(t.item())
    dog.disarm()
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    ms_per_step = 1e3 * elapsed / args.steps
    total_mbp = (G_total if strong else G_all_ranks) / 1e6
    value = total_mbp / (elapsed / args.steps)
    b_alg = algorithmic_bytes(job)
    if strong:  # this rank's share of the one job: its records, its positions
        e = job["emit"].astype(np.int64)
        b_alg = job["part"].n_aln * (job["read_len"] + 16) + 4 * job["part"].n_cig_total + 2 * int((e[:, 1] - e[:, 0]).sum())
    dom_avg_ms = float(np.mean(dom_ms)) if dom_ms else 0.0
    achieved = b_alg / (dom_avg_ms * 1e-3) / 1e9 if dom_avg_ms > 0 else 0.0
    peak = 8000.0
    polished, offs, stats = ctx.result()
    recovered = None
    if not strong and not repeat:  # (repeat copies that differ by SNPs are out-voted by their siblings' reads: not expected)
        recovered = synthjob.recovered(job, polished, offs)

    # HBM traffic of the dominant kernel (PMC counters): measured in this run by two rocprofv3 child passes of this very
    # command (N = 1); failing that, the committed figure of the same workload (profiles/traffic.json), else null.
    traffic = traffic_source = None
    if world == 1 and dom_name and not args.no_live_traffic:
        tail = ["--config", str(args.config), "--seq-pitch", str(args.seq_pitch), "--seq4", args.seq4, "--wo", args.wo, "--seq-layout", args.seq_layout, "--recipe", args.recipe, "--indel-frac", repr(args.indel_frac), "--sub-rate", repr(args.sub_rate),
                "--n-rate", repr(args.n_rate), "--read-len", str(args.read_len), "--repeat-bp", str(args.repeat_bp), "--nd-frac", repr(args.nd_frac)]
        if args.genome is not None:
            tail += ["--genome", str(args.genome)]
        if args.coverage is not None:
            tail += ["--coverage", str(args.coverage)]
        lt = live_traffic(tail, dom_kernel)
        if lt is not None:
            traffic = lt["hbm_bytes"]
            traffic_source = ("measured in this run: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE child passes of this command (3 steps "
                              f"each), mean per dispatch; 2 x FETCH_SIZE ({lt['fetch_raw_bytes']} B raw) + WRITE_SIZE ({lt['write_raw_bytes']} B)")
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if traffic is None and default_shape and args.config == 1 and dom_name and os.path.exists(tpath):
        with open(tpath) as f:
            kk = json.load(f).get("kernels", {})
            traffic = next((v.get("hbm_bytes") for k, v in kk.items() if k == dom_kernel or k.startswith(str(dom_kernel) + "<")), None)
        if traffic is not None:
            traffic_source = "profiles/traffic.json (committed rocprofv3 --pmc passes of this command, not this run)"

    second = None
    if world == 1 and args.seq_layout == "window" and args.seq4 == "on" and args.wo == "on" and args.seq_pitch < 0 and not args.no_second_layout:
        # The same job as rounds 1-3 measured it (and as the ingests still deliver it with PP_SEQ_LAYOUT=file PP_SEQ4=0): SEQ bytes
        # in the order of the records, no 4-bit mirror -- the lane-group plain class over the bytes.  Measured like the headline.
        def other_layout(wj, what, tail):
            ctx.set_profiling(0)
            for _ in range(args.warmup):
                run_job(ctx, pp, wj)
            ctx.set_profiling(2)
            w_ms = []
            torch.cuda.synchronize()
            tw = time.perf_counter()
            for _ in range(args.steps):
                run_job(ctx, pp, wj)
                kt = ctx.kernel_times()["ms"]
                if kt:
                    w_ms.append(next(iter(kt.values())))
            ctx.sync()
            torch.cuda.synchronize()
            w_step = 1e3 * (time.perf_counter() - tw) / args.steps
            ctx.set_profiling(0)
            w_polished, _, _ = ctx.result()
            w_kernel = float(np.mean(w_ms)) if w_ms else 0.0
            w_traffic = None
            if dom_name and not args.no_live_traffic:
                lt = live_traffic(["--config", str(args.config), "--seq-pitch", str(args.seq_pitch), "--recipe", args.recipe,
                                   "--indel-frac", repr(args.indel_frac)] + tail, "k_" + dom_name)
                w_traffic = lt["hbm_bytes"] if lt else None
            return {"layout": what, "kernel": "k_" + (dom_name or "?"), "kernel_ms": round(w_kernel, 4),
                    "achieved": round(b_alg / (w_kernel * 1e-3) / 1e9, 1) if w_kernel else 0.0, "peak": peak, "unit": "GB/s",
                    "frac": round(b_alg / (w_kernel * 1e-3) / 1e9 / peak, 4) if w_kernel else 0.0, "traffic": w_traffic,
                    "ms_per_step": round(w_step, 4), "mbp_per_s": round(G_total / 1e6 / (w_step * 1e-3), 1) if w_step else None,
                    "same_polished_bytes": bool(w_polished == polished)}
        wj = synthjob.with_wo(synthjob.with_seq4(synthjob.file_ordered(job), on=False), on=False)
        second = other_layout(wj, "SEQ bytes in the order of the records, no 4-bit mirror of the seq array, no window-order mirror of the "
                                  "records (PP_SEQ_LAYOUT=file PP_SEQ4=0 PP_WO=0: the resident batch of rounds 1-3; every result unchanged)",
                              ["--seq-layout", "file", "--seq4", "off", "--wo", "off"])
        del wj
    # What the dominant kernel actually moves, as a rate: the algorithmic fraction above prices a 150-byte read at 150
    # bytes, the memory system fetches the 128-byte lines it touches (2.16 of them at an arbitrary byte offset).  6290 GB/s
    # is the copy rate an MI355X reaches in practice (MI355X_MICROARCH.md, chip-level parameters).
    PRACTICAL = 6290.0
    def moved(traffic_bytes, kernel_ms):
        if not traffic_bytes or not kernel_ms:
            return None
        rate = traffic_bytes / (kernel_ms * 1e-3) / 1e9
        return {"rate": round(rate, 1), "unit": "GB/s", "frac_of_peak": round(rate / peak, 4),
                "frac_of_practical_copy_rate": round(rate / PRACTICAL, 4), "practical_copy_rate": PRACTICAL,
                "bytes_moved_per_algorithmic_byte": round(traffic_bytes / b_alg, 3)}
    if second is not None:
        second["hbm_actual"] = moved(second.get("traffic"), second.get("kernel_ms"))
    out = {
        "metric": METRIC,
        "value": round(value, 2),
        "unit": "Mbp/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True,
        "scaling": "strong" if strong else "weak",
        "vs_baseline": None,
        "dtype": "u8/u32 counts + f64 depth",
        "data": "synthetic",
        "config": {"workload": label + f" alignment records resident in HBM ({job['n_aln']} records"
                                       f"{' in total' if strong else ''}; recipe '{args.recipe}': assembly errors {job['planted']}, "
                                       f"reads aligned to the assembly (I/D runs over the planted indels), {100 * args.indel_frac:g}% with a 1-bp sequencing indel"
                                       + ("; resident as the product's ingests lay a batch out (include/polypolish_hip.h): records in SAM order, every "
                                          "record's SEQ on a 32-byte boundary of the seq array, the SEQ bytes of each of the two SAM files "
                                          "window-grouped (PP_SEQ_WINDOW_GROUPED, the default of pp_ingest_* / pp_dev_ingest_* since round 4)"
                                          if args.seq_layout == "window" and args.seq_pitch < 0 else
                                          ("; SEQ bytes in the order of the records (PP_SEQ_LAYOUT=file: not the default layout)" if args.seq_pitch < 0
                                           else ("; SEQ packed back to back in file order" if args.seq_pitch == 0 else f"; SEQ pitch {args.seq_pitch} bytes, file order")))
                                       + ("; with the 4-bit mirror of the seq array the device tokenizer hands over (pp_aln_batch.seq4)" if args.seq4 == "on"
                                          else "; WITHOUT the 4-bit mirror (PP_SEQ4=0: not the default batch)")
                                       + ("; with the window-order mirror of the records both ingests hand over (pp_aln_batch.wo)" if args.wo == "on"
                                          else "; WITHOUT the window-order mirror of the records (PP_WO=0: not the default batch)") + ")",
                   "parallelism": (("contig-shard" if len(lens) > 1 else "window-tile") if strong else "contig-shard") + f" x{world}"
                   if world > 1 else "single GPU",
                   "alignments_per_gpu": job["part"].n_aln if strong else job["n_aln"]},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": peak, "unit": "GB/s",
                     "frac": round(achieved / peak, 4), "traffic": traffic, "traffic_source": traffic_source,
                     "hbm_actual": moved(traffic, dom_avg_ms),
                     "kernel": dom_kernel or "?", "kernel_ms": round(dom_avg_ms, 4), "algorithmic_bytes": b_alg,
                     "path": ("direct: k_prepd (one pass over the window-order mirror: validation, where the windows begin, extras; the records that are not bulk behind its loop) -> k_winplan -> "
                              "k_tile_direct (a window's bulk straight from the mirror) -- no bucketing pass, no work items for 93 % of the records"
                              if direct_path else "bucketing: k_prep -> k_scan_cols / k_scan / k_fill (16-byte work items) -> k_tile"),
                     "whole_path_frac": round(b_alg / (ms_per_step * 1e-3) / 1e9 / peak, 4) if world == 1 else None},
        "roofline_file_order_seq": second,
        "kernel_ms_per_step": {k: round(v / max(n_break, 1), 4) for k, v in sorted(all_ms.items())},
        "work": work,
        "planted_errors_recovered": recovered,
        "gather_verified": gather_ok,
        "per_rank": per_rank,
        "multi_gpu_split": None if per_rank is None else {
            "compute_ms_per_step_max": max(r["compute_ms_per_step"] for r in per_rank),
            "gather_ms_per_step_max": max(r["gather_ms_per_step"] for r in per_rank),
            "whole_job_ms_on_one_gpu": None if one_gpu_ms is None else round(one_gpu_ms, 4),
            "compute_only_speedup": None if one_gpu_ms is None else round(one_gpu_ms / max(max(r["compute_ms_per_step"] for r in per_rank), 1e-9), 3),
            "note": "strong scaling: the whole job on rank 0's GPU alone (after the timed region) over the slowest rank's compute half of a "
                    "step -- the curve without the gather; the driver computes the efficiency with the gather from `value`"},
        "changed_positions": int(sum(s["changed"] for s in stats)),
    }

    if not args.no_cpu_baseline and world == 1:
        from oracle import orc  # the checker / reported baseline -- never the measured product path
        # a bounded sample: the first S bp of the contig with the most records (config 1: the whole job)
        c_best = int(np.argmax(np.diff(job["contig_off"].astype(np.int64))))
        clen = int(job["contig_off"][c_best + 1] - job["contig_off"][c_best])
        S = min(args.cpu_sample * 200 // coverage if args.config != 1 else args.cpu_sample, clen)
        if repeat:  # make sure the sample holds a repeat locus
            S = min(clen, max(S, job["repeat_loci"][0] + 200_000))
        whole = S == job["G"] and len(lens) == 1
        sub = job if whole else subset_job(job, 0, S, c_best)
        torch.cuda.synchronize()
        run_job(ctx, pp, sub)
        got, _, _ = ctx.result()
        host = to_host_records(sub)
        hb = sub["bases"].cpu().numpy()
        off = np.array([0, S], dtype=np.uint64)
        t1 = time.perf_counter()
        want = orc.polish_records(off, hb, host)
        cpu_s = time.perf_counter() - t1
        out["cpu_baseline"] = {
            "value": round(S / 1e6 / cpu_s, 4), "unit": "Mbp/s", "cores": 1, "kind": "port",
            "sample": f"first {S} bp of contig {c_best} with the {sub['n_aln']} records that lie inside it "
                      f"({cpu_s:.1f} s of single-thread CPU work, pileup+vote from parsed records, no text parsing)",
            "parity_on_sample": bool(got == want["polished"]),
        }
        if got != want["polished"]:
            out["cpu_baseline"]["parity_note"] = "MISMATCH between device and oracle on the sample"
    if world == 1 and not args.no_second_layout:
        try:
            out["roofline_filter"] = filter_roofline(ctx, pp, device, int(sum(lens)) * coverage // (2 * args.read_len), int(sum(lens)))
        except Exception as e:  # noqa: BLE001 -- (a report, not the metric)
            out["roofline_filter"] = {"error": str(e)}
    if world == 1:
        # (1) a fresh context's FIRST job of this very batch: what the steady-state step does not pay -- buffer allocation, the
        # contig table's upload, k_meta_init in front, the exact replays' launches not yet left out, a pass that only finds
        # out how much room the windows' extras need.  Host clock around begin / add / finish.
        torch.cuda.synchronize()
        fresh = pp.Context(dev_index)
        fresh.trust_mirrors(True)
        fresh.sync()
        t1 = time.perf_counter()
        run_job(fresh, pp, job)
        first_ms = (time.perf_counter() - t1) * 1e3
        fresh.set_profiling(1)
        t1 = time.perf_counter()
        run_job(fresh, pp, job)
        second_ms = (time.perf_counter() - t1) * 1e3
        passes2 = fresh.kernel_times()["n_passes"]
        fresh.set_profiling(0)
        out["first_job_ms"] = round(first_ms, 3)
        out["first_job_note"] = ("a fresh context's first begin / add / finish of this batch, host clock: device buffers allocated, contig table "
                                 f"uploaded, capacities found (reruns); its second job {second_ms:.3f} ms in {passes2} pass(es), before the replays' "
                                 "launches are left out and the next job's metadata are set up ahead (ms_per_step is the steady state)")
        del fresh
        # (2) what the check of a FOREIGN window-order mirror costs (the same batch, not taken on trust): per step
        ctx.trust_mirrors(False)
        for _ in range(3):
            run_job(ctx, pp, job)
        ctx.sync()
        t1 = time.perf_counter()
        for _ in range(10):
            run_job(ctx, pp, job)
        ctx.sync()
        out["foreign_mirror_check_ms"] = round((time.perf_counter() - t1) * 1e2 - ms_per_step, 4)
        ctx.trust_mirrors(True)
    if world == 1 and not args.no_other_configs and not args.no_live_traffic and default_shape and args.config == 1:
        # (3) the other GPU configurations, compact and driver-timed: child runs of this command (their jobs do not fit next to
        # this one's), and configs[1] once more without the two steady-state shortcuts (PP_SPECULATE=0 PP_INIT_AHEAD=0)
        def child(extra, env=None):
            cmd = [sys.executable, os.path.abspath(__file__), "--no-e2e", "--no-cpu-baseline", "--no-live-traffic", "--no-second-layout",
                   "--no-other-configs", "--steps", "10", "--warmup", "3"] + extra
            try:
                r = subprocess.run(cmd, capture_output=True, timeout=400, env=dict(os.environ, **(env or {})))
                return json.loads(r.stdout.decode().strip().splitlines()[-1])
            except Exception as e:  # noqa: BLE001 -- (a report, not the metric)
                return {"error": str(e)[:200]}
        del job
        torch.cuda.empty_cache()
        job = None
        others = {}
        for c in (2, 3, 4):
            d = child(["--config", str(c)])
            others[f"configs[{c}]"] = d if "error" in d else {
                "workload": d["config"]["workload"][:80], "ms_per_step": d["ms_per_step"], "mbp_per_s": d["value"],
                "kernel": d["roofline"]["kernel"], "kernel_ms": d["roofline"]["kernel_ms"], "frac": d["roofline"]["frac"],
                "whole_path_frac": d["roofline"]["whole_path_frac"], "first_job_ms": d.get("first_job_ms"),
                "parity": ("planted errors recovered: the polished contigs equal the truth" if d.get("planted_errors_recovered")
                           else ("all-hits repeats: planted-error check does not apply" if d.get("planted_errors_recovered") is None else "MISMATCH")),
                "positions_replayed_exactly": d["work"]["positions_replayed_exactly"]}
        out["other_configs"] = others
        d = child([], {"PP_SPECULATE": "0", "PP_INIT_AHEAD": "0"})
        out["ms_per_step_without_steady_state_shortcuts"] = d if "error" in d else {
            "ms_per_step": d["ms_per_step"], "env": "PP_SPECULATE=0 PP_INIT_AHEAD=0",
            "note": "every step launches the exact replays (five kernels) and k_meta_init in front, as a context's first jobs do"}
    if world == 1 and not args.no_e2e:
        job = None
        torch.cuda.empty_cache()
        out["e2e"] = end_to_end(device, args.config, lens, coverage, repeat, seed=4242 + args.config, keep_dir=args.e2e_dir,
                                recipe=args.recipe)
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
