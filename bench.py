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
  e2e          (N=1, config 1) the drop-in CLI from SAM TEXT to FASTA on files generated on the box: wall
               seconds of `polypolish polish` (device tokenizer and host ingest) and `filter-polish`, the
               oracle's CLI on one core on the same files, and whether the output bytes are identical.
"""
from __future__ import annotations

import argparse
import ctypes
import hashlib
import json
import os
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


def make_job(device, contig_lens=(5_000_000,), coverage=200, read_len=150, seed=42, sub_rate=0.002, n_rate=1e-4,
             asm_sub_rate=1e-4, indel_read_frac=0.01, repeat=None, repeat_bp=0, repeat_k=5, pairs=False,
             unaligned_frac=0.0, G=None):
    """Synthetic polish job resident on `device` (SURVEY.md section 8d recipe, vectorised): uniform
    random truth, assembly = truth with substitutions at `asm_sub_rate`, reads = truth substrings
    with 0.2 % substitutions and 1e-4 N; `indel_read_frac` of the reads carry one 1-bp insertion
    or deletion (CIGAR aM1IbM / aM1DbM), the rest are a single M run.  Records are in random
    (read) order, as a SAM from an aligner is.

    repeat=(seg, copies): a segment of `seg` bp is present `copies` times (each later copy diverged by 0 or 1
    SNP); a read that lies inside a copy gets `copies` adjacent records, one per copy, all with k = copies
    (what the host ingest makes of an all-hits group: the secondary records' SEQ "*" already filled).
    pairs=True: records [0, n/2) are mate 1 and [n/2, n) mate 2 of fragment i (insert ~ N(350, 35) clipped to
    [160, 700], orientation fr), for the two-file SAM pair of the end-to-end leg; extra columns (flag, pnext,
    tlen, nm, read) are returned under "sam".  Returns dict of torch tensors + the truth."""
    if G is not None:
        contig_lens = (G,)
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    L = read_len
    lens = torch.tensor(list(contig_lens), dtype=torch.int64, device=device)
    nc = len(contig_lens)
    coff = torch.zeros(nc + 1, dtype=torch.int64, device=device)
    coff[1:] = torch.cumsum(lens, 0)
    G = int(coff[-1].item())
    truth = torch.randint(0, 4, (G,), dtype=torch.uint8, device=device, generator=g)
    loci = None
    if repeat:
        seg, copies = repeat
        loci = [(j + 1) * (G // (copies + 1)) for j in range(copies)]
        base = truth[loci[0]:loci[0] + seg].clone()
        for j in range(1, copies):
            cp = base.clone()
            if j % 2 == 1:  # every other copy carries one SNP
                p = int(torch.randint(0, seg, (1,), device=device, generator=g).item())
                cp[p] = (cp[p] + 1) % 4
            truth[loci[j]:loci[j] + seg] = cp
    asm = truth.clone()
    err = torch.rand(G, device=device, generator=g) < asm_sub_rate
    shift = torch.randint(1, 4, (G,), dtype=torch.uint8, device=device, generator=g)
    asm = torch.where(err, (asm + shift) % 4, asm)
    del err, shift
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=device)
    bases = lut[asm.long()]

    n = G * coverage // L
    margin = (700 if pairs else L) + 2
    room = torch.clamp(lens - margin, min=1)
    cum = torch.cumsum(room, 0)
    n_frag = n // 2 if pairs else n
    u = (torch.rand(n_frag, device=device, generator=g, dtype=torch.float64) * float(cum[-1].item())).long()
    u = torch.clamp(u, max=int(cum[-1].item()) - 1)
    contig = torch.searchsorted(cum, u, right=True)
    rs = u - (cum - room)[contig]
    sam = None
    if pairs:
        ins = torch.clamp(torch.round(torch.randn(n_frag, device=device, generator=g) * 35 + 350), 160, 700).long()
        fwd1 = torch.rand(n_frag, device=device, generator=g) < 0.5   # mate 1 on the forward strand
        left, right = rs, rs + ins - L
        s1 = torch.where(fwd1, left, right)
        s2 = torch.where(fwd1, right, left)
        una = torch.rand(n_frag, device=device, generator=g) < unaligned_frac
        f1 = torch.where(una, 77, torch.where(fwd1, 99, 83))
        f2 = torch.where(una, 141, torch.where(fwd1, 147, 163))
        t1 = torch.where(fwd1, ins, -ins)
        sam = {"flag": torch.cat([f1, f2]).int(), "pnext": torch.cat([s2, s1]).int(), "tlen": torch.cat([t1, -t1]).int(),
               "read": torch.cat([torch.arange(n_frag, device=device)] * 2).int()}
        contig = torch.cat([contig, contig])
        rs = torch.cat([s1, s2])
        n = 2 * n_frag
        del ins, fwd1, left, right, s1, s2, una, f1, f2, t1
    start = coff[contig] + rs  # global start
    del u
    kind = torch.zeros(n, dtype=torch.int64, device=device)
    sel = torch.rand(n, device=device, generator=g) < indel_read_frac
    kind[sel] = torch.randint(1, 3, (int(sel.sum()),), device=device, generator=g)
    del sel
    a = torch.randint(5, L - 5, (n,), device=device, generator=g)
    seq = torch.empty(n * L, dtype=torch.uint8, device=device)
    nm = torch.empty(n, dtype=torch.int32, device=device) if pairs else None
    j = torch.arange(L, device=device)[None, :]
    CH = 1 << 20
    for lo in range(0, n, CH):
        hi = min(n, lo + CH)
        k_, a_, s_ = kind[lo:hi, None], a[lo:hi, None], start[lo:hi, None]
        off = j + torch.where(k_ == 1, -(j > a_).long(), torch.where(k_ == 2, (j >= a_).long(), 0))
        codes = truth[s_ + off]
        m = hi - lo
        rnd = torch.randint(0, 4, (m, 1), dtype=torch.uint8, device=device, generator=g)
        codes = torch.where((k_ == 1) & (j == a_), rnd, codes)
        sub = torch.rand(m, L, device=device, generator=g) < sub_rate
        sh = torch.randint(1, 4, (m, L), dtype=torch.uint8, device=device, generator=g)
        codes = torch.where(sub, (codes + sh) % 4, codes)
        s = lut[codes.long()]
        s[torch.rand(m, L, device=device, generator=g) < n_rate] = ord("N")
        seq[lo * L:hi * L] = s.reshape(-1)
        if nm is not None:  # edit distance to the ASSEMBLY: mismatching aligned columns + the indel
            differ = (s != bases[s_ + off]) & ~((k_ == 1) & (j == a_))
            nm[lo:hi] = (differ.sum(1) + (kind[lo:hi] != 0)).int()
    n_cig = torch.where(kind == 0, 1, 3).to(torch.int32)
    k = torch.where((start >= G // 5) & (start < G // 5 + repeat_bp), repeat_k, 1).int()
    rs32 = rs.int()
    contig32 = contig.int()
    if repeat:
        # all-hits expansion: a read inside copy c becomes `copies` adjacent records (its own locus first)
        seg, copies = repeat
        loc = torch.tensor(loci, dtype=torch.int64, device=device)
        inside = (start[:, None] >= loc[None, :]) & (start[:, None] + L + 1 <= loc[None, :] + seg)
        own = torch.where(inside.any(1), inside.float().argmax(1), -1)
        cnt = torch.where(own >= 0, copies, 1)
        src = torch.repeat_interleave(torch.arange(n, device=device), cnt)
        first = torch.cumsum(cnt, 0) - cnt
        within = torch.arange(len(src), device=device) - first[src]
        own_s = own[src]
        cp = torch.where(own_s >= 0, (own_s + within) % copies, 0)
        new_start = torch.where(own_s >= 0, loc[cp] + (start[src] - loc[torch.clamp(own_s, min=0)]), start[src])
        k = torch.where(own_s >= 0, copies, 1).int()
        seq = seq.view(n, L)[src].reshape(-1).contiguous()
        kind, a, n_cig = kind[src], a[src], n_cig[src]
        contig32 = contig32[src]
        rs32 = (new_start - coff[contig32.long()]).int()
        start = new_start
        n = len(src)
        del inside, own, cnt, src, first, within, own_s, cp, new_start
    cig_off = torch.cumsum(n_cig.long(), 0) - n_cig.long()
    cigar = torch.zeros(int(n_cig.sum()), dtype=torch.int32, device=device)
    plain = kind == 0
    cigar[cig_off[plain]] = (L << 4) | OP_M
    ix = torch.nonzero(kind == 1)[:, 0]
    b = cig_off[ix]
    cigar[b] = ((a[ix] << 4) | OP_M).int()
    cigar[b + 1] = (1 << 4) | OP_I
    cigar[b + 2] = (((L - a[ix] - 1) << 4) | OP_M).int()
    dx = torch.nonzero(kind == 2)[:, 0]
    b = cig_off[dx]
    cigar[b] = ((a[dx] << 4) | OP_M).int()
    cigar[b + 1] = (1 << 4) | OP_D
    cigar[b + 2] = (((L - a[dx]) << 4) | OP_M).int()
    recs = {
        "contig": contig32.contiguous(),
        "ref_start": rs32.contiguous(),
        "k": k.contiguous(),
        "seq_off": torch.arange(n, device=device, dtype=torch.int64) * L,
        "seq_len": torch.full((n,), L, dtype=torch.int32, device=device),
        "cig_off": cig_off,
        "n_cig": n_cig.contiguous(),
        "seq": seq,
        "cigar": cigar,
    }
    if sam is not None:
        sam["nm"] = nm
    return {"G": G, "contig_off": coff.cpu().numpy().astype(np.uint64), "bases": bases, "recs": recs,
            "truth": lut[truth.long()], "read_len": L, "n_runs": int(n_cig.sum()), "n_aln": n, "sam": sam,
            "repeat_loci": loci, "gstart": start}


def subset_job(job, lo, hi, contig=0):
    """Records lying entirely inside [lo, hi) of contig `contig`, re-based to a single contig of hi-lo bp."""
    r = job["recs"]
    L = job["read_len"]
    rs = r["ref_start"].long()
    keep = (r["contig"] == contig) & (rs >= lo) & (rs + L + 1 <= hi)
    idx = torch.nonzero(keep)[:, 0]
    n = len(idx)
    j = torch.arange(L, device=idx.device)[None, :]
    seq = r["seq"][(r["seq_off"][idx][:, None] + j).reshape(-1)]
    n_cig = r["n_cig"][idx]
    cig_off = torch.cumsum(n_cig.long(), 0) - n_cig.long()
    pos = torch.repeat_interleave(torch.arange(n, device=idx.device), n_cig.long())
    within = torch.arange(len(pos), device=idx.device) - cig_off[pos]
    cigar = r["cigar"][r["cig_off"][idx][pos] + within]
    recs = {
        "contig": torch.zeros(n, dtype=torch.int32, device=idx.device),
        "ref_start": (rs[idx] - lo).int(),
        "k": r["k"][idx].contiguous(),
        "seq_off": torch.arange(n, device=idx.device, dtype=torch.int64) * L,
        "seq_len": r["seq_len"][idx].contiguous(),
        "cig_off": cig_off,
        "n_cig": n_cig.contiguous(),
        "seq": seq.contiguous(),
        "cigar": cigar.contiguous(),
    }
    g0 = int(job["contig_off"][contig])
    return {"G": hi - lo, "contig_off": np.array([0, hi - lo], dtype=np.uint64),
            "bases": job["bases"][g0 + lo:g0 + hi].contiguous(), "recs": recs, "read_len": L,
            "n_runs": int(n_cig.sum()), "n_aln": n}


def run_job(ctx, pp, job, params=(5, 0.5, 0.2)):
    """One step: begin + add (device-resident, borrowed) + finish.  The arguments of the three C calls are marshalled
    once per job (ctx.prepared_job): a step is then the calls themselves, not tens of microseconds of Python between
    them while the GPU waits."""
    key = (id(ctx), params, None if job.get("emit") is None else id(job["emit"]), job["bases"].data_ptr(),
           job["recs"]["seq"].data_ptr(), job["n_aln"], job["contig_off"].tobytes() if len(job["contig_off"]) < 64 else id(job["contig_off"]))
    run = job.setdefault("_prepared", {}).get(key)
    if run is None:
        r = job["recs"]
        run = ctx.prepared_job(job["contig_off"], job["bases"].data_ptr(), pp.MEM_DEVICE, job["n_aln"],
                               {k: v.data_ptr() for k, v in r.items()}, r["seq"].numel(), r["cigar"].numel(), pp.MEM_DEVICE,
                               *params, emit=job.get("emit"))
        job["_prepared"][key] = run
    run()


def algorithmic_bytes(job):
    """per good alignment seq_len + 16 B record + 4 B per CIGAR run;
    per assembly position 1 B read + 1 B written."""
    return job["n_aln"] * (job["read_len"] + 16) + 4 * job["n_runs"] + 2 * job["G"]


def to_host_records(job):
    dt = {"contig": np.uint32, "ref_start": np.uint32, "k": np.uint32, "seq_off": np.uint64, "seq_len": np.uint32,
          "cig_off": np.uint64, "n_cig": np.uint32, "seq": np.uint8, "cigar": np.uint32}
    return {k: v.cpu().numpy().astype(dt[k], copy=False) if v.dtype != torch.uint8 else v.cpu().numpy()
            for k, v in job["recs"].items()}


# ---- the end-to-end leg: SAM text -> FASTA through the drop-in CLI ----------------------------------------
def _samgen():
    path = os.path.join(ROOT, "tools", "_build", "libsamgen.so")
    if not os.path.exists(path):
        return None
    lib = ctypes.CDLL(path)
    lib.samgen_write_sam.argtypes = [ctypes.c_char_p, ctypes.c_uint32, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_void_p]
    lib.samgen_write_fasta.argtypes = [ctypes.c_char_p, ctypes.c_uint32, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_void_p]
    return lib


class _SamRecords(ctypes.Structure):
    _fields_ = [("n", ctypes.c_uint64)] + [(k, ctypes.c_void_p) for k in (
        "read", "flag", "contig", "ref_start", "cig_off", "n_cig", "cigar", "pnext", "tlen", "seq_off", "seq_len", "seq",
        "nm")] + [("qual", ctypes.c_int)]


def write_sam_pair(job, outdir, qual=True):
    """FASTA + the two SAM files (mate 1 / mate 2, same read order) of a make_job(pairs=True) job."""
    lib = _samgen()
    if lib is None:
        raise RuntimeError("tools/_build/libsamgen.so is missing (make)")
    h = to_host_records(job)
    sam = {k: v.cpu().numpy() for k, v in job["sam"].items()}
    n = job["n_aln"]
    half = n // 2
    off = job["contig_off"]
    names = b"".join(b"contig_%d\0" % (i + 1) for i in range(len(off) - 1))
    lens = np.ascontiguousarray(np.diff(off.astype(np.int64)).astype(np.uint64))
    bases = job["bases"].cpu().numpy()
    fa = os.path.join(outdir, "asm.fasta")
    if lib.samgen_write_fasta(fa.encode(), len(off) - 1, names, off.ctypes.data, bases.ctypes.data):
        raise RuntimeError("writing the FASTA failed")
    paths = []
    for f, (lo, hi) in enumerate(((0, half), (half, n))):
        cols = {"read": sam["read"][lo:hi].astype(np.uint32), "flag": sam["flag"][lo:hi].astype(np.uint32),
                "contig": h["contig"][lo:hi], "ref_start": h["ref_start"][lo:hi], "cig_off": h["cig_off"][lo:hi],
                "n_cig": h["n_cig"][lo:hi], "cigar": h["cigar"], "pnext": sam["pnext"][lo:hi].astype(np.uint32),
                "tlen": sam["tlen"][lo:hi].astype(np.int32), "seq_off": h["seq_off"][lo:hi], "seq_len": h["seq_len"][lo:hi],
                "seq": h["seq"], "nm": sam["nm"][lo:hi].astype(np.uint32)}
        cols = {k: np.ascontiguousarray(v) for k, v in cols.items()}
        rec = _SamRecords(hi - lo, *[cols[k].ctypes.data for k in ("read", "flag", "contig", "ref_start", "cig_off", "n_cig",
                                                                    "cigar", "pnext", "tlen", "seq_off", "seq_len", "seq", "nm")],
                          int(qual))
        p = os.path.join(outdir, f"reads_{f + 1}.sam")
        if lib.samgen_write_sam(p.encode(), len(off) - 1, names, lens.ctypes.data, ctypes.byref(rec)):
            raise RuntimeError("writing the SAM failed")
        paths.append(p)
    return fa, paths


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


def end_to_end(device, genome, coverage, seed, keep_dir=None):
    """SAM text in, FASTA out (reference contract: src/main.rs:112-126 -> src/polish.rs:26-38, 196-203), on
    configs[1]-shaped files generated here: bin/polypolish (both ingests; fused filter-polish) against the
    oracle's CLI on one core.  Page-cache-warm files, best of two runs for the product."""
    exe = os.path.join(ROOT, "bin", "polypolish")
    orc_exe = os.path.join(ROOT, "oracle", "_build", "pp_oracle")
    if not (os.path.exists(exe) and os.path.exists(orc_exe) and _samgen() is not None):
        return {"skipped": "bin/polypolish, oracle/_build/pp_oracle or tools/_build/libsamgen.so is missing"}
    tmp = keep_dir or tempfile.mkdtemp(prefix="pp_e2e_", dir=os.environ.get("TMPDIR", "/tmp"))
    t0 = time.perf_counter()
    job = make_job(device, contig_lens=[genome], coverage=coverage, seed=seed, pairs=True, unaligned_frac=1e-3)
    torch.cuda.synchronize()
    fa, sams = write_sam_pair(job, tmp, qual=True)
    n_rec = job["n_aln"]
    del job
    torch.cuda.empty_cache()
    gen_s = time.perf_counter() - t0
    text_bytes = sum(os.path.getsize(p) for p in sams)
    sha = lambda b: hashlib.sha256(b).hexdigest()
    out = {"files": f"{genome / 1e6:g} Mbp FASTA + 2 SAM files, {n_rec} records, {text_bytes / 1e9:.2f} GB of text "
                    f"(QUAL strings included), generated in {gen_s:.1f} s",
           "host_cores": os.cpu_count(), "text_bytes": text_bytes}
    try:
        env = dict(os.environ)
        env["PP_DEVICE_INGEST"] = "1"
        t_dev, r_dev = _timed([exe, "polish", fa] + sams, env, repeat=2)
        env["PP_DEVICE_INGEST"] = "0"
        t_host, r_host = _timed([exe, "polish", fa] + sams, env, repeat=2)
        t_cpu, r_cpu = _timed([orc_exe, "polish", fa] + sams)
        if None in (t_dev, t_host, t_cpu):
            bad = [r for t, r in ((t_dev, r_dev), (t_host, r_host), (t_cpu, r_cpu)) if t is None][0]
            out["error"] = bad.stderr.decode(errors="replace")[-400:]
            return out
        want = sha(r_cpu.stdout)
        out["polish"] = {"wall_s": round(t_dev, 3), "mbp_per_s": round(genome / 1e6 / t_dev, 2), "ingest": "device tokenizer (default)",
                         "parity": sha(r_dev.stdout) == want}
        out["polish_host_ingest"] = {"wall_s": round(t_host, 3), "mbp_per_s": round(genome / 1e6 / t_host, 2),
                                     "parity": sha(r_host.stdout) == want}
        out["oracle_polish"] = {"wall_s": round(t_cpu, 2), "mbp_per_s": round(genome / 1e6 / t_cpu, 4), "cores": 1,
                                "sha256": want[:16]}
        out["speedup_polish"] = round(t_cpu / min(t_dev, t_host), 1)
        # the chain filter -> polish: fused in one process against the oracle's two commands
        del env["PP_DEVICE_INGEST"]
        t_fp, r_fp = _timed([exe, "filter-polish", "--in1", sams[0], "--in2", sams[1], fa], env, repeat=2)
        f1, f2 = os.path.join(tmp, "f_1.sam"), os.path.join(tmp, "f_2.sam")
        t = time.perf_counter()
        ra = subprocess.run([orc_exe, "filter", "--in1", sams[0], "--in2", sams[1], "--out1", f1, "--out2", f2], capture_output=True)
        t_orc_filter = time.perf_counter() - t
        rb = subprocess.run([orc_exe, "polish", fa, f1, f2], capture_output=True)
        t_chain = time.perf_counter() - t
        # `filter` on its own: two tagged SAM files out (reference contract: src/filter.rs:26-37, 309-349)
        g1, g2 = os.path.join(tmp, "g_1.sam"), os.path.join(tmp, "g_2.sam")
        t_f, r_f = _timed([exe, "filter", "--in1", sams[0], "--in2", sams[1], "--out1", g1, "--out2", g2], env, repeat=2)
        if t_f is not None and ra.returncode == 0:
            def file_sha(path):
                h = hashlib.sha256()
                with open(path, "rb") as fh:
                    for blk in iter(lambda: fh.read(1 << 24), b""):
                        h.update(blk)
                return h.hexdigest()
            out["filter"] = {"wall_s": round(t_f, 3), "parity": file_sha(g1) == file_sha(f1) and file_sha(g2) == file_sha(f2),
                             "oracle_wall_s": round(t_orc_filter, 2), "speedup": round(t_orc_filter / t_f, 1)}
        if t_fp is None or ra.returncode or rb.returncode:
            out["filter_polish"] = {"error": (r_fp.stderr if t_fp is None else (ra.stderr + rb.stderr)).decode(errors="replace")[-400:]}
        else:
            out["filter_polish"] = {"wall_s": round(t_fp, 3), "mbp_per_s": round(genome / 1e6 / t_fp, 2),
                                    "parity": sha(r_fp.stdout) == sha(rb.stdout),
                                    "oracle_chain_wall_s": round(t_chain, 2), "speedup": round(t_chain / t_fp, 1)}
        out["parity"] = bool(out["polish"]["parity"] and out["polish_host_ingest"]["parity"] and
                             out.get("filter_polish", {}).get("parity", False) and out.get("filter", {}).get("parity", False))
    finally:
        if keep_dir is None:
            for p in os.listdir(tmp):
                os.unlink(os.path.join(tmp, p))
            os.rmdir(tmp)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=1, choices=[1, 2, 3, 4],
                    help="BASELINE.json configs index: 1 (default, the metric's configuration), 2 repeats / all-hits, "
                         "3 100-contig metagenome, 4 one 250 Mbp contig")
    ap.add_argument("--genome", type=int, default=None, help="override the configuration's total assembly length (bp)")
    ap.add_argument("--coverage", type=int, default=None, help="override the configuration's coverage")
    ap.add_argument("--cpu-sample", type=int, default=5_000_000,
                    help="bp of one contig given to the CPU oracle (default: the whole 5 Mbp job of config 1, ~12 s)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end CLI leg (config 1, one GPU)")
    ap.add_argument("--e2e-dir", default=None, help="keep the end-to-end files in this directory")
    ap.add_argument("--indel-frac", type=float, default=0.01, help="experiments only: fraction of reads with a 1-bp indel")
    ap.add_argument("--sub-rate", type=float, default=0.002, help="experiments only: per-base substitution rate")
    ap.add_argument("--n-rate", type=float, default=1e-4, help="experiments only: per-base N rate")
    ap.add_argument("--read-len", type=int, default=150, help="experiments only: read length (coverage is kept)")
    ap.add_argument("--repeat-bp", type=int, default=0,
                    help="experiments only: reads starting in a region of this many bp get depth share 1/5 "
                         "(order-dependent f64 depth -> exact replay kernel) without the extra records of config 2")
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
    if world > 1:
        import torch.distributed as dist
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)

    import polypolish_amd as pp
    ctx = pp.Context(dev_index)

    lens, coverage, repeat, label = config_shape(args.config, args.genome, args.coverage)
    G_total = int(sum(lens))
    strong = world > 1 and args.config in (3, 4)
    default_shape = (args.indel_frac == 0.01 and args.sub_rate == 0.002 and args.n_rate == 1e-4 and args.read_len == 150 and
                     args.repeat_bp == 0 and args.nd_frac == 0.0 and args.genome is None and args.coverage is None)
    # config 1: this rank's own 5 Mbp contig (seed differs per rank); configs 3 / 4 with N > 1: every rank builds
    # the same job and keeps its shard
    job = make_job(device, contig_lens=lens, coverage=coverage, read_len=args.read_len,
                   seed=42 + args.config + 1 + (0 if strong else 1000 * rank),
                   indel_read_frac=args.indel_frac, sub_rate=args.sub_rate, n_rate=args.n_rate, repeat=repeat,
                   repeat_bp=args.repeat_bp)
    if args.nd_frac > 0:
        gg = torch.Generator(device=device)
        gg.manual_seed(7)
        nd = torch.rand(job["n_aln"], device=device, generator=gg) < args.nd_frac
        job["recs"]["k"] = torch.where(nd, 3, job["recs"]["k"]).int().contiguous()
    plan = None
    if strong:
        # ONE job, resident on every rank; a rank polishes with the ranges of its units (pp_shard_plan_create: whole
        # contigs by longest-processing-time, the single contig in one window per rank) -- the device drops the records
        # that do not reach them and skips the windows outside them (pp_polish_set_emit)
        counts = torch.bincount(job["recs"]["contig"].long(), minlength=len(lens)).cpu().numpy()
        plan = pp.Plan(job["contig_off"], counts, world)
        job["emit"] = plan.emit_ranges(rank)
    torch.cuda.synchronize()
    nc_job = len(job["contig_off"]) - 1
    gdev = "cpu" if share else device
    cap = job["G"] + job["G"] // 16 + (1 << 16)          # polished bytes of one rank, at most
    total_cap = cap if strong else world * cap
    rank_lens = rank_offs = None
    gbuf = None
    if world > 1 and not share:
        # the exchange of the path, inside the library: pp_polish_gather (RCCL over xGMI)
        ident = [pp.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ident, src=0, device=device)
        ctx.comm_init(rank, world, ident[0])
        gbuf = torch.zeros(total_cap, dtype=torch.uint8, device=device) if rank == 0 else None
    elif world > 1:
        # one GPU shared by all ranks (tests): RCCL refuses that, the bytes travel over gloo
        sbuf = torch.zeros(cap, dtype=torch.uint8, device=device)
        gathered = [torch.empty(cap, dtype=torch.uint8) for _ in range(world)] if rank == 0 else None
    last = {}

    def step():
        run_job(ctx, pp, job)
        if world > 1 and not share:
            last["lens"], last["offs"] = ctx.gather(gbuf.data_ptr() if rank == 0 else None, total_cap if rank == 0 else 0)
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

    # Timed region: only the dominant kernel carries an event pair (on the library's stream), so that the
    # timers do not perturb what `value` measures.  The per-group breakdown (kernel_ms_per_step) comes from
    # a few extra, untimed steps afterwards with every kernel group under HIP events.
    ctx.set_profiling(0)
    for _ in range(args.warmup):
        step()
    ctx.set_profiling(2)
    dom_ms = []
    dom_name = None
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        kt = ctx.kernel_times()["ms"]
        if kt:
            dom_name = next(iter(kt))
            dom_ms.append(kt[dom_name])
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
    ctx.set_profiling(1)
    all_ms, n_break = {}, 5
    for _ in range(n_break):
        step()
        for k, v in ctx.kernel_times()["ms"].items():
            all_ms[k] = all_ms.get(k, 0.0) + v
    ctx.set_profiling(0)

    gather_ok = None
    if world > 1:
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
                full = dict(job)
                full["emit"] = None
                run_job(ctx, pp, full)
                ref, _, _ = ctx.result()
                gather_ok = bool(gather_ok and whole == ref)
                run_job(ctx, pp, job)  # leave the rank's own result in the context for the report below
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    ms_per_step = 1e3 * elapsed / args.steps
    total_mbp = (G_total if strong else world * G_total) / 1e6
    value = total_mbp / (elapsed / args.steps)
    b_alg = algorithmic_bytes(job) // (world if strong else 1)  # strong scaling: a rank's share of the one job
    dom_avg_ms = float(np.mean(dom_ms)) if dom_ms else 0.0
    achieved = b_alg / (dom_avg_ms * 1e-3) / 1e9 if dom_avg_ms > 0 else 0.0
    peak = 8000.0
    polished, offs, stats = ctx.result()
    recovered = None
    if not strong and not repeat:
        truth = job["truth"].cpu().numpy()
        got = np.frombuffer(polished, dtype=np.uint8)
        if len(got) == job["G"]:
            ok = got == truth
            for c in range(len(job["contig_off"]) - 1):  # contig ends have no coverage: ignore 1 kbp either side
                a, b = int(job["contig_off"][c]), int(job["contig_off"][c + 1])
                ok[a:a + 1000] = True
                ok[max(a, b - 1000):b] = True
            recovered = bool(ok.all())
        else:
            recovered = False

    # HBM traffic of the dominant kernel: PMC counters cannot be read from inside this process, so the
    # figure comes from the committed rocprofv3 counter passes of this same command (tools/profile_round.sh
    # -> profiles/traffic.json); null when the workload differs from the profiled one.
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if default_shape and args.config == 1 and dom_name and os.path.exists(tpath):
        with open(tpath) as f:
            traffic = json.load(f).get("kernels", {}).get("k_" + dom_name, {}).get("hbm_bytes")

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
                                       f"{' in total' if strong else ''}, {100 * args.indel_frac:g}% with a 1-bp indel)",
                   "parallelism": (("contig-shard" if len(lens) > 1 else "window-tile") if strong else "contig-shard") + f" x{world}"
                   if world > 1 else "single GPU",
                   "alignments_per_gpu": job["n_aln"] // (world if strong else 1)},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": peak, "unit": "GB/s",
                     "frac": round(achieved / peak, 4), "traffic": traffic,
                     "traffic_source": "profiles/traffic.json (committed rocprofv3 --pmc passes of this command, not this run)"
                     if traffic is not None else None,
                     "kernel": "k_" + (dom_name or "?"), "kernel_ms": round(dom_avg_ms, 4), "algorithmic_bytes": b_alg,
                     "whole_path_frac": round(b_alg / (ms_per_step * 1e-3) / 1e9 / peak, 4) if world == 1 else None},
        "kernel_ms_per_step": {k: round(v / max(n_break, 1), 4) for k, v in sorted(all_ms.items())},
        "planted_errors_recovered": recovered,
        "gather_verified": gather_ok,
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
    if world == 1 and args.config == 1 and not args.no_e2e:
        del job
        torch.cuda.empty_cache()
        out["e2e"] = end_to_end(device, G_total, coverage, seed=4242, keep_dir=args.e2e_dir)
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
