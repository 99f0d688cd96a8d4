#!/usr/bin/env python3
"""bench.py -- throughput of the polish hot path on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N>1 it is launched with
torch.distributed.run, one rank per GPU.  One *step* = one full pass of the device hot path
(CIGAR walk + trim, bucketing, pileup accumulate, vote, exact replay, emit) over one synthetic
job of BASELINE.json's configs[1] shape -- a 5 Mbp single-contig assembly with 200x coverage of
2x150 bp reads -- whose assembly bases and parsed alignment records are ALREADY RESIDENT IN HBM
when the timed region starts.  With N ranks every rank polishes its own 5 Mbp contig shard
(contigs shard across GPUs with no data-path collective; "weak" scaling) and the polished bytes
are gathered to rank 0 with one RCCL all_gather inside the timed step.

Rank 0 prints ONE JSON line with metric/value/unit, plus
  roofline     achieved HBM GB/s of the dominant kernel (k_tile) = algorithmic bytes per launch /
               mean launch duration measured with HIP events on the library's stream
  cpu_baseline the single-threaded C oracle (kind "port": the Rust reference cannot be built
               here) timed on a bounded sample of the same workload, with a live parity check.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

OP_M, OP_I, OP_D = 0, 1, 2


def make_job(device, G=5_000_000, coverage=200, read_len=150, seed=42, sub_rate=0.002, n_rate=1e-4,
             asm_sub_rate=1e-4, indel_read_frac=0.01, repeat_bp=0, repeat_k=5):
    """Synthetic polish job resident on `device` (SURVEY.md section 8d recipe, vectorised): uniform
    random truth, assembly = truth with substitutions at `asm_sub_rate`, reads = truth substrings
    with 0.2 % substitutions and 1e-4 N; `indel_read_frac` of the reads carry one 1-bp insertion
    or deletion (CIGAR aM1IbM / aM1DbM), the rest are a single M run.  Records are in random
    (read) order, as a SAM from an aligner is.  Returns dict of torch tensors + the truth."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    L = read_len
    truth = torch.randint(0, 4, (G,), dtype=torch.uint8, device=device, generator=g)
    asm = truth.clone()
    err = torch.rand(G, device=device, generator=g) < asm_sub_rate
    shift = torch.randint(1, 4, (G,), dtype=torch.uint8, device=device, generator=g)
    asm = torch.where(err, (asm + shift) % 4, asm)
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=device)
    bases = lut[asm.long()]

    n = G * coverage // L
    start = (torch.rand(n, device=device, generator=g, dtype=torch.float64) * (G - L - 2)).long()
    kind = torch.zeros(n, dtype=torch.int64, device=device)
    sel = torch.rand(n, device=device, generator=g) < indel_read_frac
    kind[sel] = torch.randint(1, 3, (int(sel.sum()),), device=device, generator=g)
    a = torch.randint(5, L - 5, (n,), device=device, generator=g)
    seq = torch.empty(n * L, dtype=torch.uint8, device=device)
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
    n_cig = torch.where(kind == 0, 1, 3).to(torch.int32)
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
        "contig": torch.zeros(n, dtype=torch.int32, device=device),
        "ref_start": start.int(),
        "k": torch.where((start >= G // 5) & (start < G // 5 + repeat_bp), repeat_k, 1).int(),
        "seq_off": torch.arange(n, device=device, dtype=torch.int64) * L,
        "seq_len": torch.full((n,), L, dtype=torch.int32, device=device),
        "cig_off": cig_off,
        "n_cig": n_cig,
        "seq": seq,
        "cigar": cigar,
    }
    return {"G": G, "bases": bases, "recs": recs, "truth": lut[truth.long()], "read_len": L,
            "n_runs": int(n_cig.sum()), "n_aln": n}


def subset_job(job, lo, hi):
    """Records lying entirely inside [lo, hi) of the contig, re-based to a contig of hi-lo bp."""
    r = job["recs"]
    L = job["read_len"]
    keep = (r["ref_start"].long() >= lo) & (r["ref_start"].long() + L + 1 <= hi)
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
        "ref_start": (r["ref_start"][idx].long() - lo).int(),
        "k": r["k"][idx].contiguous(),
        "seq_off": torch.arange(n, device=idx.device, dtype=torch.int64) * L,
        "seq_len": r["seq_len"][idx].contiguous(),
        "cig_off": cig_off,
        "n_cig": n_cig.contiguous(),
        "seq": seq.contiguous(),
        "cigar": cigar.contiguous(),
    }
    return {"G": hi - lo, "bases": job["bases"][lo:hi].contiguous(), "recs": recs, "read_len": L,
            "n_runs": int(n_cig.sum()), "n_aln": n}


def run_job(ctx, pp, job, params=(5, 0.5, 0.2)):
    """One step: begin + add (device-resident, borrowed) + finish."""
    r = job["recs"]
    off = np.array([0, job["G"]], dtype=np.uint64)
    ctx.polish_begin(off, job["bases"].data_ptr(), pp.MEM_DEVICE, *params)
    ctx.polish_add_ptrs(job["n_aln"], {k: v.data_ptr() for k, v in r.items()}, r["seq"].numel(),
                        r["cigar"].numel(), pp.MEM_DEVICE)
    ctx.polish_finish()


def algorithmic_bytes(job):
    """per good alignment seq_len + 16 B record + 4 B per CIGAR run;
    per assembly position 1 B read + 1 B written."""
    return job["n_aln"] * (job["read_len"] + 16) + 4 * job["n_runs"] + 2 * job["G"]


def to_host_records(job):
    dt = {"contig": np.uint32, "ref_start": np.uint32, "k": np.uint32, "seq_off": np.uint64, "seq_len": np.uint32,
          "cig_off": np.uint64, "n_cig": np.uint32, "seq": np.uint8, "cigar": np.uint32}
    return {k: v.cpu().numpy().astype(dt[k], copy=False) if v.dtype != torch.uint8 else v.cpu().numpy()
            for k, v in job["recs"].items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--genome", type=int, default=5_000_000)
    ap.add_argument("--coverage", type=int, default=200)
    ap.add_argument("--cpu-sample", type=int, default=5_000_000,
                    help="bp of the contig given to the CPU oracle (default: the whole 5 Mbp job, ~15-20 s)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--indel-frac", type=float, default=0.01, help="experiments only: fraction of reads with a 1-bp indel")
    ap.add_argument("--sub-rate", type=float, default=0.002, help="experiments only: per-base substitution rate")
    ap.add_argument("--n-rate", type=float, default=1e-4, help="experiments only: per-base N rate")
    ap.add_argument("--read-len", type=int, default=150, help="experiments only: read length (coverage is kept)")
    ap.add_argument("--seq-layout", choices=["file", "window"], default="file",
                    help="experiments only: where the SEQ bytes of a record live in the seq array -- 'file': in record (file) "
                         "order, as a streaming ingest delivers them (the benchmark's layout); 'window': grouped by the 2048-bp "
                         "window of the read's start, as an ingest that buckets while it copies could deliver them")
    ap.add_argument("--repeat-bp", type=int, default=0,
                    help="experiments only: reads starting in a region of this many bp get depth share 1/5 "
                         "(order-dependent f64 depth -> exact replay kernel), as in configs[2]")
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

    # contig shard of this rank: its own 5 Mbp contig (seed differs per rank)
    job = make_job(device, G=args.genome, coverage=args.coverage, read_len=args.read_len, seed=42 + 2 + 1000 * rank,
                   indel_read_frac=args.indel_frac, sub_rate=args.sub_rate, n_rate=args.n_rate,
                   repeat_bp=args.repeat_bp)
    if args.seq_layout == "window":  # records stay in file order; only the placement of their bytes changes
        r = job["recs"]
        L = job["read_len"]
        order = torch.argsort(r["ref_start"].long() // 2048, stable=True)
        rank_of = torch.empty_like(order)
        rank_of[order] = torch.arange(len(order), device=device)
        r["seq"] = r["seq"].view(-1, L)[order].reshape(-1).contiguous()
        r["seq_off"] = (rank_of * L).contiguous()
    torch.cuda.synchronize()
    # two send buffers in turn: the RCCL gather of step i (enqueued, not waited for) may still be reading its
    # buffer while step i+1 polishes and fills the other one; the final synchronize closes the timed region
    gather_bufs = [torch.zeros(args.genome + (1 << 16), dtype=torch.uint8, device=device) for _ in range(2)]
    gather_buf = gather_bufs[0]
    gdev = "cpu" if share else device
    gathered = [torch.empty(gather_buf.shape, dtype=torch.uint8, device=gdev) for _ in range(world)] \
        if (world > 1 and rank == 0) else None
    n_steps_done = [0]

    def step():
        run_job(ctx, pp, job)
        if world > 1:
            # the only exchange of the path: polished contig bytes -> rank 0 (RCCL over xGMI)
            buf = gather_bufs[n_steps_done[0] & 1]
            n_steps_done[0] += 1
            pp.lib().pp_polish_result(ctx._h, buf.data_ptr(), pp.MEM_DEVICE, None, None)
            dist.gather(buf.cpu() if share else buf, gathered, dst=0)

    # Timed region: only the dominant kernel carries an event pair (on the library's stream), so that the
    # timers do not perturb what `value` measures.  The per-group breakdown (kernel_ms_per_step) comes from
    # a few extra, untimed steps afterwards with every kernel group under HIP events.
    ctx.set_profiling(0)
    for _ in range(args.warmup):
        step()
    ctx.set_profiling(2)
    tile_ms = []
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        tile_ms.append(ctx.kernel_times()["ms"].get("tile", 0.0))
    ctx.sync()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
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
        # every rank's polished bytes must have arrived on rank 0 unchanged: compare byte sums
        mine, _, _ = ctx.result()
        sums = torch.zeros(2 * world, dtype=torch.int64, device=gdev)
        sums[rank] = int(np.frombuffer(mine, dtype=np.uint8).sum(dtype=np.int64))
        sums[world + rank] = len(mine)
        dist.all_reduce(sums)
        if rank == 0:
            gather_ok = all(int(gathered[r][:int(sums[world + r].item())].sum(dtype=torch.int64).item()) == int(sums[r].item())
                            for r in range(world)) and bytes(gathered[0][:len(mine)].cpu().numpy()) == mine
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    ms_per_step = 1e3 * elapsed / args.steps
    total_mbp = world * args.genome / 1e6
    value = total_mbp / (elapsed / args.steps)
    b_alg = algorithmic_bytes(job)
    tile_avg_ms = float(np.mean(tile_ms)) if tile_ms else 0.0
    achieved = b_alg / (tile_avg_ms * 1e-3) / 1e9 if tile_avg_ms > 0 else 0.0
    peak = 8000.0
    polished, offs, stats = ctx.result()
    truth = bytes(job["truth"].cpu().numpy())
    interior = slice(1000, args.genome - 1000)
    recovered = polished[interior] == truth[interior] if len(polished) == args.genome else False

    # HBM traffic of the dominant kernel: PMC counters cannot be read from inside this process, so the
    # figure comes from the committed rocprofv3 counter passes of this same command (tools/profile_round.sh
    # -> profiles/traffic.json); null when the workload differs from the profiled one.
    traffic = None
    tpath = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "traffic.json")
    default_shape = (args.genome == 5_000_000 and args.coverage == 200 and args.repeat_bp == 0 and args.seq_layout == "file" and
                     args.read_len == 150 and
                     args.indel_frac == 0.01 and args.sub_rate == 0.002 and args.n_rate == 1e-4)
    if default_shape and os.path.exists(tpath):
        with open(tpath) as f:
            traffic = json.load(f).get("kernels", {}).get("k_tile", {}).get("hbm_bytes")

    out = {
        "metric": "assembly Mbp polished/sec at 200x coverage; bit-identical FASTA vs reference",
        "value": round(value, 2),
        "unit": "Mbp/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u8/u32 counts + f64 depth",
        "data": "synthetic",
        "config": {"workload": f"configs[1]: {args.genome / 1e6:g} Mbp single-contig assembly per GPU, "
                               f"{args.coverage}x 2x{args.read_len} bp alignment records resident in HBM "
                               f"({job['n_aln']} records, 1% with a 1-bp indel)",
                   "parallelism": f"contig-shard x{world}" if world > 1 else "single GPU",
                   "alignments_per_gpu": job["n_aln"]},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": peak, "unit": "GB/s",
                     "frac": round(achieved / peak, 4), "traffic": traffic, "kernel": "k_tile",
                     "kernel_ms": round(tile_avg_ms, 4), "algorithmic_bytes": b_alg},
        "kernel_ms_per_step": {k: round(v / max(n_break, 1), 4) for k, v in sorted(all_ms.items())},
        "planted_errors_recovered": bool(recovered),
        "gather_verified": gather_ok,
        "changed_positions": stats[0]["changed"],
    }

    if not args.no_cpu_baseline and world == 1:
        from oracle import orc  # the checker / reported baseline -- never the measured product path
        S = min(args.cpu_sample, args.genome)
        sub = job if S == args.genome else subset_job(job, 0, S)
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
            "sample": f"first {S} bp of the same contig with the {sub['n_aln']} records that lie inside it "
                      f"({cpu_s:.1f} s of single-thread CPU work, pileup+vote from parsed records, no text parsing)",
            "parity_on_sample": bool(got == want["polished"]),
        }
        if got != want["polished"]:
            out["cpu_baseline"]["parity_note"] = "MISMATCH between device and oracle on the sample"
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
