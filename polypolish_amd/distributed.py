"""Multi-GPU polish: contigs (and windows of large contigs) shard across ranks, one process per GPU,
no data-path collective.

Every assembly position's counters depend only on the alignments that cover it
(src/pileup.rs:56-65) and the only cross-alignment state -- the read group's share 1/k
(src/alignment.rs:288) -- is fixed by the host ingest BEFORE sharding, so whole contigs can be
polished independently.  The one exchange of the path is the final collection of polished bytes
on rank 0, in FASTA order: an all_reduce of the per-unit lengths followed by one gather of the
padded byte payloads (RCCL over xGMI when the backend is "nccl"; gloo in the CPU tests).

A contig that carries more than one rank's share of the alignments (config C5: one 250 Mbp contig)
is cut into contiguous windows.  A rank polishes its window plus a halo of one alignment span on
either side, receives every alignment that overlaps the window (so each owned position sees all of
its alignments, in file order -- the f64 depth stays exact), and emits only the window
(pp_polish_set_emit).
"""
from __future__ import annotations

import numpy as np


def assign_contigs(weights, world: int) -> np.ndarray:
    """Longest-processing-time greedy assignment of contigs to ranks (deterministic)."""
    weights = np.asarray(weights, dtype=np.float64)
    owner = np.zeros(len(weights), dtype=np.int64)
    load = np.zeros(world, dtype=np.float64)
    for c in sorted(range(len(weights)), key=lambda i: (-weights[i], i)):
        r = int(np.argmin(load))
        owner[c] = r
        load[r] += weights[c]
    return owner


def shard_job(contig_off, bases, recs, owner, rank):
    """The sub-job of `rank`: its contigs (FASTA order kept) and their alignment records (file order
    kept, k untouched).  Returns (local_contig_ids, contig_off, bases, recs)."""
    contig_off = np.asarray(contig_off, dtype=np.uint64)
    mine = np.nonzero(owner == rank)[0]
    lens = (contig_off[1:] - contig_off[:-1])[mine]
    loc_off = np.zeros(len(mine) + 1, dtype=np.uint64)
    loc_off[1:] = np.cumsum(lens)
    loc_bases = np.concatenate([bases[int(contig_off[c]):int(contig_off[c + 1])] for c in mine]) if len(mine) \
        else np.zeros(0, np.uint8)
    remap = np.full(len(owner), -1, dtype=np.int64)
    remap[mine] = np.arange(len(mine))
    sel = np.nonzero(remap[recs["contig"]] >= 0)[0]
    seq_len = recs["seq_len"][sel].astype(np.int64)
    n_cig = recs["n_cig"][sel].astype(np.int64)

    def gather(src, starts, lengths):
        total = int(lengths.sum())
        if total == 0:
            return src[:0].copy()
        row = np.repeat(np.arange(len(lengths)), lengths)
        first = np.cumsum(lengths) - lengths
        return src[(starts[row] + (np.arange(total) - first[row])).astype(np.int64)]

    out = {
        "contig": remap[recs["contig"][sel]].astype(np.uint32),
        "ref_start": recs["ref_start"][sel],
        "k": recs["k"][sel],
        "seq_off": (np.cumsum(seq_len) - seq_len).astype(np.uint64),
        "seq_len": recs["seq_len"][sel],
        "cig_off": (np.cumsum(n_cig) - n_cig).astype(np.uint64),
        "n_cig": recs["n_cig"][sel],
        "seq": gather(recs["seq"], recs["seq_off"][sel].astype(np.int64), seq_len),
        "cigar": gather(recs["cigar"], recs["cig_off"][sel].astype(np.int64), n_cig),
    }
    return mine, loc_off, loc_bases, out


REF_CONSUMING = (0, 2, 3, 7, 8)  # M D N = X  (get_ref_end, src/alignment.rs:138-149)
WINDOW_ALIGN = 2048              # the device's tile width: windows start on tile boundaries


def ref_spans(recs) -> np.ndarray:
    """Reference span of every record from its packed CIGAR runs."""
    cig = np.asarray(recs["cigar"], dtype=np.uint32)
    consumes = np.isin(cig & 15, REF_CONSUMING)
    cs = np.concatenate([[0], np.cumsum((cig >> 4).astype(np.int64) * consumes)])
    lo = np.asarray(recs["cig_off"], dtype=np.int64)
    return cs[lo + np.asarray(recs["n_cig"], dtype=np.int64)] - cs[lo]


def plan_units(contig_off, recs, world: int, min_window: int = 1 << 16):
    """Cut the assembly into units (contig, lo, hi): whole contigs, except that a contig holding more
    than one rank's share of the alignments becomes up to `world` windows.  Deterministic; FASTA order,
    then position order.  Returns (unit_contig, unit_lo, unit_hi, unit_weight)."""
    contig_off = np.asarray(contig_off, dtype=np.int64)
    n_contigs = len(contig_off) - 1
    lens = contig_off[1:] - contig_off[:-1]
    per_contig = np.bincount(np.asarray(recs["contig"], dtype=np.int64), minlength=n_contigs).astype(np.float64)
    share = per_contig.sum() / max(world, 1)
    uc, ulo, uhi, uw = [], [], [], []
    for c in range(n_contigs):
        pieces = 1
        if world > 1 and per_contig[c] > share > 0:
            pieces = int(min(world, np.ceil(per_contig[c] / share), max(1, lens[c] // min_window)))
        cuts = [0]
        for j in range(1, pieces):
            x = int(lens[c] * j // pieces) // WINDOW_ALIGN * WINDOW_ALIGN
            if x > cuts[-1]:
                cuts.append(x)
        cuts.append(int(lens[c]))
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            uc.append(c); ulo.append(lo); uhi.append(hi)
            uw.append(per_contig[c] * (hi - lo) / lens[c] + 1e-9 * (hi - lo))
    return (np.array(uc, np.int64), np.array(ulo, np.int64), np.array(uhi, np.int64), np.array(uw, np.float64))


def shard_units(contig_off, bases, recs, units, owner, rank):
    """The sub-job of `rank` over its units.  Every unit becomes one local contig = the unit's window
    plus a halo of max-alignment-span bases on either side (clipped to the contig), with the records
    that overlap the window, rebased; file order is kept.  Returns (my_units, contig_off, bases, recs,
    emit) where emit[j] = [lo, hi) of local contig j that the rank owns."""
    contig_off = np.asarray(contig_off, dtype=np.int64)
    uc, ulo, uhi, _ = units
    mine = np.nonzero(owner == rank)[0]
    rc = np.asarray(recs["contig"], dtype=np.int64)
    start = np.asarray(recs["ref_start"], dtype=np.int64)
    span = ref_spans(recs)
    n_contigs = len(contig_off) - 1
    halo = np.zeros(n_contigs, dtype=np.int64)
    if len(rc):
        np.maximum.at(halo, rc[rc < n_contigs], span[rc < n_contigs])
    loc_off = [0]
    pieces, sel_idx, sel_unit, sel_start, emit = [], [], [], [], []
    for j, u in enumerate(mine):
        c, lo, hi = int(uc[u]), int(ulo[u]), int(uhi[u])
        clen = int(contig_off[c + 1] - contig_off[c])
        whole = lo == 0 and hi == clen
        left = 0 if whole else min(lo, int(halo[c]))
        right = 0 if whole else min(clen - hi, int(halo[c]))
        pieces.append(bases[int(contig_off[c]) + lo - left:int(contig_off[c]) + hi + right])
        loc_off.append(loc_off[-1] + (hi - lo) + left + right)
        emit.append((left, left + hi - lo))
        if whole:
            idx = np.nonzero(rc == c)[0]
        else:
            idx = np.nonzero((rc == c) & (start < hi) & (start + span > lo))[0]
        sel_idx.append(idx)
        sel_unit.append(np.full(len(idx), j, dtype=np.int64))
        sel_start.append(start[idx] - (lo - left))
    if sel_idx:
        idx = np.concatenate(sel_idx)
        order = np.argsort(idx, kind="stable")  # back to file order
        sel = idx[order]
        loc_contig = np.concatenate(sel_unit)[order]
        loc_start = np.concatenate(sel_start)[order]
    else:
        sel = np.zeros(0, np.int64); loc_contig = sel; loc_start = sel
    seq_len = np.asarray(recs["seq_len"])[sel].astype(np.int64)
    n_cig = np.asarray(recs["n_cig"])[sel].astype(np.int64)

    def gather(src, starts, lengths):
        total = int(lengths.sum())
        if total == 0:
            return np.asarray(src)[:0].copy()
        row = np.repeat(np.arange(len(lengths)), lengths)
        first = np.cumsum(lengths) - lengths
        return np.asarray(src)[(starts[row] + (np.arange(total) - first[row])).astype(np.int64)]

    out = {
        "contig": loc_contig.astype(np.uint32),
        "ref_start": loc_start.astype(np.uint32),  # a start past 2^32 cannot occur: contigs are < 2^32 bp
        "k": np.asarray(recs["k"])[sel],
        "seq_off": (np.cumsum(seq_len) - seq_len).astype(np.uint64),
        "seq_len": np.asarray(recs["seq_len"])[sel],
        "cig_off": (np.cumsum(n_cig) - n_cig).astype(np.uint64),
        "n_cig": np.asarray(recs["n_cig"])[sel],
        "seq": gather(recs["seq"], np.asarray(recs["seq_off"])[sel].astype(np.int64), seq_len),
        "cigar": gather(recs["cigar"], np.asarray(recs["cig_off"])[sel].astype(np.int64), n_cig),
    }
    loc_bases = np.concatenate(pieces) if pieces else np.zeros(0, np.uint8)
    return mine, np.array(loc_off, dtype=np.uint64), loc_bases, out, np.array(emit, dtype=np.uint64).reshape(-1, 2)


def gather_polished(local_contigs, local_bytes, local_off, n_contigs, rank, world, device="cpu"):
    """Collect the polished contigs on rank 0 in FASTA order.  local_bytes/local_off describe this
    rank's contigs (in the order of local_contigs).  Returns a list of bytes on rank 0, None elsewhere."""
    import torch
    import torch.distributed as dist
    lengths = torch.zeros(n_contigs, dtype=torch.int64, device=device)
    for j, c in enumerate(local_contigs):
        lengths[int(c)] = int(local_off[j + 1]) - int(local_off[j])
    if world > 1:
        dist.all_reduce(lengths, op=dist.ReduceOp.SUM)
    lengths = lengths.cpu().numpy()
    if world == 1:
        return [local_bytes[int(local_off[j]):int(local_off[j + 1])] for j in range(len(local_contigs))]
    # payload of every rank: its contigs concatenated in FASTA order, padded to the largest shard
    owner_len = np.zeros(world, dtype=np.int64)
    counts = torch.zeros(world, dtype=torch.int64, device=device)
    counts[rank] = len(local_bytes)
    dist.all_reduce(counts, op=dist.ReduceOp.SUM)
    owner_len[:] = counts.cpu().numpy()
    pad = int(owner_len.max()) if owner_len.max() > 0 else 1
    buf = torch.zeros(pad, dtype=torch.uint8, device=device)
    if len(local_bytes):
        buf[:len(local_bytes)] = torch.frombuffer(bytearray(local_bytes), dtype=torch.uint8).to(device)
    # who owns which contig: ranks announce their contig lists through a second all_reduce
    who = torch.zeros(n_contigs, dtype=torch.int64, device=device)
    for c in local_contigs:
        who[int(c)] = rank
    dist.all_reduce(who, op=dist.ReduceOp.SUM)
    who = who.cpu().numpy()
    gathered = [torch.empty_like(buf) for _ in range(world)] if rank == 0 else None
    dist.gather(buf, gathered, dst=0)
    if rank != 0:
        return None
    cursor = np.zeros(world, dtype=np.int64)
    payload = [g.cpu().numpy().tobytes() for g in gathered]
    out = []
    for c in range(n_contigs):
        r = int(who[c])
        out.append(payload[r][cursor[r]:cursor[r] + int(lengths[c])])
        cursor[r] += int(lengths[c])
    return out


def polish_sharded(engine, names, descs, contig_off, bases, recs, rank, world, device="cpu", min_window=1 << 16,
                   **params):
    """Sharded polish.  `engine(contig_off, bases, recs, emit=..., **params)` polishes one shard and
    returns {"polished": bytes, "offsets": array} (Context.polish_records on a GPU).  Rank 0 returns
    the FASTA text of the whole assembly (src/polish.rs:196-203), other ranks None."""
    contig_off = np.asarray(contig_off, dtype=np.uint64)
    n_contigs = len(contig_off) - 1
    units = plan_units(contig_off, recs, world, min_window)
    owner = assign_contigs(units[3], world)
    mine, loc_off, loc_bases, loc_recs, emit = shard_units(contig_off, bases, recs, units, owner, rank)
    if len(mine):
        res = engine(loc_off, loc_bases, loc_recs, emit=emit, **params)
        polished, offs = res["polished"], res["offsets"]
    else:
        polished, offs = b"", np.zeros(1, dtype=np.uint64)
    pieces = gather_polished(mine, polished, offs, len(units[0]), rank, world, device)
    if rank != 0:
        return None
    per_contig = [[] for _ in range(n_contigs)]
    for u, c in enumerate(units[0]):
        per_contig[int(c)].append(pieces[u])
    out = []
    for c in range(n_contigs):
        out.append(b">" + names[c].encode() + ((b" " + descs[c].encode()) if descs[c] else b"") + b" polypolish\n"
                   + b"".join(per_contig[c]) + b"\n")
    return b"".join(out)


def main(argv=None):
    """`polypolish polish` across the GPUs of one node, one process per GPU:

        python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \\
            -m polypolish_amd.distributed polish [options] assembly.fasta a_1.sam a_2.sam > polished.fasta

    Options as the reference's (src/main.rs:78-108) except --debug.  Every rank runs the host ingest
    (the 1/k shares are fixed before sharding), polishes its contigs / windows on its own GPU and rank 0
    prints the FASTA.  PP_SHARE_GPU=1 (testing on a one-GPU box): all ranks use GPU 0, gather over gloo."""
    import argparse
    import os
    import sys
    import torch
    import torch.distributed as dist
    import polypolish_amd as pp
    ap = argparse.ArgumentParser(prog="polypolish_amd.distributed")
    ap.add_argument("command", choices=["polish"])
    ap.add_argument("-i", "--fraction_invalid", type=float, default=0.2)
    ap.add_argument("-v", "--fraction_valid", type=float, default=0.5)
    ap.add_argument("-m", "--max_errors", type=int, default=10)
    ap.add_argument("-d", "--min_depth", type=int, default=5)
    ap.add_argument("--careful", action="store_true")
    ap.add_argument("assembly")
    ap.add_argument("sam", nargs="*")
    a = ap.parse_args(argv)
    # stdout carries the FASTA and nothing else: libraries that chat on fd 1 (gloo's "[Gloo] Rank ..." lines)
    # are pointed at stderr for the whole run
    sys.stdout.flush()
    fasta_fd = os.dup(1)
    os.dup2(2, 1)
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    share = os.environ.get("PP_SHARE_GPU") == "1"
    if not torch.cuda.is_available():
        raise SystemExit("Error: no usable MI355X (HIP) device -- this build has no CPU path")
    dev = 0 if share else local
    torch.cuda.set_device(dev)
    if world > 1:
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
    try:
        names, descs, off, bases, recs, _ = pp.ingest(a.assembly, a.sam, max_errors=a.max_errors, careful=a.careful)
        ctx = pp.Context(dev)
        out = polish_sharded(ctx.polish_records, names, descs, off, bases, recs, rank, world,
                             device="cpu" if share or world == 1 else f"cuda:{dev}", min_depth=a.min_depth,
                             fraction_valid=a.fraction_valid, fraction_invalid=a.fraction_invalid)
    except pp.PolypolishError as e:
        sys.stderr.write(f"\nError: {e.msg}\n")
        raise SystemExit(101 if e.code == pp.ERR_PANIC else 1)
    if rank == 0:
        view = memoryview(out)
        while len(view):
            view = view[os.write(fasta_fd, view):]
    os.close(fasta_fd)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
