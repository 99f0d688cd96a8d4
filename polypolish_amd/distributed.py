"""Multi-GPU polish: contigs shard across ranks, one process per GPU, no data-path collective.

Every assembly position's counters depend only on the alignments that cover it
(src/pileup.rs:56-65) and the only cross-alignment state -- the read group's share 1/k
(src/alignment.rs:288) -- is fixed by the host ingest BEFORE sharding, so whole contigs can be
polished independently.  The one exchange of the path is the final collection of polished bytes
on rank 0, in FASTA order: an all_reduce of the per-contig lengths followed by one gather of the
padded byte payloads (RCCL over xGMI when the backend is "nccl"; gloo in the CPU tests).
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


def polish_sharded(engine, names, descs, contig_off, bases, recs, rank, world, device="cpu", **params):
    """Contig-sharded polish.  `engine(contig_off, bases, recs, **params)` polishes one shard and
    returns {"polished": bytes, "offsets": array} (Context.polish_records on a GPU).  Rank 0 returns
    the FASTA text of the whole assembly (src/polish.rs:196-203), other ranks None."""
    contig_off = np.asarray(contig_off, dtype=np.uint64)
    n_contigs = len(contig_off) - 1
    # weight = alignments per contig (the pileup work), ties broken by length
    w = np.bincount(recs["contig"], minlength=n_contigs).astype(np.float64) + 1e-9 * (contig_off[1:] - contig_off[:-1])
    owner = assign_contigs(w, world)
    mine, loc_off, loc_bases, loc_recs = shard_job(contig_off, bases, recs, owner, rank)
    if len(mine):
        res = engine(loc_off, loc_bases, loc_recs, **params)
        polished, offs = res["polished"], res["offsets"]
    else:
        polished, offs = b"", np.zeros(1, dtype=np.uint64)
    pieces = gather_polished(mine, polished, offs, n_contigs, rank, world, device)
    if rank != 0:
        return None
    out = []
    for c in range(n_contigs):
        out.append(b">" + names[c].encode() + ((b" " + descs[c].encode()) if descs[c] else b"") + b" polypolish\n"
                   + pieces[c] + b"\n")
    return b"".join(out)
