"""Multi-GPU polish, one process per GPU: contigs (and windows of a large contig) shard across ranks with no
data-path collective; the only exchange is the collection of the polished bytes on rank 0.

Everything that decides or moves data is in libpolypolish_hip.so (include/polypolish_hip.h, "multi-GPU"):

  pp_shard_plan_create   whole contigs by longest-processing-time on their alignment counts; a contig that carries
                         more than one rank's share (config C5: one 250 Mbp contig) is cut into windows on 2048-bp
                         boundaries, one per rank
  pp_shard_split         the records that reach a rank's units: its contigs' records, on a tiled contig its window's
                         records plus those that reach in from the neighbours (an alignment only touches its own
                         contig's positions ref_start + j: src/alignment.rs:297-303, src/pileup.rs:189-200)
  pp_polish_set_emit     a rank polishes those records with the ranges of its units: the device works on its windows
                         only.  Every owned position sees all of its alignments in file order, so the order-dependent
                         f64 depth stays exact (src/pileup.rs:64), and the read group's share 1/k
                         (src/alignment.rs:288) was fixed by the ingest before anything was partitioned
  pp_polish_gather       ncclAllGather of byte counts + one group of ncclSend / ncclRecv into rank 0 (RCCL over xGMI)
  pp_shard_assemble      the ranks' bytes back in FASTA order

This module is the launcher glue for `python -m torch.distributed.run ... -m polypolish_amd.distributed polish ...`:
it hands the ncclUniqueId around and prints the FASTA.  With PP_SHARE_GPU=1 (tests on a one-GPU box: RCCL refuses two
ranks on one device) and in the CPU tests the bytes travel as torch.distributed objects over gloo instead.
"""
from __future__ import annotations

import numpy as np


def gather_objects(polished: bytes, offs, rank: int, world: int):
    """Transport for tests / shared-GPU runs: every rank's (bytes, contig_out_off) to rank 0 over torch.distributed
    (gloo).  Returns (list of bytes, list of offset arrays) on rank 0, (None, None) elsewhere."""
    if world == 1:
        return [polished], [np.asarray(offs, dtype=np.uint64)]
    import torch.distributed as dist
    got = [None] * world if rank == 0 else None
    dist.gather_object((polished, np.asarray(offs, dtype=np.uint64)), got, dst=0)
    if rank != 0:
        return None, None
    return [g[0] for g in got], [g[1] for g in got]


def polish_sharded(engine, names, descs, contig_off, bases, recs, rank, world, gather=None, min_window=0, locate=None,
                   **params):
    """Sharded polish.  `engine(contig_off, bases, recs, emit=..., **params)` polishes the records it is given restricted
    to the emit ranges and returns {"polished": bytes, "offsets": array} (Context.polish_records on a GPU; the oracle in
    the CPU tests).  A rank gives it only the records that reach its units (pp_shard_split).  `gather(polished, offsets)`
    collects every rank's result on rank 0 (default: gloo objects).  `locate(exc)` (optional) returns the rank-local
    number of the record an engine failure is about: the ranks then agree on the job's FIRST bad record (the reference
    streams, so that is the one it reports) and all of them raise.  Rank 0 returns the FASTA text of the whole assembly
    (src/polish.rs:196-203), other ranks None."""
    import polypolish_amd as pp
    contig_off = np.asarray(contig_off, dtype=np.uint64)
    n_contigs = len(contig_off) - 1
    counts = np.bincount(np.asarray(recs["contig"], dtype=np.int64), minlength=n_contigs)[:n_contigs]
    plan = pp.Plan(contig_off, counts, world, min_window)
    mine, orig = pp.shard_split_host(plan, rank, recs)
    res, failure = None, None
    try:
        res = engine(contig_off, bases, mine, emit=plan.emit_ranges(rank), **params)
    except Exception as e:  # noqa: BLE001 -- whatever the engine raises travels to every rank
        local = locate(e) if locate else None
        failure = (int(orig[local]) if local is not None and local < len(orig) else -1, e)
    if world > 1:
        import torch.distributed as dist
        seen = [None] * world
        dist.all_gather_object(seen, None if failure is None else (failure[0], type(failure[1]).__name__, str(failure[1]),
                                                                   getattr(failure[1], "code", 1)))
        bad = [(s[0], r, s) for r, s in enumerate(seen) if s is not None]
        if bad:
            _, r, s = min(bad)
            if failure is not None and r == rank:
                raise failure[1]
            raise pp.PolypolishError(s[3], s[2])
    elif failure is not None:
        raise failure[1]
    if gather is None:
        gather = lambda b, o: gather_objects(b, o, rank, world)  # noqa: E731
    all_bytes, all_offs = gather(res["polished"], res["offsets"])
    if rank != 0:
        return None
    data, out_off = plan.assemble(all_bytes, all_offs)
    out = []
    for c in range(n_contigs):
        out.append(b">" + names[c].encode() + ((b" " + descs[c].encode()) if descs[c] else b"") + b" polypolish\n"
                   + data[int(out_off[c]):int(out_off[c + 1])] + b"\n")
    return b"".join(out)


def main(argv=None):
    """`polypolish polish` across the GPUs of one node, one process per GPU:

        python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \\
            -m polypolish_amd.distributed polish [options] assembly.fasta a_1.sam a_2.sam > polished.fasta

    Options as the reference's (src/main.rs:78-108) except --debug.  Every rank runs the host ingest (the 1/k shares
    are fixed before anything is partitioned), polishes its contigs / windows on its own GPU, the polished bytes go
    to rank 0 over RCCL and rank 0 prints the FASTA.  PP_SHARE_GPU=1 (testing on a one-GPU box): all ranks use GPU 0
    and the bytes travel over gloo."""
    import argparse
    import ctypes as C
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
        dist.init_process_group("gloo")  # control plane only: the unique id and (shared-GPU runs) the bytes
    try:
        names, descs, off, bases, recs, _ = pp.ingest(a.assembly, a.sam, max_errors=a.max_errors, careful=a.careful)
        ctx = pp.Context(dev)
        gather = None
        if world > 1 and not share:
            ident = [pp.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(ident, src=0)
            ctx.comm_init(rank, world, ident[0])

            def gather(polished, offs):  # RCCL: the bytes are still on the device (pp_polish_result_device)
                buf = torch.empty(int(off[-1]) + int(off[-1]) // 8 + (1 << 20), dtype=torch.uint8, device=f"cuda:{dev}") \
                    if rank == 0 else None
                lens, offs_all = ctx.gather(buf.data_ptr() if rank == 0 else None, buf.numel() if rank == 0 else 0)
                if rank != 0:
                    return None, None
                host = buf.cpu().numpy()
                starts = np.concatenate([[0], np.cumsum(lens.astype(np.int64))])
                return [host[int(starts[r]):int(starts[r + 1])].tobytes() for r in range(world)], list(offs_all)
        def locate(exc):  # the rank-local record a device error of the CIGAR walk is about
            rec, kind = C.c_uint64(), C.c_uint32()
            return int(rec.value) if pp.lib().pp_polish_error_record(ctx._h, C.byref(rec), C.byref(kind)) else None
        out = polish_sharded(ctx.polish_records, names, descs, off, bases, recs, rank, world, gather=gather, locate=locate,
                             min_depth=a.min_depth, fraction_valid=a.fraction_valid, fraction_invalid=a.fraction_invalid)
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
