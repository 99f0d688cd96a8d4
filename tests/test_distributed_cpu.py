"""The N>1 path on CPU: world_size-2 (and 3) gloo processes run the contig-sharded driver
(polypolish_amd/distributed.py) with the oracle standing in for the per-rank device engine (test
only), and rank 0's gathered FASTA must equal the unsharded oracle output byte for byte -- i.e.
sharding after the host ingest (k fixed before sharding, file order kept inside a shard) and the
length exchange + byte gather preserve the reference's result."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

import polypolish_amd as pp
import synth
from polypolish_amd import distributed as D

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, fasta, sams, outfile, min_window=1 << 16):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from oracle import orc
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    names, descs, off, bases, recs, _ = pp.ingest(fasta, sams)

    out = D.polish_sharded(synth.oracle_engine(orc), names, descs, off, bases, recs, rank, world, device="cpu",
                           min_window=min_window, min_depth=5)
    if rank == 0:
        open(outfile, "wb").write(out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_contig_sharded_polish_matches_unsharded(orc, tmp_path, world):
    ds = synth.rich_dataset(str(tmp_path), seed=51, contig_lens=(3000, 900, 2000, 700, 1500), coverage=25,
                            repeat_len=300, repeat_copies=3)
    sams = [ds["sam1"], ds["sam2"]]
    outfile = str(tmp_path / "gathered.fasta")
    port = 29500 + (os.getpid() % 2000) + world
    mp.spawn(_worker, args=(world, port, ds["fasta"], sams, outfile), nprocs=world, join=True)
    assert open(outfile, "rb").read() == orc.polish_files(ds["fasta"], sams)["fasta"]


@pytest.mark.parametrize("world", [2, 3])
def test_window_tiled_polish_matches_unsharded(orc, tmp_path, world):
    """Config C5 in miniature: one contig carries (almost) all alignments, so it is cut into windows;
    each rank polishes window + halo and emits only the window.  Indels at window edges, multi-mapped
    reads (k = 3, non-dyadic f64 shares) and a small second contig are in the mix."""
    ds = synth.rich_dataset(str(tmp_path), seed=52, contig_lens=(14000, 600), coverage=25, repeat_len=300,
                            repeat_copies=3)
    sams = [ds["sam1"], ds["sam2"]]
    names, descs, off, bases, recs, _ = pp.ingest(ds["fasta"], sams)
    units = D.plan_units(off, recs, world, 2048)
    assert (units[0] == 0).sum() == world, "the large contig was not cut into one window per rank"
    outfile = str(tmp_path / "gathered.fasta")
    port = 31500 + (os.getpid() % 2000) + world
    mp.spawn(_worker, args=(world, port, ds["fasta"], sams, outfile, 2048), nprocs=world, join=True)
    assert open(outfile, "rb").read() == orc.polish_files(ds["fasta"], sams)["fasta"]


def test_units_tile_every_contig_exactly_once():
    contig_off, bases, recs = synth.fast_records(seed=4, contig_lens=(60000, 400, 9000), coverage=20, read_len=100,
                                                 k_choices=(1, 2, 3), indel_read_frac=0.2)
    span = D.ref_spans(recs)
    for world in (1, 2, 4, 8):
        units = D.plan_units(contig_off, recs, world, 4096)
        uc, ulo, uhi, uw = units
        for c in range(3):
            lo, hi = ulo[uc == c], uhi[uc == c]
            assert lo[0] == 0 and hi[-1] == contig_off[c + 1] - contig_off[c] and np.array_equal(lo[1:], hi[:-1])
            assert np.all(lo % D.WINDOW_ALIGN == 0)
        assert (uc == 0).sum() >= max(1, world - 1) and (uc == 1).sum() == 1
        owner = D.assign_contigs(uw, world)
        covered = np.zeros(int(contig_off[-1]), dtype=np.int64)
        for r in range(world):
            mine, off, b, rr, emit = D.shard_units(contig_off, bases, recs, units, owner, r)
            assert len(b) == off[-1] and len(emit) == len(mine)
            for j, u in enumerate(mine):
                g0 = int(contig_off[uc[u]])
                own = slice(int(off[j] + emit[j, 0]), int(off[j] + emit[j, 1]))
                assert np.array_equal(b[own], bases[g0 + ulo[u]:g0 + uhi[u]])
                covered[g0 + ulo[u]:g0 + uhi[u]] += 1
            # every record lies inside its local contig and overlaps the owned window
            ln = (off[1:] - off[:-1]).astype(np.int64)[rr["contig"]]
            sp = D.ref_spans(rr)
            st = rr["ref_start"].astype(np.int64)
            assert np.all(st + sp <= ln)
            e = emit.astype(np.int64)[rr["contig"]]
            assert np.all((st < e[:, 1]) & (st + sp > e[:, 0]))
        assert np.all(covered == 1)
    assert span.min() >= 1


def test_assignment_and_shards_are_a_partition():
    contig_off, bases, recs = synth.fast_records(seed=3, contig_lens=(5000, 400, 2500, 2500, 900, 7000), coverage=20,
                                                 read_len=100, k_choices=(1, 2, 3))
    w = np.bincount(recs["contig"], minlength=6)
    for world in (1, 2, 4, 8):
        owner = D.assign_contigs(w, world)
        assert owner.min() >= 0 and owner.max() < world
        loads = [w[owner == r].sum() for r in range(world)]
        assert max(loads) <= sum(loads) / world + w.max()  # LPT bound
        seen = 0
        for r in range(world):
            mine, off, b, rr = D.shard_job(contig_off, bases, recs, owner, r)
            seen += len(rr["contig"])
            assert int(off[-1]) == len(b) == sum(int(contig_off[c + 1] - contig_off[c]) for c in mine)
            full_idx = np.nonzero(owner[recs["contig"]] == r)[0]
            assert np.array_equal(rr["k"], recs["k"][full_idx])  # k fixed before sharding, order kept
            for j in (0, len(full_idx) // 2, len(full_idx) - 1) if len(full_idx) else ():
                i = full_idx[j]
                a = recs["seq"][int(recs["seq_off"][i]):int(recs["seq_off"][i]) + int(recs["seq_len"][i])]
                c = rr["seq"][int(rr["seq_off"][j]):int(rr["seq_off"][j]) + int(rr["seq_len"][j])]
                assert np.array_equal(a, c)
        assert seen == len(recs["contig"])
