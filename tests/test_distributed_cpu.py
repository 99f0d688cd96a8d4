"""The N>1 path on CPU: world_size-2 (and 3) gloo processes run the sharded driver (polypolish_amd/distributed.py)
with the oracle standing in for the per-rank device engine (test only), and rank 0's assembled FASTA must equal the
unsharded oracle output byte for byte.  The partition is the PRODUCT's (pp_shard_plan_create / pp_shard_emit_ranges /
pp_shard_assemble in libpolypolish_hip.so, no GPU needed for those): whole contigs by longest-processing-time, a
dominant contig cut into one window per rank; every rank gets the FULL record set and its emit ranges."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

import polypolish_amd as pp
import synth
from polypolish_amd import distributed as D

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, fasta, sams, outfile, min_window=0):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from oracle import orc
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    names, descs, off, bases, recs, _ = pp.ingest(fasta, sams)

    out = D.polish_sharded(synth.oracle_engine(orc), names, descs, off, bases, recs, rank, world,
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
    """Config C5 in miniature: one contig carries (almost) all alignments, so it is cut into windows, one per rank;
    a rank emits only its window but sees every record that reaches it.  Indels at window edges, multi-mapped reads
    (k = 3, non-dyadic f64 shares) and a small second contig are in the mix."""
    ds = synth.rich_dataset(str(tmp_path), seed=52, contig_lens=(14000, 600), coverage=25, repeat_len=300,
                            repeat_copies=3)
    sams = [ds["sam1"], ds["sam2"]]
    names, descs, off, bases, recs, _ = pp.ingest(ds["fasta"], sams)
    plan = pp.Plan(off, np.bincount(recs["contig"], minlength=2), world, 2048)
    assert (plan.unit_contig == 0).sum() == world, "the large contig was not cut into one window per rank"
    outfile = str(tmp_path / "gathered.fasta")
    port = 31500 + (os.getpid() % 2000) + world
    mp.spawn(_worker, args=(world, port, ds["fasta"], sams, outfile, 2048), nprocs=world, join=True)
    assert open(outfile, "rb").read() == orc.polish_files(ds["fasta"], sams)["fasta"]


def test_units_tile_every_contig_exactly_once():
    contig_off, bases, recs = synth.fast_records(seed=4, contig_lens=(60000, 400, 9000), coverage=20, read_len=100,
                                                 k_choices=(1, 2, 3), indel_read_frac=0.2)
    counts = np.bincount(recs["contig"], minlength=3)
    for world in (1, 2, 4, 8):
        plan = pp.Plan(contig_off, counts, world, 4096)
        uc, ulo, uhi, ur = plan.unit_contig, plan.unit_lo.astype(np.int64), plan.unit_hi.astype(np.int64), plan.unit_rank
        assert np.all(np.diff(uc.astype(np.int64)) >= 0), "units are listed contig by contig"
        for c in range(3):
            lo, hi = ulo[uc == c], uhi[uc == c]
            assert lo[0] == 0 and hi[-1] == contig_off[c + 1] - contig_off[c] and np.array_equal(lo[1:], hi[:-1])
            assert np.all(lo % 2048 == 0)
            assert len(set(ur[uc == c].tolist())) == (uc == c).sum(), "a rank holds at most one window of a contig"
        assert (uc == 0).sum() >= max(1, world - 1) and (uc == 1).sum() == 1
        assert ur.max() < world
        covered = np.zeros(int(contig_off[-1]), dtype=np.int64)
        for r in range(world):
            e = plan.emit_ranges(r).astype(np.int64)
            for c in range(3):
                covered[int(contig_off[c]) + e[c, 0]:int(contig_off[c]) + e[c, 1]] += 1
        assert np.all(covered == 1), "the ranks' emit ranges partition the assembly"


def test_assignment_is_balanced_and_assembly_restores_the_order():
    contig_off, bases, recs = synth.fast_records(seed=3, contig_lens=(5000, 400, 2500, 2500, 900, 7000), coverage=20,
                                                 read_len=100, k_choices=(1, 2, 3))
    w = np.bincount(recs["contig"], minlength=6).astype(np.int64)
    rng = np.random.default_rng(0)
    for world in (1, 2, 4, 8):
        plan = pp.Plan(contig_off, w, world, 1 << 20)  # windows never smaller than 1 Mbp here: whole contigs only
        assert plan.unit_lo.max() == 0 and len(plan.unit_contig) == 6
        loads = [w[plan.unit_contig[plan.unit_rank == r]].sum() for r in range(world)]
        assert max(loads) <= sum(loads) / world + w.max()  # LPT bound
        # every rank "polishes" its contigs to a marker string of another length; assembly must restore FASTA order
        marks = [bytes([65 + c]) * int(rng.integers(1, 50)) for c in range(6)]
        rank_bytes, rank_offs = [], []
        for r in range(world):
            mine = [c for c in range(6) if plan.unit_rank[list(plan.unit_contig).index(c)] == r]
            offs = np.zeros(7, dtype=np.uint64)
            buf = b""
            for c in range(6):
                offs[c] = len(buf)
                if c in mine:
                    buf += marks[c]
            offs[6] = len(buf)
            rank_bytes.append(buf)
            rank_offs.append(offs)
        data, out_off = plan.assemble(rank_bytes, rank_offs)
        assert data == b"".join(marks)
        assert [int(x) for x in out_off] == [0] + list(np.cumsum([len(m) for m in marks]))
