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


def _worker(rank, world, port, fasta, sams, outfile):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from oracle import orc
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    names, descs, off, bases, recs, _ = pp.ingest(fasta, sams)

    def engine(o, b, r, **kw):
        return orc.polish_records(o, b, r, **kw)
    out = D.polish_sharded(engine, names, descs, off, bases, recs, rank, world, device="cpu", min_depth=5)
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
