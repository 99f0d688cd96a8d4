"""The N>1 path on CPU: world_size-2 (and 3) gloo processes run the sharded driver (polypolish_amd/distributed.py)
with the oracle standing in for the per-rank device engine (test only), and rank 0's assembled FASTA must equal the
unsharded oracle output byte for byte.  The partition is the PRODUCT's (pp_shard_plan_create / pp_shard_emit_ranges /
pp_shard_assemble in libpolypolish_hip.so, no GPU needed for those): whole contigs by longest-processing-time, a
dominant contig cut into one window per rank; a rank is handed ONLY the records that reach its units (pp_shard_split:
its contigs' records, plus -- on the tiled contig -- those that reach into its window) and its emit ranges."""
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


def _ref_span(recs, i):
    co, nc = int(recs["cig_off"][i]), int(recs["n_cig"][i])
    ops = recs["cigar"][co:co + nc]
    return max(1, int(sum(int(o) >> 4 for o in ops if (int(o) & 15) in (0, 2, 3, 7, 8))))


def _split_reference(plan, dest, contig_off, recs):
    """The routing rule in plain Python: a record goes to the rank of every unit its [ref_start, ref_start + span) touches."""
    keep = []
    nc = len(contig_off) - 1
    last_rank = int(plan.unit_rank[-1])
    for i in range(len(recs["contig"])):
        c, rs, span = int(recs["contig"][i]), int(recs["ref_start"][i]), _ref_span(recs, i)
        if c >= nc:
            owners = {last_rank}
        else:
            us = [u for u in range(len(plan.unit_contig)) if plan.unit_contig[u] == c]
            owners = {int(plan.unit_rank[u]) for u in us if rs < int(plan.unit_hi[u]) and rs + span > int(plan.unit_lo[u])}
            if not owners:
                owners = {int(plan.unit_rank[us[-1]])}
        if dest in owners:
            keep.append(i)
    return np.array(keep, dtype=np.int64)


@pytest.mark.parametrize("world", [2, 3, 8])
def test_split_hands_every_rank_the_records_that_reach_its_units(orc, world):
    """pp_shard_split on host batches against the rule in plain Python, on a job with whole-contig units AND a tiled contig,
    reads with indels (D runs lengthen the span), a record with a contig index outside the assembly and one that starts
    beyond its contig's end (each must still reach exactly one rank, which reports it).  Then the property the partition
    exists for: the oracle polishing a rank's PART with its emit ranges gives the bytes it gives with ALL records."""
    contig_off, bases, recs = synth.fast_records(seed=14, contig_lens=(30000, 400, 5000, 2600), coverage=25, read_len=100,
                                                 k_choices=(1, 2, 3), k_probs=(0.8, 0.1, 0.1), indel_read_frac=0.2)
    n = len(recs["contig"])
    plan = pp.Plan(contig_off, np.bincount(recs["contig"], minlength=4), world, 2048)
    if world > 1:
        assert (plan.unit_contig == 0).sum() > 1, "the large contig is tiled"
    odd = {k: v.copy() for k, v in recs.items()}
    odd["contig"][7] = 99            # not in the assembly
    odd["ref_start"][11] = 10 ** 6   # beyond the end of its contig
    seen = np.zeros(n, dtype=np.int64)
    for r in range(world):
        part, orig = pp.shard_split_host(plan, r, odd)
        want = _split_reference(plan, r, contig_off, odd)
        assert np.array_equal(orig.astype(np.int64), want), r
        seen[want] += 1
        for k in ("contig", "ref_start", "k", "seq_len", "n_cig"):
            assert np.array_equal(part[k], odd[k][want]), (r, k)
        so, sl = part["seq_off"].astype(np.int64), part["seq_len"].astype(np.int64)
        room = (sl + 31) & ~31                                  # a part's SEQ records start on PP_SEQ_ALIGN boundaries
        assert np.array_equal(so, np.concatenate([[0], np.cumsum(room)[:-1]]) if len(sl) else so)
        for j in (0, len(want) // 2, len(want) - 1) if len(want) else ():
            i = int(want[j])
            assert bytes(part["seq"][so[j]:so[j] + sl[j]]) == bytes(odd["seq"][int(odd["seq_off"][i]):int(odd["seq_off"][i]) + int(odd["seq_len"][i])])
            a, m = int(part["cig_off"][j]), int(part["n_cig"][j])
            assert np.array_equal(part["cigar"][a:a + m], odd["cigar"][int(odd["cig_off"][i]):int(odd["cig_off"][i]) + m])
    assert seen.min() >= 1 and seen[7] == 1 and seen[11] == 1
    if world > 1:
        assert 1 < seen.max() <= 2 and (seen > 1).mean() < 0.2, "only reads across a window boundary are handed to two ranks"
        assert seen.sum() < 1.2 * n
    # the property: a rank's part + its emit ranges == all records + its emit ranges (oracle as the engine)
    eng = synth.oracle_engine(orc)
    for r in range(world):
        part, _ = pp.shard_split_host(plan, r, recs)
        e = plan.emit_ranges(r)
        assert eng(contig_off, bases, part, emit=e)["polished"] == eng(contig_off, bases, recs, emit=e)["polished"], r
    cnt = pp.shard_count(None, n, recs["contig"].ctypes.data, pp.MEM_HOST, 4, {k: v.ctypes.data for k, v in recs.items()})
    assert np.array_equal(cnt.astype(np.int64), np.bincount(recs["contig"], minlength=4))


@pytest.mark.parametrize("world", [2, 4])
def test_a_parts_mirror_keeps_its_runs(world):
    """pp_shard_split restricts the window-order mirror of a batch to a part's records AND the table of the mirror's runs
    (pp_aln_batch.wo_run_end): the part's runs are the source's, entry for entry, every run ascending in the records'
    home windows -- what lets a rank of a sharded job take the direct path.  Two SAM files, an empty run in front and one
    between them (a file without usable records)."""
    contig_off, bases, recs = synth.fast_records(seed=15, contig_lens=(30000, 400, 5000, 2600), coverage=25, read_len=100,
                                                 k_choices=(1, 2), indel_read_frac=0.2)
    n = len(recs["contig"])
    per_file = [0, n // 3, 0, n - n // 3]
    recs = dict(recs)
    recs["wo"] = pp.window_order_mirror(recs, contig_off, used_per_file=per_file)
    recs["wo_runs"] = np.cumsum(per_file).astype(np.uint64)
    plan = pp.Plan(contig_off, np.bincount(recs["contig"], minlength=4), world, 2048)
    off = np.asarray(contig_off, dtype=np.int64)
    for r in range(world):
        part, orig = pp.shard_split_host(plan, r, recs)
        runs = [int(x) for x in part["wo_runs"]]
        assert len(runs) == 4 and runs[0] == 0 and runs[1] == runs[2] and runs[-1] == len(part["contig"])
        wo = part["wo"]
        src_file = np.searchsorted(recs["wo_runs"], orig[wo["file_idx"]].astype(np.uint64), side="right")
        home = (off[wo["contig"].astype(np.int64)] + wo["ref_start"].astype(np.int64)) // 2048
        for i, (a, b) in enumerate(zip([0] + runs[:-1], runs)):
            assert (src_file[a:b] == i).all(), (r, i)          # a run holds the records of its SAM file ...
            assert (np.diff(home[a:b]) >= 0).all(), (r, i)     # ... in window order
    # a mirror without a run table: the part has none either
    plain = {k: v for k, v in recs.items() if k != "wo_runs"}
    assert "wo_runs" not in pp.shard_split_host(plan, 0, plain)[0]
