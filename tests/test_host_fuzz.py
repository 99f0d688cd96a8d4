"""CPU differential fuzzing of the host parsers against the oracle: SAM files with random defects and oddities
(tests/synth.py: mutate_sam) must end the same way -- identical polished FASTA / identical error code and
message -- whether the text goes through the product's host ingest or through the oracle's text path."""
import numpy as np

import polypolish_amd as pp
import synth

LOAD_ERRORS = ("too few columns", "unable to load alignments", "no alignments found")


def test_fuzz_polish_ingest_against_the_oracle(orc, tmp_path):
    ds = synth.rich_dataset(str(tmp_path), seed=92, contig_lens=(1200, 700), coverage=6, repeat_len=200, repeat_copies=2,
                            zp_frac=0.05, lowercase_frac=0.1)
    base1, base2 = open(ds["sam1"]).read(), open(ds["sam2"]).read()
    rng = np.random.default_rng(77)
    f1, f2 = str(tmp_path / "f1.sam"), str(tmp_path / "f2.sam")
    n_ok = n_err = 0
    for trial in range(250):
        open(f1, "w").write(synth.mutate_sam(base1, rng, int(rng.integers(1, 5))))
        open(f2, "w").write(synth.mutate_sam(base2, rng, int(rng.integers(0, 3))))
        kw = dict(max_errors=int(rng.choice([0, 2, 10])), careful=bool(trial % 5 == 0))
        try:
            want = orc.polish_files(ds["fasta"], [f1, f2], **kw)
            we = None
        except orc.OrcError as e:
            want, we = None, (e.code, e.msg)
        try:
            names, descs, off, bases, recs, counts = pp.ingest(ds["fasta"], [f1, f2], **kw)
            ge = None
        except pp.PolypolishError as e:
            ge = (e.code, e.msg)
        if we is not None and ("unexpected character" in we[1] or "does not match read sequence" in we[1]
                               or "out of bounds" in we[1] or "past the end" in we[1]):
            # raised by the CIGAR walk / the pileup, i.e. by the device in the product, after the whole ingest: a
            # file with a second, later defect of the parse kind reports that one instead (DESIGN: known deviations)
            continue
        if we is not None and we[0] == 101:
            assert ge is not None and ge[0] == 101 or ge is None, (trial, ge, we)   # panics: only the exit code compares
            n_err += 1
            continue
        assert ge == we, (trial, ge, we)
        if want is None:
            n_err += 1
            continue
        n_ok += 1
        got = orc.polish_records(off, bases, recs)
        seqs = "".join(l for l in want["fasta"].decode().split("\n") if l and not l.startswith(">")).encode()
        assert got["polished"] == seqs, trial
        assert (sum(c[0] for c in counts), sum(c[1] for c in counts), sum(c[2] for c in counts)) == want["counts"], trial
    assert n_ok >= 30 and n_err >= 30, (n_ok, n_err)


def test_fuzz_filter_load_against_the_oracle(orc, tmp_path):
    ds = synth.rich_dataset(str(tmp_path), seed=93, contig_lens=(1200, 700), coverage=6, repeat_len=200, repeat_copies=2)
    base1, base2 = open(ds["sam1"]).read(), open(ds["sam2"]).read()
    rng = np.random.default_rng(78)
    f1, f2, o1, o2 = (str(tmp_path / n) for n in ("f1.sam", "f2.sam", "o1.sam", "o2.sam"))
    n_same = 0
    for trial in range(250):
        open(f1, "w").write(synth.mutate_sam(base1, rng, int(rng.integers(1, 4))))
        open(f2, "w").write(synth.mutate_sam(base2, rng, int(rng.integers(0, 3))))
        try:
            rep = orc.filter_files(f1, f2, o1, o2)
            we = None
        except orc.OrcError as e:
            rep, we = None, (e.code, e.msg)
        try:
            L = pp.FilterLoaded(f1, f2)
            ge = None
        except pp.PolypolishError as e:
            L, ge = None, (e.code, e.msg)
        if we is not None and (we[0] == 101 or any(m in we[1] for m in LOAD_ERRORS)):
            assert ge is not None and (ge == we or we[0] == 101 == ge[0] or ge[0] == pp.ERR_LIMIT), (trial, ge, we)
            n_same += 1
            continue
        # later stages (thresholds) are the device's and the driver's business; PP_ERR_LIMIT is the product's documented
        # 32-bit position limit
        assert ge is None or ge[0] == pp.ERR_LIMIT, (trial, ge, we)
        if L is None:
            continue
        if rep is not None:
            assert sum(c[0] for c in L.counts) == rep["before"], trial
        L.close()
    assert n_same >= 20
