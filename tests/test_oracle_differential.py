"""Differential test of the two independent restatements (C oracle vs pure-Python pyref) on
randomised synthetic SAM/FASTA inputs: FASTA bytes, --debug TSV bytes and filter outputs must be
identical.  This is the cross-check SURVEY.md section 8c asks for in place of a reference build
(no Rust toolchain here).  CPU only."""
import numpy as np
import pytest

import synth
from oracle import pyref

CASES = [
    dict(seed=1),
    dict(seed=2, contig_lens=(3000,), coverage=60, indel_rate=0.006, asm_err_rate=0.01),
    dict(seed=3, contig_lens=(6000, 1200, 900), coverage=30, repeat_len=400, repeat_copies=3),
    dict(seed=4, contig_lens=(5000,), coverage=50, repeat_len=300, repeat_copies=5, inverted=False,
         n_rate=0.01),
    dict(seed=5, contig_lens=(2500, 2500), coverage=8, sub_rate=0.02, zp_frac=0.1),
]


@pytest.mark.parametrize("case", CASES, ids=[f"seed{c['seed']}" for c in CASES])
def test_polish_c_vs_python(orc, tmp_path, case):
    ds = synth.rich_dataset(str(tmp_path), **case)
    sams = [ds["sam1"], ds["sam2"]]
    for kw in (dict(), dict(careful=True), dict(min_depth=2, fraction_invalid=0.1, max_errors=3)):
        c = orc.polish_files(ds["fasta"], sams, debug=True, positions=True, **kw)
        py_fasta, py_dbg, per_pos = pyref.polish(ds["fasta"], sams, debug=True, **kw)
        assert c["fasta"] == py_fasta.encode()
        assert c["debug"] == py_dbg.encode()
        depth = np.array([p[0] for p in per_pos])
        assert np.array_equal(c["positions"]["depth"], depth)  # bit-exact f64
    # the data must actually exercise the interesting paths
    pos = orc.polish_files(ds["fasta"], sams, positions=True)["positions"]
    assert (pos["status"] == 1).sum() > 0, "no position was changed"
    assert (pos["count_other"] > 0).sum() > 0, "no indel / N evidence anywhere"


@pytest.mark.parametrize("case", CASES[2:4], ids=["seed3", "seed4"])
def test_filter_c_vs_python(orc, tmp_path, case):
    ds = synth.rich_dataset(str(tmp_path), **case)
    o1, o2 = str(tmp_path / "f_1.sam"), str(tmp_path / "f_2.sam")
    for kw in (dict(), dict(orientation="fr", low=5.0, high=95.0), dict(orientation="rf", low=1.0, high=60.0)):
        try:
            rep = orc.filter_files(ds["sam1"], ds["sam2"], o1, o2, **kw)
        except orc.OrcError as e:
            with pytest.raises(pyref.Quit) as pe:
                pyref.filter_pairs(ds["sam1"], ds["sam2"], **kw)
            assert str(pe.value) == e.msg
            continue
        p1, p2, prep = pyref.filter_pairs(ds["sam1"], ds["sam2"], **kw)
        assert open(o1, "rb").read() == p1
        assert open(o2, "rb").read() == p2
        for k in ("before", "after", "low", "high", "orientation", "counts"):
            assert rep[k] == prep[k], k
        if not kw:
            assert rep["after"] < rep["before"], "filter failed nothing: repeats not exercised"
    # filtered output feeds polish identically in both restatements
    orc.filter_files(ds["sam1"], ds["sam2"], o1, o2)
    c = orc.polish_files(ds["fasta"], [o1, o2], debug=True)
    py_fasta, py_dbg, _ = pyref.polish(ds["fasta"], [o1, o2], debug=True)
    assert c["fasta"] == py_fasta.encode() and c["debug"] == py_dbg.encode()


def test_records_entry_matches_text_path(orc, tmp_path):
    """orc_polish_records (the SoA entry used to check the device) == the text path."""
    contig_off, bases, recs = synth.fast_records(seed=7, contig_lens=(3000, 2000), coverage=30, read_len=80,
                                                 indel_read_frac=0.05, n_rate=0.002)
    fa, sam = str(tmp_path / "a.fasta"), str(tmp_path / "a.sam")
    synth.records_to_sam(contig_off, bases, recs, path_fasta=fa, path_sam=sam)
    text = orc.polish_files(fa, [sam], positions=True, max_errors=1000)
    rec = orc.polish_records(contig_off, bases, recs, positions=True)
    seqs = [l for l in text["fasta"].decode().split("\n") if l and not l.startswith(">")]
    assert rec["polished"] == "".join(seqs).encode()
    for k in ("depth", "count_a", "count_c", "count_g", "count_t", "count_other", "status"):
        assert np.array_equal(text["positions"][k], rec["positions"][k]), k
