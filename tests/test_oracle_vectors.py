"""Pin the oracle (C restatement, and the independent Python restatement) against every
known-answer vector the reference's own unit tests hold, plus the
hand-derived vectors D1..D12 of SURVEY.md section 8c for the functions the reference never tests.
CPU only."""
import gzip
import json
import os

import pytest

from oracle import pyref

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_unit_vectors.json")))


# ----------------------------------------------------------------------------- T1 / T2
def test_T1_expanded_cigar_good(orc):
    for cigar, exp in GOLD["T1_expanded_cigar_good"]["cases"]:
        assert orc.get_expanded_cigar(cigar) == exp
        assert pyref.get_expanded_cigar(cigar) == exp


def test_T2_expanded_cigar_bad(orc):
    for cigar in GOLD["T2_expanded_cigar_bad"]["cases"]:
        assert orc.get_expanded_cigar(cigar) is None
        assert pyref.get_expanded_cigar(cigar) is None


# ----------------------------------------------------------------------------- T3
def test_T3_ref_positions(orc):
    t = GOLD["T3_ref_positions"]
    for cigar, start, end in t["cases"]:
        line = t["line_template"].format(cigar=cigar)
        assert orc.parse_positions(line) == (start, end)
        a = pyref.Alignment.new(line)
        assert (a.ref_start, a.ref_end()) == (start, end)


# ----------------------------------------------------------------------------- T4
def test_T4_orientation(orc):
    t = GOLD["T4_orientation"]
    for p1, p2, f1, f2, want in t["cases"]:
        # the reference builds SAM lines with POS = p (1-based) -> ref_start = p - 1
        assert orc.get_orientation(f1, p1 - 1, t["cigar"], f2, p2 - 1, t["cigar"]) == want
        l1 = f"r_1\t{f1}\tx\t{p1}\t60\t150M\t*\t0\t0\tACTG\tKKKK\tNM:i:0"
        l2 = f"r_2\t{f2}\tx\t{p2}\t60\t150M\t*\t0\t0\tACTG\tKKKK\tNM:i:0"
        assert pyref.get_orientation(pyref.Alignment.new_quick(l1), pyref.Alignment.new_quick(l2)) == want


# ----------------------------------------------------------------------------- T5 / T6
def test_T5_auto_orientation(orc):
    for counts, want in GOLD["T5_auto_orientation"]["cases"]:
        assert orc.auto_determine_orientation(counts) == want
    assert orc.auto_determine_orientation([2, 2, 0, 0]) is None  # tie -> "could not automatically determine"


def test_T6_percentile(orc):
    t = GOLD["T6_percentile"]
    for p, want in t["cases"]:
        assert orc.get_percentile(t["nums"], p) == want
        assert pyref.get_percentile(t["nums"], p) == want


# ----------------------------------------------------------------------------- T8
def test_T8_pileup_base(orc):
    t = GOLD["T8_pileup_base"]
    for case in t["cases"]:
        b = orc.PileupBase(case["original"])
        p = pyref.Position(case["original"])
        for s, n, dc in case["adds"]:
            for _ in range(n):
                b.add_seq(s, dc)
                p.add(s, dc)
        assert b.get_count_str() == case["count_str"]
        assert p.count_str() == case["count_str"]
        got = b.get_polished_seq(t["min_depth"], t["fraction_valid"], case["fraction_invalid"])
        assert got == (case["polished"], case["status"])
        new, status, _, _ = p.vote(t["min_depth"], t["fraction_valid"], case["fraction_invalid"])
        assert (new, status) == (case["polished"], case["status"])


# ----------------------------------------------------------------------------- T9 / T10
def test_T9_bankers_rounding(orc):
    for x, want in GOLD["T9_bankers_rounding"]["cases"]:
        assert orc.bankers_rounding(x) == want
        assert pyref.bankers_rounding(x) == want


def test_T10_reverse_complement(orc):
    for s, want in GOLD["T10_reverse_complement"]["cases"]:
        assert orc.reverse_complement(s) == want
        assert pyref.reverse_complement(s) == want


# ----------------------------------------------------------------------------- T11
@pytest.mark.parametrize("gz", [False, True])
def test_T11_load_fasta(orc, tmp_path, gz):
    import ctypes as C
    t = GOLD["T11_load_fasta"]
    path = tmp_path / ("test.fasta.gz" if gz else "test.fasta")
    if gz:
        with gzip.open(path, "wb") as f:
            f.write(t["contents"].encode())
    else:
        path.write_text(t["contents"])
    assert [list(r) for r in pyref.load_fasta(str(path))] == t["records"]

    class Fasta(C.Structure):
        _fields_ = [("n", C.c_size_t), ("name", C.POINTER(C.c_char_p)), ("desc", C.POINTER(C.c_char_p)),
                    ("seq", C.POINTER(C.c_char_p)), ("len", C.POINTER(C.c_size_t))]
    fa, err = Fasta(), C.create_string_buffer(512)
    L = orc.lib()
    assert L.orc_load_fasta(str(path).encode(), C.byref(fa), err, 512) == 0, err.value
    got = [[fa.name[i].decode(), fa.desc[i].decode(), fa.seq[i].decode()] for i in range(fa.n)]
    assert got == t["records"]
    L.orc_fasta_free(C.byref(fa))


# ============================================================================= derived D1..D12
def test_D1_trim_doc_example(orc):
    # alignment.rs:349-363: ...TGAGTACAGG trims to ...TGAGTAC
    assert orc.read_slices("10M", "TGAGTACAGG") == list("TGAGTAC")


def test_D2_insertion(orc):
    assert orc.read_slices("3M1I5M", "ACGTTACGA") == ["A", "C", "GT", "T", "A", "C"]


def test_D3_deletion(orc):
    assert orc.read_slices("2M2D4M", "ACGTAC") == ["A", "C", "", "", "G", "T"]


def test_trim_corner_cases(orc):
    assert orc.read_slices("6M", "AAAAAA") == []  # all one base: contributes nothing
    assert orc.read_slices("4M", "ACGG") == ["A"]  # pops G,G then one more (C)
    assert orc.read_slices("2M1D2M", "ACGG") == ["A", "C"]  # pops G,G; deletion slot stops the run; popped
    assert orc.read_slices("2M1I2M", "ACTGG") == ["A"]  # 'CT' slot stops the run, then is popped
    assert orc.read_slices("2M1D1I2M", "ACGGG") == ["A"]  # D slot rewritten to 'G' by the I: joins the run, then C is popped
    for bad in ("2M1S2M", "2M1N2M", "2M1H2M", "2M1P2M"):
        with pytest.raises(orc.OrcError) as e:
            orc.read_slices(bad, "ACGTA")
        assert e.value.code == orc.QUIT
    with pytest.raises(orc.OrcError):
        orc.read_slices("4M", "ACGTA")  # CIGAR does not match read sequence


def _write(tmp_path, fasta, sams):
    fa = tmp_path / "a.fasta"
    fa.write_text(fasta)
    paths = []
    for i, s in enumerate(sams):
        p = tmp_path / f"s{i}.sam"
        p.write_text(s)
        paths.append(str(p))
    return str(fa), paths


def _line(name, flag, ref, pos, cigar, seq, nm=0, extra=""):
    return f"{name}\t{flag}\t{ref}\t{pos}\t60\t{cigar}\t*\t0\t0\t{seq}\t*\tNM:i:{nm}{extra}\n"


def _both(orc, fa, sams, **kw):
    c = orc.polish_files(fa, sams, debug=True, positions=True, **kw)
    py_fasta, py_dbg, _ = pyref.polish(fa, sams, debug=True, **kw)
    assert c["fasta"] == py_fasta.encode()
    assert c["debug"] == py_dbg.encode()
    return c


def test_D4_f64_order_dependence(orc, tmp_path):
    # fifteen good alignments with k=3 over one position: depth = 4.999999999999999 -> low_depth at -d 5
    ref = "ACGTACGTACGTTGCA" * 3
    sam = ""
    for r in range(15):
        for copy in range(3):
            sam += _line(f"r{r}", 0 if copy == 0 else 256, "c", 1 + 16 * copy, "16M", ref[:16] if copy == 0 else "*")
    fa, sams = _write(tmp_path, f">c\n{ref}\n", [sam])
    c = _both(orc, fa, sams)
    assert c["positions"]["depth"][0] == 4.999999999999999
    assert orc.STATUS[c["positions"]["status"][0]] == "low_depth"


def test_D5_percentile_rank_1000(orc):
    nums = list(range(1, 1001))
    assert orc.get_percentile(nums, 99.9) == 1000  # 99.9/100*1000 = 999.0000000000001 -> ceil 1000
    assert pyref.get_percentile(nums, 99.9) == 1000


def test_D6_zero_invalid_threshold(orc, tmp_path):
    # -d 1, depth 2, A=2: invalid_thr = round(0.4) = 0 -> C,G,T (count 0) are intermediate -> too_close
    ref = "CCCCCCCCCCAGGT"
    sam = _line("r1", 0, "c", 1, "14M", "ACCCCCCCCCAGGT") + _line("r2", 0, "c", 1, "14M", "ACCCCCCCCCAGGT")
    fa, sams = _write(tmp_path, f">c\n{ref}\n", [sam])
    c = _both(orc, fa, sams, min_depth=1)
    assert orc.STATUS[c["positions"]["status"][0]] == "too_close"
    assert c["fasta"] == f">c polypolish\n{ref}\n".encode()


def test_D7_D8_multimap_share_and_star_fill(orc, tmp_path):
    # primary on the - strand with SEQ, secondary (256) on + with SEQ '*' -> uses revcomp(primary SEQ)
    unit = "ACGGTCATTGCA"
    ref = unit + "TTTTT" + pyref.reverse_complement(unit) + "G"
    sam = ""
    for r in range(6):
        sam += _line(f"r{r}", 16, "c", 1, "12M", unit) + _line(f"r{r}", 256, "c", 18, "12M", "*")
    fa, sams = _write(tmp_path, f">c\n{ref}\n", [sam])
    c = _both(orc, fa, sams)
    pos = c["positions"]
    assert pos["depth"][0] == 3.0 and pos["count_a"][0] == 6  # counts are whole alignments, depth is shared
    assert pos["depth"][17] == 3.0 and pos["count_t"][17] == 6  # revcomp(unit) starts with T


def test_D9_careful(orc, tmp_path):
    ref = "ACGGTCATTGCAACGGTCATTGCA"
    sam = ""
    for r in range(8):
        sam += _line(f"r{r}", 0, "c", 1, "12M", ref[:12]) + _line(f"r{r}", 256, "c", 13, "5S7M", "*")
    sam += _line("solo", 0, "c", 1, "12M", ref[:12])
    fa, sams = _write(tmp_path, f">c\n{ref}\n", [sam])
    c = _both(orc, fa, sams, careful=True)
    assert c["positions"]["depth"][0] == 1.0  # groups of two are dropped even though one member is bad
    c = _both(orc, fa, sams, careful=False)
    assert c["positions"]["depth"][0] == 9.0


def test_D10_filter_asymmetry(orc):
    # same-strand pair: evaluated from file 2 the roles swap and the answer flips (filter.rs:205-206)
    assert orc.get_orientation(0, 100, "50M", 0, 400, "50M") == "ff"
    assert orc.get_orientation(0, 400, "50M", 0, 100, "50M") == "rr"


def test_D11_header_forms(orc, tmp_path):
    fa, sams = _write(tmp_path, ">c1 foo bar\nACGT\n>c2\nGG-TT\n>c3\tx\nacgtn\n", [])
    c = _both(orc, fa, sams)
    assert c["fasta"] == b">c1 foo bar polypolish\nACGT\n>c2 polypolish\nGGTT\n>c3 x polypolish\nACGTN\n"


def test_D12_empty_lines(orc, tmp_path):
    ref = "ACGGTCATTGCA"
    sam = "\n" + _line("r1", 0, "c", 1, "12M", ref) + "\n"
    fa, sams = _write(tmp_path, f">c\n{ref}\n", [sam, sam])
    _both(orc, fa, [sams[0]])  # polish skips empty lines (alignment.rs:241)
    with pytest.raises(orc.OrcError) as e:  # filter dies on them (filter.rs:126-130)
        orc.filter_files(sams[0], sams[1], str(tmp_path / "o1"), str(tmp_path / "o2"))
    assert e.value.code == orc.QUIT and "too few columns" in e.value.msg and "(line 1)" in e.value.msg
    with pytest.raises(pyref.Quit):
        pyref.filter_pairs(sams[0], sams[1])


def test_error_paths(orc, tmp_path):
    ref = "ACGGTCATTGCA"
    fa, _ = _write(tmp_path, f">c\n{ref}\n", [])

    def run(sam_text, **kw):
        p = tmp_path / "e.sam"
        p.write_text(sam_text)
        try:
            orc.polish_files(fa, [str(p)], **kw)
        except orc.OrcError as e:
            c_res = (e.code, e.msg)
        else:
            c_res = (0, "")
        try:
            pyref.polish(fa, [str(p)], **kw)
        except pyref.Quit as e:
            py_res = (1, str(e))
        except pyref.Panic:
            py_res = (101, None)
        else:
            py_res = (0, "")
        assert c_res[0] == py_res[0], (c_res, py_res)
        if py_res[1] is not None:
            assert c_res[1] == py_res[1]
        return c_res

    assert run(_line("r", 0, "c", 1, "12M", ref))[0] == 0
    assert "missing NM tag" in run("r\t0\tc\t1\t60\t12M\t*\t0\t0\t" + ref + "\t*\n")[1]
    assert "too few columns" in run("r\t0\tc\t1\t60\t12M\n")[1]
    assert "invalid CIGAR" in run(_line("r", 0, "c", 1, "12Q", ref))[1]
    assert "not in assembly" in run(_line("r", 0, "zzz", 1, "12M", ref))[1]
    assert "does not match read sequence" in run(_line("r", 0, "c", 1, "11M", ref))[1]
    assert "unexpected character" in run(_line("r", 0, "c", 1, "4M2N6M", ref[:10]))[1]
    assert "contain sequence" in run(_line("r", 0, "c", 1, "12M", "*"))[1]
    assert run("@HD\tVN:1\n")[0] == 101  # EOF flush of an empty group panics (alignment.rs:265,319)
    assert run(_line("r", 0, "c", 5, "12M", ref))[0] == 101  # runs past the contig end (pileup.rs:194-196)
    assert run(_line("r", "x", "c", 1, "12M", ref))[0] == 101  # FLAG parse unwrap
    assert run(_line("r", 0, "c", 1, "12M", ref), fraction_invalid=0.6)[0] == 1


def test_polish_without_sams(orc, tmp_path):
    fa, _ = _write(tmp_path, ">a d\nAC-GT\n", [])
    assert orc.polish_files(fa, [])["fasta"] == b">a d polypolish\nACGT\n"


def test_derived_end_to_end_fixture_is_reproduced_by_the_oracle(orc, tmp_path):
    """tests/golden/derived_e2e (made by make_derived_fixtures.py): the oracle's filter + polish of a committed
    SAM pair must keep giving the committed bytes (a drift guard; derived, not reference-supplied)."""
    import hashlib
    import json
    import os
    g = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "derived_e2e")
    exp = json.load(open(os.path.join(g, "expected.json")))
    s1, s2, fa = (os.path.join(g, n) for n in ("case_1.sam", "case_2.sam", "case.fasta"))
    f1, f2 = str(tmp_path / "f1.sam"), str(tmp_path / "f2.sam")
    assert orc.filter_files(s1, s2, f1, f2) == exp["filter_report"]
    assert hashlib.sha256(open(f1, "rb").read()).hexdigest() == exp["filtered_1_sha256"]
    assert hashlib.sha256(open(f2, "rb").read()).hexdigest() == exp["filtered_2_sha256"]
    got = orc.polish_files(fa, [f1, f2])
    assert got["fasta"] == open(os.path.join(g, "polished_after_filter.fasta"), "rb").read()
    assert list(got["counts"]) == exp["counts_after_filter"]
    got = orc.polish_files(fa, [s1, s2], careful=True, min_depth=3)
    assert got["fasta"] == open(os.path.join(g, "polished_raw_careful_d3.fasta"), "rb").read()
