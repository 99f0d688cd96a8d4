"""CPU-side tests of the product's host code (no GPU): the C-ABI library loads and exports every
symbol include/polypolish_hip.h declares, and the host ingest (FASTA/SAM text -> SoA: grouping,
gates, 1/k share, '*' fill + reverse complement, upper-casing; alignment.rs:49-128,225-322)
produces exactly the records the oracle's text path implies."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

import polypolish_amd as pp
import synth
from layout_check import check_seq_layout, check_window_order_mirror, same_records

ROOT = pp.ROOT


def _seqs(fasta_bytes):
    return [l for l in fasta_bytes.decode().split("\n") if l and not l.startswith(">")]


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "polypolish_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(pp_[a-z0-9_]+)\s*\(", header))
    assert declared == set(pp.EXPORTS), declared ^ set(pp.EXPORTS)
    L = pp.lib()
    for name in sorted(declared):
        assert hasattr(L, name), f"{name} is declared in the header but not exported"
    assert b"polypolish-mi355x" in L.pp_version()


def test_no_cpu_fallback_without_a_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(pp.PolypolishError) as e:
        pp.Context(0)
    assert e.value.code == pp.ERR_HIP
    r = subprocess.run([os.path.join(ROOT, "bin", "polypolish"), "polish", "x.fasta"], capture_output=True)
    assert r.returncode == 1 and b"no CPU path" in r.stderr and r.stdout == b""


def test_header_is_c99_and_the_c_example_fails_loudly_without_a_device(tmp_path):
    """include/polypolish_hip.h is a C header (strict C99, no warnings); examples/polish_min.c, which sees nothing
    else, builds against the library and refuses to run without an MI355X."""
    src = tmp_path / "hdr.c"
    src.write_text('#include "polypolish_hip.h"\nint main(void) { return pp_version() ? 0 : 1; }\n')
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"),
                        "-c", str(src), "-o", str(tmp_path / "hdr.o")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    exe = os.path.join(ROOT, "bin", "polish_min")
    assert os.path.exists(exe), "make builds examples/polish_min.c"
    import torch
    if not torch.cuda.is_available():
        r = subprocess.run([exe, "x.fasta", "y.sam"], capture_output=True)
        assert r.returncode == 1 and r.stdout == b"" and b"no CPU fallback" in r.stderr


def test_product_does_not_touch_the_oracle():
    """The shipped path must never import, link or execute anything under oracle/."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "polypolish_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "pp_oracle" not in text and "oracle." not in text and "pyref" not in text, f
    out = subprocess.run(["ldd", pp.LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle" not in out


CASES = [
    dict(seed=11),
    dict(seed=12, contig_lens=(5000, 1500), coverage=30, repeat_len=350, repeat_copies=4, lowercase_frac=0.3),
    dict(seed=13, contig_lens=(3000,), coverage=25, repeat_len=300, repeat_copies=3, inverted=False, zp_frac=0.2,
         clip_frac=0.2),
]


@pytest.mark.parametrize("case", CASES, ids=[f"seed{c['seed']}" for c in CASES])
@pytest.mark.parametrize("careful", [False, True])
def test_ingest_records_reproduce_the_text_path(orc, tmp_path, case, careful):
    ds = synth.rich_dataset(str(tmp_path), **case)
    sams = [ds["sam1"], ds["sam2"]]
    names, descs, off, bases, recs, counts = pp.ingest(ds["fasta"], sams, max_errors=10, careful=careful)
    want = orc.polish_files(ds["fasta"], sams, positions=True, careful=careful)
    got = orc.polish_records(off, bases, recs, positions=True)
    assert got["polished"] == "".join(_seqs(want["fasta"])).encode()
    for k in ("depth", "count_a", "count_c", "count_g", "count_t", "count_other", "status"):
        assert np.array_equal(want["positions"][k], got["positions"][k]), k
    assert (sum(c[0] for c in counts), sum(c[1] for c in counts), sum(c[2] for c in counts)) == want["counts"]
    assert len(recs["contig"]) == want["counts"][1]
    if not careful and case.get("repeat_copies"):
        assert recs["k"].max() > 1, "no multi-mapped read survived: the 1/k path is not exercised"
    assert names == [c.name for c in ds["contigs"]] and descs[0] == "some description"
    # the layout of the seq array (include/polypolish_hip.h: PP_SEQ_ALIGN, PP_SEQ_WINDOW_GROUPED): every record's SEQ on a
    # 32-byte boundary, the bytes up to the next record zero, nothing else in the array; window-grouped per file by default
    # (file order inside a window, whatever the thread count), in the order of the records on request -- same records
    used = [c[1] for c in counts]
    check_seq_layout(recs, off, used, grouped=True, file_order_inside=True)
    check_window_order_mirror(recs, off, used, file_order_inside=True)   # pp_aln_batch.wo: the records once more, in window order
    for layout, env in ((0, None), (None, "file")):
        if env:
            os.environ["PP_SEQ_LAYOUT"] = env
        try:
            _, _, _, _, flat, counts_f = pp.ingest(ds["fasta"], sams, max_errors=10, careful=careful, seq_layout=layout)
        finally:
            os.environ.pop("PP_SEQ_LAYOUT", None)
        assert counts_f == counts
        check_seq_layout(flat, off, used, grouped=False)
        check_window_order_mirror(flat, off, used, file_order_inside=True)   # (the mirror does not depend on where the SEQ bytes go)
        same_records(flat, recs)


def _line(name, flag, ref, pos, cigar, seq, tags="NM:i:0"):
    return f"{name}\t{flag}\t{ref}\t{pos}\t60\t{cigar}\t*\t0\t0\t{seq}\t*\t{tags}\n"


def test_ingest_details(tmp_path):
    ref = "ACGGTCATTGCAACGGTTATTGCA"
    fa = tmp_path / "a.fasta"
    fa.write_text(f">c d1 d2\n{ref[:10]}\n{ref[10:].lower()}\n>e\nGGGG\n")
    sam = tmp_path / "a.sam"
    sam.write_text(
        "@HD\tVN:1\n\n"
        + _line("r1", 16, "c", 1, "12M", "acggtcattgca", "AS:i:3\tNM:i:2\tXX:Z:y")   # lower-case SEQ, NM not first
        + _line("r1", 256, "c", 13, "0S12M", "*")                                     # zero-length run dropped, '*' fill (revcomp)
        + _line("r1", 272, "c", 13, "5S7M", "*")                                      # soft clip: not good
        + _line("r2", 4, "*", 0, "*", "ACGT", "")                                     # unaligned, no NM: skipped
        + _line("r3", 0, "c", 1, "6=1X5=", ref[:12], "NM:i:11")                       # NM > max_errors
        + _line("r4", 0, "c", 0, "12M", ref[:12], "NM:i:0\tzp:z:FAIL")                # ZP tag, any case; POS 0 -> 0
        + _line("r5", 0, "e", 1, "2M1I1M", "GGAG", "NM:i:1\tNM:i:0"))                 # last NM wins
    names, descs, off, bases, recs, counts = pp.ingest(str(fa), [str(sam)])
    assert names == ["c", "e"] and descs == ["d1 d2", ""]
    assert bytes(bases) == (ref + "GGGG").encode() and list(off) == [0, 24, 28]
    assert counts == [(6, 3, 4)]
    assert list(recs["contig"]) == [0, 0, 1] and list(recs["ref_start"]) == [0, 12, 0] and list(recs["k"]) == [2, 2, 1]
    seqs = [bytes(recs["seq"][o:o + l]).decode() for o, l in zip(recs["seq_off"], recs["seq_len"])]
    assert seqs == ["ACGGTCATTGCA", "TGCAATGACCGT", "GGAG"]
    cig = [[(int(x) >> 4, pp.OPS[int(x) & 15]) for x in recs["cigar"][o:o + n]] for o, n in zip(recs["cig_off"], recs["n_cig"])]
    assert cig == [[(12, "M")], [(12, "M")], [(2, "M"), (1, "I"), (1, "M")]]


def test_ingest_errors_match_the_oracle(orc, tmp_path):
    ref = "ACGGTCATTGCA"
    fa = tmp_path / "a.fasta"
    fa.write_text(f">c\n{ref}\n")

    def both(text):
        p = tmp_path / "e.sam"
        p.write_text(text)
        try:
            pp.ingest(str(fa), [str(p)])
            got = (0, "")
        except pp.PolypolishError as e:
            got = (e.code, e.msg)
        try:
            orc.polish_files(str(fa), [str(p)])
            want = (0, "")
        except orc.OrcError as e:
            want = (e.code, e.msg)
        return got, want

    for text in [
        "r\t0\tc\t1\t60\t12M\n",                                    # too few columns
        "r\t0\tc\t1\t60\t12M\t*\t0\t0\t" + ref + "\t*\n",            # missing NM tag
        _line("r", 0, "c", 1, "12Q", ref),                          # invalid CIGAR
        _line("r", 0, "c", 1, "12M", "*"),                          # no alignment contains sequence
        _line("r", 0, "zzz", 1, "12M", ref),                        # contig not in assembly
    ]:
        got, want = both(text)
        assert got == want and got[0] == 1, (got, want)
    for text in ["@HD\tVN:1\n", _line("r", "x", "c", 1, "12M", ref), _line("r", 0, "c", 1, "*", ref)]:
        got, want = both(text)  # the reference panics: only the exit code is comparable
        assert got[0] == want[0] == 101, (got, want)
    # errors the reference raises inside the CIGAR walk are raised by the device, not by the ingest
    got, want = both(_line("r", 0, "c", 1, "4M2N6M", ref[:10]))
    assert got[0] == 0 and want[0] == 1


def test_fasta_errors_match_the_oracle(orc, tmp_path):
    import ctypes
    for text in ["", "A", ">\nACGT\n", "ACGT\n", ">a\n>b\nAC\n", ">a\nAC\n>a\nGG\n", ">a b\n"]:
        p = tmp_path / "f.fasta"
        p.write_text(text)
        try:
            pp.ingest(str(p), [])
            got = (0, "")
        except pp.PolypolishError as e:
            got = (e.code, e.msg)
        try:
            orc.polish_files(str(p), [])
            want = (0, "")
        except orc.OrcError as e:
            want = (e.code, e.msg)
        assert got == want and got[0] == 1, (text, got, want)


def test_ingest_is_independent_of_the_thread_count(orc, tmp_path, monkeypatch):
    """The SAM ingest parses line-aligned slices in parallel; records, counts and errors (with their
    line numbers, in the order the reference's streaming loop would hit them) must not depend on it."""
    ds = synth.rich_dataset(str(tmp_path), seed=61, contig_lens=(5000, 1500), coverage=30, repeat_len=350,
                            repeat_copies=4, lowercase_frac=0.2, zp_frac=0.05)
    sams = [ds["sam1"], ds["sam2"]]
    ref = None
    for t in ("1", "2", "7", "64"):
        monkeypatch.setenv("PP_INGEST_THREADS", t)
        names, descs, off, bases, recs, counts = pp.ingest(ds["fasta"], sams, max_errors=10)
        cur = (counts, {k: v.tobytes() for k, v in recs.items()})
        if ref is None:
            ref = cur
        assert cur == ref, f"thread count {t} changed the ingest"
    # ... nor on how many rows the window multisplit's tables have (parts that share a row are placed one after the other,
    # in file order: a large assembly's tables are bounded this way)
    monkeypatch.setenv("PP_INGEST_THREADS", "7")
    for rows in ("1", "3", "7"):
        monkeypatch.setenv("PP_INGEST_ROWS", rows)
        names, descs, off, bases, recs, counts = pp.ingest(ds["fasta"], sams, max_errors=10)
        assert (counts, {k: v.tobytes() for k, v in recs.items()}) == ref, f"{rows} rows changed the ingest"
    monkeypatch.delenv("PP_INGEST_ROWS")
    # an error deep in the file: same message and line number whatever the slicing
    lines = open(ds["sam1"]).read().split("\n")
    n = len(lines)
    broken = list(lines)
    broken[int(n * 0.8)] = "bad\t0\tcontig_1\t1\t60\t10M"          # too few columns late in the file
    p1 = tmp_path / "late_error.sam"
    p1.write_text("\n".join(broken))
    broken2 = list(lines)
    first = next(i for i, l in enumerate(broken2) if l and not l.startswith("@"))
    f = broken2[first].split("\t"); f[9] = "*"; f[10] = "*"
    broken2[first] = "\t".join(f)                                      # first read group has no sequence ...
    broken2[int(n * 0.8)] = "bad\t0\tcontig_1\t1\t60\t10M"          # ... and a parse error much later
    p2 = tmp_path / "two_errors.sam"
    p2.write_text("\n".join(broken2))
    for path in (p1, p2):
        try:
            orc.polish_files(ds["fasta"], [str(path)])
            want = (0, "")
        except orc.OrcError as e:
            want = (e.code, e.msg)
        assert want[0] == 1
        for t in ("1", "3", "16"):
            monkeypatch.setenv("PP_INGEST_THREADS", t)
            with pytest.raises(pp.PolypolishError) as e:
                pp.ingest(ds["fasta"], [str(path)])
            assert (e.value.code, e.value.msg) == want, (t, path.name)


def test_read_groups_across_slice_boundaries(orc, tmp_path, monkeypatch):
    """Groups are cut in parallel from a local predicate (previous QNAME empty or equal): a group as
    long as the file, groups straddling every slice boundary and empty QNAMEs must all behave as in the
    reference's streaming loop."""
    ref = "ACGGTCATTGCAACGGTTATTGCAGGATCCATTGACCAGTA"
    fa = tmp_path / "a.fasta"
    fa.write_text(f">c\n{ref}\n")
    texts = {
        "one_group": "".join(_line("same", 0 if i == 0 else 256, "c", 1 + (i % 20), "12M", ref[i % 20:i % 20 + 12] if i == 0 else "*")
                             for i in range(400)),
        "triples": "".join(_line(f"r{i // 3}", 0 if i % 3 == 0 else 256, "c", 1 + (i % 25), "12M",
                                 ref[i % 25:i % 25 + 12] if i % 3 == 0 else "*") for i in range(600)),
        "empty_names": "".join(_line("" if i % 4 == 1 else f"r{i}", 0, "c", 1 + (i % 25), "12M", ref[i % 25:i % 25 + 12])
                               for i in range(300)),
    }
    for name, text in texts.items():
        p = tmp_path / f"{name}.sam"
        p.write_text(text)
        want = orc.polish_files(str(fa), [str(p)], positions=True)
        for t in ("1", "5", "64"):
            monkeypatch.setenv("PP_INGEST_THREADS", t)
            names, descs, off, bases, recs, counts = pp.ingest(str(fa), [str(p)])
            assert tuple(counts[0]) == want["counts"], (name, t)
            got = orc.polish_records(off, bases, recs, positions=True)
            for k in ("depth", "count_a", "count_c", "count_g", "count_t", "status"):
                assert np.array_equal(want["positions"][k], got["positions"][k]), (name, t, k)


def test_sam_from_a_pipe(orc, tmp_path):
    """`polypolish polish asm.fasta <(samtools view ...)`: a SAM that is not a regular file cannot be mapped; the
    ingest and the filter loader read it through instead."""
    import threading
    ds = synth.rich_dataset(str(tmp_path), seed=71, contig_lens=(2000,), coverage=15)
    want = pp.ingest(ds["fasta"], [ds["sam1"], ds["sam2"]])

    def feed(path, src):
        with open(path, "wb") as f:
            f.write(open(src, "rb").read())
    fifos = []
    for i, src in enumerate((ds["sam1"], ds["sam2"])):
        p = str(tmp_path / f"pipe{i}.sam")
        os.mkfifo(p)
        threading.Thread(target=feed, args=(p, src), daemon=True).start()
        fifos.append(p)
    got = pp.ingest(ds["fasta"], fifos)
    assert got[5] == want[5] and all(np.array_equal(got[4][k], want[4][k]) for k in want[4])
    H = pp.FilterLoaded(ds["sam1"], ds["sam2"])
    fifos = []
    for i, src in enumerate((ds["sam1"], ds["sam2"])):
        p = str(tmp_path / f"pipe_f{i}.sam")
        os.mkfifo(p)
        threading.Thread(target=feed, args=(p, src), daemon=True).start()
        fifos.append(p)
    P = pp.FilterLoaded(*fifos)
    assert P.counts == H.counts and all(np.array_equal(P.files[f][k], H.files[f][k]) for f in range(2) for k in H.files[f])
    H.close(); P.close()


@pytest.mark.parametrize("gz", [False, True])
def test_reference_fasta_vector_through_the_product_loader(tmp_path, gz):
    """T11 (src/misc.rs:245-267: three records incl. a multi-word and an empty description), plain and gzipped
    (src/misc.rs:81-99,136-167), through the PRODUCT's pp_assembly_load -- the oracle is not involved."""
    import ctypes as C
    import gzip
    import json
    import polypolish_amd as pp
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_unit_vectors.json")))["T11_load_fasta"]
    path = tmp_path / ("test.fasta.gz" if gz else "test.fasta")
    if gz:
        with gzip.open(path, "wb") as f:
            f.write(gold["contents"].encode())
    else:
        path.write_text(gold["contents"])
    L = pp.lib()
    a, err = C.c_void_p(), C.create_string_buffer(512)
    assert L.pp_assembly_load(str(path).encode(), C.byref(a), err, 512) == 0, err.value
    try:
        n = L.pp_assembly_n_contigs(a)
        off = np.ctypeslib.as_array(L.pp_assembly_offsets(a), shape=(n + 1,))
        bases = np.ctypeslib.as_array(L.pp_assembly_bases(a), shape=(int(off[-1]),))
        got = [[L.pp_assembly_name(a, i).decode(), L.pp_assembly_description(a, i).decode(),
                bytes(bases[int(off[i]):int(off[i + 1])]).decode()] for i in range(n)]
        assert got == gold["records"]
    finally:
        L.pp_assembly_free(a)


def test_slices_of_a_sam_file_start_on_read_group_boundaries(tmp_path):
    """The multi-GPU driver gives every GPU a byte range of each SAM file to upload and tokenize; a cut must not split a
    read group (src/alignment.rs:255-263: adjacent aligned lines with one QNAME; an aligned line also joins a group whose
    QNAME is empty; header, empty and unaligned lines neither join nor close a group).  Every cut the driver would make
    is checked against the grouping of the whole file, and the ingest of the slices, one after the other, gives the
    records of the whole file (same k per record)."""
    import ctypes as C
    import polypolish_amd as pp
    L = pp.lib()
    L.pp_sam_group_cut_.argtypes = [C.c_char_p, C.c_uint64, C.c_uint64]
    L.pp_sam_group_cut_.restype = C.c_uint64
    rng = np.random.default_rng(5)
    lines = ["@HD\tVN:1.6", "@SQ\tSN:c1\tLN:5000"]
    seq = "ACGT" * 10
    def rec(name, flag, pos):
        return f"{name}\t{flag}\tc1\t{pos}\t60\t40M\t*\t0\t0\t{seq if not flag & 256 else '*'}\t{'I' * 40 if not flag & 256 else '*'}\tNM:i:0"
    r = 0
    for _ in range(400):
        kind = rng.integers(0, 10)
        name = f"read{r}"
        r += 1
        if kind < 5:
            lines.append(rec(name, 0, int(rng.integers(1, 4000))))
        elif kind < 8:  # an all-hits group, sometimes with an unaligned or empty line in its middle
            for j in range(int(rng.integers(2, 7))):
                lines.append(rec(name, 256 if j else 16, int(rng.integers(1, 4000))))
                if rng.random() < 0.15:
                    lines.append(rng.choice(["", f"u{r}\t4\t*\t0\t0\t*\t*\t0\t0\t{seq}\t*", "@CO\tmid-file comment"]))
        elif kind == 8:
            lines.append(f"{name}\t4\t*\t0\t0\t*\t*\t0\t0\t{seq}\t*")
        else:  # an empty QNAME does not close its group: the next record joins it
            lines.append(rec("", 0, int(rng.integers(1, 4000))))
            lines.append(rec(name, 0, int(rng.integers(1, 4000))))
    text = ("\n".join(lines) + "\n").encode()
    # the grouping of the whole file: start offsets of the lines that open a group
    starts, prev, p = set(), None, 0
    for ln in text.split(b"\n")[:-1]:
        cols = ln.split(b"\t")
        if ln and not ln.startswith(b"@") and len(cols) > 1 and not int(cols[1]) & 4:
            if prev is None or (prev != b"" and prev != cols[0]):
                starts.add(p)
            prev = cols[0]
        p += len(ln) + 1
    cuts = set()
    for frm in list(range(1, len(text), 97)) + [len(text) - 1, len(text)]:
        c = int(L.pp_sam_group_cut_(text, len(text), frm))
        assert c == len(text) or (c >= frm and c in starts), (frm, c)
        nxt = min([s0 for s0 in starts if s0 >= frm], default=len(text))
        assert c == nxt, (frm, c, nxt)
        cuts.add(c)
    assert len(cuts) > 50
    # A long run of unaligned records (unmapped reads grouped together) in front of the target: the aligned record before it
    # is out of the cut search's sight, the first aligned line after it may still belong to that record's group (unaligned
    # lines do not close a group) -- the cut then falls on the first QNAME change after it: a boundary whatever came
    # before, and not the end of the file (which left one GPU with the rest of the text).
    una = [f"u{i}\t4\t*\t0\t0\t*\t*\t0\t0\t{seq}\t*" for i in range(5000)]
    long_lines = ["@HD\tVN:1.6", rec("g1", 16, 10)] + una + [rec("g1", 256, 900), rec("g1", 256, 950), rec("g2", 0, 20), rec("g3", 0, 30)]
    ltext = ("\n".join(long_lines) + "\n").encode()
    after_una = ltext.index(b"g1\t256")
    c = int(L.pp_sam_group_cut_(ltext, len(ltext), after_una - 200))
    assert c == ltext.index(b"g2\t0"), (c, ltext.index(b"g2\t0"))
    # ingest of the slices == ingest of the file
    fa = tmp_path / "a.fasta"
    fa.write_text(">c1\n" + "ACGT" * 1250 + "\n")
    whole = tmp_path / "whole.sam"
    whole.write_bytes(text)
    _, _, _, _, want, _ = pp.ingest(str(fa), [str(whole)])
    edges = [0] + sorted(c for c in cuts if min(starts) < c < len(text))[::7] + [len(text)]  # (a file of header lines only would be an error)
    paths = []
    for i, (a0, b0) in enumerate(zip(edges[:-1], edges[1:])):
        pth = tmp_path / f"slice{i}.sam"
        pth.write_bytes(text[a0:b0])
        paths.append(str(pth))
    _, _, _, _, got, _ = pp.ingest(str(fa), paths)
    same_records(want, got)   # (the SEQ bytes are window-grouped per FILE: the slices' stretches differ from the whole file's)
    _, _, _, _, want_f, _ = pp.ingest(str(fa), [str(whole)], seq_layout=0)
    _, _, _, _, got_f, _ = pp.ingest(str(fa), paths, seq_layout=0)
    for k in want_f:
        if k not in ("wo", "wo_runs"):  # (the window-order mirror is per FILE, like the window-grouped SEQ bytes)
            assert np.array_equal(want_f[k], got_f[k]), k
