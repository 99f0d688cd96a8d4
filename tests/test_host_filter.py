"""CPU-side tests of the filter's host half (pp_filter_load / pp_filter_write, no GPU): the SoA, the
QNAME interning + per-file read groups and the re-emitted SAM text.  The three device kernels are
replaced here by a few lines of numpy (ref_end, orientation, insert size, pass rule: alignment.rs:138-149,
filter.rs:189-218,352-377) so that the whole `filter` command can be compared with the oracle's files."""
import os

import numpy as np
import pytest

import polypolish_amd as pp
import synth


def _ref_end(F):
    cig = F["cigar"].astype(np.int64)
    consumes = np.isin(cig & 15, (0, 2, 3, 7, 8))
    cs = np.concatenate([[0], np.cumsum((cig >> 4) * consumes)])
    lo = F["cig_off"].astype(np.int64)
    return F["ref_start"].astype(np.int64) + cs[lo + F["n_cig"].astype(np.int64)] - cs[lo]


def _orientation(f1, s1, e1, f2, s2, e2):  # filter.rs:189-209 -> 0 fr, 1 rf, 2 ff, 3 rr
    fwd1, fwd2 = (f1 & 16) == 0, (f2 & 16) == 0
    p1, p2 = (s1 if fwd1 else e1), (s2 if fwd2 else e2)
    if fwd1 != fwd2:
        first_fwd = fwd1 if p1 < p2 else fwd2
        return 0 if first_fwd else 1
    if fwd1:
        return 2 if p1 < p2 else 3
    return 2 if p2 < p1 else 3


def _emulate(loaded, low_pct, high_pct, orientation):
    A, B = loaded.files
    ea, eb = _ref_end(A), _ref_end(B)
    files, ends = (A, B), (ea, eb)
    counts, sizes = [0, 0, 0, 0], [[], [], [], []]
    for r in range(loaded.n_reads):
        ga = A["grp_idx"][A["grp_off"][r]:A["grp_off"][r + 1]]
        gb = B["grp_idx"][B["grp_off"][r]:B["grp_off"][r + 1]]
        if len(ga) == 1 and len(gb) == 1 and A["ref_id"][ga[0]] == B["ref_id"][gb[0]]:
            i, j = int(ga[0]), int(gb[0])
            o = _orientation(int(A["flags"][i]), int(A["ref_start"][i]), int(ea[i]), int(B["flags"][j]),
                             int(B["ref_start"][j]), int(eb[j]))
            pts = (int(A["ref_start"][i]), int(ea[i]), int(B["ref_start"][j]), int(eb[j]))
            counts[o] += 1
            sizes[o].append(max(pts) - min(pts))
    if orientation == "auto":
        correct = int(np.argmax(counts))
        assert counts.count(max(counts)) == 1
    else:
        correct = ("fr", "rf", "ff", "rr").index(orientation)
    s = sorted(sizes[correct])

    def pct(p):
        rank = max(1, int(np.ceil((p / 100.0) * len(s))))
        return s[rank - 1]
    lo, hi = pct(low_pct), pct(high_pct)
    passes = []
    for f in range(2):
        T, P, et, ep = files[f], files[1 - f], ends[f], ends[1 - f]
        ok = np.ones(len(T["flags"]), dtype=np.uint8)
        for r in range(loaded.n_reads):
            gt = T["grp_idx"][T["grp_off"][r]:T["grp_off"][r + 1]]
            gp = P["grp_idx"][P["grp_off"][r]:P["grp_off"][r + 1]]
            if len(gp) == 0 or len(gt) == 1:
                continue
            for i in gt:
                good = False
                for j in gp:
                    if T["ref_id"][i] != P["ref_id"][j]:
                        continue
                    pts = (int(T["ref_start"][i]), int(et[i]), int(P["ref_start"][j]), int(ep[j]))
                    ins = max(pts) - min(pts)
                    o = _orientation(int(T["flags"][i]), int(T["ref_start"][i]), int(et[i]), int(P["flags"][j]),
                                     int(P["ref_start"][j]), int(ep[j]))
                    if lo <= ins <= hi and o == correct:
                        good = True
                        break
                ok[i] = good
        passes.append(ok)
    return lo, hi, correct, counts, passes


CASES = [
    dict(seed=31),
    dict(seed=32, contig_lens=(5000, 1500), coverage=30, repeat_len=350, repeat_copies=4),
    dict(seed=33, contig_lens=(3000,), coverage=25, repeat_len=300, repeat_copies=3, inverted=False),
]


@pytest.mark.parametrize("case", CASES, ids=[f"seed{c['seed']}" for c in CASES])
def test_filter_host_half_reproduces_the_oracle_files(orc, tmp_path, monkeypatch, case):
    ds = synth.rich_dataset(str(tmp_path), **case)
    o1, o2, g1, g2 = (str(tmp_path / n) for n in ("o1.sam", "o2.sam", "g1.sam", "g2.sam"))
    want = orc.filter_files(ds["sam1"], ds["sam2"], o1, o2)
    ref = None
    for t in ("1", "3", "64"):
        monkeypatch.setenv("PP_INGEST_THREADS", t)
        L = pp.FilterLoaded(ds["sam1"], ds["sam2"])
        lo, hi, correct, counts, passes = _emulate(L, 0.1, 99.9, "auto")
        assert (lo, hi, ("fr", "rf", "ff", "rr")[correct], counts) == (want["low"], want["high"], want["orientation"],
                                                                       want["counts"])
        n = [L.write(f, passes[f], p) for f, p in ((0, g1), (1, g2))]
        assert open(g1, "rb").read() == open(o1, "rb").read() and open(g2, "rb").read() == open(o2, "rb").read()
        assert n[0][0] + n[1][0] == want["after"] and sum(c[0] for c in L.counts) == want["before"]
        cur = (L.n_reads, L.counts, [{k: v.tobytes() for k, v in F.items()} for F in L.files])
        if ref is None:
            ref = cur
            assert any((np.diff(F["grp_off"].astype(np.int64)) > 1).any() for F in L.files) or not case.get("repeat_copies")
        assert cur == ref, f"thread count {t} changed the loaded filter input"
        L.close()


def _line(name, flag, ref, pos, cigar, rest="*\t0\t0\tACGT\t*"):
    return f"{name}\t{flag}\t{ref}\t{pos}\t60\t{cigar}\t{rest}\n"


def test_name_interning_under_contention(tmp_path, monkeypatch):
    """Many records per QNAME, in both files and in random order: every thread count must produce the same read
    numbering (representative = first record of the name), groups and counts -- the lock-free table's CAS-min and
    its claim/publish hand-over are hit thousands of times per name here."""
    rng = np.random.default_rng(77)
    n_names, per = 3000, 24
    names = [f"read{rng.integers(0, 10**9)}_{i}" + "x" * int(rng.integers(0, 30)) for i in range(n_names)]
    paths = []
    for f in range(2):
        order = rng.permutation(np.repeat(np.arange(n_names), per))
        if f == 1:
            order = order[order % 7 != 3]          # some names only in file 1 ...
        lines = [_line(names[i], int(rng.integers(0, 2)) * 16, "c", int(rng.integers(1, 5000)), "20M") for i in order]
        if f == 1:
            lines += [_line(f"only2_{j}", 0, "c", 5, "20M") for j in range(500)]     # ... and some only in file 2
        p = tmp_path / f"dup{f}.sam"
        p.write_text("@SQ\tSN:c\tLN:6000\n" + "".join(lines))
        paths.append(str(p))
    ref = None
    for t in ("1", "2", "5", "8", "64"):
        monkeypatch.setenv("PP_INGEST_THREADS", t)
        for rep in range(2 if t != "1" else 1):
            L = pp.FilterLoaded(*paths)
            cur = (L.n_reads, L.counts, [{k: v.tobytes() for k, v in F.items()} for F in L.files])
            if ref is None:
                ref = cur
                assert L.n_reads == n_names + 500 and L.counts[0] == (n_names * per, n_names)
                # read numbers follow first appearance (file 1 first)
                first = {}
                for r in L.files[0]["read"]:
                    first.setdefault(int(r), len(first))
                assert all(k == v for k, v in first.items())
            assert cur == ref, f"{t} threads changed the result"
            L.close()


def test_filter_load_details_and_errors(orc, tmp_path, monkeypatch):
    a, b = tmp_path / "a.sam", tmp_path / "b.sam"
    # names out of order and repeated non-adjacently (the reference groups by HashMap key, not adjacency);
    # CR line ends, no final newline, odd CIGAR text (regex semantics), an unaligned record
    a.write_text("@HD\tVN:1\r\n" + _line("x", 0, "c", 10, "5M2D3M") + _line("y", 16, "d", 0, "4Mzz3=1Q2X") +
                 _line("x", 256, "c", 100, "*") + _line("u", 4, "*", 0, "*") + _line("z", 0, "c", 7, "10M").rstrip("\n"))
    b.write_text(_line("y", 0, "d", 50, "10M") + _line("w", 0, "c", 1, "3M") + _line("x", 16, "c", 30, "10M"))
    L = pp.FilterLoaded(str(a), str(b))
    A, B = L.files
    assert L.counts == [(4, 3), (3, 3)] and L.n_reads == 4
    assert list(A["ref_start"]) == [9, 0, 99, 6] and list(A["flags"]) == [0, 16, 256, 0]
    assert list(A["read"]) == [0, 1, 0, 2] and list(B["read"]) == [1, 3, 0]
    assert list(A["ref_id"]) == [0, 1, 0, 0] and list(B["ref_id"]) == [1, 0, 0]
    runs = [[(int(x) >> 4, pp.OPS[int(x) & 15]) for x in A["cigar"][o:o + n]] for o, n in zip(A["cig_off"], A["n_cig"])]
    assert runs == [[(5, "M"), (2, "D"), (3, "M")], [(4, "M"), (3, "="), (2, "X")], [], [(10, "M")]]
    assert [list(A["grp_idx"][A["grp_off"][r]:A["grp_off"][r + 1]]) for r in range(4)] == [[0, 2], [1], [3], []]
    assert [list(B["grp_idx"][B["grp_off"][r]:B["grp_off"][r + 1]]) for r in range(4)] == [[2], [0], [], [1]]
    out = tmp_path / "o.sam"
    assert L.write(0, np.array([1, 0, 0, 1], np.uint8), out) == (2, 2)
    lines = out.read_text().split("\n")
    assert lines[0] == "@HD\tVN:1" and lines[-1] == "" and len(lines) == 7          # CR dropped, final newline added
    assert [l.endswith("\tZP:Z:fail") for l in lines[1:6]] == [False, True, True, False, False]
    L.close()

    def both(t1, t2):
        a.write_text(t1); b.write_text(t2)
        o = [str(tmp_path / n) for n in ("e1", "e2")]
        try:
            orc.filter_files(str(a), str(b), *o)
            want = (0, "")
        except orc.OrcError as e:
            want = (e.code, e.msg)
        got = []
        for t in ("1", "5"):
            monkeypatch.setenv("PP_INGEST_THREADS", t)
            try:
                pp.FilterLoaded(str(a), str(b)).close()
                got.append((0, ""))
            except pp.PolypolishError as e:
                got.append((e.code, e.msg))
        assert got[0] == got[1]
        return got[0], want
    ok = _line("r", 0, "c", 1, "4M")
    many = "".join(_line(f"r{i}", 0, "c", 1 + i, "4M") for i in range(200))
    for t1, t2 in [(many + "\n" + ok, ok),                       # an empty line is fatal in filter (line 201)
                   (many, many[:900] + "bad\t0\tc\n" + many),     # too few columns in file 2
                   ("@HD\tVN:1\n", ok),                           # no alignments in file 1
                   (many + "r\t0\tc\t1\n", "zzz")]:               # file 1's error wins over file 2's
        got, want = both(t1, t2)
        assert got == want and got[0] == 1, (got, want)
    got, want = both(ok, "r\tx\tc\t1\t60\t4M\t*\t0\t0\tA\t*\n")   # FLAG does not parse: the reference panics
    assert got[0] == want[0] == 101
    with pytest.raises(pp.PolypolishError) as e:
        pp.FilterLoaded(str(tmp_path / "missing.sam"), str(b))
    assert e.value.code == 1 and "unable to load alignments" in e.value.msg


def test_write_text_equals_the_loaded_writer(tmp_path, monkeypatch):
    """pp_filter_write_text (tag splicing with pwritev straight from the input mapping) against pp_filter_write
    (from the parsed lines): same bytes for LF / CRLF input, with and without a final newline."""
    import ctypes
    rng = np.random.default_rng(9)
    lines = ["@HD\tVN:1", "@SQ\tSN:c\tLN:100"]
    for i in range(5000):
        flag = 4 if i % 50 == 7 else (16 if i % 3 == 0 else 0)
        lines.append(f"r{i}\t{flag}\tc\t{1 + i % 90}\t60\t4M\t*\t0\t0\tACGT\t*\tNM:i:0")
    other = tmp_path / "other.sam"
    other.write_text(_line("zz", 0, "c", 1, "4M"))
    for eol, final in (("\n", True), ("\r\n", True), ("\n", False), ("\r\n", False)):
        text = eol.join(lines) + (eol if final else "")
        p = tmp_path / "in.sam"
        p.write_bytes(text.encode())
        for t in ("1", "7"):
            monkeypatch.setenv("PP_INGEST_THREADS", t)
            L = pp.FilterLoaded(str(p), str(other))
            n = L.counts[0][0]
            verdicts = (rng.random(n) < 0.7).astype(np.uint8)
            want_counts = L.write(0, verdicts, tmp_path / "want.sam")
            raw = p.read_bytes()
            ok, bad = ctypes.c_uint64(), ctypes.c_uint64()
            err = ctypes.create_string_buffer(600)
            rc = pp.lib().pp_filter_write_text(raw, len(raw), verdicts.ctypes.data, n, str(tmp_path / "got.sam").encode(),
                                               ctypes.byref(ok), ctypes.byref(bad), err, 600)
            assert rc == 0, err.value
            assert (ok.value, bad.value) == want_counts
            assert (tmp_path / "got.sam").read_bytes() == (tmp_path / "want.sam").read_bytes(), (eol, final, t)
            L.close()
    rc = pp.lib().pp_filter_write_text(raw, len(raw), verdicts.ctypes.data, n - 1, str(tmp_path / "got.sam").encode(), None, None, err, 600)
    assert rc == pp.ERR_ARG


def test_tagged_output_into_a_pipe(tmp_path, monkeypatch):
    """--out1/--out2 may be a FIFO or /dev/stdout (the reference streams through a BufWriter, src/filter.rs:305-306):
    positional writes are impossible there, the slices go out in order -- same bytes as into a regular file; and a
    QNAME of more than 65535 bytes (its length saturates in the name table) is interned and written like any other."""
    import ctypes
    import threading
    rng = np.random.default_rng(11)
    long_a, long_b = "L" * 70000, "L" * 70000 + "x"
    lines = ["@SQ\tSN:c\tLN:100"]
    for i in range(3000):
        name = long_a if i == 100 else (long_b if i == 101 else f"r{i}")
        lines.append(f"{name}\t{16 if i % 3 == 0 else 0}\tc\t{1 + i % 90}\t60\t4M\t*\t0\t0\tACGT\t*\tNM:i:0")
    p = tmp_path / "in.sam"
    p.write_bytes(("\n".join(lines) + "\n").encode())
    other = tmp_path / "other.sam"
    other.write_text(_line("zz", 0, "c", 1, "4M") + _line(long_b, 0, "c", 5, "4M") + _line(long_a, 0, "c", 9, "4M"))
    for t in ("1", "5"):
        monkeypatch.setenv("PP_INGEST_THREADS", t)
        L = pp.FilterLoaded(str(p), str(other))
        n = L.counts[0][0]
        assert L.counts[0] == (3000, 3000) and L.counts[1] == (3, 3), "the two long names are different reads"
        verdicts = (rng.random(n) < 0.6).astype(np.uint8)
        want_counts = L.write(0, verdicts, tmp_path / "want.sam")
        fifo = str(tmp_path / f"out{t}.fifo")
        os.mkfifo(fifo)
        got = {}
        reader = threading.Thread(target=lambda: got.setdefault("bytes", open(fifo, "rb").read()))
        reader.start()
        assert L.write(0, verdicts, fifo) == want_counts
        reader.join(60)
        assert got["bytes"] == (tmp_path / "want.sam").read_bytes()
        # the splicing writer (pp_filter_write_text, what the CLI uses) into a pipe as well
        fifo2 = str(tmp_path / f"out{t}_text.fifo")
        os.mkfifo(fifo2)
        got2 = {}
        reader = threading.Thread(target=lambda: got2.setdefault("bytes", open(fifo2, "rb").read()))
        reader.start()
        raw = p.read_bytes()
        err = ctypes.create_string_buffer(600)
        rc = pp.lib().pp_filter_write_text(raw, len(raw), verdicts.ctypes.data, n, fifo2.encode(), None, None, err, 600)
        reader.join(60)
        assert rc == 0 and got2["bytes"] == (tmp_path / "want.sam").read_bytes(), err.value
        L.close()
