"""The reference's own known-answer vectors (tests/golden/reference_unit_vectors.json: the assert lines of its inline unit
tests) ON THE DEVICE, not through the oracle: every expected value below is one the reference holds.

  T8  src/pileup.rs:208-295   eight PileupBase tallies + votes        -> records through pp_polish_* (k_prep .. k_tile / k_exact)
  T9  src/misc.rs:279-296     twelve bankers_rounding values          -> the invalid threshold the per-position planes record
  T10 src/misc.rs:298-304     reverse_complement                      -> the device tokenizer's "*" fill of an inverted secondary
  T3  src/alignment.rs:402-422  ref_start / ref_end of four CIGARs    -> k_ref_end through the filter's insert sizes
(T4, src/filter.rs:384-424, is test_gpu_parity.py::test_reference_orientation_vectors_on_device.)
Plus SURVEY 8f-4: a gzipped assembly through bin/polypolish (src/misc.rs:81-99,136-167).  Needs an MI355X: `-m gpu`."""
import ctypes as C
import gzip
import json
import os
import subprocess

import numpy as np
import pytest

import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_unit_vectors.json")))
STATUS = ("kept", "changed", "low_depth", "none", "multiple", "too_close")  # PP_ST_* in the order of src/pileup.rs:156-163


@pytest.fixture(scope="module")
def pp():
    import polypolish_amd
    return polypolish_amd


@pytest.fixture(scope="module")
def ctx(pp):
    c = pp.Context(0)
    yield c
    c.close()


# A 64-base assembly without two equal neighbours; position P in the middle of 24-base reads that start at P - 10 and end in
# two different bases (the trim of src/alignment.rs:364-378 then pops exactly the last two entries, far from P).
P, READ_START, READ_LEN = 30, 20, 24


def _assembly(original):
    ref = list("ACGTCAGTCTGATCGACTGCATGCTAGCATCGATGCATCAGTCAGTGCATGCATCGATCGTAGC")
    assert len(ref) == 64 and all(a != b for a, b in zip(ref, ref[1:]))
    ref[P] = original  # (equal neighbours around P would not matter: the trim only looks at a read's right end)
    return "".join(ref)


def _records(ref, adds):
    """adds: [(base at P, how many reads, depth share 1/k)] in the order the reference's test calls add_seq."""
    seqs, ks = [], []
    for base, count, share in adds:
        k = int(round(1.0 / share))
        assert 1.0 / k == share
        read = ref[READ_START:P] + base + ref[P + 1:READ_START + READ_LEN]
        seqs += [read] * count
        ks += [k] * count
    n = len(seqs)
    return {"contig": np.zeros(n, np.uint32), "ref_start": np.full(n, READ_START, np.uint32), "k": np.array(ks, np.uint32),
            "seq_off": np.arange(n, dtype=np.uint64) * READ_LEN, "seq_len": np.full(n, READ_LEN, np.uint32),
            "cig_off": np.arange(n, dtype=np.uint64), "n_cig": np.ones(n, np.uint32),
            "seq": np.frombuffer("".join(seqs).encode(), np.uint8), "cigar": np.full(n, (READ_LEN << 4) | 0, np.uint32)}


def _device_batch(ctx, pp, off, bases, recs, positions, **kw):
    """one device-resident batch with both mirrors, polished in place (what the tokenizer hands over)"""
    import test_gpu_parity as tg
    return tg._polish_device_batch(ctx, pp, off, bases, recs, True, positions=positions, wo=True, **kw)


def _count_str(pos, p):
    """get_count_str, src/pileup.rs:137-148: the non-zero tallies as "<key>x<count>", sorted as strings"""
    items = [f"{b}x{int(pos['count_' + b.lower()][p])}" for b in "ACGT" if pos["count_" + b.lower()][p]]
    assert pos["count_other"][p] == 0
    return ",".join(sorted(items))


@pytest.mark.parametrize("i", range(len(GOLD["T8_pileup_base"]["cases"])))
def test_T8_pileup_base_vectors_on_device(ctx, pp, i):
    """src/pileup.rs:208-295: add_seq tallies and get_polished_seq(5, 0.5, fraction_invalid) of one position, as records
    whose only disagreement with the assembly is the base at that position.  Case 5 is 444 reads of depth share 0.1."""
    t8 = GOLD["T8_pileup_base"]
    case = t8["cases"][i]
    ref = _assembly(case["original"])
    bases = np.frombuffer(ref.encode(), np.uint8)
    off = np.array([0, len(ref)], np.uint64)
    recs = _records(ref, [tuple(a) for a in case["adds"]])
    kw = dict(min_depth=t8["min_depth"], fraction_valid=t8["fraction_valid"], fraction_invalid=case["fraction_invalid"])
    runs = {
        "host batch, --debug planes": ctx.polish_records(off, bases, recs, positions=True, **kw),
        "host batch, k_tile's own votes": ctx.polish_records(off, bases, recs, positions=3, **kw),
        "device batch + mirrors, --debug planes": _device_batch(ctx, pp, off, bases, recs, True, **kw),
        "device batch + mirrors, k_tile's own votes": _device_batch(ctx, pp, off, bases, recs, 3, **kw),
    }
    for name, got in runs.items():
        pos = got["positions"]
        assert _count_str(pos, P) == case["count_str"], (name, _count_str(pos, P))
        assert STATUS[pos["status"][P]] == case["status"], (name, STATUS[pos["status"][P]])
        assert len(got["polished"]) == len(ref) and chr(got["polished"][P]) == case["polished"], (name, got["polished"])
    for name, got in (("host batch", ctx.polish_records(off, bases, recs, **kw)),
                      ("device batch + mirrors", _device_batch(ctx, pp, off, bases, recs, False, **kw))):
        assert len(got["polished"]) == len(ref) and chr(got["polished"][P]) == case["polished"], (name, got["polished"])
        assert got["stats"][0]["changed"] == (1 if case["status"] == "changed" else 0), name


def _exact_product(x):
    """(n reads, fraction) with float(n) * fraction == x exactly (one f64 multiply, as src/pileup.rs:70-72), fraction < 0.9"""
    if x == 0.0:
        return 0, 0.5
    for n in range(max(1, int(x / 0.899) + 1), int(x / 0.899) + 200000):
        f = x / n
        if 0.0 < f < 0.9 and float(n) * f == x:
            return n, f
    raise AssertionError(f"no exact product for {x}")


@pytest.mark.parametrize("x,want", [tuple(c) for c in GOLD["T9_bankers_rounding"]["cases"]])
def test_T9_bankers_rounding_vectors_on_device(ctx, pp, x, want):
    """src/misc.rs:279-296 through the vote's invalid threshold, bankers_rounding(depth * fraction_invalid)
    (src/pileup.rs:70-72): n matching reads give depth n, and n * fraction_invalid is the vector's argument bit for bit."""
    n, fi = _exact_product(x)
    assert float(n) * fi == x
    ref = _assembly("A")
    bases = np.frombuffer(ref.encode(), np.uint8)
    off = np.array([0, len(ref)], np.uint64)
    recs = _records(ref, [("A", n, 1.0)])
    kw = dict(min_depth=5, fraction_valid=0.95, fraction_invalid=fi)
    for name, got in (("host batch", ctx.polish_records(off, bases, recs, positions=True, **kw)),
                      ("k_tile's own votes", ctx.polish_records(off, bases, recs, positions=3, **kw)),
                      ("device batch + mirrors", _device_batch(ctx, pp, off, bases, recs, True, **kw) if n else None)):
        if got is None:
            continue
        pos = got["positions"]
        assert pos["depth"][P] == float(n) and pos["count_a"][P] == n, (name, pos["depth"][P])
        assert pos["invalid_thr"][P] == want, (name, x, n, fi, int(pos["invalid_thr"][P]))


def test_T10_reverse_complement_vectors_through_the_device_tokenizer(ctx, pp, tmp_path):
    """src/misc.rs:298-304 through the "*" fill of src/alignment.rs:290-295,161-167: a primary on the reverse strand carries
    the SEQ, its secondary on the forward strand carries "*" and gets reverse_complement(SEQ).  Alignment::new upper-cases SEQ
    first (src/alignment.rs:94), so the lower-case vectors come out as the upper case of the reference's expectation."""
    cases = GOLD["T10_reverse_complement"]["cases"]
    contig = "".join("ACGT"[(i * 7 + i // 3) % 4] for i in range(400))
    fa = tmp_path / "a.fasta"
    fa.write_text(f">c\n{contig}\n")
    lines = ["@SQ\tSN:c\tLN:400"]
    for j, (seq, _) in enumerate(cases):
        cig = f"{len(seq)}M"
        lines.append(f"r{j}\t16\tc\t{10 + j}\t60\t{cig}\t*\t0\t0\t{seq}\t*\tNM:i:0")
        lines.append(f"r{j}\t256\tc\t{200 + j}\t0\t{cig}\t*\t0\t0\t*\t*\tNM:i:0")
    sam = tmp_path / "a.sam"
    sam.write_text("\n".join(lines) + "\n")
    for ingest in (pp.ingest_device, lambda c, f, s, **kw: pp.ingest(f, s, **kw)):
        recs = ingest(ctx, str(fa), [str(sam)], max_errors=1000)[4]
        assert len(recs["contig"]) == 2 * len(cases) and (recs["k"] == 2).all()
        for j, (seq, rc) in enumerate(cases):
            got = {}
            for r in (2 * j, 2 * j + 1):
                so, sl = int(recs["seq_off"][r]), int(recs["seq_len"][r])
                got[int(recs["ref_start"][r])] = recs["seq"][so:so + sl].tobytes().decode()
            assert got[9 + j] == seq.upper(), (j, got)
            assert got[199 + j] == rc.upper(), (j, got)


def test_T3_ref_end_vectors_through_the_filter_kernels(ctx, pp):
    """src/alignment.rs:402-422: 4M, 2=1X1=, 2M1I1M, 2M1D1M at POS 1000 -> ref_start 999, ref_end 1003 / 1003 / 1002 / 1003.
    k_ref_end's value is read off the insert size of a pair whose mate is one base at the same start
    (get_insert_size, src/filter.rs:212-218: max - min over the four ends)."""
    cases = GOLD["T3_ref_positions"]["cases"]
    n = len(cases)
    ops = "MIDNSHP=X"

    def runs(cigar):
        out, num = [], ""
        for ch in cigar:
            if ch.isdigit():
                num += ch
            else:
                out.append((int(num) << 4) | ops.index(ch))
                num = ""
        return out

    def ffile(cigars, pos):
        packed = [runs(c) for c in cigars]
        n_cig = np.array([len(p) for p in packed], np.uint32)
        arr = dict(ref_id=np.zeros(n, np.uint32), ref_start=np.array(pos, np.uint32) - 1, flags=np.zeros(n, np.uint32),
                   cig_off=(np.cumsum(n_cig) - n_cig).astype(np.uint64), n_cig=n_cig,
                   cigar=np.array([r for p in packed for r in p], np.uint32), read=np.arange(n, dtype=np.uint32),
                   grp_off=np.arange(n + 1, dtype=np.uint32), grp_idx=np.arange(n, dtype=np.uint32))
        f = pp.FilterFile(n, arr["ref_id"].ctypes.data, arr["ref_start"].ctypes.data, arr["flags"].ctypes.data,
                          arr["cig_off"].ctypes.data, arr["n_cig"].ctypes.data, arr["cigar"].ctypes.data, len(arr["cigar"]),
                          arr["read"].ctypes.data, arr["grp_off"].ctypes.data, arr["grp_idx"].ctypes.data)
        return f, arr
    f1, k1 = ffile([c[0] for c in cases], [1000] * n)
    f2, k2 = ffile(["1M"] * n, [1000] * n)
    inp = pp.FilterInput(n, (pp.FilterFile * 2)(f1, f2))
    L = pp.lib()
    assert L.pp_filter_begin(ctx._h, C.byref(inp), pp.MEM_HOST) == 0, L.pp_last_error(ctx._h)
    orient, insert = np.zeros(n, np.uint8), np.zeros(n, np.uint32)
    assert L.pp_filter_samples(ctx._h, orient.ctypes.data, insert.ctypes.data) == 0
    assert [c[1] for c in cases] == [999] * n  # ref_start = POS - 1 (src/alignment.rs:58-61)
    assert [999 + int(v) for v in insert] == [c[2] for c in cases], list(insert)


def test_gzipped_assembly_through_the_cli(tmp_path):
    """SURVEY 8f-4 (src/misc.rs:81-99,136-167): `polypolish polish assembly.fasta.gz ...` against the oracle's CLI, plain and
    gzipped, and the two errors of is_file_gzipped / load_fasta on files that are too small or not a gzip stream."""
    from oracle import orc
    orc.build()
    ds = synth.rich_dataset(str(tmp_path), seed=77, contig_lens=(6000, 2500), coverage=35, lowercase_frac=0.1)
    gz = str(tmp_path / "asm.fasta.gz")
    with open(ds["fasta"], "rb") as f, gzip.open(gz, "wb") as g:
        g.write(f.read())
    exe = os.path.join(ROOT, "bin", "polypolish")
    outs = {}
    for name, binary in (("product", exe), ("oracle", orc.BIN_PATH)):
        for asm in (ds["fasta"], gz):
            r = subprocess.run([binary, "polish", asm, ds["sam1"], ds["sam2"]], capture_output=True)
            assert r.returncode == 0, (name, asm, r.stderr[-400:])
            outs[(name, asm.endswith(".gz"))] = r.stdout
    assert outs[("product", True)] == outs[("product", False)] == outs[("oracle", False)] == outs[("oracle", True)]
    assert outs[("product", True)].count(b">") == 2 and b" polypolish\n" in outs[("product", True)]
    # a file of one byte: "<path> is too small" (src/misc.rs:92-95); a gzip magic with garbage behind it: the load fails
    tiny = tmp_path / "tiny.fasta"
    tiny.write_bytes(b">")
    bad = tmp_path / "bad.fasta.gz"
    bad.write_bytes(bytes([31, 139]) + b"this is not a deflate stream at all" * 4)
    for path, text in ((tiny, b"is too small"), (bad, None)):
        got = subprocess.run([exe, "polish", str(path), ds["sam1"]], capture_output=True)
        want = subprocess.run([orc.BIN_PATH, "polish", str(path), ds["sam1"]], capture_output=True)
        assert got.returncode == want.returncode == 1 and got.stdout == b"", (path, got.returncode, want.returncode)
        got_msg = got.stderr[got.stderr.index(b"Error:"):].strip()
        want_msg = want.stderr[want.stderr.index(b"Error:"):].strip()
        assert got_msg == want_msg, (got_msg, want_msg)
        if text:
            assert text in got_msg
