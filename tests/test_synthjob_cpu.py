"""tools/synthjob.py (the synthetic workloads of SURVEY.md section 8d, used by bench.py and the full-size GPU tests)
checked on the CPU at small sizes against the oracle and against the product's host ingest: what the generator calls
the truth is what the reference's algorithm (as restated by the oracle) recovers, its text form and its resident-record
form describe the same job, and the product's host parser turns the text into exactly the resident records."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import synthjob  # noqa: E402

CPU = torch.device("cpu")


@pytest.fixture(scope="module")
def orc():
    from oracle import orc as o
    o.lib()
    return o


def _fasta_of(job, res):
    out = b""
    for c in range(len(job["contig_off"]) - 1):
        out += b">contig_%d polypolish\n" % (c + 1) + res["polished"][int(res["offsets"][c]):int(res["offsets"][c + 1])] + b"\n"
    return out


@pytest.mark.parametrize("recipe", ["survey", "subs"])
def test_planted_errors_are_what_the_reference_algorithm_repairs(orc, recipe):
    job = synthjob.make_job(CPU, contig_lens=[60_000, 9_000], coverage=60, seed=3, asm_err_rate=2e-3, recipe=recipe)
    p = job["planted"]
    if recipe == "survey":
        assert min(p["substitutions"], p["deletions"], p["insertions"]) >= 20 and p["indels_in_homopolymers"] >= 20
        assert job["G"] == len(job["truth"]) - p["deletions"] + p["insertions"]
        assert int((job["recs"]["n_cig"] > 1).sum()) > 0.1 * job["n_aln"]  # reads over planted indels carry I / D runs
    else:
        assert p["deletions"] == p["insertions"] == 0 and job["G"] == len(job["truth"])
    res = orc.polish_records(job["contig_off"], job["bases"].numpy(), synthjob.to_host_records(job))
    assert synthjob.recovered(job, res["polished"], res["offsets"])
    assert not synthjob.recovered(job, bytes(job["bases"].numpy()), job["contig_off"])  # the unpolished assembly is not the truth


def test_text_and_records_describe_the_same_job(orc, tmp_path):
    """configs[2]'s shape: repeat copies on both strands, all-hits groups (primary + secondary records with SEQ '*'),
    pairs, unaligned records, reads the gates reject.  Oracle on the text == oracle on the resident records, and the
    product's host ingest makes exactly those records out of the text (array by array)."""
    import polypolish_amd as pp
    job = synthjob.make_job(CPU, contig_lens=[120_000], coverage=40, seed=9, asm_err_rate=1e-3, pairs=True, unaligned_frac=1e-2,
                            repeat=(3000, 5))
    S = job["sam"]
    assert int((S["seq_len"] == 0).sum()) > 1000 and int(((S["flag"] & 256) != 0).sum()) == int((S["seq_len"] == 0).sum())
    assert int(((S["flag"] & 272) == 272).sum()) > 100  # secondary records on the reverse strand (inverted copies)
    assert job["n_records"] > job["n_aln"]  # some records are unaligned or fail the gates
    fa, sams = synthjob.write_sam_pair(job, str(tmp_path))
    h = synthjob.to_host_records(job)
    want = orc.polish_records(job["contig_off"], job["bases"].numpy(), h)
    got = orc.polish_files(fa, sams)
    assert got["fasta"] == _fasta_of(job, want)
    names, descs, off, bases, recs, counts = pp.ingest(fa, sams)
    assert np.array_equal(off, job["contig_off"]) and np.array_equal(bases, job["bases"].numpy())
    assert sum(c[1] for c in counts) == job["n_aln"]
    for k in h:
        assert np.array_equal(h[k], recs[k]), k
    # ... and the window-order mirror of the records the ingest hands over with them (pp_aln_batch.wo) is the one the bench
    # keeps resident (tools/synthjob.py wo_of) -- and the one the binding's numpy definition gives
    wo = synthjob.wo_of(job).numpy().view(np.uint8).reshape(-1).view(pp.WO_DTYPE)
    for k in pp.WO_DTYPE.names:
        assert np.array_equal(wo[k], recs["wo"][k]), k
    ref = pp.window_order_mirror(recs, off, [c[1] for c in counts])
    assert all(np.array_equal(ref[k], recs["wo"][k]) for k in pp.WO_DTYPE.names)
    assert [int(e) for e in recs["wo_runs"]] == synthjob.with_wo(job)["wo_runs"]   # ... with the same run table (pp_aln_batch.wo_run_end)
    assert int((h["k"] == 5).sum()) > 1000


def test_cli_argument_grammar_is_claps():
    """The reference derives its parser with clap 4 (src/main.rs:23-109): -d4, -d=4, --min_depth=4, `--`, an option given
    twice, flags with a value, unknown options.  Usage errors leave with 2 before anything else happens; accepted
    command lines get as far as the device (exit 1 here: this box has none; on a GPU box they run)."""
    exe = os.path.join(ROOT, "bin", "polypolish")
    has_gpu = torch.cuda.is_available()

    def rc(*args):
        r = subprocess.run([exe] + list(args), capture_output=True)
        return r.returncode, r.stderr

    for args in (["polish", "-d", "4", "-d", "5", "x.fa"], ["polish", "-dx", "x.fa"], ["polish", "--bogus", "x.fa"],
                 ["polish", "-q", "x.fa"], ["polish", "--careful=1", "x.fa"], ["polish", "-d"], ["polish"],
                 ["polish", "-i", "abc", "x.fa"], ["filter", "--in1", "a", "--in2", "b", "--out1", "c"],
                 ["filter", "--in1", "a", "--in2", "b", "--out1", "c", "--out2", "d", "--low", "x"],
                 ["filter", "--in1", "a", "--in2", "b", "--out1", "c", "--out2", "d", "extra"], ["bogus"],
                 # an option's value is never taken from an argument that looks like an option (clap: "a value is required"):
                 ["polish", "--debug", "--careful", "x.fa"], ["polish", "-m", "-5", "x.fa"], ["polish", "-d", "--", "x.fa"],
                 ["filter", "--in1", "--in2", "b", "--out1", "c", "--out2", "d"]):
        code, err = rc(*args)
        assert code == 2 and err.startswith(b"error:"), (args, code, err)
    for args, opt in ((["polish", "--debug", "--careful", "x.fa"], b"--debug <DEBUG>"), (["polish", "-m", "-5", "x.fa"], b"--max_errors <MAX_ERRORS>")):
        code, err = rc(*args)
        assert b"a value is required for '" + opt + b"' but none was supplied" in err, (args, err)
        assert not os.path.exists("--careful")
    for args in (["polish", "-d4", "missing.fa"], ["polish", "-d=4", "missing.fa"], ["polish", "--min_depth=4", "--", "-odd-name.fa"],
                 ["polish", "--debug", "-", "missing.fa"], ["polish", "--debug=-x.tsv", "missing.fa"],   # a lone "-" and an attached value are values
                 ["polish", "-m3", "-i0.1", "-v=0.6", "--careful", "missing.fa", "a.sam", "b.sam"]):
        code, err = rc(*args)
        assert code == 1, (args, code, err)
        assert (b"file does not exist" in err) if has_gpu else (b"no usable MI355X" in err), (args, err)
    for args in (["-h"], ["--help"], ["polish", "-h"], ["polish", "-hV"], ["filter", "--help"], ["-V"], ["polish", "-V"]):
        r = subprocess.run([exe] + args, capture_output=True)
        assert r.returncode == 0 and r.stdout, args


def test_the_4bit_mirror_of_a_seq_array_as_the_header_defines_it():
    """pp_aln_batch.seq4 (include/polypolish_hip.h): base i of the seq ARRAY in bits 4*(i&1).. of byte i >> 1, codes
    PP_SEQ4_* read off the header; the binding's packer (the tests' reference for the device tokenizer's mirror) and the
    bench's torch packer agree with that definition and with each other, for even and odd lengths."""
    import re
    import torch
    import polypolish_amd as pp
    hdr = open(os.path.join(ROOT, "include", "polypolish_hip.h")).read()
    code = {m.group(1): int(m.group(2)) for m in re.finditer(r"#define PP_SEQ4_(\w+) (\d+)", hdr)}
    assert code == {"A": 0, "C": 1, "T": 2, "G": 3, "N": 4, "DASH": 5, "OTHER": 15}
    want = {ord("A"): code["A"], ord("C"): code["C"], ord("T"): code["T"], ord("G"): code["G"], ord("N"): code["N"],
            ord("-"): code["DASH"]}
    rng = np.random.default_rng(5)
    for n in (0, 1, 2, 9, 1000, 1001):
        seq = np.frombuffer(b"ACGTNRYacgtn-.*", np.uint8)[rng.integers(0, 15, n)].copy()
        m = pp.pack_seq4(seq)
        assert len(m) >= (n + 1) // 2 + 32                     # the slack the header asks for
        for i in range(n):
            assert (int(m[i >> 1]) >> (4 * (i & 1))) & 15 == want.get(int(seq[i]), code["OTHER"]), (n, i)
        t = synthjob.seq4_of(torch.from_numpy(seq), chunk=64).numpy()
        assert np.array_equal(t[:(n + 1) // 2], m[:(n + 1) // 2])
