"""Seeded synthetic inputs for the parity tests (SURVEY.md section 8d recipe, scaled down).

Two generators:

* ``rich_dataset``  -- slow, per-read Python: truth genome -> assembly with planted
  substitutions / 1-bp indels, paired reads with sequencing errors (subs, indels, N), exact
  CIGAR / NM computed by construction against the *assembly*, repeats (direct and inverted
  copies) with all-hits secondary records (FLAG 256, SEQ '*'), soft-clipped and unaligned
  records, lower-case SEQ, stray tags.  Writes assembly FASTA + _1.sam / _2.sam.
* ``fast_records``  -- vectorised numpy: M-only reads (plus optional single-indel reads)
  delivered directly as the C-ABI structure-of-arrays; used for the medium-size kernel tests
  and mirrored (on the GPU, with torch) by bench.py.
"""
from __future__ import annotations

import os

import numpy as np

COMP = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}
OPS = "MIDNSHP=X"
OPCODE = {c: i for i, c in enumerate(OPS)}


def revcomp(s: str) -> str:
    return "".join(COMP.get(c, "N") for c in reversed(s))


def _rand_seq(rng, n):
    return "".join("ACGT"[i] for i in rng.integers(0, 4, n))


def _compress(ops):
    out, prev, run = [], None, 0
    for o in ops:
        if o == prev:
            run += 1
        else:
            if prev is not None:
                out.append(f"{run}{prev}")
            prev, run = o, 1
    if prev is not None:
        out.append(f"{run}{prev}")
    return "".join(out)


class Contig:
    """Truth sequence, its (erroneous) assembly and the column map between them."""

    def __init__(self, name, truth, rng, asm_err_rate):
        self.name, self.truth = name, truth
        asm, pos_of, extra_after = [], [], []
        for t, base in enumerate(truth):
            extra = ""
            r = rng.random()
            if 5 < t < len(truth) - 5 and r < asm_err_rate:
                kind = rng.integers(0, 3)
                if kind == 0:  # substitution in the assembly
                    pos_of.append(len(asm))
                    asm.append("ACGT"[("ACGT".index(base) + int(rng.integers(1, 4))) % 4])
                elif kind == 1:  # assembly lacks this truth base
                    pos_of.append(None)
                else:  # assembly has an extra base after this one
                    pos_of.append(len(asm))
                    asm.append(base)
                    extra = "ACGT"[int(rng.integers(0, 4))]
                    asm.append(extra)
            else:
                pos_of.append(len(asm))
                asm.append(base)
            extra_after.append(len(extra))
        self.assembly = "".join(asm)
        self.pos_of, self.extra_after = pos_of, extra_after


def _align_columns(contig: Contig, t0, cols):
    """cols[j] = (base or None, inserted_str) for truth position t0+j.  Returns
    (ref_start0, cigar, nm, seq) of the read against contig.assembly, or None."""
    ops, seq, nm = [], [], 0
    ref_start, asm = None, contig.assembly
    L = len(cols)
    for j, (base, ins) in enumerate(cols):
        t = t0 + j
        ap = contig.pos_of[t]
        if ap is None:
            if base is not None:
                ops.append("I"); seq.append(base); nm += 1
        else:
            if ref_start is None:
                ref_start = ap
            if base is None:
                ops.append("D"); nm += 1
            else:
                ops.append("M"); seq.append(base)
                if base != asm[ap]:
                    nm += 1
        for b in ins:
            ops.append("I"); seq.append(b); nm += 1
        if j < L - 1:
            for _ in range(contig.extra_after[t]):
                if ref_start is not None:
                    ops.append("D"); nm += 1
    if ref_start is None or not seq:
        return None
    while ops and ops[0] == "D":  # an aligner never reports a leading/trailing deletion
        ops.pop(0); nm -= 1; ref_start += 1
    while ops and ops[-1] == "D":
        ops.pop(); nm -= 1
    return ref_start, _compress(ops), nm, "".join(seq)


def _read_columns(rng, truth_piece, sub_rate, indel_rate, n_rate):
    cols = []
    L = len(truth_piece)
    for j, b in enumerate(truth_piece):
        r = rng.random()
        ins = ""
        if r < sub_rate:
            b = "ACGT"[("ACGT".index(b) + int(rng.integers(1, 4))) % 4]
        elif r < sub_rate + n_rate:
            b = "N"
        elif r < sub_rate + n_rate + indel_rate and 0 < j < L - 1:
            if rng.random() < 0.5:
                b = None
            else:
                ins = _rand_seq(rng, int(rng.integers(1, 3)))
        cols.append((b, ins))
    return cols


def _flip_columns(cols):
    """Columns of the same read seen from the opposite strand."""
    L = len(cols)
    out = []
    for j2 in range(L):
        j = L - 1 - j2
        base = cols[j][0]
        base = None if base is None else COMP.get(base, "N")
        ins = revcomp(cols[j - 1][1]) if j - 1 >= 0 else ""
        out.append((base, ins))
    return out


def rich_dataset(outdir, seed=1, contig_lens=(4000, 2500), coverage=40, read_len=100,
                 ins_mean=300, ins_sd=30, sub_rate=0.004, indel_rate=0.002, n_rate=0.001,
                 asm_err_rate=0.004, repeat_len=0, repeat_copies=0, inverted=True,
                 unaligned_frac=0.01, clip_frac=0.02, lowercase_frac=0.05, zp_frac=0.0,
                 qual=True, prefix="ds"):
    """Write <prefix>.fasta, <prefix>_1.sam, <prefix>_2.sam under outdir; return their paths."""
    rng = np.random.default_rng(seed)
    contigs, repeats = [], []  # repeats: (contig_idx, truth_start, inverted?)
    for ci, n in enumerate(contig_lens):
        truth = _rand_seq(rng, n)
        if ci == 0 and repeat_len and repeat_copies >= 2:
            unit = _rand_seq(rng, repeat_len)
            gap = (n - repeat_copies * repeat_len) // (repeat_copies + 1)
            assert gap > read_len, "contig too short for the requested repeats"
            t, p = list(truth), gap
            for k in range(repeat_copies):
                inv = inverted and (k % 2 == 1)
                t[p:p + repeat_len] = revcomp(unit) if inv else unit
                repeats.append((ci, p, inv))
                p += repeat_len + gap
            truth = "".join(t)
        contigs.append(Contig(f"contig_{ci + 1}", truth, rng, asm_err_rate))

    def secondary_hits(ci, t0, L):
        """Other repeat copies fully containing truth interval [t0, t0+L) of a copy."""
        for (rc, rs, rinv) in repeats:
            if rc == ci and rs <= t0 and t0 + L <= rs + repeat_len:
                x = t0 - rs
                return [(oc, os_ + (x if oinv == rinv else repeat_len - x - L), oinv != rinv)
                        for (oc, os_, oinv) in repeats if (oc, os_) != (rc, rs)]
        return []

    lines = {1: [], 2: []}
    for f in (1, 2):
        lines[f].append("@HD\tVN:1.6\tSO:unsorted")
        for c in contigs:
            lines[f].append(f"@SQ\tSN:{c.name}\tLN:{len(c.assembly)}")
        lines[f].append("@PG\tID:synth\tPN:synth")
    qual_str = "I" * read_len if qual else "*"
    ridx = 0
    for ci, c in enumerate(contigs):
        n_pairs = int(len(c.truth) * coverage / (2 * read_len))
        for _ in range(n_pairs):
            ins = int(np.clip(round(rng.normal(ins_mean, ins_sd)), read_len + 10, 3 * ins_mean))
            if ins >= len(c.truth):
                continue
            fs = int(rng.integers(0, len(c.truth) - ins))
            name = f"r{ridx}"
            ridx += 1
            flip_pair = rng.random() < 0.5  # which mate is on the forward strand
            for mate in (1, 2):
                first = (mate == 1) != flip_pair
                t0 = fs if first else fs + ins - read_len
                strand_rev = not first
                if rng.random() < unaligned_frac:
                    lines[mate].append(f"{name}\t4\t*\t0\t0\t*\t*\t0\t0\t{_rand_seq(rng, read_len)}\t{qual_str}")
                    continue
                cols = _read_columns(rng, c.truth[t0:t0 + read_len], sub_rate, indel_rate, n_rate)
                recs = []
                al = _align_columns(c, t0, cols)
                if al is not None:
                    recs.append((c.name, al, strand_rev, False))
                for (oc, ot0, flipped) in secondary_hits(ci, t0, read_len):
                    cols2 = _flip_columns(cols) if flipped else cols
                    al2 = _align_columns(contigs[oc], ot0, cols2)
                    if al2 is not None:
                        recs.append((contigs[oc].name, al2, strand_rev != flipped, True))
                if not recs:
                    continue
                if len(recs) > 1 and rng.random() < 0.3:  # primary not always first in its group
                    k = int(rng.integers(1, len(recs)))
                    recs[0], recs[k] = recs[k], recs[0]
                for (rname, (rs, cigar, nm, seq), rev, secondary) in recs:
                    flag = (16 if rev else 0) | (256 if secondary else 0)
                    s, q = (("*", "*") if secondary else (seq, qual_str[:len(seq)] if qual else "*"))
                    if len(recs) == 1 and rng.random() < clip_frac:
                        k = int(rng.integers(1, 6))
                        cigar = f"{k}S{cigar}"
                        s = _rand_seq(rng, k) + s
                        q = "I" * len(s) if qual else "*"
                    if not secondary and rng.random() < lowercase_frac:
                        s = s.lower()
                    tags = [f"NM:i:{nm}"]
                    if rng.random() < 0.3:
                        tags.insert(0, "AS:i:90")
                    if rng.random() < 0.3:
                        tags.append("XS:i:0")
                    if rng.random() < zp_frac:
                        tags.append("ZP:Z:fail")
                    lines[mate].append(f"{name}\t{flag}\t{rname}\t{rs + 1}\t60\t{cigar}\t*\t0\t0\t{s}\t{q}\t"
                                       + "\t".join(tags))
    os.makedirs(outdir, exist_ok=True)
    fa = os.path.join(outdir, f"{prefix}.fasta")
    with open(fa, "w") as f:
        for i, c in enumerate(contigs):
            desc = " some description" if i == 0 else ""
            f.write(f">{c.name}{desc}\n")
            for k in range(0, len(c.assembly), 70):
                f.write(c.assembly[k:k + 70] + "\n")
    paths = [fa]
    for m in (1, 2):
        p = os.path.join(outdir, f"{prefix}_{m}.sam")
        with open(p, "w") as f:
            f.write("\n".join(lines[m]) + "\n")
        paths.append(p)
    return {"fasta": fa, "sam1": paths[1], "sam2": paths[2], "contigs": contigs}


# ---------------------------------------------------------------------------------------------
def fast_records(seed=0, contig_lens=(50_000,), coverage=60, read_len=150, sub_rate=0.002,
                 n_rate=1e-4, asm_sub_rate=2e-4, indel_read_frac=0.01, k_choices=(1,),
                 k_probs=None):
    """Vectorised generator of good-alignment records in the C-ABI SoA layout.

    Reads are sampled uniformly; a fraction ``indel_read_frac`` carries one 1-bp insertion or
    deletion at an interior offset (CIGAR aM1IbM / aM1DbM); ``k`` is drawn from ``k_choices``
    (depth share 1/k, as if the read had k good alignments).  Returns (contig_off, bases, recs).
    """
    rng = np.random.default_rng(seed)
    lens = np.asarray(contig_lens, dtype=np.int64)
    contig_off = np.zeros(len(lens) + 1, dtype=np.uint64)
    contig_off[1:] = np.cumsum(lens)
    G = int(contig_off[-1])
    truth = rng.integers(0, 4, G, dtype=np.uint8)
    asm = truth.copy()
    errs = rng.random(G) < asm_sub_rate
    asm[errs] = (asm[errs] + rng.integers(1, 4, int(errs.sum()), dtype=np.uint8)) % 4
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    bases = lut[asm]

    n_per = np.maximum(1, (lens * coverage // read_len)).astype(np.int64)
    contig = np.repeat(np.arange(len(lens), dtype=np.uint32), n_per)
    n = len(contig)
    span_max = read_len + 1
    hi = (lens[contig] - span_max).astype(np.int64)
    assert (hi > 0).all(), "contigs must be longer than a read"
    ref_start = (rng.random(n) * hi).astype(np.int64)
    order = rng.permutation(n)  # SAM files are in read order, not position order
    contig, ref_start = contig[order], ref_start[order]

    kind = np.zeros(n, dtype=np.int8)  # 0 plain, 1 insertion, 2 deletion
    sel = rng.random(n) < indel_read_frac
    kind[sel] = rng.integers(1, 3, int(sel.sum()))
    a = rng.integers(5, read_len - 5, n)  # indel offset within the read
    j = np.arange(read_len)[None, :]
    # reference offset of read base j
    off = j + np.where(kind[:, None] == 1, -(j > a[:, None]).astype(np.int64),
                       np.where(kind[:, None] == 2, (j >= a[:, None]).astype(np.int64), 0))
    gpos = contig_off[contig].astype(np.int64)[:, None] + ref_start[:, None] + off
    codes = truth[gpos]
    ins_here = (kind[:, None] == 1) & (j == a[:, None])
    codes = np.where(ins_here, rng.integers(0, 4, (n, 1), dtype=np.uint8), codes)
    sub = rng.random((n, read_len)) < sub_rate
    codes = np.where(sub, (codes + rng.integers(1, 4, (n, read_len), dtype=np.uint8)) % 4, codes)
    seq = lut[codes]
    seq[rng.random((n, read_len)) < n_rate] = ord("N")

    n_cig = np.where(kind == 0, 1, 3).astype(np.uint32)
    cig_off = np.zeros(n, dtype=np.uint64)
    cig_off[1:] = np.cumsum(n_cig)[:-1]
    cigar = np.zeros(int(n_cig.sum()), dtype=np.uint32)
    plain = kind == 0
    cigar[cig_off[plain].astype(np.int64)] = (read_len << 4) | OPCODE["M"]
    ix = np.nonzero(kind == 1)[0]
    base = cig_off[ix].astype(np.int64)
    cigar[base] = (a[ix] << 4) | OPCODE["M"]
    cigar[base + 1] = (1 << 4) | OPCODE["I"]
    cigar[base + 2] = ((read_len - a[ix] - 1) << 4) | OPCODE["M"]
    dx = np.nonzero(kind == 2)[0]
    base = cig_off[dx].astype(np.int64)
    cigar[base] = (a[dx] << 4) | OPCODE["M"]
    cigar[base + 1] = (1 << 4) | OPCODE["D"]
    cigar[base + 2] = ((read_len - a[dx]) << 4) | OPCODE["M"]

    k = rng.choice(np.asarray(k_choices, dtype=np.uint32), size=n, p=k_probs).astype(np.uint32)
    recs = {
        "contig": contig.astype(np.uint32),
        "ref_start": ref_start.astype(np.uint32),
        "k": k,
        "seq_off": (np.arange(n, dtype=np.uint64) * np.uint64(read_len)),
        "seq_len": np.full(n, read_len, dtype=np.uint32),
        "cig_off": cig_off,
        "n_cig": n_cig,
        "seq": np.ascontiguousarray(seq.reshape(-1)),
        "cigar": cigar,
    }
    return contig_off, bases, recs


def records_to_sam(contig_off, bases, recs, names=None, path_fasta=None, path_sam=None):
    """Write the records as a single-end SAM (+ FASTA) so the text path can be driven with the
    same data.  Groups of k>1 are NOT reconstructed (each record is written with k=1 in mind),
    so use only with k == 1 records."""
    n_contigs = len(contig_off) - 1
    names = names or [f"c{i}" for i in range(n_contigs)]
    if path_fasta:
        with open(path_fasta, "w") as f:
            for i in range(n_contigs):
                f.write(f">{names[i]}\n{bytes(bases[int(contig_off[i]):int(contig_off[i + 1])]).decode()}\n")
    if path_sam:
        with open(path_sam, "w") as f:
            for i in range(n_contigs):
                f.write(f"@SQ\tSN:{names[i]}\tLN:{int(contig_off[i + 1] - contig_off[i])}\n")
            seq = recs["seq"]
            for i in range(len(recs["contig"])):
                so, sl = int(recs["seq_off"][i]), int(recs["seq_len"][i])
                co, nc = int(recs["cig_off"][i]), int(recs["n_cig"][i])
                cig = "".join(f"{int(x) >> 4}{OPS[int(x) & 15]}" for x in recs["cigar"][co:co + nc])
                f.write(f"r{i}\t0\t{names[int(recs['contig'][i])]}\t{int(recs['ref_start'][i]) + 1}\t60\t{cig}"
                        f"\t*\t0\t0\t{bytes(seq[so:so + sl]).decode()}\t*\tNM:i:0\n")


def oracle_engine(orc):
    """The oracle standing in for the per-rank device engine of polypolish_amd.distributed (tests only).
    An emit range is honoured by slicing the polished string with the per-position emit lengths."""
    def engine(o, b, r, emit=None, **kw):
        res = orc.polish_records(o, b, r, positions=emit is not None, **kw)
        if emit is None:
            return res
        cum = np.concatenate([[0], np.cumsum(res["positions"]["emit_len"].astype(np.int64))])
        out, offs = [], [0]
        for j, (lo, hi) in enumerate(np.asarray(emit, dtype=np.int64)):
            a, z = int(cum[int(o[j]) + lo]), int(cum[int(o[j]) + hi])
            out.append(res["polished"][a:z])
            offs.append(offs[-1] + z - a)
        return {"polished": b"".join(out), "offsets": np.array(offs, dtype=np.uint64)}
    return engine


def merge_records(a, b, seed=0):
    """Two record sets over the same assembly -> one, in a shuffled (read) order."""
    n_a, n_b = len(a["contig"]), len(b["contig"])
    out = {}
    for k in ("contig", "ref_start", "k", "seq_len", "n_cig"):
        out[k] = np.concatenate([a[k], b[k]])
    out["seq_off"] = np.concatenate([a["seq_off"], b["seq_off"] + np.uint64(len(a["seq"]))])
    out["cig_off"] = np.concatenate([a["cig_off"], b["cig_off"] + np.uint64(len(a["cigar"]))])
    out["seq"] = np.concatenate([a["seq"], b["seq"]])
    out["cigar"] = np.concatenate([a["cigar"], b["cigar"]])
    order = np.random.default_rng(seed).permutation(n_a + n_b)
    for k in ("contig", "ref_start", "k", "seq_len", "n_cig", "seq_off", "cig_off"):
        out[k] = out[k][order]
    return out


def mutate_sam(text: str, rng, n_mut=3) -> str:
    """A SAM text with a few random defects / oddities (for differential fuzzing of the parsers)."""
    lines = text.split("\n")
    body = [i for i, l in enumerate(lines) if l and not l.startswith("@")]
    if not body:
        return text
    for _ in range(n_mut):
        i = int(rng.choice(body))
        f = lines[i].split("\t")
        kind = int(rng.integers(0, 16))
        if kind == 0 and len(f) > 3:
            del f[int(rng.integers(0, len(f)))]                       # a column goes missing
        elif kind == 1 and len(f) > 1:
            f[1] = str(rng.choice(["x", "-1", "+16", "4", "99999999999", "", "0x10", "272"]))   # FLAG
        elif kind == 2 and len(f) > 3:
            f[3] = str(rng.choice(["0", "+5", "-3", "abc", "", "18446744073709551615", "18446744073709551616", "7"]))  # POS
        elif kind == 3 and len(f) > 5:
            f[5] = str(rng.choice(["*", "", "5M", "0M5M", "3S5M", "5M3S", "2M1I2M", "2M1D2M", "5Q", "M", "12", "4294967296M",
                                   "268435456M", "1=1X1=", "2M0I2M", "5M zz", "3H5M"]))      # CIGAR
        elif kind == 4 and len(f) > 9:
            f[9] = str(rng.choice(["*", "", "acgtn", "ACGTRYKM", "A" * int(rng.integers(1, 40))]))  # SEQ
        elif kind == 5:
            f = [x for x in f if not x.startswith("NM:i:")]          # NM tag removed
        elif kind == 6:
            f.append(str(rng.choice(["ZP:Z:fail", "zp:z:FAIL", "ZP:Z:failed", "NM:i:x", "NM:i:", "NM:i:+3", "NM:i:99", "XX:Z:", ""])))
        elif kind == 7:
            f[0] = str(rng.choice(["", "same", "same", "r 1", "@odd"]))  # QNAME (an empty one glues groups together)
        elif kind == 8 and len(f) > 2:
            f[2] = str(rng.choice(["*", "nope", "", "contig_1", "contig_2"]))  # RNAME
        elif kind == 9:
            lines.insert(i, "")                                      # an empty line
            body = [j if j < i else j + 1 for j in body]
            continue
        elif kind == 10:
            lines.insert(i, lines[i])                                # a duplicated record (same QNAME: one group)
            body = [j if j < i else j + 1 for j in body]
            continue
        elif kind == 11:
            f[-1] = f[-1] + "\r"                                     # CRLF on one line
        elif kind == 12 and len(f) > 10:
            f = f[:11]                                               # exactly eleven columns: no tags at all
        elif kind == 13:
            f.append("")                                             # trailing tab
        elif kind == 14 and len(f) > 1:
            f[1] = str(int(f[1]) ^ 16) if f[1].isdigit() else f[1]   # strand flipped
        else:
            j = int(rng.integers(0, max(1, len(lines[i]))))
            lines[i] = lines[i][:j] + str(rng.choice(["\t", "A", "0", ":", "*"])) + lines[i][j + 1:]
            continue
        lines[i] = "\t".join(f)
    out = "\n".join(lines)
    if rng.random() < 0.3:
        out = out.rstrip("\n")                                       # no final newline
    return out


def random_cigar_records(seed=0, contig_lens=(3000, 1200), n_reads=1500, max_ops=7, k_choices=(1, 1, 1, 2, 3), bad_frac=0.0,
                         n_rate=0.01):
    """Good-alignment records with RANDOM multi-operation CIGARs (M = X I D, several indels per read, indels next to
    each other, homopolymer ends ...) for differential fuzzing of the CIGAR walk.  With bad_frac > 0 some records
    carry a defect the walk must reject (an op it does not know, a CIGAR that disagrees with SEQ, a read running
    off its contig).  Returns (contig_off, bases, recs)."""
    rng = np.random.default_rng(seed)
    lens = np.asarray(contig_lens, dtype=np.int64)
    contig_off = np.zeros(len(lens) + 1, dtype=np.uint64)
    contig_off[1:] = np.cumsum(lens)
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    bases = lut[rng.integers(0, 4, int(contig_off[-1]))]
    # a few homopolymer stretches so that the right-end trim has something to chew on
    for _ in range(20):
        p = int(rng.integers(0, len(bases) - 12))
        bases[p:p + int(rng.integers(3, 12))] = lut[int(rng.integers(0, 4))]
    out = {k: [] for k in ("contig", "ref_start", "k", "seq_len", "n_cig")}
    seqs, cigs = [], []
    for _ in range(n_reads):
        c = int(rng.integers(0, len(lens)))
        ops = []
        n_ops = int(rng.integers(1, max_ops + 1))
        for i in range(n_ops):
            first_or_last = i == 0 or i == n_ops - 1
            op = "M=X"[int(rng.choice([0, 0, 0, 1, 2]))] if first_or_last else "M=XID"[int(rng.choice([0, 0, 1, 2, 3, 3, 4, 4]))]
            if first_or_last and op == "X":
                op = "M"
            ops.append((int(rng.integers(1, 25)) if op in "M=X" else int(rng.integers(1, 4)), op))
        span = sum(n for n, o in ops if o in "M=XD")
        if span >= lens[c] - 2:
            continue
        start = int(rng.integers(0, lens[c] - span))
        g = int(contig_off[c]) + start
        seq = bytearray()
        for n, o in ops:
            if o in "M=":
                seq += bytes(bases[g:g + n]); g += n
            elif o == "X":
                seq += bytes(lut[(np.searchsorted(lut, bases[g:g + n]) + 1) % 4]); g += n
            elif o == "I":
                seq += bytes(lut[rng.integers(0, 4, n)])
            else:
                g += n
        seq = np.frombuffer(bytes(seq), dtype=np.uint8).copy()
        if len(seq):
            sub = rng.random(len(seq)) < 0.02
            seq[sub] = lut[rng.integers(0, 4, int(sub.sum()))]
            seq[rng.random(len(seq)) < n_rate] = ord("N")
        packed = [(n << 4) | OPCODE[o] for n, o in ops]
        if rng.random() < bad_frac:
            kind = int(rng.integers(0, 4))
            if kind == 0:
                packed.insert(int(rng.integers(0, len(packed) + 1)), (int(rng.integers(1, 5)) << 4) | OPCODE[str(rng.choice(list("NSHP")))])
            elif kind == 1 and len(seq) > 2:
                seq = seq[:-1]
            elif kind == 2:
                seq = np.concatenate([seq, lut[rng.integers(0, 4, 2)]])
            else:
                start = int(lens[c]) - span + int(rng.integers(1, 4))   # runs off the contig
        out["contig"].append(c); out["ref_start"].append(start); out["k"].append(int(rng.choice(k_choices)))
        out["seq_len"].append(len(seq)); out["n_cig"].append(len(packed))
        seqs.append(seq); cigs.append(np.array(packed, dtype=np.uint32))
    recs = {k: np.array(v, dtype=np.uint32) for k, v in out.items()}
    sl, nc = recs["seq_len"].astype(np.uint64), recs["n_cig"].astype(np.uint64)
    recs["seq_off"] = (np.cumsum(sl) - sl).astype(np.uint64)
    recs["cig_off"] = (np.cumsum(nc) - nc).astype(np.uint64)
    recs["seq"] = np.concatenate(seqs) if seqs else np.zeros(0, np.uint8)
    recs["cigar"] = np.concatenate(cigs) if cigs else np.zeros(0, np.uint32)
    return contig_off, bases, recs
