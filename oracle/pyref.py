"""pyref.py -- second, independent restatement of the Polypolish hot path (pure Python).

TEST INFRASTRUCTURE, NOT PRODUCT.  Only tests/ may import this module.  It exists
so that the C oracle (pp_oracle.c) can be differentially tested: both were written
from the behaviour described in SURVEY.md section 3/8 and the cited reference lines, in
different languages and with different data structures (this one keeps whole
strings in dicts, exactly one dict per assembly position), and they must agree
byte-for-byte on randomised inputs.  Pure-Python loops: use it for small cases only.

Reference citations are file:line into /root/reference/src/.
"""
from __future__ import annotations

import gzip
import math
import re

CIGAR_TOKEN = re.compile(r"\d+[MIDNSHP=X]")  # alignment.rs:27-29


class Quit(Exception):
    """misc.rs:29-33 quit_with_error -> exit code 1."""


class Panic(Exception):
    """A Rust panic -> exit code 101."""


# ---------------------------------------------------------------- misc.rs
def bankers_rounding(x: float) -> int:  # misc.rs:208-215
    rounded_down = int(x) if x > 0 else 0
    frac = x - math.trunc(x)
    if frac < 0.5:
        return rounded_down
    if frac > 0.5:
        return rounded_down + 1
    return rounded_down + (rounded_down & 1)


_COMP = dict(zip("ATGCatgcNnRYSWKMBVDHryswkmbvdh.-?", "TACGtacgNnYRSWMKVBHDyrswmkvbhd.-?"))


def reverse_complement(seq: str) -> str:  # misc.rs:170-191
    return "".join(_COMP.get(c, "N") for c in reversed(seq))


def load_fasta(path: str):  # misc.rs:38-167
    with open(path, "rb") as f:
        magic = f.read(2)
    if len(magic) < 2:
        raise Quit(f'"{path}" is too small')
    opener = gzip.open if magic == b"\x1f\x8b" else open
    with opener(path, "rb") as f:
        raw = f.read()
    try:
        text = raw.decode("utf-8")
    except UnicodeDecodeError:
        raise Quit(f'unable to load "{path}"')
    lines = text.split("\n")
    if lines and lines[-1] == "":
        lines.pop()
    recs = []
    name, desc, seq = "", "", []
    for line in lines:
        if line.endswith("\r"):
            line = line[:-1]
        if not line:
            continue
        if line.startswith(">"):
            if name:
                recs.append((name, desc, _ascii_upper("".join(seq))))
                seq = []
            m = re.search(r"[ \t\n\x0b\x0c\r]", line[1:])
            if m:
                name, desc = line[1 : 1 + m.start()], line[2 + m.start() :]
            else:
                name, desc = line[1:], ""
        else:
            if not name:
                raise Quit(f'"{path}" is not correctly formatted')
            seq.append(line)
    if name:
        recs.append((name, desc, _ascii_upper("".join(seq))))
    if not recs:
        raise Quit(f'"{path}" contains no sequences')
    for n, _, s in recs:
        if not n:
            raise Quit(f'"{path}" has an unnamed sequence')
        if not s:
            raise Quit(f'"{path}" has an empty sequence')
    if len({n for n, _, _ in recs}) < len(recs):
        raise Quit(f'"{path}" has a duplicated name')
    return recs


def _ascii_upper(s: str) -> str:
    return "".join(chr(ord(c) - 32) if "a" <= c <= "z" else c for c in s)


def _parse_uint(s: str, bits: int) -> int:
    t = s[1:] if s.startswith("+") else s
    if not t or not all("0" <= c <= "9" for c in t):
        raise Panic(f"cannot parse {s!r} as unsigned")
    v = int(t)
    if v >= 1 << bits:
        raise Panic(f"{s!r} overflows u{bits}")
    return v


# ---------------------------------------------------------------- alignment.rs
def get_expanded_cigar(cigar: str):  # alignment.rs:325-346; None = invalid
    if cigar == "*":
        return ""
    out, total = [], 0
    for m in CIGAR_TOKEN.finditer(cigar):
        tok = m.group(0)
        if not tok[:-1].isascii():
            raise Panic("non-ASCII digit in CIGAR")
        out.append(tok[-1] * _parse_uint(tok[:-1], 32))
        total += len(tok)
    if total != len(cigar):
        return None
    return "".join(out)


def get_ref_end(ref_start: int, cigar: str) -> int:  # alignment.rs:138-149
    end = ref_start
    for m in CIGAR_TOKEN.finditer(cigar):
        tok = m.group(0)
        if tok[-1] in "MDN=X":
            end += int(tok[:-1])
    return end


class Alignment:
    __slots__ = ("read_name", "ref_name", "flags", "ref_start", "cigar", "expanded", "read_seq",
                 "mismatches", "pass_qc")

    @staticmethod
    def new(line: str):  # alignment.rs:49-98; returns Alignment or an error string
        parts = line.split("\t")
        if len(parts) < 11:
            return "too few columns"
        a = Alignment()
        a.read_name = parts[0]
        a.flags = _parse_uint(parts[1], 32)
        a.ref_name = parts[2]
        a.ref_start = max(0, _parse_uint(parts[3], 64) - 1)
        a.cigar = parts[5]
        a.mismatches = None
        a.pass_qc = True
        for p in parts[11:]:
            if p.startswith("NM:i:"):
                a.mismatches = _parse_uint(p[5:], 32)
                if a.mismatches == 0xFFFFFFFF:
                    a.mismatches = None  # u32::MAX doubles as "absent" in the reference
            if p.lower() == "zp:z:fail" and p.isascii():
                a.pass_qc = False
        if a.mismatches is None and not a.flags & 4:
            return "missing NM tag"
        a.expanded = get_expanded_cigar(a.cigar)
        if a.expanded is None:
            raise Quit(f'encountered an invalid CIGAR string for read {a.read_name}: "{a.cigar}"')
        a.read_seq = _ascii_upper(parts[9])
        if a.mismatches is None:
            a.mismatches = 0xFFFFFFFF
        return a

    @staticmethod
    def new_quick(line: str):  # alignment.rs:102-128
        parts = line.split("\t")
        if len(parts) < 11:
            return "too few columns"
        a = Alignment()
        a.read_name = parts[0]
        a.flags = _parse_uint(parts[1], 32)
        a.ref_name = parts[2]
        a.ref_start = max(0, _parse_uint(parts[3], 64) - 1)
        a.cigar = parts[5]
        return a

    def is_aligned(self):
        return not self.flags & 4

    def forward(self):
        return not self.flags & 16

    def ref_end(self):
        return get_ref_end(self.ref_start, self.cigar)


def read_slices(a: Alignment):  # alignment.rs:175-201 + 364-378
    """Strings contributed to consecutive reference positions, after the trim."""
    pieces = []  # each entry a list [start, end)
    i = 0
    for op in a.expanded:
        if op in "M=X":
            pieces.append([i, i + 1])
            i += 1
        elif op == "I":
            if not pieces:
                raise Panic("insertion with no preceding reference base")
            pieces[-1][1] = i + 1
            i += 1
        elif op == "D":
            pieces.append([i, i])
        else:
            raise Quit(f"unexpected character (other than M, =, X, I or D) in CIGAR string for read "
                       f'{a.read_name}: "{a.cigar}" - did you use BWA MEM to generate your alignments?')
    if i != len(a.read_seq):
        raise Quit(f"CIGAR string for read {a.read_name} does not match read sequence")
    strs = [a.read_seq[s:e] for s, e in pieces]
    if not strs:
        raise Panic("trim on an empty list")
    last = strs[-1]
    while strs and strs[-1] == last:
        strs.pop()
    if strs:
        strs.pop()
    return strs


# ---------------------------------------------------------------- pileup.rs
STATUS = ("kept", "changed", "low_depth", "none", "multiple", "too_close")


class Position:
    __slots__ = ("original", "depth", "counts")

    def __init__(self, original):
        self.original = original
        self.depth = 0.0
        self.counts = {}

    def add(self, s, dc):  # pileup.rs:56-65
        self.counts[s] = self.counts.get(s, 0) + 1
        self.depth += dc

    def vote(self, min_depth, fv, fi):  # pileup.rs:67-134
        valid_thr = max(min_depth, bankers_rounding(self.depth * fv))
        invalid_thr = bankers_rounding(self.depth * fi)
        valid, middle = [], 0
        keys = ["A", "C", "G", "T"] + [k for k in self.counts if k not in ("A", "C", "G", "T")]
        for k in keys:
            n = self.counts.get(k, 0)
            if n >= valid_thr:
                valid.append(k)
            elif n >= invalid_thr:
                middle += 1
        new, status = self.original, "kept"
        if self.depth < float(min_depth):
            status = "low_depth"
        elif len(valid) == 1:
            if middle:
                status = "too_close"
            else:
                new = valid[0]
                if new != self.original:
                    status = "changed"
        elif not valid:
            status = "none"
        else:
            status = "multiple"
        return new, status, valid_thr, invalid_thr

    def count_str(self):  # pileup.rs:137-148
        return ",".join(sorted(f"{k}x{n}" for k, n in self.counts.items()))


def _fmt_1(x: float) -> str:
    return f"{x:.1f}"


# ---------------------------------------------------------------- polish.rs
def _lines(path):
    with open(path, "rb") as f:
        raw = f.read()
    out = raw.split(b"\n")
    if out and out[-1] == b"":
        out.pop()
    res = []
    for b in out:
        if b.endswith(b"\r"):
            b = b[:-1]
        try:
            res.append(b.decode("utf-8"))
        except UnicodeDecodeError:
            res.append(None)
    return res


def _one_read(group, pileups, max_errors, careful):  # alignment.rs:275-305
    if careful and len(group) > 1:
        return 0
    src = next((a for a in group if a.read_seq != "*"), None)
    if src is None:
        if not group:
            raise Panic("empty read group")
        raise Quit(f"no alignments for read {group[0].read_name} contain sequence")
    seq, fwd = src.read_seq, src.forward()
    good = []
    for a in group:
        if not a.expanded:
            raise Panic("empty expanded CIGAR")
        if a.expanded[0] in "M=" and a.expanded[-1] in "M=" and a.mismatches <= max_errors and a.pass_qc:
            good.append(a)
    if not good:
        return 0
    dc = 1.0 / len(good)
    for a in good:
        if a.read_seq == "*":
            a.read_seq = seq if a.forward() == fwd else reverse_complement(seq)
    for a in good:
        if a.ref_name not in pileups:
            raise Quit(f"query name {a.ref_name} in SAM but not in assembly")
        pile = pileups[a.ref_name]
        for j, s in enumerate(read_slices(a)):
            if a.ref_start + j >= len(pile):
                raise Panic("index out of bounds")
            pile[a.ref_start + j].add(s if s else "-", dc)
    return len(good)


def polish(assembly, sams, fraction_invalid=0.2, fraction_valid=0.5, max_errors=10, min_depth=5,
           careful=False, debug=False):
    """polish.rs:26-38.  Returns (fasta_text, debug_text_or_None, per_position_list)."""
    if not 0.0 < fraction_valid < 1.0:
        raise Quit("--fraction_valid must be between 0 and 1 (exclusive)")
    if not 0.0 < fraction_invalid < 1.0:
        raise Quit("--fraction_invalid must be between 0 and 1 (exclusive)")
    if fraction_invalid >= fraction_valid:
        raise Quit("--fraction_invalid must be less than --fraction_valid")
    recs = load_fasta(assembly)
    pileups = {n: [Position(c) for c in s] for n, _, s in recs}
    for sam in sams:  # alignment.rs:225-272
        current, group, n_aligned = "", [], 0
        for ln, line in enumerate(_lines(sam), 1):
            if line is None:
                raise Quit(f'unable to load alignments from "{sam}"')
            if not line or line.startswith("@"):
                continue
            a = Alignment.new(line)
            if isinstance(a, str):
                raise Quit(f'{a} in "{sam}" (line {ln})')
            if not a.is_aligned():
                continue
            n_aligned += 1
            if current == "" or current == a.read_name:
                group.append(a)
            else:
                _one_read(group, pileups, max_errors, careful)
                group = [a]
            current = a.read_name
        _one_read(group, pileups, max_errors, careful)
        if not n_aligned:
            raise Quit(f'no alignments in "{sam}"')
    fasta, dbg, per_pos = [], [], []
    if debug:
        dbg.append("name\tpos\tbase\tdepth\tinvalid\tvalid\tpileup\tstatus\tnew_base\n")
    for name, desc, _ in recs:  # polish.rs:157-203
        out = []
        for i, pos in enumerate(pileups[name]):
            new, status, vt, it = pos.vote(min_depth, fraction_valid, fraction_invalid)
            per_pos.append((pos.depth, dict(pos.counts), vt, it, status, new))
            if debug:
                dbg.append(f"{name}\t{i}\t{pos.original}\t{_fmt_1(pos.depth)}\t{it}\t{vt}\t"
                           f"{pos.count_str()}\t{status}\t{new}\n")
            out.append(new)
        seq = "".join(out).replace("-", "")
        fasta.append(f">{name}{' ' + desc if desc else ''} polypolish\n{seq}\n")
    return "".join(fasta), ("".join(dbg) if debug else None), per_pos


# ---------------------------------------------------------------- filter.rs
ORIENT = ("fr", "rf", "ff", "rr")


def get_orientation(a1, a2) -> str:  # filter.rs:189-209
    s1, s2 = ("f" if a1.forward() else "r"), ("f" if a2.forward() else "r")
    p1 = a1.ref_start if a1.forward() else a1.ref_end()
    p2 = a2.ref_start if a2.forward() else a2.ref_end()
    if s1 != s2:
        return s1 + s2 if p1 < p2 else s2 + s1
    if s1 == "f":
        return "ff" if p1 < p2 else "rr"
    return "ff" if p2 < p1 else "rr"


def get_insert_size(a1, a2) -> int:  # filter.rs:212-218
    pos = (a1.ref_start, a1.ref_end(), a2.ref_start, a2.ref_end())
    return (max(pos) - min(pos)) & 0xFFFFFFFF


def get_percentile(sorted_list, p):  # filter.rs:249-259
    if not sorted_list:
        return 0
    rank = max(1, math.ceil((p / 100.0) * len(sorted_list)))
    return sorted_list[rank - 1] if rank - 1 < len(sorted_list) else 0


def filter_pairs(in1, in2, orientation="auto", low=0.1, high=99.9):
    """filter.rs:26-37.  Returns (out1_bytes, out2_bytes, report dict)."""
    if low <= 0.0 or low >= 50.0:
        raise Quit("--low must be greater than 0 and less than 50")
    if high <= 50.0 or high >= 100.0:
        raise Quit("--high must be greater than 50 and less than 100")
    table = {}
    for path, suffix in ((in1, "_1"), (in2, "_2")):  # filter.rs:110-145
        for ln, line in enumerate(_lines(path), 1):
            if line is None:
                raise Quit(f'unable to load alignments from "{path}"')
            if line.startswith("@"):
                continue
            a = Alignment.new_quick(line)
            if isinstance(a, str):
                raise Quit(f'{a} in "{path}" (line {ln})')
            if a.is_aligned():
                table.setdefault(a.read_name + suffix, []).append(a)
        if not table:
            raise Quit(f'no alignments found in "{path}"')
    sizes = {}
    for key, al1 in table.items():  # filter.rs:148-186
        if not key.endswith("_1") or len(al1) != 1:
            continue
        al2 = table.get(key[:-2] + "_2")
        if al2 is not None and len(al2) == 1 and al1[0].ref_name == al2[0].ref_name:
            sizes.setdefault(get_orientation(al1[0], al2[0]), []).append(get_insert_size(al1[0], al2[0]))
    if not sizes:
        raise Quit("no one-alignment-per-read pairs available to determine orientation and insert "
                   "size thresholds")
    counts = [len(sizes.get(o, ())) for o in ORIENT]
    if orientation == "auto":  # filter.rs:238-246
        best = [o for o, c in zip(ORIENT, counts) if c == max(counts)]
        if len(best) != 1:
            raise Quit("could not automatically determine read pair orientation")
        correct = best[0]
    else:
        correct = orientation
    chosen = sorted(sizes.get(correct, []))
    if not chosen:
        raise Quit("no read pairs available to determine insert size thresholds")
    lo, hi = get_percentile(chosen, low), get_percentile(chosen, high)

    outs, after = [], 0
    for path, this_sfx, pair_sfx in ((in1, "_1", "_2"), (in2, "_2", "_1")):  # filter.rs:296-349
        out = []
        for line in _lines(path):
            if line.startswith("@"):
                out.append(line + "\n")
                continue
            a = Alignment.new_quick(line)
            if isinstance(a, str):
                raise Panic(a)
            if not a.is_aligned():
                out.append(line + "\n")
                continue
            this = table[a.read_name + this_sfx]
            pair = table.get(a.read_name + pair_sfx, [])
            ok = (not pair) or len(this) == 1 or any(  # filter.rs:352-377
                a.ref_name == p.ref_name and lo <= get_insert_size(a, p) <= hi
                and get_orientation(a, p) == correct for p in pair)
            if ok:
                after += 1
                out.append(line + "\n")
            else:
                out.append(line + "\tZP:Z:fail\n")
        outs.append("".join(out).encode("utf-8"))
    before = sum(len(v) for v in table.values())
    return outs[0], outs[1], {"before": before, "after": after, "low": lo, "high": hi,
                              "orientation": correct, "counts": counts}
