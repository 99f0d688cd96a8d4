"""synthjob.py -- the synthetic polish workloads of SURVEY.md section 8(d), generated with torch on whatever device is
given (the MI355X for the full-size configurations, the CPU for small test cases).  Measurement / test infrastructure,
not product code: bench.py, the GPU tests and tools/ use it.

Recipe (`recipe="survey"`, the default; `"subs"` keeps the round-1/2 substitution-only assembly for continuity):

* truth genome: i.i.d. uniform ACGT per contig (+ repeat copies, direct and inverted, for configs[2]);
* assembly = truth with errors at `asm_err_rate` per bp: 1/3 substitutions, 1/3 1-bp deletions (the assembly lacks a
  truth base), 1/3 1-bp insertions (the assembly has an extra base), half of the indels inside homopolymers of >= 3;
  no error within `end_margin` of a contig end or inside / next to a repeat copy;
* reads: truth substrings of `read_len` bases (uniform starts; `pairs=True`: fragments with insert ~ N(350, 35) clipped
  to [160, 700], orientation fr, mates in two files) with 0.2 % substitutions, 1e-4 N and -- `indel_read_frac` of the
  reads: 0.15 % with the survey recipe (its 1e-5 per base), 1 % with "subs" as in rounds 1 and 2 -- one 1-bp
  sequencing insertion or deletion;
* alignment records are computed BY CONSTRUCTION AGAINST THE ASSEMBLY: exact CIGAR with M / I / D runs (an assembly
  deletion under a read is an I, an assembly insertion a D), NM = edit count, POS in assembly coordinates
  (reference semantics: src/alignment.rs:175-201 walks exactly these runs, :349-378 trims them);
* configs[2]: a read that lies inside a repeat copy gets one record per copy, adjacent in the file: the primary with its
  SEQ, the others as secondary records (FLAG 256, strand flipped on an inverted copy, SEQ/QUAL "*" in the text;
  src/alignment.rs:290-295,311-322 fills them), all with depth share 1/copies.

`make_job` returns the GOOD records (gates of src/alignment.rs:282-287 applied: first/last op M, NM <= 10) as the C-ABI
structure of arrays, resident on `device`, plus -- under "sam" -- the columns of EVERY record for the text writer.
"""
from __future__ import annotations

import ctypes
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OP_M, OP_I, OP_D = 0, 1, 2
MAX_ERRORS = 10  # the CLI's default --max_errors (src/main.rs:93-95)
# sequencing indel error 1e-5 per base -> a 150-base read carries one with probability 1 - (1 - 1e-5)^150
SURVEY_INDEL_READ_FRAC = 0.0015


def _excl_cumsum(x):
    c = torch.cumsum(x, 0)
    return c - x


def make_job(device, contig_lens=(5_000_000,), coverage=200, read_len=150, seed=42, sub_rate=0.002, n_rate=1e-4,
             seq_pitch=None, seq_layout="window",
             asm_err_rate=1e-4, recipe="survey", indel_read_frac=None, repeat=None, repeat_bp=0, repeat_k=5, pairs=False,
             unaligned_frac=0.0, G=None, end_margin=1000, asm_sub_rate=None):
    """One synthetic polish job (see the module docstring).  contig_lens are TRUTH lengths; the assembly's differ by
    the planted indels.  Returns a dict: G / contig_off / bases (the assembly), recs (good records, SoA on `device`),
    truth + truth_off, n_aln, n_runs, planted (counts of planted errors), sam (all records' text columns + `half`,
    the number of records of mate 1), repeat_loci (assembly coordinates), gstart (global assembly start per record)."""
    if G is not None:
        contig_lens = (G,)
    if asm_sub_rate is not None:  # the old keyword
        asm_err_rate = asm_sub_rate
    if indel_read_frac is None:
        indel_read_frac = SURVEY_INDEL_READ_FRAC if recipe == "survey" else 0.01
    dev = device
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    L = read_len
    i64 = torch.int64
    tlens = torch.tensor(list(contig_lens), dtype=i64, device=dev)
    nc = len(contig_lens)
    toff = torch.zeros(nc + 1, dtype=i64, device=dev)
    toff[1:] = torch.cumsum(tlens, 0)
    Gt = int(toff[-1].item())
    truth = torch.randint(0, 4, (Gt,), dtype=torch.uint8, device=dev, generator=g)

    # ---- repeat copies (configs[2]): direct and inverted, every other later copy with one SNP ----
    loci = inverted = None
    if repeat:
        assert nc == 1, "repeats are planted in a single-contig job"
        seg, copies = repeat
        loci = [(j + 1) * (Gt // (copies + 1)) for j in range(copies)]
        inverted = [j % 2 == 1 for j in range(copies)]
        base = truth[loci[0]:loci[0] + seg].clone()
        for j in range(1, copies):
            cp = base.clone()
            if j >= 2 and j % 4 >= 2:  # copies 2, 3, 6, 7 ... carry one SNP
                p = int(torch.randint(0, seg, (1,), device=dev, generator=g).item())
                cp[p] = (cp[p] + 1) % 4
            if inverted[j]:
                cp = (3 - cp).flip(0)  # reverse complement in code space (A0 C1 G2 T3)
            truth[loci[j]:loci[j] + seg] = cp

    # ---- assembly errors: substitutions, 1-bp deletions, 1-bp insertions; half of the indels in homopolymers ----
    n_err = int(round(asm_err_rate * Gt))
    if recipe == "survey":
        n_sub = n_err // 3
        n_del = (n_err - n_sub) // 2
        n_ins = n_err - n_sub - n_del
    elif recipe == "subs":
        n_sub, n_del, n_ins = n_err, 0, 0
    else:
        raise ValueError(f"unknown recipe {recipe!r}")
    elig = torch.ones(Gt, dtype=torch.bool, device=dev)
    for c in range(nc):
        a, b = int(toff[c].item()), int(toff[c + 1].item())
        m = min(end_margin, (b - a) // 8)
        elig[a:a + m] = False
        elig[b - m:b] = False
    if repeat:
        for lc in loci:
            elig[max(0, lc - 2 * L - 8):lc + seg + 2 * L + 8] = False
    pos_l, kind_l, hp_l = [], [], []
    def uniform_sites(k):
        if k <= 0:
            return torch.empty(0, dtype=i64, device=dev)
        p = torch.randint(0, Gt, (k,), device=dev, generator=g)
        return p[elig[p]]
    hp_sites = None
    if n_del + n_ins > 0:
        mid = torch.zeros(Gt, dtype=torch.bool, device=dev)
        mid[1:-1] = (truth[:-2] == truth[1:-1]) & (truth[1:-1] == truth[2:])
        hp_sites = torch.nonzero(mid & elig)[:, 0]
        del mid
    def hp_pick(k):
        if k <= 0 or hp_sites is None or len(hp_sites) == 0:
            return torch.empty(0, dtype=i64, device=dev)
        return hp_sites[torch.randint(0, len(hp_sites), (k,), device=dev, generator=g)]
    for kind, sites, hp in ((0, uniform_sites(n_sub), 0), (1, uniform_sites(n_del - n_del // 2), 0), (1, hp_pick(n_del // 2), 1),
                            (2, uniform_sites(n_ins - n_ins // 2), 0), (2, hp_pick(n_ins // 2), 1)):
        pos_l.append(sites)
        kind_l.append(torch.full((len(sites),), kind, dtype=i64, device=dev))
        hp_l.append(torch.full((len(sites),), hp, dtype=i64, device=dev))
    epos, ekind, ehp = torch.cat(pos_l), torch.cat(kind_l), torch.cat(hp_l)
    order = torch.argsort(epos, stable=True)
    epos, ekind, ehp = epos[order], ekind[order], ehp[order]
    if len(epos) > 1:  # planted errors stay at least 12 bp apart (one alignment event per site)
        far = torch.ones(len(epos), dtype=torch.bool, device=dev)
        far[1:] = (epos[1:] - epos[:-1]) >= 12
        # a dropped site may have been the near neighbour of the next one: one more pass is enough in practice
        epos, ekind, ehp = epos[far], ekind[far], ehp[far]
        far = torch.ones(len(epos), dtype=torch.bool, device=dev)
        far[1:] = (epos[1:] - epos[:-1]) >= 12
        epos, ekind, ehp = epos[far], ekind[far], ehp[far]
    keep = torch.ones(Gt, dtype=torch.uint8, device=dev)
    extra = torch.zeros(Gt, dtype=torch.uint8, device=dev)
    asm_t = truth.clone()
    sp = epos[ekind == 0]
    asm_t[sp] = (truth[sp] + torch.randint(1, 4, (len(sp),), dtype=torch.uint8, device=dev, generator=g)) % 4
    keep[epos[ekind == 1]] = 0
    ip, ihp = epos[ekind == 2], ehp[ekind == 2]
    extra[ip] = 1
    extra_base = torch.where(ihp == 1, truth[ip], torch.randint(0, 4, (len(ip),), dtype=torch.uint8, device=dev, generator=g))
    planted = {"substitutions": int(len(sp)), "deletions": int((ekind == 1).sum().item()), "insertions": int(len(ip)),
               "indels_in_homopolymers": int((ehp[ekind != 0] == 1).sum().item())}
    apos = torch.zeros(Gt + 1, dtype=i64, device=dev)
    torch.cumsum(keep.to(i64) + extra.to(i64), 0, out=apos[1:])
    Ga = int(apos[-1].item())
    asm = torch.empty(Ga, dtype=torch.uint8, device=dev)
    kidx = torch.nonzero(keep)[:, 0]
    asm[apos[kidx]] = asm_t[kidx]
    asm[apos[ip] + keep[ip].to(i64)] = extra_base
    del kidx, asm_t, elig, hp_sites
    aoff = apos[toff]
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    bases = lut[asm.long()]

    # ---- fragments / reads in truth coordinates ----
    n = Gt * coverage // L
    margin = (700 if pairs else L) + 2
    room = torch.clamp(tlens - margin, min=1)
    cum = torch.cumsum(room, 0)
    n_frag = n // 2 if pairs else n
    u = (torch.rand(n_frag, device=dev, generator=g, dtype=torch.float64) * float(cum[-1].item())).long()
    u = torch.clamp(u, max=int(cum[-1].item()) - 1)
    contig = torch.searchsorted(cum, u, right=True)
    rs_t = u - (cum - room)[contig]
    del u
    flag = pnext_t = tlen = read_id = None
    if pairs:
        ins = torch.clamp(torch.round(torch.randn(n_frag, device=dev, generator=g) * 35 + 350), 160, 700).long()
        fwd1 = torch.rand(n_frag, device=dev, generator=g) < 0.5   # mate 1 on the forward strand
        left, right = rs_t, rs_t + ins - L
        s1 = torch.where(fwd1, left, right)
        s2 = torch.where(fwd1, right, left)
        una = torch.rand(n_frag, device=dev, generator=g) < unaligned_frac
        f1 = torch.where(una, 77, torch.where(fwd1, 99, 83))
        f2 = torch.where(una, 141, torch.where(fwd1, 147, 163))
        t1 = torch.where(fwd1, ins, -ins)
        flag = torch.cat([f1, f2])
        pnext_t = torch.cat([s2, s1])  # truth-relative; turned into assembly coordinates below
        tlen = torch.cat([t1, -t1])
        read_id = torch.cat([torch.arange(n_frag, device=dev)] * 2)
        contig = torch.cat([contig, contig])
        rs_t = torch.cat([s1, s2])
        n = 2 * n_frag
        del ins, fwd1, left, right, s1, s2, una, f1, f2, t1
    else:
        flag = torch.zeros(n, dtype=i64, device=dev)
    half = n // 2 if pairs else n
    tstart = toff[contig] + rs_t  # global truth start
    kind = torch.zeros(n, dtype=i64, device=dev)
    sel = torch.rand(n, device=dev, generator=g) < indel_read_frac
    kind[sel] = torch.randint(1, 3, (int(sel.sum()),), device=dev, generator=g)
    del sel
    a = torch.randint(5, L - 5, (n,), device=dev, generator=g)

    seq = torch.empty(n * L, dtype=torch.uint8, device=dev)
    nm = torch.empty(n, dtype=i64, device=dev)
    gstart = torch.empty(n, dtype=i64, device=dev)
    ends_ok = torch.empty(n, dtype=torch.bool, device=dev)
    n_cig = torch.empty(n, dtype=i64, device=dev)
    cig_chunks = []
    j = torch.arange(L, device=dev)[None, :]
    CH = 1 << 20
    for lo in range(0, n, CH):
        hi = min(n, lo + CH)
        m = hi - lo
        k_, a_, s_ = kind[lo:hi, None], a[lo:hi, None], tstart[lo:hi, None]
        tc = s_ + j + torch.where(k_ == 1, -(j > a_).long(), torch.where(k_ == 2, (j >= a_).long(), 0))
        is_ins = (k_ == 1) & (j == a_)
        codes = truth[tc]
        rnd = torch.randint(0, 4, (m, 1), dtype=torch.uint8, device=dev, generator=g)
        codes = torch.where(is_ins, rnd, codes)
        sub = torch.rand(m, L, device=dev, generator=g) < sub_rate
        sh = torch.randint(1, 4, (m, L), dtype=torch.uint8, device=dev, generator=g)
        codes = torch.where(sub, (codes + sh) % 4, codes)
        s = lut[codes.long()]
        s[torch.rand(m, L, device=dev, generator=g) < n_rate] = ord("N")
        seq[lo * L:hi * L] = s.reshape(-1)
        del codes, sub, sh, rnd
        # alignment against the ASSEMBLY: column j sits on assembly index A[j] (M) or between bases (I)
        keep0 = keep[tc].long()
        A0 = apos[tc]
        End0 = A0 + keep0
        End_prev0 = torch.cat([End0[:, :1], End0[:, :-1]], 1)
        A = torch.where(is_ins, End_prev0, A0)
        keepc = torch.where(is_ins, 0, keep0)
        End = A + keepc
        gap = torch.zeros(m, L, dtype=i64, device=dev)   # assembly bases skipped in front of column j: a D run
        gap[:, 1:] = A[:, 1:] - End[:, :-1]
        opI = keepc == 0
        del keep0, A0, End0, End_prev0, End, tc, is_ins
        mism = (~opI) & (s != bases[torch.clamp(A, max=Ga - 1)])
        nm[lo:hi] = mism.sum(1) + opI.sum(1) + gap.sum(1)
        gstart[lo:hi] = A[:, 0]
        ends_ok[lo:hi] = (~opI[:, 0]) & (~opI[:, -1])
        del mism, s
        brk = torch.zeros(m, L, dtype=torch.bool, device=dev)
        brk[:, 1:] = (opI[:, 1:] != opI[:, :-1]) | (gap[:, 1:] > 0)
        nb = brk.sum(1)
        ncg = 1 + nb + (gap > 0).sum(1)
        n_cig[lo:hi] = ncg
        coff = _excl_cumsum(ncg)
        cg = torch.zeros(int(ncg.sum().item()), dtype=torch.int32, device=dev)
        simple = nb == 0
        cg[coff[simple]] = (L << 4) | OP_M
        cr = torch.nonzero(~simple)[:, 0]
        if len(cr):
            bi = torch.nonzero(brk[cr])          # (row within cr, column), row-major
            r_, j_ = bi[:, 0], bi[:, 1]
            rr = cr[r_]                          # row within the chunk
            d_ = gap[rr, j_]
            cnt = 1 + (d_ > 0).long()
            nxt_same = torch.zeros(len(r_), dtype=torch.bool, device=dev)
            nxt_same[:-1] = r_[1:] == r_[:-1]
            nj = torch.full((len(r_),), L, dtype=i64, device=dev)
            nj[:-1] = torch.where(nxt_same[:-1], j_[1:], nj[:-1])
            first_e = torch.ones(len(r_), dtype=torch.bool, device=dev)
            first_e[1:] = r_[1:] != r_[:-1]
            cs = _excl_cumsum(cnt)
            first_idx = torch.nonzero(first_e)[:, 0]     # index of each complex read's first breakpoint
            within = cs - cs[first_idx][r_]
            base_o = coff[rr] + 1 + within
            # the initial run of every complex read
            cg[coff[cr]] = ((j_[first_idx] << 4) | torch.where(opI[cr, 0], OP_I, OP_M)).int()
            hasd = d_ > 0
            cg[base_o[hasd]] = ((d_[hasd] << 4) | OP_D).int()
            cg[base_o + hasd.long()] = (((nj - j_) << 4) | torch.where(opI[rr, j_], OP_I, OP_M)).int()
        cig_chunks.append(cg)
        del brk, gap, opI, A, keepc
    cigar = torch.cat(cig_chunks) if cig_chunks else torch.zeros(0, dtype=torch.int32, device=dev)
    del cig_chunks
    cig_off = _excl_cumsum(n_cig)
    aligned = (flag & 4) == 0
    good = aligned & ends_ok & (nm <= MAX_ERRORS)
    rs_a = gstart - aoff[contig]  # assembly-relative start
    k = torch.where((gstart >= Ga // 5) & (gstart < Ga // 5 + repeat_bp), repeat_k, 1)
    seq_len_text = torch.full((n,), L, dtype=i64, device=dev)
    pnext = None
    if pairs:  # the mate's start in assembly coordinates (cosmetic columns of the SAM line)
        mate_g = torch.cat([gstart[half:], gstart[:half]])
        pnext = mate_g - aoff[contig]
        del mate_g, pnext_t

    loci_asm = None
    if repeat:
        # all-hits expansion: a read inside copy c becomes `copies` adjacent records (its own locus first)
        seg, copies = repeat
        loc = torch.tensor(loci, dtype=i64, device=dev)
        inv = torch.tensor(inverted, dtype=torch.bool, device=dev)
        span_t = L + (kind == 2).long() - (kind == 1).long()
        inside = (tstart[:, None] >= loc[None, :]) & (tstart[:, None] + span_t[:, None] <= loc[None, :] + seg) & aligned[:, None]
        own = torch.where(inside.any(1), inside.float().argmax(1), -1)
        cnt = torch.where(own >= 0, copies, 1)
        src = torch.repeat_interleave(torch.arange(n, device=dev), cnt)
        first = _excl_cumsum(cnt)
        within = torch.arange(len(src), device=dev) - first[src]
        own_s = own[src]
        in_copy = own_s >= 0
        cp = torch.where(in_copy, (own_s + within) % copies, 0)
        flipped = in_copy & (inv[cp] != inv[torch.clamp(own_s, min=0)])
        o = tstart[src] - loc[torch.clamp(own_s, min=0)]
        new_t = torch.where(in_copy, torch.where(flipped, loc[cp] + seg - o - span_t[src], loc[cp] + o), tstart[src])
        half = int(cnt[:half].sum().item())
        n_old, n = n, len(src)
        gstart = apos[new_t]   # no assembly error inside or next to a copy: exact for every copy
        contig = contig[src]
        rs_a = gstart - aoff[contig]
        secondary = in_copy & (within > 0)
        k = torch.where(in_copy, copies, 1)
        # SEQ (as the host ingest fills it): the read's bytes, reverse-complemented on a flipped copy
        comp = torch.arange(256, dtype=torch.uint8, device=dev)
        for x, y in ((b"A", b"T"), (b"C", b"G")):
            comp[x[0]], comp[y[0]] = y[0], x[0]
        seq2 = seq.view(n_old, L)[src]
        fl_idx = torch.nonzero(flipped)[:, 0]
        seq2[fl_idx] = comp[seq2[fl_idx].flip(1).long()]
        seq = seq2.reshape(-1).contiguous()
        del seq2
        # CIGAR: the read's runs, reversed on a flipped copy
        ncg_s = n_cig[src]
        new_coff = _excl_cumsum(ncg_s)
        rec_of_run = torch.repeat_interleave(torch.arange(n, device=dev), ncg_s)
        run_i = torch.arange(len(rec_of_run), device=dev) - new_coff[rec_of_run]
        src_run = torch.where(flipped[rec_of_run], ncg_s[rec_of_run] - 1 - run_i, run_i)
        cigar = cigar[cig_off[src][rec_of_run] + src_run]
        n_cig, cig_off = ncg_s, new_coff
        kind_s, a_s = kind[src], a[src]
        a_s = torch.where(flipped, torch.where(kind_s == 1, L - 1 - a_s, L - a_s), a_s)
        # NM of the records inside copies, against their own copy (copies differ by SNPs)
        nm = nm[src]
        ic = torch.nonzero(in_copy)[:, 0]
        for lo in range(0, len(ic), CH):
            ix = ic[lo:lo + CH]
            k_, a_ = kind_s[ix, None], a_s[ix, None]
            off = j + torch.where(k_ == 1, -(j > a_).long(), torch.where(k_ == 2, (j >= a_).long(), 0))
            s = seq.view(n, L)[ix]
            differ = (s != bases[gstart[ix, None] + off]) & ~((k_ == 1) & (j == a_))
            nm[ix] = differ.sum(1) + (kind_s[ix] != 0).long()
        good = good[src] & (nm <= MAX_ERRORS)
        # a group is all good or all bad up to NM; k = number of good records of the group (src/alignment.rs:288)
        if bool((in_copy & ~good).any()):
            gcount = torch.zeros(n_old, dtype=i64, device=dev).index_add_(0, src, good.long())
            k = torch.where(in_copy, gcount[src], k)
        aligned = aligned[src]
        pflag = flag[src]
        rev = ((pflag & 16) != 0) != flipped
        flag = torch.where(secondary, 256 + 16 * rev.long(), pflag)
        seq_len_text = torch.where(secondary, 0, L)
        if pairs:
            pnext, tlen, read_id = pnext[src], torch.where(secondary, 0, tlen[src]), read_id[src]
        loci_asm = [int(apos[lc].item()) for lc in loci]
        del inside, own, cnt, first, within, own_s, cp, new_t, o, flipped, in_copy, secondary, rec_of_run, run_i, src_run

    # ---- the good records, compacted: what the ingest hands to seam B ----
    gi = torch.nonzero(good)[:, 0]
    ng = len(gi)
    if ng == n:
        r_seq, r_cigar, r_ncig, r_coff = seq, cigar, n_cig, cig_off
    else:
        r_seq = seq.view(n, L)[gi].reshape(-1).contiguous()
        r_ncig = n_cig[gi]
        r_coff = _excl_cumsum(r_ncig)
        rr = torch.repeat_interleave(torch.arange(ng, device=dev), r_ncig)
        r_cigar = cigar[cig_off[gi][rr] + (torch.arange(len(rr), device=dev) - r_coff[rr])]
    recs = {
        "contig": contig[gi].int().contiguous(),
        "ref_start": rs_a[gi].int().contiguous(),
        "k": k[gi].int().contiguous(),
        "seq_off": torch.arange(ng, device=dev, dtype=i64) * L,
        "seq_len": torch.full((ng,), L, dtype=torch.int32, device=dev),
        "cig_off": r_coff.contiguous(),
        "n_cig": r_ncig.int().contiguous(),
        "seq": r_seq,
        "cigar": r_cigar.int().contiguous(),
    }
    sam = None
    if pairs:
        sam = {"flag": flag.int(), "pnext": pnext.int(), "tlen": tlen.int(), "read": read_id.int(), "nm": nm.int(),
               "contig": contig.int(), "ref_start": rs_a.int(), "cig_off": cig_off, "n_cig": n_cig.int(), "cigar": cigar.int(),
               "seq_off": torch.arange(n, device=dev, dtype=i64) * L, "seq_len": seq_len_text.int(), "seq": seq,
               "half": half, "n": n}
    job = {"G": Ga, "contig_off": aoff.cpu().numpy().astype(np.uint64), "bases": bases, "recs": recs,
           "truth": lut[truth.long()], "truth_off": toff.cpu().numpy().astype(np.uint64), "read_len": L,
           "n_runs": int(r_ncig.sum().item()), "n_aln": ng, "n_records": n, "sam": sam, "planted": planted, "recipe": recipe,
           "repeat_loci": loci_asm, "repeat_loci_truth": loci, "repeat_seg": repeat[0] if repeat else 0, "gstart": gstart[gi],
           "file_used": None}
    # good records of SAM file 1 / 2 (without the text form, pairs=False: the two halves of the records, as two files would hold them)
    n1 = int((gi < (half if pairs else n // 2)).sum().item())
    job["file_used"] = [n1, ng - n1]
    # the resident records as the product's ingests lay them out (include/polypolish_hip.h): every record's SEQ on a
    # PP_SEQ_ALIGN (32-byte) boundary of the seq array, zeros in between, and -- the default since round 4 -- the SEQ bytes of
    # a SAM file WINDOW-GROUPED (PP_SEQ_WINDOW_GROUPED; file order inside a window, as the host ingest writes them:
    # tests/test_synthjob_cpu.py compares the arrays).  seq_layout="file": in the order of the records (rounds 1-3);
    # seq_pitch=0: packed back to back as well (the layout of rounds 1-3's first bench lines)
    pitch = (L + 31) // 32 * 32 if seq_pitch is None else (seq_pitch or L)
    if pitch != L:
        job = with_pitch(job, pitch)
    return window_grouped(job) if seq_layout == "window" and seq_pitch is None else job


def seq4_of(seq, chunk=1 << 27):
    """The 4-bit mirror of a seq array (pp_aln_batch.seq4), as the device tokenizer produces it next to the bytes: base i
    of the array in bits 4*(i&1).. of byte i >> 1, codes PP_SEQ4_* (A C T G = 0..3, N 4, '-' 5, anything else 15); 64
    bytes of slack behind."""
    dev = seq.device
    lut = torch.full((256,), 15, dtype=torch.uint8, device=dev)
    for ch, v in ((65, 0), (67, 1), (84, 2), (71, 3), (78, 4), (45, 5)):
        lut[ch] = v
    n = seq.numel()
    out = torch.zeros((n + 1) // 2 + 64, dtype=torch.uint8, device=dev)
    for lo in range(0, n, chunk):                       # (chunk is even: every piece starts on a byte of the mirror)
        c = lut[seq[lo:lo + chunk].long()]
        if c.numel() & 1:
            c = torch.cat([c, torch.zeros(1, dtype=torch.uint8, device=dev)])
        out[lo // 2:lo // 2 + c.numel() // 2] = c[0::2] | (c[1::2] << 4)
    return out


def with_seq4(job, on=True):
    """The job with (or without) the 4-bit mirror of its seq array: job["seq4"], handed to the library as pp_aln_batch.seq4."""
    out = dict(job)
    out.pop("_prepared", None)
    out["seq4"] = seq4_of(job["recs"]["seq"]) if on else None
    return out


def wo_of(job, window=2048):
    """The window-order mirror of the job's records (pp_aln_batch.wo: 32 bytes per record -- contig, ref_start, k, seq_len,
    seq_off (8), op0, file_idx), as the host ingest writes it: per SAM file, the records that start in one 2048-position
    window adjacent, file order inside a window.  An int32 tensor [n, 8] on the records' device."""
    r = job["recs"]
    n = job["n_aln"]
    dev = r["seq"].device
    n_win = max(1, (int(job["G"]) + window - 1) // window)
    win = torch.clamp(job["gstart"] // window, max=n_win - 1)
    n1 = job.get("file_used", [n, 0])[0]
    file_of = (torch.arange(n, device=dev) >= n1).long()
    order = torch.argsort(file_of * n_win + win, stable=True)
    wo = torch.empty((n, 8), dtype=torch.int32, device=dev)
    wo[:, 0] = r["contig"][order]
    wo[:, 1] = r["ref_start"][order]
    wo[:, 2] = r["k"][order]
    wo[:, 3] = r["seq_len"][order]
    so = r["seq_off"][order]
    wo[:, 4] = (so & 0xFFFFFFFF).to(torch.int64).where((so & 0xFFFFFFFF) < 2 ** 31, (so & 0xFFFFFFFF) - 2 ** 32).int()
    wo[:, 5] = (so >> 32).int()
    first = r["cigar"][r["cig_off"][order]]
    wo[:, 6] = torch.where(r["n_cig"][order] == 1, first, torch.full_like(first, -1))
    wo[:, 7] = order.int()
    return wo.contiguous()


def with_wo(job, on=True):
    """The job with (or without) the window-order mirror of its records: job["wo"], handed to the library as pp_aln_batch.wo."""
    out = dict(job)
    out.pop("_prepared", None)
    out["wo"] = wo_of(job) if on else None
    # ... and its runs (pp_aln_batch.wo_run_end): one per SAM file, in ascending window order -- what both ingests hand over
    n1 = job.get("file_used", [job["n_aln"], 0])[0]
    out["wo_runs"] = [e for e in (n1, job["n_aln"]) if e > 0] if on else None
    if out["wo_runs"] and len(out["wo_runs"]) == 2 and out["wo_runs"][0] == out["wo_runs"][1]:
        out["wo_runs"] = out["wo_runs"][:1]
    return out


def with_pitch(job, pitch):
    """The same job with every read's SEQ bytes starting on a multiple of `pitch` bytes (an experiment of DESIGN.md section
    9: 150-byte reads at pitch 160 start on 32-byte boundaries and touch 2.0 instead of 2.16 lines of 128 bytes)."""
    r = job["recs"]
    L, n = job["read_len"], job["n_aln"]
    assert pitch >= L
    dev = r["seq"].device
    out = dict(job)
    out.pop("_prepared", None)
    recs = dict(r)
    padded = torch.zeros((n, pitch), dtype=torch.uint8, device=dev)
    P0 = job.get("pitch", L)
    padded[:, :L] = r["seq"].view(n, P0)[r["seq_off"] // P0][:, :L]   # (any layout of constant pitch on the way in; file order out)
    recs["seq"] = padded.reshape(-1).contiguous()
    recs["seq_off"] = (torch.arange(n, device=dev, dtype=torch.int64) * pitch).contiguous()
    out["recs"] = recs
    out["pitch"] = pitch
    out["seq4"] = None
    return out


def window_grouped(job, window=2048):
    """The same job with the SEQ bytes laid out WINDOW-GROUPED, as the product's ingests write them by default
    (PP_SEQ_WINDOW_GROUPED): per SAM file, the reads that start in one 2048-position window are adjacent in the file's
    stretch of the seq array (windows in order, file order inside a window -- the host ingest's order; the device tokenizer
    leaves the order inside a window to its atomics); every other array -- and the order of the records -- is unchanged,
    seq_off simply points there.  The C ABI allows any seq_off, so this is not a batch format but the producer's choice."""
    r = job["recs"]
    L = job["read_len"]
    n = job["n_aln"]
    dev = r["seq"].device
    n_win = max(1, (int(job["G"]) + window - 1) // window)
    win = torch.clamp(job["gstart"] // window, max=n_win - 1)
    n1 = job.get("file_used", [n, 0])[0]
    file_of = (torch.arange(n, device=dev) >= n1).long()
    order = torch.argsort(file_of * n_win + win, stable=True)   # records by (file, window), file order inside
    slot = torch.empty(n, dtype=torch.int64, device=dev)
    slot[order] = torch.arange(n, device=dev)
    out = dict(job)
    out.pop("_prepared", None)
    recs = dict(r)
    P = job.get("pitch", L)                            # bytes of the seq array per record (its SEQ up to the next boundary)
    row = r["seq_off"] // P                            # (any layout of constant pitch on the way in)
    recs["seq"] = r["seq"].view(n, P)[row[order]].reshape(-1).contiguous()
    recs["seq_off"] = (slot * P).contiguous()
    out["recs"] = recs
    out["seq_layout"] = "window"
    if job.get("seq4") is not None:
        out["seq4"] = seq4_of(recs["seq"])
    return out


def file_ordered(job):
    """The same job with the SEQ bytes in the order of the records (PP_SEQ_FILE_ORDER: the layout of rounds 1-3)."""
    r = job["recs"]
    n, P = job["n_aln"], job.get("pitch", job["read_len"])
    out = dict(job)
    out.pop("_prepared", None)
    recs = dict(r)
    recs["seq"] = r["seq"].view(n, P)[r["seq_off"] // P].reshape(-1).contiguous()
    recs["seq_off"] = (torch.arange(n, device=r["seq"].device, dtype=torch.int64) * P).contiguous()
    out["recs"] = recs
    out["seq_layout"] = "file"
    if job.get("seq4") is not None:
        out["seq4"] = seq4_of(recs["seq"])
    return out


def recovered(job, polished, offs, margin=1000, count=False):
    """Did the polish turn the assembly back into the truth?  Contig by contig (their lengths change where indels were
    repaired), ignoring `margin` bp at either end, where coverage runs out and nothing was planted, and the repeat
    copies of a configs[2] job (+- 200 bp): copies that differ by a SNP are out-voted there by their siblings' reads.
    count=True: the number of differing positions instead (-1 when a contig's length is off)."""
    truth = job["truth"].cpu().numpy()
    toff = job["truth_off"].astype(np.int64)
    got = np.frombuffer(polished, dtype=np.uint8)
    offs = np.asarray(offs).astype(np.int64)
    bad = 0
    for c in range(len(toff) - 1):
        t = truth[toff[c]:toff[c + 1]]
        p = got[offs[c]:offs[c + 1]]
        if len(t) != len(p):
            return -1 if count else False
        m = min(margin, len(t) // 8)
        ok = t == p
        ok[:m] = True
        ok[len(t) - m:] = True
        if c == 0 and job.get("repeat_loci_truth"):
            for l in job["repeat_loci_truth"]:
                ok[max(0, l - 200):l + job["repeat_seg"] + 200] = True
        bad += int((~ok).sum())
    return bad if count else bad == 0


def subset_job(job, lo, hi, contig=0):
    """Records lying entirely inside [lo, hi) of contig `contig`, re-based to a single contig of hi-lo bp."""
    r = job["recs"]
    L = job["read_len"]
    rs = r["ref_start"].long()
    keep = (r["contig"] == contig) & (rs >= lo) & (rs + L + 4 <= hi)
    idx = torch.nonzero(keep)[:, 0]
    n = len(idx)
    j = torch.arange(L, device=idx.device)[None, :]
    seq = r["seq"][(r["seq_off"][idx][:, None] + j).reshape(-1)]
    n_cig = r["n_cig"][idx]
    cig_off = torch.cumsum(n_cig.long(), 0) - n_cig.long()
    pos = torch.repeat_interleave(torch.arange(n, device=idx.device), n_cig.long())
    within = torch.arange(len(pos), device=idx.device) - cig_off[pos]
    cigar = r["cigar"][r["cig_off"][idx][pos] + within]
    recs = {
        "contig": torch.zeros(n, dtype=torch.int32, device=idx.device),
        "ref_start": (rs[idx] - lo).int(),
        "k": r["k"][idx].contiguous(),
        "seq_off": torch.arange(n, device=idx.device, dtype=torch.int64) * L,
        "seq_len": r["seq_len"][idx].contiguous(),
        "cig_off": cig_off,
        "n_cig": n_cig.contiguous(),
        "seq": seq.contiguous(),
        "cigar": cigar.contiguous(),
    }
    g0 = int(job["contig_off"][contig])
    return {"G": hi - lo, "contig_off": np.array([0, hi - lo], dtype=np.uint64),
            "bases": job["bases"][g0 + lo:g0 + hi].contiguous(), "recs": recs, "read_len": L,
            "n_runs": int(n_cig.sum()), "n_aln": n}


def algorithmic_bytes(job):
    """per good alignment seq_len + 16 B record + 4 B per CIGAR run;
    per assembly position 1 B read + 1 B written."""
    return job["n_aln"] * (job["read_len"] + 16) + 4 * job["n_runs"] + 2 * job["G"]


_DT = {"contig": np.uint32, "ref_start": np.uint32, "k": np.uint32, "seq_off": np.uint64, "seq_len": np.uint32,
       "cig_off": np.uint64, "n_cig": np.uint32, "seq": np.uint8, "cigar": np.uint32}


def to_host_records(job):
    return {k: v.cpu().numpy().astype(_DT[k], copy=False) if v.dtype != torch.uint8 else v.cpu().numpy()
            for k, v in job["recs"].items()}


# ---- SAM / FASTA text of a pairs=True job (tools/samgen.c) ---------------------------------------------------------
def samgen_lib():
    path = os.path.join(ROOT, "tools", "_build", "libsamgen.so")
    if not os.path.exists(path):
        return None
    lib = ctypes.CDLL(path)
    lib.samgen_write_sam.argtypes = [ctypes.c_char_p, ctypes.c_uint32, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_void_p]
    lib.samgen_write_fasta.argtypes = [ctypes.c_char_p, ctypes.c_uint32, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_void_p]
    return lib


class _SamRecords(ctypes.Structure):
    _fields_ = [("n", ctypes.c_uint64)] + [(k, ctypes.c_void_p) for k in (
        "read", "flag", "contig", "ref_start", "cig_off", "n_cig", "cigar", "pnext", "tlen", "seq_off", "seq_len", "seq",
        "nm")] + [("qual", ctypes.c_int)]


def write_sam_pair(job, outdir, qual=True):
    """FASTA + the two SAM files (mate 1 / mate 2, same read order) of a make_job(pairs=True) job: EVERY record (the
    ones the gates will reject and the unaligned ones too), secondary records with SEQ / QUAL "*"."""
    lib = samgen_lib()
    if lib is None:
        raise RuntimeError("tools/_build/libsamgen.so is missing (make)")
    S = job["sam"]
    n, half = S["n"], S["half"]
    off = job["contig_off"]
    names = b"".join(b"contig_%d\0" % (i + 1) for i in range(len(off) - 1))
    lens = np.ascontiguousarray(np.diff(off.astype(np.int64)).astype(np.uint64))
    bases = job["bases"].cpu().numpy()
    fa = os.path.join(outdir, "asm.fasta")
    if lib.samgen_write_fasta(fa.encode(), len(off) - 1, names, off.ctypes.data, bases.ctypes.data):
        raise RuntimeError("writing the FASTA failed")
    del bases
    whole = {"cigar": S["cigar"].cpu().numpy().astype(np.uint32, copy=False), "seq": S["seq"].cpu().numpy()}
    dt = {"read": np.uint32, "flag": np.uint32, "contig": np.uint32, "ref_start": np.uint32, "cig_off": np.uint64,
          "n_cig": np.uint32, "pnext": np.uint32, "tlen": np.int32, "seq_off": np.uint64, "seq_len": np.uint32, "nm": np.uint32}
    paths = []
    for f, (lo, hi) in enumerate(((0, half), (half, n))):
        cols = {k: np.ascontiguousarray(S[k][lo:hi].cpu().numpy().astype(t, copy=False)) for k, t in dt.items()}
        cols.update(whole)
        rec = _SamRecords(hi - lo, *[cols[k].ctypes.data for k in ("read", "flag", "contig", "ref_start", "cig_off", "n_cig",
                                                                    "cigar", "pnext", "tlen", "seq_off", "seq_len", "seq", "nm")],
                          int(qual))
        p = os.path.join(outdir, f"reads_{f + 1}.sam")
        if lib.samgen_write_sam(p.encode(), len(off) - 1, names, lens.ctypes.data, ctypes.byref(rec)):
            raise RuntimeError("writing the SAM failed")
        paths.append(p)
    return fa, paths
