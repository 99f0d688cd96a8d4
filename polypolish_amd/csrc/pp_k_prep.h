// pp_k_prep.h -- k_prep: one lane per alignment record -- run validation, spans, trim of the slow classes.
// Part of pp_kernels.hip (included there, in this order, and nowhere else: it defines __global__ kernels).
#pragma once

namespace pp {

// Profiling build only (-DPP_PREP_STAMPS; tools/exp_prep_stamps.py): thread 0 of every block of k_prep (kernel 0) and
// k_fill (kernel 1) leaves 100 MHz ticks at a few points, 8 words per block and kernel.
#ifdef PP_PREP_STAMPS
__device__ u64 *g_prep_stamps;
#define PP_STAMP(kern, slot)                                                                                  \
    do {                                                                                                      \
        if (threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.x < 16384u) g_prep_stamps[((u64)(kern) * 16384u + blockIdx.x) * 8 + (slot)] = wall_clock64(); \
    } while (0)
#else
#define PP_STAMP(kern, slot) do { } while (0)
#endif

// =============================================================================================
// k_prep
// =============================================================================================
// every record that is not a single short M run inside its contig
// g0: where the record's contig starts in the run's coordinates (k_prep's g_base); clen: the contig's length
// RUN run(i): the record's i-th CIGAR run; BYTE sb(i): byte i of its SEQ -- straight from memory (k_prep), or out of registers
// that were loaded together (k_prepd: the first four runs, the read's last eight bytes: two round trips instead of five)
template <typename RUN, typename BYTE>
__device__ __forceinline__ void prep_general_t(u64 a, u32 rs, u32 sl, u32 nc, RUN run, BYTE sb, u64 g0, u64 clen,
                                               u32 *g_out, u32 *nk_out, u8 *fl_out, u64 *status) {
    // walk the runs (alignment.rs:178-194): spans and validity
    u64 ref_span = 0, read_span = 0;
    bool indel = false;
    for (u32 r = 0; r < nc; r++) {
        u32 op = run(r), len = op >> 4, o = op & 15u;
        if (len == 0 || o > 8u) { report(status, a, DE_BAD_RUN); return; }
        if (o == PP_OP_M || o == PP_OP_EQ || o == PP_OP_X) { ref_span += len; read_span += len; }
        else if (o == PP_OP_I) { read_span += len; indel = true; }
        else if (o == PP_OP_D) { ref_span += len; indel = true; }
        else { report(status, a, DE_UNEXPECTED_OP); return; }
    }
    u32 o_first = run(0) & 15u, o_last = run(nc - 1) & 15u;
    if (!((o_first == PP_OP_M || o_first == PP_OP_EQ) && (o_last == PP_OP_M || o_last == PP_OP_EQ))) {
        report(status, a, DE_BAD_ENDS);
        return;
    }
    if (read_span != (u64)sl) { report(status, a, DE_LEN_MISMATCH); return; }
    if (ref_span >= 0x3FFFFFFFull) { report(status, a, DE_OVERFLOW); return; }

    // simple_trim_start over the bytes [from, from + len) of the read: index (relative to `from`) of the first base of the
    // trailing homopolymer
    auto trim_start = [&](u32 from, u32 len) -> u32 {
        const u8 last = sb(from + len - 1u);
        u32 i = len - 1u;
        while (i > 0 && sb(from + i - 1u) == last) i--;
        return i;
    };
    u32 n_entries = (u32)ref_span;
    if (indel && nc == 3u && sl <= FAST_MAX_LEN && (u64)rs + ref_span <= clen) {
        // ONE 1-base indel between two M/= runs (a read over a planted assembly indel, or a sequencing indel): no walk.
        // k_fill cuts it into the flank in front, the entry at the indel and the flank behind (pp_internal.h, ENT_NOTRIM /
        // ENT_POINT); the trim has to stay inside the flank behind, which is what the reads of k_tile's plain class assume.
        const u32 c0 = run(0), c1 = run(1), c2 = run(2);
        const u32 o0 = c0 & 15u, o1 = c1 & 15u, o2 = c2 & 15u, a = c0 >> 4, b = c2 >> 4;
        if ((o0 == PP_OP_M || o0 == PP_OP_EQ) && (o2 == PP_OP_M || o2 == PP_OP_EQ) && (o1 == PP_OP_I || o1 == PP_OP_D) &&
            (c1 >> 4) == 1u) {
            const bool del = o1 == PP_OP_D;
            const u32 l1 = del ? a : a - 1u;
            if (l1 >= INDEL1_MIN_SEG && b >= INDEL1_MIN_SEG && trim_start(del ? a : a + 1u, b) >= 1u) {
                *g_out = (u32)(g0 + rs);
                *nk_out = n_entries | (a << 9) | ((del ? 1u : 0u) << 17);
                *fl_out = (u8)NKW_INDEL1;
                return;
            }
        }
    }
    if (!indel && sl <= FAST_MAX_LEN && (u64)rs + ref_span <= clen) {
        // fast class (=/X runs): k_tile loads the whole read anyway and trims it there, so the read
        // bytes are not touched here; bucketed by its untrimmed span
        *g_out = (u32)(g0 + rs);
        *nk_out = n_entries;
        return;
    }
    // trim_bases_for_homopolymers (alignment.rs:364-378).  The last entry is the single base
    // seq[sl-1] (the last run is M/=).  `run` = number of trailing entries equal to it.
    u8 last = sb(sl - 1u);
    u32 trun = 0;
    if (!indel) {
        trun = sl - trim_start(0u, sl);
    } else {
        // walk the entries from the right end and stop at the first one that differs from the
        // last base (typically after 2-3 steps): runs in reverse; `pend` = bases inserted right
        // after the run being visited (they extend its last entry)
        u64 ro = sl;
        u32 pend = 0;
        bool stop = false;
        for (u32 r = nc; r-- > 0 && !stop;) {
            const u32 op = run(r), len = op >> 4, o = op & 15u;
            if (o == PP_OP_I) { ro -= len; pend += len; continue; }
            if (o == PP_OP_D) {
                // last slot of the run: empty, or rewritten to the inserted bases; the others are empty
                if (pend == 1 && sb((u32)ro) == last) { trun += 1; if (len > 1) stop = true; }
                else stop = true;
            } else {
                for (u32 t = 0; t < len; t++) {
                    const bool extended = (t == 0) && pend > 0;
                    if (!extended && sb((u32)(ro - 1 - t)) == last) trun += 1; else { stop = true; break; }
                }
                ro -= len;
            }
            pend = 0;
        }
    }
    u32 nk = (n_entries > trun) ? n_entries - trun - 1u : 0u;
    if (nk == 0) return;  // contributes nothing; the reference never indexes the pileup for it
    if ((u64)rs + nk > clen) { report(status, a, DE_OUT_OF_BOUNDS); return; }
    *g_out = (u32)(g0 + rs);
    *nk_out = nk;
    *fl_out = indel ? (u8)ENT_COMPLEX : (u8)ENT_PRETRIM;
}

// k_prep's: everything read where it is needed
__device__ __noinline__ void prep_general(u64 a, u32 rs, u32 sl, u64 so, const u32 *cg, u32 nc, const u8 *seq,
                                               u64 g0, u64 clen, u32 *g_out, u32 *nk_out, u8 *fl_out, u64 *status) {
    const u8 *s = seq + so;
    prep_general_t(a, rs, sl, nc, [&](u32 i) -> u32 { return cg[i]; }, [&](u32 i) -> u8 { return s[i]; }, g0, clen, g_out, nk_out, fl_out, status);
}

// PP_CHECK_WO=1 (opt-in): the window-order mirror against the arrays it mirrors -- every entry names a record of the batch,
// no record twice (n entries: a permutation then), and carries that record's fields.  The mirror is a hint the kernels
// TRUST (include/polypolish_hip.h): its producers are the library's own; a caller that builds one itself can have it
// checked here, at the price of one scattered read of the arrays (a few times k_prep's own time).
__global__ __launch_bounds__(256) void k_check_wo(u64 n, const pp_wo_rec *__restrict__ wo, const u32 *__restrict__ contig,
                                                  const u32 *__restrict__ ref_start, const u32 *__restrict__ kk,
                                                  const u64 *__restrict__ seq_off, const u32 *__restrict__ seq_len,
                                                  const u64 *__restrict__ cig_off, const u32 *__restrict__ n_cig,
                                                  const u32 *__restrict__ cigar, u32 *__restrict__ seen, u64 *status) {
    const u64 a = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= n) return;
    const pp_wo_rec r = wo[a];
    const u32 fi = r.file_idx;
    bool ok = fi < n;
    if (ok) {
        ok = (atomicOr(&seen[fi >> 5], 1u << (fi & 31u)) & (1u << (fi & 31u))) == 0;   // (a second entry for the same record)
        const u32 nc = n_cig[fi];
        ok = ok && r.contig == contig[fi] && r.ref_start == ref_start[fi] && r.k == kk[fi] && r.seq_off == seq_off[fi] &&
             r.seq_len == seq_len[fi] && r.op0 == (nc == 1u ? cigar[cig_off[fi]] : (u32)PP_WO_MULTI_RUN);
    }
    if (!ok) report(status, a, DE_BAD_MIRROR);
}

#ifndef PP_PLAIN_ALIGNED
#define PP_PLAIN_ALIGNED 0
#endif
constexpr u32 PLAIN_NARROW_MAX = PP_PLAIN_ALIGNED ? 129u : 160u;  // = PlainCfg<5>::MAXL below

// The bulk (one short M run inside its contig) touches 28 bytes of input per record and is done in the streaming loop;
// k and seq_off are only validated later, by k_fill, which reads them anyway.  Every other record (indels, long reads,
// contig overhang, malformed ones: ~1 % of a typical job) is only NOTED there, in an LDS list, and handled after the
// loop with one record per lane: inside the loop a single such record would hold its 63 neighbours for the several
// dependent memory round trips of the CIGAR walk and the trim -- with 1 % of them in random places that is every other
// wave (measured: half of the kernel's time).
// COUNT: the block also tallies its records' windows (single level) or coarse buckets of cw windows (two levels) in an
// LDS histogram and leaves its row of the blocks x columns matrix for k_scan_cols; k_count does that as a pass of its
// own only when the columns do not fit one LDS range.
#ifndef PP_PREP_WAVES
#define PP_PREP_WAVES 8
#endif
#ifndef PP_PREP_UNROLL
#define PP_PREP_UNROLL 1
#endif
constexpr u32 PREP_LATER_MAX = 2048;  // noted records per block; beyond that they are handled on the spot

// WO: the records are read through the batch's window-order mirror (pp_aln_batch.wo, 32 bytes per record, a coalesced
// stream like the SoA): gstart / nkeep are then indexed by the MIRROR position (k_fill walks the mirror as well), errors
// are reported by the record's file index, and -- a block's records now falling into a handful of windows -- the LDS
// histogram is fed one atomic per wave and window (ballot) for the window a record's first piece starts in: with every lane
// of a wave adding to the same counter the plain atomics serialise (measured on window-sorted records: k_prep 0.073 ->
// 0.173 ms).
template <bool COUNT, bool WO>
__global__ __launch_bounds__(1024, PP_PREP_WAVES) void k_prep(u64 n, u64 chunk, const pp_wo_rec *__restrict__ wo,
                                               const u32 *__restrict__ contig,
                                               const u32 *__restrict__ ref_start,
                                               const u64 *__restrict__ seq_off,
                                               const u32 *__restrict__ seq_len,
                                               const u64 *__restrict__ cig_off,
                                               const u32 *__restrict__ n_cig,
                                               const u32 *__restrict__ cigar,
                                               const u8 *__restrict__ seq,
                                               const u64 *__restrict__ contig_off, u32 n_contigs,
                                               const u64 *__restrict__ g_base,
                                               const u32 *__restrict__ slice,
                                               const u32 *__restrict__ own,
                                               u32 *__restrict__ gstart, u32 *__restrict__ nkeep,
                                               u32 *__restrict__ maxlen, u32 nwin, u32 cw, u32 ncols,
                                               u32 *__restrict__ hist, u64 *status) {
    __shared__ u32 h[COUNT ? COUNT_RANGE : 1];
    __shared__ u32 later[PREP_LATER_MAX], n_later;
    PP_STAMP(0, 0);
    if (threadIdx.x == 0) n_later = 0;
    if (COUNT)
        for (u32 i = threadIdx.x; i < (u32)COUNT_RANGE; i += blockDim.x) h[i] = 0;
    __syncthreads();
    PP_STAMP(0, 1);
    const u64 lo = (u64)blockIdx.x * chunk, hi = min(n, lo + chunk);
    // a record's result: stored, and (COUNT) tallied in the windows it reaches
    // contig_off / n_contigs / own are the JOB's (a record names its contig by the job's index); g_base[c] is where contig
    // c starts in the coordinates of this RUN -- the same table, or a compact one over the contigs this context owns
    // (run_pipeline), with ~0 for the others: their records are validated like any record, then dropped.
    auto finish = [&](u64 a, u64 idx, u32 c, u32 rs, u32 g_out, u32 nk_out, u32 fl_out, u32 n_counted) {  // a: where the result goes; idx: the record's file index; n_counted: leading histogram entries the wave has counted already (WO)
        // Sharded job (pp_polish_set_emit): a record that does not reach the range of its contig this context emits
        // is somebody else's -- validated like every record (all ranks report the same first bad record), then
        // dropped.  The untrimmed span of the fast class errs on the side of keeping.
        u32 word = nk_out | (fl_out << 30);  // kept entries (< 2^30) | class flags (a one-indel read: see for_each_piece)
        const u32 span = nkw_span(word);
        if (own && span && c < n_contigs && ((u64)rs + span <= own[2 * c] || rs >= own[2 * c + 1])) word = 0;
        if (c < n_contigs && g_base[c] == ~0ull) word = 0;
        // compact run: what is kept has to lie inside the stretch of its contig that the run holds (a read longer than
        // the halo does not: the host reruns the job over the whole assembly)
        if (slice && word && c < n_contigs && ((u64)rs < slice[2 * c] || (u64)rs + span > slice[2 * c + 1])) {
            report(status, idx, DE_HALO);
            word = 0;
        }
        gstart[a] = g_out;
        nkeep[a] = word;
        if (COUNT && word) {
            u32 skip = n_counted;  // (WO: the first windows of the first piece were counted by the wave, see below)
            for_each_piece(g_out, word, [&](u32, u32 g, u32 sp) {
                if (!sp) return;
                const u32 w0 = g / (u32)TILE, w1 = min((g + sp - 1u) / (u32)TILE, nwin - 1u);
                for (u32 w = w0; w <= w1; w++) {
                    if (skip) { skip--; continue; }
                    atomicAdd(&h[w / cw], 1u);  // per window, or per coarse bucket of cw windows
                }
            });
        }
    };
    // WO: one LDS atomic per wave and column for the column a record's first work item goes to (all lanes take part)
    auto count_first_by_wave = [&](bool have, u32 col) {
        u32 key = have ? col : 0xFFFFFFFFu;
        for (;;) {
            const u64 todo = __ballot(key != 0xFFFFFFFFu);
            if (!todo) break;
            const int lead = __ffsll((long long)todo) - 1;
            const u32 kl = (u32)__builtin_amdgcn_readlane((int)key, lead);
            const u64 same = __ballot(key == kl);
            if ((int)(threadIdx.x & 63u) == lead) atomicAdd(&h[kl], (u32)__popcll(same));
            if (key == kl) key = 0xFFFFFFFFu;
        }
    };
    auto general = [&](u64 a, u64 idx, u32 c, u32 nc, u32 sl, u32 rs, u64 so, u64 co, u64 c_lo, u64 c_hi) {
        u32 g_out = 0, nk_out = 0;
        u8 fl_out = 0;
        if (c >= n_contigs) report(status, idx, DE_BAD_CONTIG);
        else if (nc == 0) report(status, idx, DE_BAD_RUN);
        else prep_general(idx, rs, sl, so, cigar + co, nc, seq, g_base[c], c_hi - c_lo, &g_out, &nk_out, &fl_out, status);
        finish(a, idx, c, rs, g_out, nk_out, fl_out, 0u);
    };
    u32 fast_len = 0;  // the longest fast-class read this thread saw: picks the lane-group width of k_tile's plain class
    if (WO) {
        // the mirror: one 32-byte record per lane as two 16-byte loads, two records per trip in flight; nothing dependent
        // but the contig table
#ifndef PP_WO_UNROLL
#define PP_WO_UNROLL 2
#endif
        constexpr int WU = PP_WO_UNROLL;
        const uint4 *wq = (const uint4 *)wo;
        const u64 trip = (u64)WU * blockDim.x;
        const u64 span = (hi - lo + trip - 1) / trip * trip;  // whole waves and whole trips: the ballots below need every lane
        for (u64 a0 = lo + threadIdx.x; a0 < lo + span; a0 += trip) {
            uint4 qa[WU], qb[WU];
#pragma unroll
            for (int u = 0; u < WU; u++) {
                const u64 a = min(a0 + (u64)u * blockDim.x, n - 1);  // clamped: the loads are unconditional
                qa[u] = wq[2 * a];
                qb[u] = wq[2 * a + 1];
            }
#pragma unroll
            for (int u = 0; u < WU; u++) {
                const u64 a = a0 + (u64)u * blockDim.x;
                const bool in = a < hi;
                pp_wo_rec r;
                r.contig = qa[u].x; r.ref_start = qa[u].y; r.k = qa[u].z; r.seq_len = qa[u].w;
                r.seq_off = (u64)qb[u].x | ((u64)qb[u].y << 32); r.op0 = qb[u].z; r.file_idx = qb[u].w;
                const u32 cc = min(r.contig, n_contigs - 1u);
                const u64 c_lo = contig_off[cc], c_hi = contig_off[cc + 1], gb = g_base[cc];
                // the mirror is a hint, but its file index is what the arrays are read by (here for the records with several
                // runs, in k_tile for every slow item): an entry that names no record of the batch is the caller's error
                if (in && r.file_idx >= n) { report(status, a, DE_BAD_MIRROR); r.file_idx = 0; }
                bool bulk = in && r.contig < n_contigs && r.op0 != PP_WO_MULTI_RUN && (r.op0 & 15u) == PP_OP_M && (r.op0 >> 4) == r.seq_len &&
                            r.seq_len > 0 && r.seq_len <= FAST_MAX_LEN && (u64)r.ref_start + r.seq_len <= c_hi - c_lo;
                // what finish() will make of a bulk record, as far as the histogram's first entry goes (the same tests)
                u32 g_out = (u32)(gb + r.ref_start);
                bool kept = bulk;
                if (bulk) {
                    if (own && ((u64)r.ref_start + r.seq_len <= own[2 * r.contig] || r.ref_start >= own[2 * r.contig + 1])) kept = false;
                    if (gb == ~0ull) kept = false;
                    if (slice && kept && ((u64)r.ref_start < slice[2 * r.contig] || (u64)r.ref_start + r.seq_len > slice[2 * r.contig + 1])) kept = false;  // (finish reports DE_HALO)
                }
                // a bulk record is one piece of at most 252 positions: one window, or two -- both counted by the wave (with
                // the records in window order the second one is the same window for a whole wave as well: left to plain
                // atomics, those 7 % of the records serialised on one address)
                const u32 w0 = g_out / (u32)TILE, w1 = min((g_out + r.seq_len - 1u) / (u32)TILE, nwin - 1u);
                if (COUNT) {
                    count_first_by_wave(kept, w0 / cw);
                    count_first_by_wave(kept && w1 > w0, w1 / cw);
                }
                if (bulk) {
                    fast_len = max(fast_len, r.seq_len);
                    finish(a, r.file_idx, r.contig, r.ref_start, g_out, r.seq_len, 0u, kept && COUNT ? (w1 > w0 ? 2u : 1u) : 0u);
                } else if (in) {
                    const u32 slot = atomicAdd(&n_later, 1u);
                    if (slot < PREP_LATER_MAX) later[slot] = (u32)(a - lo);
                    else {
                        const u32 fi = r.file_idx;
                        const bool multi = r.op0 == PP_WO_MULTI_RUN;
                        general(a, fi, r.contig, multi ? n_cig[fi] : 1u, r.seq_len, r.ref_start, r.seq_off, cig_off[fi], c_lo, c_hi);
                    }
                }
            }
        }
        PP_STAMP(0, 2);
        __syncthreads();
        PP_STAMP(0, 3);
        for (u32 i = threadIdx.x; i < min(n_later, PREP_LATER_MAX); i += blockDim.x) {
            const u64 a = lo + later[i];
            const pp_wo_rec r = wo[a];
            const u32 cc = min(r.contig, n_contigs - 1u), fi = r.file_idx < n ? r.file_idx : 0u;  // (out of range: reported in the loop)
            general(a, fi, r.contig, r.op0 == PP_WO_MULTI_RUN ? n_cig[fi] : 1u, r.seq_len, r.ref_start, r.seq_off, cig_off[fi], contig_off[cc],
                    contig_off[cc + 1]);
        }
    } else {
    for (u64 a0 = lo + threadIdx.x; a0 < hi; a0 += (u64)PP_PREP_UNROLL * blockDim.x) {
        u32 c[PP_PREP_UNROLL], nc[PP_PREP_UNROLL], sl[PP_PREP_UNROLL], rs[PP_PREP_UNROLL];
        u64 co[PP_PREP_UNROLL];
#pragma unroll
        for (int u = 0; u < PP_PREP_UNROLL; u++) {  // the independent loads of all records of the trip, ...
            const u64 a = min(a0 + (u64)u * blockDim.x, hi - 1);  // clamped: the loads are unconditional
            c[u] = contig[a]; nc[u] = n_cig[a]; sl[u] = seq_len[a]; rs[u] = ref_start[a]; co[u] = cig_off[a];
        }
        u64 c_lo[PP_PREP_UNROLL], c_hi[PP_PREP_UNROLL], gb[PP_PREP_UNROLL];
        u32 op0[PP_PREP_UNROLL];
#pragma unroll
        for (int u = 0; u < PP_PREP_UNROLL; u++) {  // ... then the dependent ones
            const u32 cc = min(c[u], n_contigs - 1u);
            c_lo[u] = contig_off[cc]; c_hi[u] = contig_off[cc + 1]; gb[u] = g_base[cc];
            op0[u] = nc[u] ? cigar[co[u]] : 0u;
        }
#pragma unroll
        for (int u = 0; u < PP_PREP_UNROLL; u++) {
            const u64 a = a0 + (u64)u * blockDim.x;
            if (a >= hi) break;
            if (c[u] < n_contigs && nc[u] == 1 && (op0[u] & 15u) == PP_OP_M && (op0[u] >> 4) == sl[u] && sl[u] > 0 &&
                sl[u] <= FAST_MAX_LEN && (u64)rs[u] + sl[u] <= c_hi[u] - c_lo[u]) {
                // the bulk: one M run, short, inside its contig -> fast class, trimmed later by k_tile
                fast_len = max(fast_len, sl[u]);
                finish(a, a, c[u], rs[u], (u32)(gb[u] + rs[u]), sl[u], 0u, 0u);
            } else {
                const u32 slot = atomicAdd(&n_later, 1u);
                if (slot < PREP_LATER_MAX) later[slot] = (u32)(a - lo);
                else general(a, a, c[u], nc[u], sl[u], rs[u], seq_off[a], co[u], c_lo[u], c_hi[u]);
            }
        }
    }
    __syncthreads();
    for (u32 i = threadIdx.x; i < min(n_later, PREP_LATER_MAX); i += blockDim.x) {
        const u64 a = lo + later[i];
        const u32 c = contig[a], cc = min(c, n_contigs - 1u);
        general(a, a, c, n_cig[a], seq_len[a], ref_start[a], seq_off[a], cig_off[a], contig_off[cc], contig_off[cc + 1]);
    }
    }
    PP_STAMP(0, 4);
    // the job's longest fast-class read, once per wave and only beyond the narrowest lane group (<= 160 bases); the
    // word is read from L2, not from a possibly stale CU-local copy
    if (__ballot(fast_len > PLAIN_NARROW_MAX)) {
        for (int o = 32; o > 0; o >>= 1) fast_len = max(fast_len, (u32)__shfl_xor((int)fast_len, o, 64));
        if ((threadIdx.x & 63u) == 0 && fast_len > __hip_atomic_load(maxlen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            atomicMax(maxlen, fast_len);
    }
    if (COUNT) {
        __syncthreads();
        PP_STAMP(0, 5);
        for (u32 i = threadIdx.x; i < ncols; i += blockDim.x) hist[(u64)blockIdx.x * ncols + i] = h[i];
    }
    PP_STAMP(0, 6);
}

}  // namespace pp
