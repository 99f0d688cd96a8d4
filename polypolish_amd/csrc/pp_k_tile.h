// pp_k_tile.h -- k_tile: pileup accumulate + vote for one window (the dominant kernel).
// Part of pp_kernels.hip (included there, in this order, and nowhere else: it defines __global__ kernels).
#pragma once

namespace pp {

// =============================================================================================
// k_tile: pileup accumulate + vote for one 2048-position window
// =============================================================================================
struct TileArgs {
    const uint4 *entA;
    const u32 *win_off;
    u32 nwin;
    const u8 *seq;
    const u8 *seq4;   // optional 4-bit mirror of seq (pp_aln_batch.seq4): the plain class reads the bases from it
    const u64 *seq_off;
    const u64 *cig_off;
    const u32 *n_cig;
    const u32 *cigar;
    const u32 *kk;    // k per record (only for the rare share classes that do not carry their k: KCLASS_OTHER)
    const u8 *bases;
    u64 G;
    const u64 *contig_off;
    u32 n_contigs;
    u32 min_depth;
    double fv, fi;
    u8 *code;
    u32 *win_len;
    u32 *win_coarse, *win_coarse2;  // output bytes per WIN_COARSE / WIN_COARSE2 consecutive windows (k_emit's offsets: pp_k_emit.h)
    u32 *counters;  // [0] positions on the global replay list, [1] n_multi, [2] all flagged positions
    u32 cap_flag;
    const u32 *vote_tab;  // (valid, invalid) thresholds per integer depth below VOTE_TAB_N (k_meta_init)
    MultiEnt *multi;  // positions whose polished string has two or more bytes (counters[1] of them; k_exact adds its own)
    u32 cap_multi;
    u32 *flag_bits;   // per window: 2048-bit map of flagged positions (64 words)
    u32 *win_nflag;   // per window: number of flagged positions
    u32 *win_slab;    // per window: index of its tally slab (6 x 2048 u32), or ~0
    u32 *slab_win;    // per tally slab: its window
    u32 *slabs;
    u32 cap_slabs;
    u32 *flag_pos;
    u32 *flag_cov;
    u64 *flag_scr;  // per listed position: where its covering alignments go in the replay scratch
    u64 *scr_need;  // replay scratch the listed positions will need (sum of their coverage), counted past cap_flag too
    ContigStatsDev *stats;
    const u32 *maxlen;  // longest fast-class read (written by k_prep)
    u64 seq_bytes;
    const u32 *own;   // optional (lo, hi) emit range per contig, relative to the contig (pp_polish_set_emit)
    const u32 *own_win;  // with it: the ranges of windows that touch those ranges [n | first window of each | windows before each]
    u32 *heavy;           // HEAVY_WORDS: number of heavy windows | the listed ones | their arrival tickets
    const u8 *win_heavy;  // per window: 0, or 1 + its slot in the list
    u32 *hslab;           // per slot and part: HSLAB_WORDS partial tallies of that helper block
#ifdef PP_TILE_STAMPS
    u64 *stamps;          // profiling build: per block (start, items done, end, window) in 100 MHz ticks
#endif
    double *dbg_depth;
    u32 *dbg_counts;  // 7 planes of G: a, c, g, t, other, valid_thr, invalid_thr
    u8 *dbg_status;
    u64 *status;
    int dbg;
    // ---- the direct path (pp_k_direct.h): the window's bulk comes straight from the window-order mirror ----
    const uint4 *wo;      // the mirror, two 16-byte words per record
    const u32 *first;     // [n_runs][nwin + 1]: where the entries of window w begin in run r
    u32 n_runs;
    u32 xcap;             // room for extras per window
    const u32 *x_cnt;     // extras of each window ...
    const uint4 *xent;    // ... at xent[w * xcap ..]
    u32 *need_win;        // the windows the exact replays will read items of (k_xmat writes them out) ...
    u64 *n_need;          // ... and how many
};

// direct path: how a window's virtual item index maps to mirror entries (LDS), and the window's contig when it has only one
struct RecMap {
    const u32 *pre;    // [R + 1] entries of the runs before run r in this window (prefix sums of the stretches)
    const u32 *first;  // [R] first entry of the window in run r
    u32 R;
    u32 v0, nv;        // the entries this workgroup takes: [v0, v0 + nv) of the window's (all of them, or a heavy window's helper's share)
    u32 w;             // the window
    bool one_contig;
    u32 c0;            // its contig (one_contig) ...
    u64 c_lo, clen;    // ... where that starts, how long it is
    u32 w_in_contig;   // ... and where the window starts in it (window start minus contig start, modulo 2^32)
};

// the window's fixed-point bits and whether any item of it had a depth share other than 1 (then the deficit row is scanned)
struct TileShare {
    u32 b;            // win_fx_bits of the window
    u32 *any_shared;  // LDS flag
    const u32 *kk;
    u32 *pt, *pt_over;  // the window's table of two-byte keys (pt_insert) and its overflow flag
    const u32 *pmask;   // LDS table: pmask[PMASK_BASE + b] = pmask4(clamp(b, 0, 32)) (see wide4_pass)
};

// The depth share of ONE work item of class kc != 0 over the window positions [p0, p1) it covers (every entry of an
// alignment is one add_seq with the read's share, pileup.rs:56-65, and an item's entries are consecutive positions): its
// deficit against 1 goes into the deficit row as a DIFFERENCE (+d at p0, -d at p1; prefix-summed before the vote, like
// the coverage row), and an inexact share marks [p0, p1) in the window's bitmap of inexact positions.  Called by one lane
// per item.
__device__ __forceinline__ void share_range(u32 *cnt, u32 *ndbits, const TileShare &S, int p0, int p1, u32 kc, u32 rec) {
    if (kc == 0 || p1 <= p0) return;
    bool inexact;
    const u32 d = share_deficit(kc, S.b, S.kk, rec, &inexact);
    atomicAdd(&cnt[ROW_DEF * TILE + p0], d);
    if (p1 < TILE) atomicAdd(&cnt[ROW_DEF * TILE + p1], 0u - d);
    *S.any_shared = 1u;
    if (inexact) {
        const u32 a = (u32)p0, b = (u32)p1;
        for (u32 wd = a >> 5; wd <= (b - 1u) >> 5; wd++) {
            const u32 from = wd == (a >> 5) ? (a & 31u) : 0u, to = wd == ((b - 1u) >> 5) ? ((b - 1u) & 31u) : 31u;
            atomicOr(&ndbits[wd], (0xFFFFFFFFu >> (31u - to)) & (0xFFFFFFFFu << from));
        }
    }
}

__device__ __forceinline__ void tile_add(u32 *cnt, int row, int p) { atomicAdd(&cnt[row * TILE + p], 1u); }

// ---- the two-byte keys of the window, tallied by string ---------------------------------------------------------------
// An assembly that lacks a base makes every read over that spot vote for a two-byte key there (alignment.rs:175-201: the I
// run extends the entry before it; pileup.rs:56-63 counts it by string) -- the sites polishing is about, and all of their
// ~200 items are ENT_POINT items.  Their keys are tallied in a small LDS hash table, (position, two bytes) -> count and
// where the bytes stand in the seq array, so that such a position is voted right here like any other instead of being
// listed for k_exact (a workgroup per position that scans the window's items again: 0.05 ms per job for 167 sites).
// Whatever else lands in the position's string-keyed row -- N, longer insertions, the same key from a read that took the
// slow class -- is not in the table: the vote notices (the table's counts do not add up to the row) and lists the position
// as before.  Entry: tag = 1 << 31 | position << 16 | bytes; count in bits 0..23 of the second word, bits 32..39 of the
// offset above them; the offset's low word.
constexpr u32 PT_SLOTS = 10;  // (a window has a planted site or two; more distinct keys than slots: listed as before)
constexpr u32 FEW_FLAGGED = 4;  // windows with at most this many positions left for the ordered replay hand them to k_exact
__device__ __noinline__ void pt_insert(u32 *pt, u32 *over, int p, const u8 *seq, u64 so) {  // (rare: kept out of the item loop's registers)
    const u32 tag = 0x80000000u | ((u32)p << 16) | (u32)seq[so] | ((u32)seq[so + 1] << 8);
    u32 h = ((tag * 2654435761u) >> 16) % PT_SLOTS;
#pragma unroll 1
    for (u32 t = 0; t < PT_SLOTS; t++) {
        const u32 old = atomicCAS(&pt[3 * h], 0u, tag);
        if (old == 0u) {
            pt[3 * h + 2] = (u32)so;
            atomicAdd(&pt[3 * h + 1], 1u | ((u32)(so >> 32) << 24));
            return;
        }
        if (old == tag) { atomicAdd(&pt[3 * h + 1], 1u); return; }
        h = h + 1u == PT_SLOTS ? 0u : h + 1u;
    }
    *over = 1u;  // more distinct keys than the table holds: the window's string-keyed positions are listed as before
}

__device__ __forceinline__ u32 find_contig(const u64 *contig_off, u32 n_contigs, u64 p) {
    u32 lo = 0, hi = n_contigs;  // contig_off[lo] <= p < contig_off[hi]
    while (hi - lo > 1) {
        u32 mid = (lo + hi) >> 1;
        if (contig_off[mid] <= p) lo = mid; else hi = mid;
    }
    return lo;
}

// The same search by a whole wave, 64 ways per round: two dependent loads for up to 4096 contigs where the binary search
// above needs a dozen, one after the other (0.2 ms of the 100-contig job's k_tile, on every block's critical path).
__device__ __forceinline__ u32 find_contig_wave(const u64 *contig_off, u32 n_contigs, u64 p, u32 lane) {
    u32 lo = 0, hi = n_contigs;  // contig_off[lo] <= p < contig_off[hi]
    while (hi - lo > 1) {
        const u32 step = (hi - lo + 63u) / 64u;
        const u32 idx = lo + (lane + 1u) * step;
        const u64 v = idx < hi ? contig_off[idx] : ~0ull;
        const u32 k = (u32)__popcll(__ballot(v <= p));  // the offsets ascend: the lanes that pass are the first k
        lo += k * step;
        hi = min(hi, lo + step);
    }
    return lo;
}

struct VoteOut {
    u8 out;     // byte to emit (0 = nothing)
    u8 status;  // PP_ST_*
    u32 vthr, ithr;
};

// pileup.rs:67-134 restricted to the keys A,C,G,T and "-"; callers guarantee that no other key
// can reach either threshold.  vote5_thr: with the two thresholds and the depth test already evaluated.
__device__ __forceinline__ VoteOut vote5_thr(u32 nA, u32 nC, u32 nG, u32 nT, u32 nDel, u32 vthr, u32 ithr, bool low_depth, u8 orig) {
    VoteOut v;
    v.vthr = vthr;
    v.ithr = ithr;
    v.out = orig;
    v.status = PP_ST_KEPT;
    if (low_depth) {
        v.status = PP_ST_LOW_DEPTH;
    } else {
        int nv = 0, ni = 0;
        u8 win = 0;
        if (nA >= v.vthr) { nv++; win = 'A'; } else if (nA >= v.ithr) ni++;
        if (nC >= v.vthr) { if (!nv) win = 'C'; nv++; } else if (nC >= v.ithr) ni++;
        if (nG >= v.vthr) { if (!nv) win = 'G'; nv++; } else if (nG >= v.ithr) ni++;
        if (nT >= v.vthr) { if (!nv) win = 'T'; nv++; } else if (nT >= v.ithr) ni++;
        if (nDel > 0) {
            if (nDel >= v.vthr) { if (!nv) win = '-'; nv++; } else if (nDel >= v.ithr) ni++;
        }
        if (nv == 1) {
            if (ni > 0) v.status = PP_ST_TOO_CLOSE;
            else { v.out = win; if (win != orig) v.status = PP_ST_CHANGED; }
        } else if (nv == 0) {
            v.status = PP_ST_NONE;
        } else {
            v.status = PP_ST_MULTIPLE;
        }
    }
    if (v.out == (u8)'-') v.out = 0;  // polish.rs:188
    return v;
}
__device__ __forceinline__ VoteOut vote5(u32 nA, u32 nC, u32 nG, u32 nT, u32 nDel, double depth,
                                         u8 orig, u32 min_depth, double fv, double fi) {
    const u32 vt = d_bankers(__dmul_rn(depth, fv));
    return vote5_thr(nA, nC, nG, nT, nDel, max(min_depth, vt), d_bankers(__dmul_rn(depth, fi)), depth < (double)min_depth, orig);
}

// The vote of pileup.rs:77-134 over A C G T, "-" and the two-byte keys the window's table holds for position p -- for a
// position whose string-keyed row the table accounts for entirely (its counts add up to n_oth; else handled = false and the
// position is listed for k_exact).  When exactly one key is valid it wins whatever the order of the keys, otherwise
// nothing changes.  multi_eff > 0: a two-byte key won and leaves that many bytes (polish.rs:188 drops '-'), at multi_off in
// the seq array.  Rare (a few positions per window at most): not inlined, so that its registers are not the vote loop's.
struct KeyVote {
    u32 handled, out, status, multi_eff;
    u64 multi_off;
};
__device__ __noinline__ KeyVote vote_with_keys(const u32 *pt, u32 p, u32 nA, u32 nC, u32 nG, u32 nT, u32 nDel, u32 n_oth,
                                               u32 vthr, u32 ithr, u32 orig) {
    KeyVote r{0u, orig == (u32)'-' ? 0u : orig, (u32)PP_ST_KEPT, 0u, 0ull};
    u32 n_tab = 0;
#pragma unroll 1
    for (u32 t = 0; t < PT_SLOTS; t++) {
        const u32 tag = pt[3 * t];
        if (tag && ((tag >> 16) & 0x7FFFu) == p) n_tab += pt[3 * t + 1] & 0xFFFFFFu;
    }
    if (n_tab != n_oth) return r;
    r.handled = 1u;
    int nv = 0, ni = 0;
    u32 win = 0, win_t = 0xFFFFFFFFu;
    const u32 c5[5] = {nA, nC, nG, nT, nDel};
    const u32 k5[5] = {'A', 'C', 'G', 'T', '-'};
#pragma unroll
    for (int j = 0; j < 5; j++) {
        if (j == 4 && nDel == 0) break;
        if (c5[j] >= vthr) { if (!nv) win = k5[j]; nv++; } else if (c5[j] >= ithr) ni++;
    }
#pragma unroll 1
    for (u32 t = 0; t < PT_SLOTS; t++) {
        const u32 tag = pt[3 * t];
        if (!tag || ((tag >> 16) & 0x7FFFu) != p) continue;
        const u32 c = pt[3 * t + 1] & 0xFFFFFFu;
        if (c >= vthr) { if (!nv) win_t = t; nv++; } else if (c >= ithr) ni++;
    }
    if (nv == 1) {
        if (ni > 0) r.status = PP_ST_TOO_CLOSE;
        else if (win_t == 0xFFFFFFFFu) {
            r.out = (win == (u32)'-') ? 0u : win;
            if (win != orig) r.status = PP_ST_CHANGED;
        } else {
            r.status = PP_ST_CHANGED;  // a two-byte string is never the original base
            const u32 tag = pt[3 * win_t];
            const u32 b0 = tag & 0xFFu, b1 = (tag >> 8) & 0xFFu;
            const u32 eff = (u32)(b0 != (u32)'-') + (u32)(b1 != (u32)'-');
            const u32 only = b1 != (u32)'-' ? b1 : b0;
            if (eff == 0) r.out = 0;
            else if (eff == 1 && only < 0x80u) r.out = only;
            else {
                r.multi_eff = eff;
                r.multi_off = (u64)pt[3 * win_t + 2] | ((u64)(pt[3 * win_t + 1] >> 24) << 32);
            }
        }
    } else {
        r.status = (nv == 0) ? PP_ST_NONE : PP_ST_MULTIPLE;
    }
    return r;
}

// LDS copy of the window's assembly bytes: ASM_PAD bytes of slack in front, >= 20 behind, so that a
// lane may read the five dwords around any window position it owns a byte of.
constexpr int ASM_PAD = 32;
constexpr int ASM_WORDS = TILE / 4 + 24;
constexpr u32 PLAIN_MIN_LEN = 8;    // the trim reads the last four bases; shorter reads take the scalar path

// ---- plain class: fast class, depth share 1 (or non-dyadic), 8..32*GW bases ------------------------
// A group of GW lanes owns one work item; lane s of the group owns read bytes [32s, 32s+32), fetched with
// two 16-byte global loads at the read's own (arbitrary) byte offset -- gfx950 global loads need no
// alignment -- so a lane's bytes line up with window positions rel + 32s .. and only the END of a read
// (trimmed tail, bytes past the read) needs masking.  GW is picked per job from the longest fast-class
// read: 5 lanes (12 items per wave pass) up to 160 bases, 6 (10 items) up to 192, 8 (8 items) up to 252.
// Everything per item lives in vector registers (no v_readlane, no per-item branches).
// bit 7 of every non-zero byte
__device__ __forceinline__ u32 nz_flags(u32 x) {
    return (((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & 0x80808080u;
}
__device__ __forceinline__ u32 splat8(u32 n) {  // n * 0x01010101 for n < 256 (one v_perm_b32)
    return __builtin_amdgcn_perm(n, n, 0u);
}
// 4-bit mask of the non-zero bytes of x (v_dot4_u32_u8 of the 0/1 bytes with weights 1, 2, 4, 8)
__device__ __forceinline__ u32 nz_mask4(u32 x) {
    return __builtin_amdgcn_udot4(nz_flags(x) >> 7, 0x08040201u, 0u, false);
}
// (a vector type of alignment 1: ONE global_load_dwordx4 -- a 16-byte memcpy from a byte pointer is lowered to an
// overlapping dwordx3 + dwordx2 and moves that have to wait for the first of them)
typedef u32 u32x4_unaligned __attribute__((ext_vector_type(4), aligned(1)));
__device__ __forceinline__ uint4 load16_unaligned(const u8 *p) {
    const u32x4_unaligned v = *(const u32x4_unaligned *)p;
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ u32 load4_unaligned(const u8 *p) {
    u32 v;
    __builtin_memcpy(&v, p, 4);
    return v;
}

// PP_PLAIN_ALIGNED=1 (compile-time alternative, same speed on MI355X): lanes own 32-byte ALIGNED blocks of
// memory instead of read-relative chunks; a group then spans 32*GW-31 bases.
template <int GW>
struct PlainCfg {
    static constexpr u32 IPP = 64 / GW;                    // items per wave pass
    static constexpr u32 BATCH = (64 / IPP) * IPP;         // items per batch: whole passes only
    static constexpr u32 SPAN = PP_PLAIN_ALIGNED ? 32 * GW - 31 : 32 * GW;
    static constexpr u32 MAXL = SPAN < FAST_MAX_LEN ? SPAN : FAST_MAX_LEN;
    static_assert(GW != 5 || MAXL == PLAIN_NARROW_MAX, "k_prep's threshold");
    __device__ static __forceinline__ u32 group(u32 lane) {
        return GW == 8 ? lane >> 3 : (GW == 5 ? (lane * 52u) >> 8 : (lane * 43u) >> 8);
    }
    // work-item words x, y: no flags (any depth share: it goes into the deficit row as a range, share_range), length in
    // range, and every 32-byte chunk of the read inside the seq array
    __device__ static __forceinline__ bool ok(u32 ex, u32 ey, u64 seq_bytes) {
        const u32 L = ey >> 24;
        const u64 so = (u64)ex | ((u64)(ey & 0xFFu) << 32);
        return (ey & 0x00FF0000u) == 0 && L >= PLAIN_MIN_LEN && L <= MAXL && so + ((L + 31u) & ~31u) <= seq_bytes;
    }
};

struct PlainItem {  // per lane
    uint4 Wa, Wb;      // this lane's 32 read bytes
    u32 tail;          // the last four bases of the read (group-uniform)
    const u8 *lane_p;  // address of this lane's byte 0
    int rel;           // global start of the read minus the window start
    int ib;            // read index of this lane's byte 0
    bool first;        // lane 0 of the group
    u32 L;
    bool plain, active;
    u32 kc, rec;       // depth-share class (0: share 1) and record index (the share of the rare class that looks its k up)
    bool notrim;       // the flank in front of a read's single indel: its end is not the read's end, nothing to trim
};

// fields of the group's item (ds_bpermute from the batch registers) and the read loads, issued early
template <int GW>
__device__ __forceinline__ PlainItem plain_fetch(const u8 *seq, u64 seq_bytes, const uint4 &my, u32 nb, u32 first,
                                                 u32 lane) {
    typedef PlainCfg<GW> C;
    PlainItem it;
    const u32 g = C::group(lane), s = lane - (u32)GW * g;
    const u32 j = first + g;  // item of the batch owned by this group
    const int src = (int)(min(j, nb - 1u) << 2);
    const u32 ex = (u32)__builtin_amdgcn_ds_bpermute(src, (int)my.x), ey = (u32)__builtin_amdgcn_ds_bpermute(src, (int)my.y);
    const u32 ez = (u32)__builtin_amdgcn_ds_bpermute(src, (int)my.z);
    it.rel = item_rel(ez);
    it.notrim = ((ez >> 30) & 1u) != 0;
    it.L = ey >> 24;
    it.kc = (ey >> 8) & 0xFFu;
    it.rec = (u32)__builtin_amdgcn_ds_bpermute(src, (int)my.w);
    it.plain = g < C::IPP && j < nb && (ez >> 31) == 0 && C::ok(ex, ey, seq_bytes);
    const u8 *rp = seq + ((u64)ex | ((u64)(ey & 0xFFu) << 32));
    const u32 mis = PP_PLAIN_ALIGNED ? (u32)((uintptr_t)rp & 31u) : 0u;
    it.ib = (int)(32u * s) - (int)mis;
    it.first = s == 0;
    it.active = it.plain && 32u * s < mis + it.L;
    it.lane_p = rp + it.ib;
    // Loads only where there is something to load (exec-masked): measured faster than unconditional loads
    // from substitute addresses, and than prefetching the next pass across this pass's work.
    it.Wa = make_uint4(0, 0, 0, 0);
    it.Wb = make_uint4(0, 0, 0, 0);
    it.tail = 0;
    if (it.plain) it.tail = load4_unaligned(rp + (it.L - 4u));
    if (it.active) {
        it.Wa = load16_unaligned(it.lane_p);
        it.Wb = load16_unaligned(it.lane_p + 16);
    }
    return it;
}

__device__ __forceinline__ void plain_apply(u32 *cnt, u32 *ndbits, const TileShare &S, const u32 *asm_w, const PlainItem &it, u32 lane) {
    const int rel = it.rel;
    const u32 L = it.L;

    // ---- trim (alignment.rs:364-378): nkeep = index of the last base that differs from the last base,
    // read off the last four bases; a trailing homopolymer of four or more takes the byte loop
    const u32 last = it.tail >> 24;
    const u32 tf = nz_flags(it.tail ^ splat8(last));
    int nkeep = (int)L - 4 + ((31 - __clz((int)tf)) >> 3);
    if (it.notrim) nkeep = (int)L;
    else if (it.plain && tf == 0) {  // rare: walk left over the homopolymer
        const u8 *rp = it.lane_p - it.ib;
        u32 i = L - 4u;
        while (i > 0 && rp[i - 1] == (u8)last) i--;
        nkeep = i > 0 ? (int)i - 1 : 0;
    }
    const int lo = max(0, -rel), hi = min(nkeep, TILE - rel);
    const bool live = it.plain && hi > lo;

    // ---- coverage difference array (two atomics per read) ----
    if (live && it.first) {
        atomicAdd(&cnt[ROW_COV * TILE + rel + lo], 1u);
        if (rel + hi < TILE) atomicAdd(&cnt[ROW_COV * TILE + rel + hi], 0xFFFFFFFFu);
        share_range(cnt, ndbits, S, rel + lo, rel + hi, it.kc, it.rec);  // a depth share other than 1
    }
    // ---- compare this lane's 32 bases with the assembly; tally only the differing ones ----
    const int ib = it.ib;
    const int b0 = min(max(lo - ib, 0), 32), b1 = min(max(hi - ib, 0), 32);
    if (live && it.active && b1 > b0) {
        const int P0 = rel + ib;  // window position of byte 0 (> -32 here)
        const u32 ai = (u32)(P0 + ASM_PAD);
        const u32 *ap = asm_w + (ai >> 2);
        const u32 a0 = ap[0], a1 = ap[1], a2 = ap[2], a3 = ap[3], a4 = ap[4], a5 = ap[5], a6 = ap[6], a7 = ap[7], a8 = ap[8];
        const u32 sh = ai & 3u;
#define PP_D(k, w, x0, x1) (nz_mask4((w) ^ __builtin_amdgcn_alignbyte(x1, x0, sh)) << (4 * (k)))
        u32 D = PP_D(0, it.Wa.x, a0, a1) | PP_D(1, it.Wa.y, a1, a2) | PP_D(2, it.Wa.z, a2, a3) | PP_D(3, it.Wa.w, a3, a4) |
                PP_D(4, it.Wb.x, a4, a5) | PP_D(5, it.Wb.y, a5, a6) | PP_D(6, it.Wb.z, a6, a7) | PP_D(7, it.Wb.w, a7, a8);
#undef PP_D
        // bit i of D <=> byte i of this lane differs from the assembly; keep bytes [b0, b1) only
        D &= (0xFFFFFFFFu << b0) & (0xFFFFFFFFu >> (32 - b1));
        while (D) {  // one trip per differing base
            const int i = __ffs((int)D) - 1;
            D &= D - 1u;
            // byte i of the lane's eight dwords, by a select tree on the bits of i (no memory access: a
            // load here would have to wait for the next pass's prefetch as well)
            const u32 m4 = (u32)(((int)((u32)i << 29)) >> 31), m8 = (u32)(((int)((u32)i << 28)) >> 31),
                      m16 = (u32)(((int)((u32)i << 27)) >> 31);
#define PP_SEL(m, b, a) (((m) & (b)) | (~(m) & (a)))
            const u32 w01 = PP_SEL(m4, it.Wa.y, it.Wa.x), w23 = PP_SEL(m4, it.Wa.w, it.Wa.z);
            const u32 w45 = PP_SEL(m4, it.Wb.y, it.Wb.x), w67 = PP_SEL(m4, it.Wb.w, it.Wb.z);
            const u32 wlo = PP_SEL(m8, w23, w01), whi = PP_SEL(m8, w67, w45);
            const u32 c = (PP_SEL(m16, whi, wlo) >> (8 * (i & 3))) & 0xFFu;
#undef PP_SEL
            const int p = P0 + i;
            atomicAdd(&cnt[row_of(c) * TILE + p], 1u);
            atomicAdd(&cnt[ROW_MIS * TILE + p], 1u);
        }
    }
}

// ---- the plain class over the 4-bit mirror of the seq array (TileArgs::seq4) ------------------------------------------
// k_tile is HBM-bound on what the memory system fetches for a read, the 128-byte lines it touches: 2.16 for 150 bytes at
// an arbitrary offset, 1.58 for the 75 bytes of its mirror.  Lane s of a group owns bases [32s, 32s + 32) as before -- 16
// bytes now, ONE load -- and the compare runs on nibbles: the window's assembly bases are kept a second time in LDS, packed
// with the same codes (asm4), a funnel shift lines them up with the lane's bases, XOR, one bit per differing nibble.
// Codes (PP_SEQ4_*): A C T G = their counter rows, N, '-', 15 for any other read byte and 14 for any other assembly byte --
// the two never compare equal, so such a base is tallied explicitly where the byte compare might have found it equal to
// the assembly's: the same integers come out (the explicit tally and the mismatch row go up together, position_tallies).
// What needs a byte as it was delivered reads seq as before: the walk over a long homopolymer tail, and the whole trim
// when the read ends in a byte that has no code of its own (two different bytes may share code 15).
constexpr int ASM4_PAD = 256;                    // nibbles in front of the window's first position: a fast-class read that reaches in from the
                                                 // window before starts up to FAST_MAX_LEN - 1 positions in front of it (DirectBulk indexes without a clamp)
constexpr int ASM4_WORDS = (ASM4_PAD + TILE) / 8 + 24;        // 8 positions per dword; a lane reads five dwords from (P0 + ASM4_PAD) / 8 on -- and
                                                 // (direct_bulk) the 21 dwords under a 160-base read that starts at the window's last position
constexpr u32 SEQ4_ASM_OTHER = 14;

// bit k of the result <=> nibble k of x is not zero
__device__ __forceinline__ u32 nz_nibbles(u32 x) {
    u32 y = x | (x >> 1);
    y |= y >> 2;
    y &= 0x11111111u;                                  // bit 4k <=> nibble k
    const u32 z = (y | (y >> 3)) & 0x03030303u;        // two flags per byte: bits 0, 1 of byte b <=> nibbles 2b, 2b + 1
    return __builtin_amdgcn_udot4(z, 0x40100401u, 0u, false);
}
// The same for the four dwords of a 32-base chunk at once, WITHOUT moving the flags together: bit 4k + d of the result
// <=> nibble k of x_d is not zero, i.e. base 8d + k of the chunk differs -- a permutation of the chunk's bitmap, which is
// all the loop over the differing bases needs (19 instructions for 32 bases where four nz_nibbles and their three shifts are
// 39; the chunk compare is what k_tile's item loop spends its VALU issue on).  (n & 7) + 7 carries into bit 3 of its own
// nibble exactly when n & 7 is not zero and never beyond it.
__device__ __forceinline__ u32 nz_perm(u32 x0, u32 x1, u32 x2, u32 x3) {
    constexpr u32 m7 = 0x77777777u;
    const u32 u0 = ((x0 & m7) + m7) | x0, u1 = ((x1 & m7) + m7) | x1, u2 = ((x2 & m7) + m7) | x2, u3 = ((x3 & m7) + m7) | x3;
    u32 f = (u0 >> 3) & 0x11111111u;
    f = ((u1 >> 2) & 0x22222222u) | (f & ~0x22222222u);
    f = ((u2 >> 1) & 0x44444444u) | (f & ~0x44444444u);
    return (u3 & 0x88888888u) | (f & ~0x88888888u);
}
// the bases [0, b) of a chunk in nz_perm's bit order, b = 0 .. 32 (k_tile keeps a table of them in LDS: TileShare::pmask)
__device__ __forceinline__ u32 pmask4(u32 b) {
    u32 m = 0;
#pragma unroll
    for (u32 d = 0; d < 4u; d++) {
        const u32 n = (u32)min(max((int)b - 8 * (int)d, 0), 8);
        m |= (n == 8u ? 0x11111111u : (0x11111111u & ((1u << (4u * n)) - 1u))) << d;
    }
    return m;
}
__device__ __forceinline__ int row_of_code(u32 code) {
    return code < 4u ? (int)code : (code == (u32)PP_SEQ4_DASH ? ROW_DEL : ROW_OTH);
}
// the same in two instructions: a byte table in a register pair (v_perm_b32 with the selector min(code, 7): selector bytes
// 0..7 pick that byte of the table; every code from 6 on lands on a ROW_OTH entry)
__device__ __forceinline__ int row_of_code8(u32 code) {
    static_assert(ROW_A == 0 && ROW_C == 1 && ROW_T == 2 && ROW_G == 3 && ROW_DEL == 4 && ROW_OTH == 5 && PP_SEQ4_N == 4 && PP_SEQ4_DASH == 5 &&
                  PP_SEQ4_OTHER == 15, "the table");
    return (int)__builtin_amdgcn_perm(0x05050405u, 0x03020100u, min(code, 7u));
}

// The trim (alignment.rs:364-378) off the codes of the read's last EIGHT bases: `t` = the four bytes of the mirror that end
// with the read's last base (nibble index e = so + L - 1), loaded from seq4 + (e >> 1) - 3.  Returns the number of kept
// entries, or -1 when the bytes have to decide: the last seven bases all equal the last one (4^-6 of the reads), or the
// last byte has no code of its own (two different bytes may share code 15).  With a whole read per lane 64 reads share a
// pass, and a walk through the bytes -- a chain of dependent loads -- in two passes of three held up all of them.
__device__ __forceinline__ int trim4(u32 t, u64 e, u32 L) {
    const bool even = ((u32)e & 1u) == 0;   // then the top nibble of t belongs to the next read: move it out
    if (even) t <<= 4;
    const u32 last = t >> 28;
    u32 f = t ^ (last * 0x11111111u);
    f = (f | (f >> 1) | (f >> 2) | (f >> 3)) & (even ? 0x01111110u : 0x01111111u);  // bit 4p <=> base L-8+p differs (p <= 6)
    if (f == 0 || last == (u32)PP_SEQ4_OTHER) return -1;
    return (int)L - 8 + ((31 - __clz((int)f)) >> 2);
}
__device__ __forceinline__ int trim_bytes(const u8 *rp, u32 L) {  // the same through the bytes
    const u8 lastc = rp[L - 1u];
    u32 i = L - 1u;
    while (i > 0 && rp[i - 1] == lastc) i--;
    return i > 0 ? (int)i - 1 : 0;
}

struct PlainItem4 {  // per lane
    uint4 W;           // this lane's 32 bases, four bits each, base 0 in bits 0-3 of W.x
    u32 tail;          // the four bytes of the mirror that end with the read's last base (trim4)
    u64 e;             // index of the read's last base
    const u8 *rp;      // the read's bytes in seq (the rare trims that need them)
    int rel, ib;
    bool first;
    u32 L, kc, rec;
    bool plain, active, notrim;
};

template <int GW>
__device__ __forceinline__ PlainItem4 plain_fetch4(const u8 *seq, const u8 *seq4, u64 seq_bytes, const uint4 &my, u32 nb, u32 first,
                                                   u32 lane) {
    typedef PlainCfg<GW> C;
    static_assert(!PP_PLAIN_ALIGNED, "the 4-bit plain class owns read-relative chunks");
    PlainItem4 it;
    const u32 g = C::group(lane), s = lane - (u32)GW * g;
    const u32 j = first + g;
    const int src = (int)(min(j, nb - 1u) << 2);
    const u32 ex = (u32)__builtin_amdgcn_ds_bpermute(src, (int)my.x), ey = (u32)__builtin_amdgcn_ds_bpermute(src, (int)my.y);
    const u32 ez = (u32)__builtin_amdgcn_ds_bpermute(src, (int)my.z);
    it.rel = item_rel(ez);
    it.notrim = ((ez >> 30) & 1u) != 0;
    it.L = ey >> 24;
    it.kc = (ey >> 8) & 0xFFu;
    it.rec = (u32)__builtin_amdgcn_ds_bpermute(src, (int)my.w);
    it.plain = g < C::IPP && j < nb && (ez >> 31) == 0 && C::ok(ex, ey, seq_bytes);
    const u64 so = (u64)ex | ((u64)(ey & 0xFFu) << 32);  // index of the piece's first base: a byte of seq, a nibble of seq4
    it.rp = seq + so;
    it.ib = (int)(32u * s);
    it.first = s == 0;
    it.active = it.plain && 32u * s < it.L;
    it.W = make_uint4(0, 0, 0, 0);
    it.tail = 0;
    if (it.plain) it.tail = load4_unaligned(seq4 + ((so + (it.L - 1u)) >> 1) - 3);
    it.e = so + (it.L - 1u);
    if (it.active) {
        const u64 n0 = so + 32u * s;
        const u8 *q = seq4 + (n0 >> 1);
        it.W = load16_unaligned(q);
        if ((u32)n0 & 1u) {  // the piece starts on an odd base of the array (the flank behind an indel, a read after an odd-length one)
            const u32 xb = q[16];
            it.W = make_uint4(__builtin_amdgcn_alignbit(it.W.y, it.W.x, 4), __builtin_amdgcn_alignbit(it.W.z, it.W.y, 4),
                              __builtin_amdgcn_alignbit(it.W.w, it.W.z, 4), __builtin_amdgcn_alignbit(xb, it.W.w, 4));
        }
    }
    return it;
}

__device__ __forceinline__ void plain_apply4(u32 *cnt, u32 *ndbits, const TileShare &S, const u32 *asm4, const PlainItem4 &it, u32 lane) {
    const int rel = it.rel;
    const u32 L = it.L;
    int nkeep = trim4(it.tail, it.e, L);
    if (it.notrim) nkeep = (int)L;
    else if (it.plain && nkeep < 0) nkeep = trim_bytes(it.rp, L);  // rare
    const int lo = max(0, -rel), hi = min(nkeep, TILE - rel);
    const bool live = it.plain && hi > lo;
    if (live && it.first) {
        atomicAdd(&cnt[ROW_COV * TILE + rel + lo], 1u);
        if (rel + hi < TILE) atomicAdd(&cnt[ROW_COV * TILE + rel + hi], 0xFFFFFFFFu);
        share_range(cnt, ndbits, S, rel + lo, rel + hi, it.kc, it.rec);
    }
    const int ib = it.ib;
    const int b0 = min(max(lo - ib, 0), 32), b1 = min(max(hi - ib, 0), 32);
    if (live && it.active && b1 > b0) {
        const int P0 = rel + ib;  // window position of base 0 (> -32 here)
        const u32 ai = (u32)(P0 + ASM4_PAD);
        const u32 *ap = asm4 + (ai >> 3);
        const u32 a0 = ap[0], a1 = ap[1], a2 = ap[2], a3 = ap[3], a4 = ap[4];
        const u32 sh = 4u * (ai & 7u);
        u32 D = nz_nibbles(it.W.x ^ __builtin_amdgcn_alignbit(a1, a0, sh)) |
                (nz_nibbles(it.W.y ^ __builtin_amdgcn_alignbit(a2, a1, sh)) << 8) |
                (nz_nibbles(it.W.z ^ __builtin_amdgcn_alignbit(a3, a2, sh)) << 16) |
                (nz_nibbles(it.W.w ^ __builtin_amdgcn_alignbit(a4, a3, sh)) << 24);
        D &= (0xFFFFFFFFu << b0) & (0xFFFFFFFFu >> (32 - b1));
        while (D) {  // one trip per differing base
            const int i = __ffs((int)D) - 1;
            D &= D - 1u;
            const u32 m8 = (u32)(((int)((u32)i << 28)) >> 31), m16 = (u32)(((int)((u32)i << 27)) >> 31);
            const u32 wlo = (m8 & it.W.y) | (~m8 & it.W.x), whi = (m8 & it.W.w) | (~m8 & it.W.z);
            const u32 code = (((m16 & whi) | (~m16 & wlo)) >> (4 * (i & 7))) & 15u;
            const int p = P0 + i;
            atomicAdd(&cnt[row_of_code(code) * TILE + p], 1u);
            atomicAdd(&cnt[ROW_MIS * TILE + p], 1u);
        }
    }
}

// ---- the plain class, ONE LANE PER READ (4-bit mirror, reads up to 160 bases) ------------------------------------------
// What the item loop costs is VALU issue and the round trips of a pass.  A CDNA SIMD is 16 lanes wide, a wave instruction
// takes four cycles, and a pass of the lane-group scheme above is ~350 of them for 12 reads -- the fields of the item, the
// trim, the coverage atomics are worked out by all five lanes of a group, once per read.  With 75 bytes per read a lane can
// hold a whole read: NCH 16-byte loads (20 VGPRs at 160 bases), and ~450 instructions then serve 64 reads instead of 12.
// Lane l works on item l of the batch, straight from the batch registers (no ds_bpermute); its chunks are compared one
// after the other against asm4.  Round 4, from the ISA listing of this function:
//  * ALL loads of a pass -- the read's chunks, the four bytes the trim looks at, the next batch of items -- go out in one
//    basic block, one after the other: with a condition per chunk every load sat in a block of its own, followed by the
//    wait for it (five dependent round trips per pass; k_tile 0.244 -> 0.221 ms on configs[1]);
//  * a read that starts on an odd base of the array (the flank behind an indel, a read after one of odd length) is NOT moved
//    down by a nibble any more (twenty funnel shifts and the byte behind the last chunk, for the whole wave whenever one
//    lane had such a read): its window coordinates move instead -- it is compared as if it started one base earlier, with
//    that base outside [lo, hi);
//  * which bases of a chunk count -- [lo, hi) cut to the chunk -- is a pair of LDS table reads at constant offsets
//    (TileShare::pmask) instead of four clamps and a branch per chunk; a chunk with no base in range is compared like any
//    other and masked to nothing.
constexpr int PMASK_BASE = 128;                   // TileShare::pmask[PMASK_BASE + b] = pmask4(clamp(b, 0, 32)), b = -128 .. 161
constexpr int PMASK_WORDS = PMASK_BASE + 162;
// ... the table, worked out by the compiler (k_tile's workgroups copy it into LDS)
constexpr u32 pmask4_c(u32 b) {
    u32 m = 0;
    for (u32 d = 0; d < 4u; d++) {
        const int n0 = (int)b - 8 * (int)d;
        const u32 n = (u32)(n0 < 0 ? 0 : (n0 > 8 ? 8 : n0));
        m |= (n == 8u ? 0x11111111u : (0x11111111u & ((1u << (4u * n)) - 1u))) << d;
    }
    return m;
}
struct PmaskTab {
    u32 v[PMASK_WORDS];
    constexpr PmaskTab() : v{} {
        for (int i = 0; i < PMASK_WORDS; i++) v[i] = pmask4_c((u32)(i - PMASK_BASE < 0 ? 0 : (i - PMASK_BASE > 32 ? 32 : i - PMASK_BASE)));
    }
};
__device__ const PmaskTab g_pmask_tab{};

// A read the pass cannot take although PlainCfg<5>::ok says plain: an odd start whose LAST base counts (the flank in front
// of an indel is not trimmed) when the length fills its last chunk -- one base too many for NCH chunks.  (A trimmed read
// never looks at its last base.)  Such an item goes the way of the other fast classes.
__device__ __forceinline__ bool wide4_takes(u32 ex, u32 ey, u32 ez) {
    return !((ex & 1u) && ((ez >> 30) & 1u) && ((ey >> 24) & 31u) == 0);
}

template <int NCH, typename ASK>
__device__ __forceinline__ void wide4_pass(u32 *cnt, u32 *ndbits, const TileShare &S, const u32 *asm4, const u8 *seq, const u8 *seq4,
                                           const uint4 &my, bool mine, ASK &&ask_for_next) {
    static_assert(32 * (NCH - 1) <= PMASK_BASE && 32 * NCH + 1 < PMASK_WORDS - PMASK_BASE, "the range table");
    const u32 ex = my.x, ey = my.y, ez = my.z;
    const int rel = item_rel(ez);
    const bool notrim = ((ez >> 30) & 1u) != 0;
    const u32 L = ey >> 24;
    const u32 kc = (ey >> 8) & 0xFFu;
    const u64 so = (u64)ex | ((u64)(ey & 0xFFu) << 32);  // index of the piece's first base: a byte of seq, a nibble of seq4
    const u8 *q = seq4 + (so >> 1);
    const u32 adj = ex & 1u;  // an odd start: base i of the read is nibble i + 1 of what is loaded
    {
        // The next batch's items are asked for HERE, next to the pass's own loads, not at the top of tile_items' loop: a
        // request that is out before the item's fields are unpacked is waited for on the spot (the fields' registers may
        // be the target of a load from the slow classes' code further down the loop, and the wait the compiler puts in
        // front of the first write to them drains everything in flight).  The empty asm (in the caller's ask_for_next) ties
        // the index to the unpacked offset, which keeps the scheduler from moving the load back up.
        ask_for_next((u32)(so >> 32));
    }
    uint4 W[NCH];
    u32 tail = 0;
    const u32 nch = (L + 31u) >> 5;  // chunks of this lane's read
#pragma unroll
    for (int c = 0; c < NCH; c++) W[c] = make_uint4(0, 0, 0, 0);
    if (mine) {
        // A chunk past the read's last one repeats the last one's address (PlainCfg::ok vouches for the room up to there)
        // and is never looked at: no base of it is in range.
        tail = load4_unaligned(seq4 + ((so + (L - 1u)) >> 1) - 3);
#pragma unroll
        for (int c = 0; c < NCH; c++) W[c] = load16_unaligned(q + 16u * min((u32)c, nch - 1u));
    }
    int nkeep = trim4(tail, so + (L - 1u), L);
    if (notrim) nkeep = (int)L;
    else if (mine && nkeep < 0) nkeep = trim_bytes(seq + so, L);  // rare: see trim4
    const int lo = max(0, -rel), hi = min(nkeep, TILE - rel);
    const bool live = mine && hi > lo;
    if (live) {
        atomicAdd(&cnt[ROW_COV * TILE + rel + lo], 1u);
        if (rel + hi < TILE) atomicAdd(&cnt[ROW_COV * TILE + rel + hi], 0xFFFFFFFFu);
        share_range(cnt, ndbits, S, rel + lo, rel + hi, kc, my.w);
    }
    if (!__ballot(live)) return;
    // nibble x of the loaded chunks <-> window position relc + x; the nibbles [xlo, xhi) count (none of a lane that is not live)
    const int relc = rel - (int)adj;
    const u32 xlo = live ? (u32)lo + adj : 0u, xhi = live ? (u32)hi + adj : 0u;
    const u32 *mlo = S.pmask + xlo, *mhi = S.pmask + xhi;
    const u32 sh = (u32)(relc + ASM4_PAD) << 2;  // (v_alignbit takes it modulo 32: 4 * (the position's nibble in its dword), the same for all chunks)
    const int dw0 = (relc + ASM4_PAD) >> 3;      // the dword of asm4 that holds chunk 0's nibble 0 (below zero: a read that starts well before the window)
#pragma unroll
    for (int c = 0; c < NCH; c++) {
        const int P0 = relc + 32 * c;  // window position of the chunk's nibble 0
        // (clamped only for a chunk with nothing in range; one add and one v_med3 on the dword index -- the clamp on the
        // nibble index was seven instructions per chunk)
#ifndef PP_WIDE4_OLD_CLAMP
        const u32 *ap = asm4 + min(max(dw0 + 4 * c, 0), ASM4_WORDS - 5);
#else
        const u32 ai = (u32)min(max(P0 + ASM4_PAD, 0), 8 * (ASM4_WORDS - 5));
        const u32 *ap = asm4 + (ai >> 3);
#endif
        const u32 a0 = ap[0], a1 = ap[1], a2 = ap[2], a3 = ap[3], a4 = ap[4];
        u32 F = nz_perm(W[c].x ^ __builtin_amdgcn_alignbit(a1, a0, sh), W[c].y ^ __builtin_amdgcn_alignbit(a2, a1, sh),
                        W[c].z ^ __builtin_amdgcn_alignbit(a3, a2, sh), W[c].w ^ __builtin_amdgcn_alignbit(a4, a3, sh));
        F &= mhi[PMASK_BASE - 32 * c] & ~mlo[PMASK_BASE - 32 * c];
        while (F) {  // one trip per differing base: bit t = 4k + d <=> nibble k of dword d
            const u32 t = (u32)__ffs((int)F) - 1u;
            F &= F - 1u;
            const u32 md0 = 0u - (t & 1u), md1 = 0u - ((t >> 1) & 1u);
            const u32 wlo = (md0 & W[c].y) | (~md0 & W[c].x), whi = (md0 & W[c].w) | (~md0 & W[c].z);
            const u32 code = (((md1 & whi) | (~md1 & wlo)) >> (t & 28u)) & 15u;
            const int p = P0 + (int)(((t & 3u) << 3) | (t >> 2));
            atomicAdd(&cnt[row_of_code(code) * TILE + p], 1u);
            atomicAdd(&cnt[ROW_MIS * TILE + p], 1u);
        }
    }
}

// ---- fast class of work items: a read without indels, <= FAST_MAX_LEN bases, inside its contig ----
struct FastItem {  // wave-uniform (built from v_readlane results)
    u64 so;    // offset of the read in the seq array
    int rel;   // global start of the read minus the window start
    u32 L;     // read length == number of entries before the trim
    u32 kc;    // depth-share class of 1/k
    u32 mis;   // (address of the read) & 3
    bool on;
    bool notrim;  // see PlainItem
};

__device__ __forceinline__ FastItem fast_fetch(const uint4 &my, u32 j, u32 nb, const u8 *seq) {
    const int jj = (int)min(j, nb - 1u);
    const u32 x = (u32)__builtin_amdgcn_readlane((int)my.x, jj), y = (u32)__builtin_amdgcn_readlane((int)my.y, jj);
    FastItem f;
    f.so = (u64)x | ((u64)(y & 0xFFu) << 32);
    const u32 z = (u32)__builtin_amdgcn_readlane((int)my.z, jj);
    f.rel = item_rel(z);
    f.notrim = ((z >> 30) & 1u) != 0;
    f.L = y >> 24;
    f.kc = (y >> 8) & 0xFFu;
    f.mis = (u32)(((uintptr_t)(seq + f.so)) & 3u);
    f.on = j < nb && ((y >> 16) & 0xFFu) == 0;
    return f;
}

// One aligned dword per lane covers the whole read (<= 252 bases + <= 3 bytes of misalignment).
// An aligned dword that holds at least one byte of the read never leaves the read's pages.
__device__ __forceinline__ u32 fast_load(const u8 *seq, const FastItem &f, u32 lane) {
    u32 w = 0;
    if (f.on && 4u * lane < f.mis + f.L) w = *((const u32 *)(seq + f.so - f.mis) + lane);
    return w;
}

// trim (alignment.rs:364-378) by ballot; then (pileup.rs:56-65,189-200) either explicit LDS atomics
// per kept base (reads whose depth share is not 1) or, for the bulk, two coverage-difference
// atomics per read plus a 4-bases-at-a-time comparison against the assembly window in LDS, with
// per-base atomics only where the read differs from the assembly.
__device__ __forceinline__ void fast_apply(u32 *cnt, u32 *ndbits, const TileShare &S, const u32 *asm_w, const FastItem &f, u32 rec,
                                           u32 word, u32 lane) {
    if (!f.on) return;
    const u32 mis = f.mis;
    const int ib = (int)(4u * lane) - (int)mis;      // read index of this lane's byte 0
    const u32 tl = mis + f.L - 1u;                    // byte position of the last base in the wave load
    const u32 lw = (u32)__builtin_amdgcn_readlane((int)word, (int)(tl >> 2));
    const u32 c_last = (lw >> (8u * (tl & 3u))) & 0xFFu;
    int hi_i = -1;  // highest read index in this lane whose base differs from the last base
#pragma unroll
    for (int b = 0; b < 4; b++) {
        const int i = ib + b;
        const u32 c = (word >> (8 * b)) & 0xFFu;
        if (i >= 0 && i < (int)f.L && c != c_last) hi_i = i;
    }
    const u64 m = __ballot(hi_i >= 0);
    int nkeep = 0;  // index of the last base that differs: the run after it and that base are popped
    if (m) nkeep = __builtin_amdgcn_readlane(hi_i, 63 - __clzll((long long)m));
    if (f.notrim) nkeep = (int)f.L;
    const int lo = max(0, -f.rel), hi = min(nkeep, TILE - f.rel);
    if (hi <= lo) return;
    if (lane == 0) {
        atomicAdd(&cnt[ROW_COV * TILE + f.rel + lo], 1u);
        if (f.rel + hi < TILE) atomicAdd(&cnt[ROW_COV * TILE + f.rel + hi], 0xFFFFFFFFu);
        share_range(cnt, ndbits, S, f.rel + lo, f.rel + hi, f.kc, rec);
    }
    const int lowb = max(0, lo - ib), highb = min(4, hi - ib);
    if (lowb < highb) {
        const u32 M = (0xFFFFFFFFu >> (8 * (4 - highb))) & (0xFFFFFFFFu << (8 * lowb));
        const int P0 = f.rel + ib;                 // window position of byte 0 (>= -3 here)
        const u32 ai = (u32)(P0 + ASM_PAD);        // asm_w holds the window bytes at byte offset ASM_PAD
        const u32 w0 = asm_w[ai >> 2], w1 = asm_w[(ai >> 2) + 1];
        const u32 av = __builtin_amdgcn_alignbyte(w1, w0, ai & 3u);
        const u32 diff = (word ^ av) & M;
        if (diff) {
#pragma unroll
            for (int b = 0; b < 4; b++) {
                if ((diff >> (8 * b)) & 0xFFu) {
                    atomicAdd(&cnt[row_of((word >> (8 * b)) & 0xFFu) * TILE + P0 + b], 1u);
                    atomicAdd(&cnt[ROW_MIS * TILE + P0 + b], 1u);
                }
            }
        }
    }
}

// exact integer tallies of one window position from the LDS rows: explicit tallies plus, for the
// assembly's own base, the fast-class bases that were never tallied one by one
__device__ __forceinline__ void position_tallies(const u32 *cnt, u8 orig, u32 p, u32 &nA, u32 &nC, u32 &nG, u32 &nT,
                                                 u32 &nDel, u32 &nOth) {
    nA = cnt[ROW_A * TILE + p]; nC = cnt[ROW_C * TILE + p]; nT = cnt[ROW_T * TILE + p];
    nG = cnt[ROW_G * TILE + p]; nDel = cnt[ROW_DEL * TILE + p]; nOth = cnt[ROW_OTH * TILE + p];
    const u32 same = cnt[ROW_COV * TILE + p] - cnt[ROW_MIS * TILE + p];
    const int ro = row_of(orig);
    nA += (ro == ROW_A) ? same : 0u; nC += (ro == ROW_C) ? same : 0u; nT += (ro == ROW_T) ? same : 0u;
    nG += (ro == ROW_G) ? same : 0u; nDel += (ro == ROW_DEL) ? same : 0u; nOth += (ro == ROW_OTH) ? same : 0u;
}

// ---- slow class with indels, the usual shape: at most four CIGAR runs and at most 256 kept entries --------------------
// The runs come in scalar registers (fetched for all slow items of a batch at once), lane l owns entries l, l + 64,
// l + 128, l + 192: one sweep over the runs decides for each of them whether it is a deletion, a multi-base key or one
// read base (alignment.rs:175-201), ALL read bytes are then loaded together and tallied -- one memory round trip per
// item, where walking run by run with the runs in memory took eight or nine dependent ones (measured: 1 % of such
// reads cost k_tile 0.09 of its 0.49 ms).
__device__ __forceinline__ void slow_short(u32 *cnt, const TileShare &S, const u8 *seq, u64 so, int rel, int nkeep, u32 nc, u32 r0,
                                           u32 r1, u32 r2, u32 r3, u32 lane) {
    constexpr int SRC_NONE = -1, SRC_DEL = -2, SRC_OTH = -3;
    const u8 *s = seq + so;
    int src[4] = {SRC_NONE, SRC_NONE, SRC_NONE, SRC_NONE};
    int key2[4] = {-1, -1, -1, -1};  // read offset of a TWO-byte key (an M entry extended by one inserted base): tallied in the window's table too
    int ent0 = 0;
    u32 ro = 0;
    for (u32 r = 0; r < nc && ent0 < nkeep; r++) {
        const u32 op = r == 0 ? r0 : (r == 1 ? r1 : (r == 2 ? r2 : r3)), len = op >> 4, o = op & 15u;
        if (o == PP_OP_I) { ro += len; continue; }
        u32 ins = 0;  // bases inserted right after this run: they extend its last entry
        for (u32 q2 = r + 1; q2 < nc; q2++) {
            const u32 op2 = q2 == 1 ? r1 : (q2 == 2 ? r2 : r3);
            if ((op2 & 15u) != PP_OP_I) break;
            ins += op2 >> 4;
        }
        const int a = max(ent0, -rel), b = min(min(ent0 + (int)len, nkeep), TILE - rel);
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const int q = (int)lane + 64 * t;
            if (q < a || q >= b) continue;
            const bool ext = (q == ent0 + (int)len - 1) && ins > 0;
            if (o == PP_OP_D) src[t] = ext ? (ins == 1 ? (int)ro : SRC_OTH) : SRC_DEL;
            else {
                src[t] = ext ? SRC_OTH : (int)ro + (q - ent0);
                if (ext && ins == 1u) key2[t] = (int)ro + (q - ent0);
            }
        }
        ent0 += (int)len;
        if (o != PP_OP_D) ro += len;
    }
    u32 c[4];
#pragma unroll
    for (int t = 0; t < 4; t++) c[t] = src[t] >= 0 ? (u32)s[src[t]] : 0u;  // all loads of the item in flight together
#pragma unroll
    for (int t = 0; t < 4; t++) {
        if (src[t] == SRC_NONE) continue;
        const int row = src[t] >= 0 ? row_of(c[t]) : (src[t] == SRC_DEL ? ROW_DEL : ROW_OTH);
        tile_add(cnt, row, rel + (int)lane + 64 * t);
        if (key2[t] >= 0) pt_insert(S.pt, S.pt_over, rel + (int)lane + 64 * t, seq, so + (u64)key2[t]);
    }
}

// the items a wave takes when they are not the window's mirror entries (tile_items, !REC)
struct ItemList {
    u32 lo, n, stride;
    const u32 *indirect;  // (LDS) the items are ent[indirect[lo + stride * u]] instead of ent[lo + stride * u], or null
    // With `defer` a SLOW item (indels in several runs, a long read: one memory round trip each, one after the other) is
    // not tallied: its index goes into the list (LDS, `defer_cap` entries, `*defer_n` counts them) for a later call in which
    // all waves take a share (tile_window) -- a window whose reads cross two planted indels has hundreds of them, and a pass
    // worth of extras holds a sixteenth or a fifth of the window's, not a two-hundredth.  An item the list has no room for is
    // tallied on the spot.
    u32 *defer, *defer_n;
    u32 defer_cap;
    bool only_slow;  // nothing but the slow items is tallied (the others have been: DirectBulk's round over extras it could not list)
};
// ... a wave's contiguous slice of [e0, e1), whole passes of IPP items (the bucketing path)
template <u32 IPP>
__device__ __forceinline__ ItemList wave_slice(u32 e0, u32 e1, u32 wave) {
    constexpr u32 WAVES = TILE_THREADS / 64;
    const u32 per_wave = ((e1 - e0 + WAVES - 1u) / WAVES + IPP - 1u) / IPP * IPP;
    const u32 lo = min(e1, e0 + wave * per_wave);
    return ItemList{lo, min(e1, lo + per_wave) - lo, 1u, nullptr, nullptr, nullptr, 0u, false};
}

// two 1024-thread workgroups per CU (8 waves per SIMD): at most 64 VGPRs
// The work items of one window, one batch per wave at a time: one coalesced load of the batch's 16-byte
// records, then the plain class IPP items per pass, then the other classes one item per pass.  Latency is
// hidden by the other 7 waves of the SIMD, not by software pipelining of the passes (which measured slower).
template <int GW, bool P4, bool REC>
__device__ __forceinline__ void tile_items(const TileArgs &A, u32 *cnt, u32 *s_ndbits, const TileShare &S, const u32 *asm_w,
                                           const u32 *asm4, const uint4 *ent, const RecMap &M, u32 e0, u32 e1, const ItemList &X, u32 wave, u32 lane) {
    typedef PlainCfg<GW> C;
    // with the 4-bit mirror and reads of up to 192 bases: one lane per read, 64 items per batch and pass (wide4_pass)
    constexpr bool WIDE = P4 && GW == 5;
    constexpr u32 BATCH = WIDE ? 64u : C::BATCH;
    // REC (direct path): a wave's work is ONE list -- its share of the window's mirror entries (M.nv of them from M.v0 on,
    // numbered run after run) and, behind them, its share of the window's items in memory [e0, e1) (the extras) -- so that a
    // pass is as full of the one kind as of the other: a pass costs what it costs however many of its lanes have work, and
    // the ~300 extras of a window of 2,700 records as a list of their own were a fourth pass for every wave.  Every wave
    // takes a sixteenth of BOTH: the extras hold all of the window's slow items (one round trip each, one after the other),
    // which two waves at the tail of one long list would have had to themselves.  Entry v is record rec_at(v) of the mirror
    // (two 16-byte words), and its work item is made up here, in registers (wo_item: what k_fill would have written for
    // it); a record that is not bulk (its pieces are among the extras) is passed over.
    // Every wave takes one contiguous slice, equal to within one pass (the order of the items does not matter: the
    // counters are integers).
    constexpr u32 WAVES = TILE_THREADS / 64;
    const u32 nv_all = REC ? M.nv : 0u;
    const u32 per_wave_r = (nv_all + WAVES - 1u) / WAVES;
    const u32 r_lo = min(nv_all, wave * per_wave_r);
    const u32 nv = min(nv_all, r_lo + per_wave_r) - r_lo;   // this wave's mirror entries: [M.v0 + r_lo, ... + nv)
    // ... and its items: a contiguous slice [x_lo, x_lo + nx) of the bucketing's items; every sixteenth of the extras, x_lo +
    // 16 j (k_prepd writes a block's extras for the records with indels behind those for its bulk reads: in contiguous
    // slices two waves would get all of a window's slow items -- one window with two planted indels 100 bases apart, whose
    // reads have five runs, kept its workgroup for 190 us where the others take 25)
    // (!REC: the caller names the wave's items -- X.lo, X.lo + X.stride, ... X.n of them: a contiguous slice of the bucketing's
    // items, or one pass worth of a window's extras, see tile_window)
    const u32 XS = REC ? WAVES : X.stride;
    const u32 x_lo = REC ? e0 + wave : X.lo;
    const u32 nx = REC ? (e1 - e0 > wave ? (e1 - e0 - wave - 1u) / WAVES + 1u : 0u) : X.n;
    const u32 nv_last = (u32)__builtin_amdgcn_readfirstlane((int)(nv ? nv - 1u : 0u));  // (uniform: kept in a scalar register)
    const u32 lo_w = 0u, hi_w = nv + nx;  // the list: u < nv a mirror entry, else item x_lo + XS * (u - nv)
    if (lo_w >= hi_w) return;
    auto rec_at = [&](u32 v) -> u32 {
        u32 a = M.first[0] + v;
        for (u32 r = 1; r < M.R; r++)
            if (v >= M.pre[r]) a = M.first[r] + (v - M.pre[r]);
        return a;
    };
    // where entry u of the list lies: index of a mirror entry (two 16-byte words), or of an item
    const uint4 *const wq = A.wo;
    // (a lookup with scalar operands for a wave whose entries lie in one run or two -- base + u -- measured the same: 0.2196 vs 0.2185 ms)
    auto index_of = [&](u32 u) -> u32 {
        if (REC && u < nv) return rec_at(M.v0 + r_lo + min(u, nv_last));
        const u32 x = x_lo + XS * (u - min(u, nv));
        return !REC && X.indirect ? X.indirect[x] : x;
    };
    // ... and its words: both words of the mirror entry, or the item's one word twice (no branch)
    auto words_at = [&](bool rec, u32 idx, uint4 &a, uint4 &b) {
        if (REC) {
            const uint4 *const p1 = rec ? wq + 2ull * idx : ent + idx, *const p2 = rec ? wq + 2ull * idx + 1 : ent + idx;
            a = *p1;
            b = *p2;
        } else a = ent[idx];
    };
    // the records of the batch after the current one are asked for before the current one is worked on (2-5 % of the
    // kernel: a wave's chain of dependent round trips is what its time consists of)
    uint4 nxt, nxt2 = make_uint4(0, 0, 0, 0);
    {
        const u32 u = lo_w + min(lane, min(BATCH, hi_w - lo_w) - 1u);
        words_at(u < nv, index_of(u), nxt, nxt2);
    }
    for (u32 eb = lo_w; eb < hi_w; eb += BATCH) {
        const u32 nb = min(BATCH, hi_w - eb);
        uint4 my = nxt;
        bool rec_ok = true, is_rec = false;
        if (REC) {
            is_rec = eb + min(lane, nb - 1u) < nv;
            const uint4 qa = nxt, qb = nxt2;  // contig, ref_start, k, seq_len | seq_off (two words), op0, file index
            u64 c_lo = M.c_lo, clen = M.clen;
            bool c_ok = qa.x == M.c0;
            if (!M.one_contig) {  // (a window with a contig boundary in it)
                const u32 cc = min(qa.x, A.n_contigs - 1u);
                c_lo = A.contig_off[cc];
                clen = A.contig_off[cc + 1] - c_lo;
                c_ok = qa.x < A.n_contigs;
            }
            rec_ok = !is_rec || wo_bulk(c_ok, qa.y, qa.w, qb.z, clen);
            // (the depth-share class of k = 1 is 0: worked out only when some entry of the batch has another k)
            const u32 kc = __ballot(is_rec && qa.z != 1u) ? kclass_of(qa.z) : 0u;
            uint4 made = wo_item((u64)qb.x | ((u64)qb.y << 32), qa.w, kc, c_lo + qa.y, M.w, qb.w);
            // (one contig: the entry's place in the window is ref_start minus where the window starts in the contig -- a 32-bit
            // difference, exact for every entry that lies in the window)
            if (M.one_contig) made.z = (qa.y - M.w_in_contig) & 0x3FFFFFFFu;
            if (is_rec) my = made;
        }
        const bool more = eb + BATCH < hi_w;
        const u32 nxt_u = more ? eb + BATCH + min(lane, min(BATCH, hi_w - eb - BATCH) - 1u) : eb + min(lane, nb - 1u);
        const u32 nxt_idx = index_of(nxt_u);  // (out of LDS: worked out here, not among the pass's loads)
        const bool nxt_rec = nxt_u < nv;
        auto ask_for_next = [&](u32 tie) {
            u32 idx = nxt_idx;
            asm volatile("" : "+v"(idx) : "v"(tie));
            words_at(nxt_rec, idx, nxt, nxt2);
        };
        if (!WIDE && more) ask_for_next(0u);  // (WIDE: wide4_pass asks for them, together with its own loads)
        const u32 my_flags = is_rec ? 0u : item_flags(my.y, my.z);
        bool my_slow = lane < nb && (my_flags & 3u) != 0;
        bool deferred = false;
        if (!REC && X.defer && __ballot(my_slow)) {  // (rare)
            if (my_slow) {
                const u32 slot = atomicAdd(X.defer_n, 1u);
                if (slot < X.defer_cap) {
                    X.defer[slot] = index_of(eb + lane);
                    deferred = true;
                    my_slow = false;
                }
            }
        }
        const bool skip = !REC && X.only_slow;
        const bool my_point = lane < nb && !skip && (my_flags & ENT_POINT) != 0;
        const bool my_plain = lane < nb && !skip && rec_ok && !my_point && C::ok(my.x, my.y, A.seq_bytes) && (!WIDE || wide4_takes(my.x, my.y, my.z));
        // the slow items' record fields, one item per lane: asked for now, needed after the plain passes
        u64 sl_so = 0, sl_co = 0;
        u32 sl_nc = 0;
        if (!WIDE && my_slow) { sl_so = A.seq_off[my.w]; sl_co = A.cig_off[my.w]; sl_nc = A.n_cig[my.w]; }
        if constexpr (WIDE) {
            wide4_pass<GW>(cnt, s_ndbits, S, asm4, A.seq, A.seq4, my, my_plain, ask_for_next);
            if (my_slow) { sl_so = A.seq_off[my.w]; sl_co = A.cig_off[my.w]; sl_nc = A.n_cig[my.w]; }  // (a whole read per lane: no registers to spare across the pass)
        } else for (u32 first = 0; first < nb; first += C::IPP) {
            // (REC: a group picks its item out of the batch registers; a record that is not bulk shows as an item of length 0)
            uint4 mine = my;
            if (REC && !rec_ok) mine = make_uint4(0, 0, 0, 0);
            if (P4) plain_apply4(cnt, s_ndbits, S, asm4, plain_fetch4<GW>(A.seq, A.seq4, A.seq_bytes, mine, nb, first, lane), lane);
            else plain_apply(cnt, s_ndbits, S, asm_w, plain_fetch<GW>(A.seq, A.seq_bytes, mine, nb, first, lane), lane);
        }
        // the entry AT a read's single indel (ENT_POINT): one tally, one item per lane -- the two-byte key of an
        // insertion is counted by string (pileup.rs:56-63), the empty slot of a deletion is the "-" key
        if (my_point) {
            const int p = item_rel(my.z);
            if (p >= 0 && p < TILE) {
                tile_add(cnt, (my.y >> 24) ? ROW_OTH : ROW_DEL, p);
                share_range(cnt, s_ndbits, S, p, p + 1, (my.y >> 8) & 0xFFu, my.w);
                if ((my.y >> 24) == 2u) pt_insert(S.pt, S.pt_over, p, A.seq, (u64)my.x | ((u64)(my.y & 0xFFu) << 32));
            }
        }
        // fast class that is not plain (shared depth, or 242..252 bases): one item per pass
        u64 rest = __ballot(lane < nb && !skip && rec_ok && !my_slow && !deferred && !my_plain && !my_point);
        while (rest) {
            const u32 j = (u32)__ffsll((long long)rest) - 1u;
            rest &= rest - 1;
            const FastItem f = fast_fetch(my, j, nb, A.seq);
            fast_apply(cnt, s_ndbits, S, asm_w, f, (u32)__builtin_amdgcn_readlane((int)my.w, (int)j), fast_load(A.seq, f, lane), lane);
        }
        u64 slow = __ballot(my_slow);
        if (slow) {
            // the first four runs of every item with indels, again one item per lane
            u32 rr0 = 0, rr1 = 0, rr2 = 0, rr3 = 0;
            if (my_slow && (my_flags & ENT_COMPLEX)) {
                const u32 *cg = A.cigar + sl_co;
                rr0 = cg[0];
                if (sl_nc > 1) rr1 = cg[1];
                if (sl_nc > 2) rr2 = cg[2];
                if (sl_nc > 3) rr3 = cg[3];
            }
            while (slow) {
                const int j = __ffsll((long long)slow) - 1;
                slow &= slow - 1;
                const u32 ey = (u32)__builtin_amdgcn_readlane((int)my.y, j);
                const int rel = item_rel((u32)__builtin_amdgcn_readlane((int)my.z, j)), nkeep = __builtin_amdgcn_readlane((int)my.x, j);
                const u32 kc = (ey >> 8) & 0xFFu;
                const u64 so = (u64)(u32)__builtin_amdgcn_readlane((int)(u32)sl_so, j) |
                               ((u64)(u32)__builtin_amdgcn_readlane((int)(u32)(sl_so >> 32), j) << 32);
                const u8 *s = A.seq + so;
                {   // the item's depth share over the window positions its kept entries cover (entry q <-> position rel + q)
                    const int lo = max(0, -rel), hi = min(nkeep, TILE - rel);
                    const u32 recj = (u32)__builtin_amdgcn_readlane((int)my.w, j);
                    if (lane == 0) share_range(cnt, s_ndbits, S, rel + lo, rel + hi, kc, recj);
                }
                if (!((ey >> 16) & ENT_COMPLEX)) {
                    // no indels, trim precomputed by k_prep (long read or contig overhang): entry i is base i
                    const int lo = max(0, -rel), hi = min(nkeep, TILE - rel);
                    for (int i = lo + (int)lane; i < hi; i += 64) tile_add(cnt, row_of(s[i]), rel + i);
                    continue;
                }
                const u32 nc = (u32)__builtin_amdgcn_readlane((int)sl_nc, j);
                if (nc <= 4u && nkeep <= 256) {
                    slow_short(cnt, S, A.seq, so, rel, nkeep, nc, (u32)__builtin_amdgcn_readlane((int)rr0, j),
                               (u32)__builtin_amdgcn_readlane((int)rr1, j), (u32)__builtin_amdgcn_readlane((int)rr2, j),
                               (u32)__builtin_amdgcn_readlane((int)rr3, j), lane);
                    continue;
                }
                // many runs or a long read: run by run, with the runs read as they come
                const u64 co = (u64)(u32)__builtin_amdgcn_readlane((int)(u32)sl_co, j) |
                               ((u64)(u32)__builtin_amdgcn_readlane((int)(u32)(sl_co >> 32), j) << 32);
                const u32 *cg = A.cigar + co;
                int ent0 = 0;
                u64 ro = 0;
                for (u32 r = 0; r < nc && ent0 < nkeep; r++) {
                    const u32 op = cg[r], len = op >> 4, o = op & 15u;
                    if (o == PP_OP_I) { ro += len; continue; }
                    u32 ins = 0;
                    for (u32 r2 = r + 1; r2 < nc && (cg[r2] & 15u) == PP_OP_I; r2++) ins += cg[r2] >> 4;
                    const int a = max(ent0, -rel), b = min(min(ent0 + (int)len, nkeep), TILE - rel);
                    for (int q = a + (int)lane; q < b; q += 64) {
                        const bool ext = (q == ent0 + (int)len - 1) && ins > 0;
                        int row;
                        if (o == PP_OP_D) row = ext ? (ins == 1 ? row_of(s[ro]) : ROW_OTH) : ROW_DEL;
                        else row = ext ? ROW_OTH : row_of(s[ro + (u64)(q - ent0)]);
                        tile_add(cnt, row, rel + q);
                        if (o != PP_OP_D && ext && ins == 1u) pt_insert(S.pt, S.pt_over, rel + q, A.seq, so + ro + (u64)(q - ent0));
                    }
                    ent0 += (int)len;
                    if (o != PP_OP_D) ro += len;
                }
            }
        }
    }
}


// ---- the direct path's bulk, one lane per MIRROR ENTRY (round 6) --------------------------------------------------------
// Until round 6 a window's mirror entries went through tile_items as work items made up in registers (wo_item, unpacked
// again by wide4_pass) in ONE list with the window's extras, a sixteenth of it per wave: ~610 vector instructions per pass
// of 64 entries (rocprofv3 SQ_INSTS_VALU over the phase builds, profiles/r6_phases_*.txt) -- three quarters of the
// kernel's instructions on the 200x job -- and, at 50x, sixteen passes two thirds full where eleven full ones would do.
// A mirror entry of window w STARTS in w: no part of it lies in front of the window (lo = 0, no clamp of the assembly index:
// asm4 has room for a read's 160 bases behind the window), its five chunks are loaded at constant offsets from one
// address, the assembly's 21 dwords at constant offsets from another, its fields are used as they come: ~330 instructions
// a pass.  Passes are dealt to the waves one by one (pass p, p + 16, ... of the window's entries; the window's extras
// follow as passes of their own, tile_window): every pass but the window's last is full.  What a pass cannot take -- a bulk
// read of fewer than 8 bases, one at the very end of the seq array -- goes the way of the other fast classes, an entry that
// is not bulk is passed over (its pieces are among the extras).
//
// With the instructions down, what a wave's time consists of is its chain of memory round trips -- entries, then the
// reads they name, pass after pass: 24 us of items per workgroup on the 200x job whatever the pass costs (the block
// timeline of the first version of this function: as before it).  So the loads run AHEAD of the work:
//  * a pass's entries are asked for two passes before it (begin / the top of the pass before the pass before), reduced
//    to three registers (where the read lies, and rel | length | flags) one pass before it;
//  * its chunks are asked for DURING the pass before it, chunk c into the registers chunk c of the current pass has
//    just left (no second set of 20 registers: the workgroup's two-per-CU residency allows 64);
//  * the first pass's entries and chunks are asked for before the workgroup's prologue has zeroed its counters (begin /
//    stage between the prologue's own loads and its barrier): they need nothing the prologue writes.
struct BulkRuns {  // the window's stretches of the mirror in two registers: lane r holds pre[r] (its entries in the runs before r; lane R: all of them) and first[r]
    u32 pre_v, first_v, R;
};
// -DPP_TILE_SLOAD (measured, not the default: k_tile_direct 0.1925 vs 0.1952 ms on configs[1], nothing on configs[4] --
// inside the noise, not worth inline assembly in the product): a handful of uniform words at the top of a workgroup -- the
// window's stretch of every run, its extras' count -- come through the SCALAR cache (s_load_dword, inline: the compiler only uses scalar loads for memory it can prove nobody writes): as vector
// loads they queue behind everything else the CU's vector memory pipeline has in flight (2 us of every workgroup's chain
// entries -> chunks -> first pass).  What the words hold was written by the kernels before this one (the scalar cache is
// invalidated at the start of a kernel).  All requests go out before the one wait.
// (one asm statement: the requests and the wait together -- between two statements the compiler may copy a destination register
// that the load has not written yet)
constexpr u32 SLOAD_RUNS = 2;  // runs whose stretches are fetched this way (a job of more SAM files: vector loads, as before)
__device__ __forceinline__ void sload5(const u32 *p0, const u32 *p1, const u32 *p2, u32 &a0, u32 &a1, u32 &b0, u32 &b1, u32 &c) {
    asm volatile("s_load_dword %0, %5, 0x0\n\ts_load_dword %1, %5, 0x4\n\ts_load_dword %2, %6, 0x0\n\ts_load_dword %3, %6, 0x4\n\t"
                 "s_load_dword %4, %7, 0x0\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(a0), "=&s"(a1), "=&s"(b0), "=&s"(b1), "=&s"(c)
                 : "s"(p0), "s"(p1), "s"(p2)
                 : "memory");
}
struct BulkRunsRaw {  // as loaded: lane r's stretch of the window in run r, and the window's count of extras
    u32 f0, len, n_extras;
};
__device__ __forceinline__ BulkRunsRaw bulk_runs_load(const TileArgs &A, u32 w, u32 lane) {
    BulkRunsRaw raw{0u, 0u, 0u};
    const u32 R = A.n_runs;
#ifdef PP_TILE_SLOAD
    if (R <= SLOAD_RUNS) {
        u32 a0, a1, b0, b1, xc;
        const u32 *const row0 = A.first + w, *const row1 = A.first + (u64)(R - 1u) * (A.nwin + 1u) + w;  // (one run: its row twice)
        sload5(row0, row1, A.x_cnt + w, a0, a1, b0, b1, xc);
        raw.n_extras = xc;
        if (lane == 0) { raw.f0 = a0; raw.len = a1 - a0; }
        if (lane == 1 && R > 1u) { raw.f0 = b0; raw.len = b1 - b0; }
    } else
#endif
    {
        // (no load under a condition -- a lane past the last run reads the last run's words and drops them: a conditional load
        // is waited for where its branch ends, and the words asked for in front of it with it)
        const u32 r = min(lane, R - 1u);
        const u32 xc = A.x_cnt[w], a = A.first[(u64)r * (A.nwin + 1u) + w], b = A.first[(u64)r * (A.nwin + 1u) + w + 1u];
        raw.n_extras = xc;
        raw.f0 = lane < R ? a : 0u;
        raw.len = lane < R ? b - a : 0u;
    }
    return raw;
}
__device__ __forceinline__ BulkRuns bulk_runs_scan(const TileArgs &A, const BulkRunsRaw &raw, u32 lane) {
    BulkRuns B;
    const u32 R = A.n_runs;
    const u32 inc = wave_scan_incl(raw.len);  // (the lanes from R on hold 0)
    const u32 total = (u32)__builtin_amdgcn_readlane((int)inc, (int)(R - 1u));
    B.pre_v = lane < R ? inc - raw.len : total;
    B.first_v = raw.f0;
    B.R = R;
    return B;
}
// (experiment builds, tools/exp_variants.sh: -DPP_EXP_NOLOAD = the chunks are not loaded, the pass works on a constant;
// -DPP_EXP_NOCOMPARE = they are loaded and waited for, nothing is compared)
#ifdef PP_EXP_NOLOAD
#define PP_EXP_LOAD16(p) make_uint4((u32)(uintptr_t)(p), 0x01230123u, 0x32103210u, 0x11112222u)
#else
#define PP_EXP_LOAD16(p) load16_unaligned(p)
#endif
#ifndef PP_GROUP_SPLIT
#define PP_GROUP_SPLIT 5
#endif
// -DPP_EXP_COALESCED: chunk c of a pass's 64 reads is loaded as ONE contiguous kilobyte (lane l: 16 bytes at 1024 c + 16 l from
// where lane 0's read lies -- the same 5 KB the pass touches, wrong bytes in every lane but one: what a wave-interleaved
// layout of the reads would cost the memory pipeline)
#ifdef PP_EXP_COALESCED
#define PP_EXP_CHUNK(q, c, lane) ((const u8 *)(((uintptr_t)(u32)__builtin_amdgcn_readfirstlane((int)(u32)(uintptr_t)(q)) | ((uintptr_t)(u32)__builtin_amdgcn_readfirstlane((int)(u32)((uintptr_t)(q) >> 32)) << 32)) & ~(uintptr_t)15) + 1024 * (c) + 16 * (lane))
#else
#define PP_EXP_CHUNK(q, c, lane) ((q) + 16 * (c))
#endif
template <int NCH>
struct DirectBulk {
    BulkRuns B;
    u32 v0, nv, pm;       // the mirror entries this workgroup takes: [v0, v0 + nv) of the window's (v0: a multiple of 64), in pm passes
    const uint4 *items;   // ... and its extras: items[x0 .. x0 + nx) (16-byte work items, k_prepd / k_prepg), in px passes behind them
    u32 x0, nx, px;
    u32 pass;             // this wave's next pass (uniform): < pm one over mirror entries, else one over extras
    uint4 ea, eb;         // its sources: a mirror entry (contig, ref_start, k, seq_len | seq_off (two words), op0, file index) or an item (ea)
    uint4 na, nb;         // ... and those of the pass after it
    uint4 W[NCH];         // its chunks
    u32 tail;             // the four bytes of the mirror that end with the read's last base (trim4)
    static constexpr u32 WAVES = TILE_THREADS / 64;

    __device__ __forceinline__ u32 n_pass() const { return pm + px; }
    __device__ __forceinline__ u32 entry_of(u32 p, u32 lane) const {  // the mirror entry this lane takes in pass p < pm (past the end: the last one again)
        const u32 vs = v0 + 64u * p;  // the pass's first entry (uniform)
        const u32 v = min(vs + lane, v0 + nv - 1u);
        u32 r = 0;
        for (u32 j = 1; j < B.R; j++)  // (scalar: the run of the pass's first entry)
            if (vs >= (u32)__builtin_amdgcn_readlane((int)B.pre_v, (int)j)) r = j;
        u32 pr = (u32)__builtin_amdgcn_readlane((int)B.pre_v, (int)r), nx_ = (u32)__builtin_amdgcn_readlane((int)B.pre_v, (int)(r + 1u));
        u32 a = (u32)__builtin_amdgcn_readlane((int)B.first_v, (int)r) + (v - pr);
        while (__ballot(v >= nx_)) {  // a pass across the end of a run (one pass per run and window)
            r++;
            pr = nx_;
            nx_ = (u32)__builtin_amdgcn_readlane((int)B.pre_v, (int)(r + 1u));
            if (v >= pr) a = (u32)__builtin_amdgcn_readlane((int)B.first_v, (int)r) + (v - pr);
        }
        return a;
    }
    // the sources of pass p into (qa, qb): this lane's mirror entry, or its item (qb: the same again, not looked at)
    __device__ __forceinline__ void ask_sources(const TileArgs &A, u32 p, u32 lane, uint4 &qa, uint4 &qb) const {
        if (p < pm) {
            const u32 a = entry_of(p, lane);
            qa = A.wo[2ull * a];
            qb = A.wo[2ull * a + 1];
        } else {
            const u32 i = x0 + min(64u * (p - pm) + lane, nx - 1u);
            qa = items[i];
            qb = qa;
        }
    }
    __device__ __forceinline__ bool in_list(u32 p, u32 lane) const { return p < pm ? 64u * p + lane < nv : 64u * (p - pm) + lane < nx; }
    // What a pass needs of a source, in three registers: where the read's chunks are loaded from (its place in the mirror, or 0
    // when the pass does not take it) and (rel + LN_REL0) | length << 12 | flags.  Everything else (k, the file index, an
    // offset that is not loadable) is fetched again by the rare paths that need it.
    struct Lean {
        u64 so_l;
        u32 pack;
    };
    static constexpr u32 LN_REL0 = 256;  // (rel > -256: a fast-class read has at most FAST_MAX_LEN bases)
    static constexpr u32 LN_FAST = 1u << 20, LN_PLAIN = 1u << 21, LN_SHARED = 1u << 22, LN_NOTRIM = 1u << 23, LN_POINT = 1u << 24, LN_SLOW = 1u << 25;
    // the read's chunks are loaded from its place in the mirror when all NCH chunk loads stay inside it: 1 .. 32 NCH bases, 32 NCH
    // nibbles from its start on inside the array (seq4 has 32 bytes of slack behind it), and the four bytes that end with its last
    // base as well (so + L >= 8).  A piece of fewer than eight bases (the short flank of a read with its indel near an end) is
    // fine: the trim looks at the last eight bases of the READ, which are the read's own on either side of the indel; where they
    // are not (a whole read that short whose bases are all the same) trim4 comes out negative and the bytes decide.
    __device__ static __forceinline__ bool loadable(const TileArgs &A, u64 so, u32 L) {
        return L - 1u <= 32u * NCH - 1u && so + L >= 8u && so + 32u * NCH <= A.seq_bytes;
    }
    __device__ __forceinline__ Lean distill(const TileArgs &A, const RecMap &M, u32 p, const uint4 &qa, const uint4 &qb, bool listed) const {
        Lean n{0ull, 0u};
        if (p < pm) {  // a mirror entry (uniform branch)
            const u32 ref_start = qa.y, L = qa.w;
            u32 c_lo32 = (u32)M.c_lo, clen32 = (u32)M.clen;  // (a contig's length fits 32 bits: G < 2^32 - 4096)
            bool c_ok = qa.x == M.c0;
            if (!M.one_contig) {  // (a window with a contig boundary in it)
                const u32 cc = min(qa.x, A.n_contigs - 1u);
                const u64 lo64 = A.contig_off[cc];
                c_lo32 = (u32)lo64;
                clen32 = (u32)(A.contig_off[cc + 1] - lo64);
                c_ok = qa.x < A.n_contigs;
            }
            const u32 rel = c_lo32 + ref_start - (u32)((u64)M.w * TILE);  // where the read starts in the window
            // wo_bulk, in 32 bits (ref_start + L <= clen without the sum), and the entry's place in THIS window
            const bool bulk = listed && c_ok && qb.z == ((L << 4) | (u32)PP_OP_M) && L - 1u < FAST_MAX_LEN && L <= clen32 && ref_start <= clen32 - L &&
                              rel < (u32)TILE;
            const u64 so = (u64)qb.x | ((u64)qb.y << 32);
            const bool plain = bulk && loadable(A, so, L);  // the pass takes it (else: the other fast classes)
            n.so_l = plain ? so : 0ull;
            n.pack = bulk ? ((rel + LN_REL0) | (L << 12) | LN_FAST | (plain ? LN_PLAIN : 0u) | (qa.z != 1u ? LN_SHARED : 0u)) : 0u;
        } else {       // a work item (k_fill's format, pp_k_bucket.h)
            const u32 ex = qa.x, ey = qa.y, ez = qa.z;
            const u32 flags = item_flags(ey, ez), L = ey >> 24;
            const bool slow = (flags & 3u) != 0, point = (flags & ENT_POINT) != 0, notrim = (flags & ENT_NOTRIM) != 0;
            const u64 so = (u64)ex | ((u64)(ey & 0xFFu) << 32);
            const int rel = item_rel(ez);
            const bool fast = listed && !slow && rel > -(int)LN_REL0 && rel < TILE;
            const bool plain = fast && !point && loadable(A, so, L) && wide4_takes(ex, ey, ez);
            n.so_l = plain ? so : 0ull;
            n.pack = listed ? (slow ? LN_SLOW : (fast ? (((u32)(rel + (int)LN_REL0)) | (L << 12) | LN_FAST | (plain ? LN_PLAIN : 0u) |
                                                         (((ey >> 8) & 0xFFu) ? LN_SHARED : 0u) | (notrim ? LN_NOTRIM : 0u) | (point ? LN_POINT : 0u))
                                                       : 0u))
                            : 0u;
        }
        return n;
    }
    // (1) as soon as the window's stretches and the number of its extras are known: the sources of this wave's first pass
    __device__ __forceinline__ void begin(const TileArgs &A, const BulkRuns &runs, u32 v0_, u32 nv_, const uint4 *items_, u32 x0_, u32 nx_, u32 wave, u32 lane) {
        B = runs;
        v0 = v0_;
        nv = nv_;
        pm = (nv + 63u) >> 6;
        items = items_;
        x0 = x0_;
        nx = nx_;
        px = (nx + 63u) >> 6;
        pass = wave;
        ea = eb = make_uint4(0, 0, 0, 0);
        if (pass < n_pass()) ask_sources(A, pass, lane, ea, eb);
    }
    // where the chunks and the tail bytes of a source are loaded from, before anything else about it is known (stage)
    __device__ __forceinline__ void raw_place(const TileArgs &A, u32 p, const uint4 &qa, const uint4 &qb, u64 &so, u32 &L) const {
        if (p < pm) {
            so = (u64)qb.x | ((u64)qb.y << 32);
            L = qa.w;
        } else {
            so = (u64)qa.x | ((u64)(qa.y & 0xFFu) << 32);
            L = (qa.y & 0x00030000u) ? 0u : qa.y >> 24;  // (a slow item's x is not an offset)
        }
    }
    // (2) when those sources are there (still in the prologue): the first pass's chunks -- from where the source says its read
    // lies when that is inside the mirror, whatever else it turns out to be --, and the sources of the pass after it
    __device__ __forceinline__ void stage(const TileArgs &A, u32 lane) {
        if (pass >= n_pass()) return;
        const bool more = pass + WAVES < n_pass();
        ask_sources(A, more ? pass + WAVES : pass, lane, na, nb);
        u64 so;
        u32 L;
        raw_place(A, pass, ea, eb, so, L);
        const bool ld = loadable(A, so, L);
        tail = load4_unaligned(A.seq4 + ((ld ? so + (L - 1u) : 7ull) >> 1) - 3);
        const u8 *const q = A.seq4 + ((ld ? so : 0ull) >> 1);
#pragma unroll
        for (int c = 0; c < NCH; c++) W[c] = PP_EXP_LOAD16(PP_EXP_CHUNK(q, c, lane));
    }
    // (3) behind the prologue's barrier: the passes.  `defer`: the list for the extras' slow items (ItemList::defer)
    __device__ __forceinline__ void run(const TileArgs &A, u32 *cnt, u32 *ndbits, const TileShare &S, const u32 *asm_w, const u32 *asm4, const RecMap &M,
                                        u32 *defer, u32 *defer_n, u32 defer_cap, u32 lane) {
        const u32 np = n_pass();
        if (pass >= np) return;
        Lean cur = distill(A, M, pass, ea, eb, in_list(pass, lane));
        for (; pass < np; pass += WAVES) {
            const bool more = pass + WAVES < np;
            // ---- what the pass itself does not tally (all of it rare; the source is fetched again) -- in front of everything
            // else of the pass, where the fewest registers are live (behind the next pass's loads the allocator spilled four
            // of the chunk registers around it, on every pass)
            if (__ballot((cur.pack & (LN_SLOW | LN_POINT)) != 0 || (cur.pack & (LN_FAST | LN_PLAIN)) == LN_FAST)) {
                const int rel_r = (int)(cur.pack & 0xFFFu) - (int)LN_REL0;
                uint4 my;  // the source as a work item
                u32 src;
                if (pass < pm) {
                    src = entry_of(pass, lane);
                    const uint4 qa = A.wo[2ull * src], qb = A.wo[2ull * src + 1];
                    my = wo_item((u64)qb.x | ((u64)qb.y << 32), qa.w, kclass_of(qa.z), (u64)M.w * TILE + (u32)rel_r, M.w, qb.w);
                } else {
                    src = x0 + min(64u * (pass - pm) + lane, nx - 1u);
                    my = items[src];
                }
                // a slow item (indels in several runs, a long read): listed for the slow round (tile_window), or -- no room --
                // counted for a round over all of the window's extras
#ifndef PP_EXP_NOSLOW
                if (cur.pack & LN_SLOW) {
                    const u32 slot = atomicAdd(defer_n, 1u);
                    if (slot < defer_cap) defer[slot] = src;
                }
#endif
                // the entry AT a read's single indel (ENT_POINT): one tally -- the two-byte key of an insertion is counted by
                // string (pileup.rs:56-63), the empty slot of a deletion is the "-" key
#ifdef PP_EXP_NOPOINT
                if (false) {
#else
                if ((cur.pack & LN_POINT) && rel_r >= 0) {
#endif
                    tile_add(cnt, (my.y >> 24) ? ROW_OTH : ROW_DEL, rel_r);
                    share_range(cnt, ndbits, S, rel_r, rel_r + 1, (my.y >> 8) & 0xFFu, my.w);
                    if ((my.y >> 24) == 2u) pt_insert(S.pt, S.pt_over, rel_r, A.seq, (u64)my.x | ((u64)(my.y & 0xFFu) << 32));
                }
                // a fast-class read the pass does not take (fewer than 8 bases; the last reads of the seq array; an odd-start flank
                // that fills its last chunk): one item per pass, as the fast class that is not plain in tile_items
                u64 rest = __ballot((cur.pack & (LN_FAST | LN_PLAIN | LN_POINT)) == LN_FAST);
#ifdef PP_EXP_NOREST
                rest = 0;
#endif
                while (rest) {
                    const u32 j = (u32)__ffsll((long long)rest) - 1u;
                    rest &= rest - 1;
                    const FastItem f = fast_fetch(my, j, 64u, A.seq);
                    fast_apply(cnt, ndbits, S, asm_w, f, (u32)__builtin_amdgcn_readlane((int)my.w, (int)j), fast_load(A.seq, f, lane), lane);
                }
            }
            // the pass after this one: its sources are here (asked for a pass ago) and are reduced to what it needs; the sources
            // of the pass after THAT are asked for now, then its tail bytes -- and its chunks one by one below, as this pass's
            // chunks leave their registers
            const Lean nxt = distill(A, M, pass + WAVES, na, nb, more && in_list(pass + WAVES, lane));
            if (pass + 2u * WAVES < np) ask_sources(A, pass + 2u * WAVES, lane, na, nb);
            const u32 tail_now = tail;
            const u8 *const q_next = A.seq4 + (nxt.so_l >> 1);
            const u8 *const t_next = A.seq4 + (((nxt.pack & LN_PLAIN) ? nxt.so_l + (((nxt.pack >> 12) & 0xFFu) - 1u) : 7ull) >> 1) - 3;

            const int rel = (int)(cur.pack & 0xFFFu) - (int)LN_REL0;
            const u32 L = (cur.pack >> 12) & 0xFFu;
            const bool plain = (cur.pack & LN_PLAIN) != 0;
            const u64 so = cur.so_l;
            // ---- the plain reads ----
            const u32 adj = (u32)so & 1u;  // an odd start: base i of the read is nibble i + 1 of what is loaded
            int nkeep = trim4(tail_now, so + (L - 1u), L);
            if (cur.pack & LN_NOTRIM) nkeep = (int)L;  // (the flank in front of a read's single indel: its end is not the read's end)
            else if (plain && nkeep < 0) nkeep = trim_bytes(A.seq + so, L);  // rare: see trim4
            const int lo = max(0, -rel), hi = min(nkeep, TILE - rel);  // the kept entries in the window: [lo, hi) of the read
            const bool live = plain && hi > lo;
            if (live) {
                atomicAdd(&cnt[ROW_COV * TILE + rel + lo], 1u);
                if (rel + hi < TILE) atomicAdd(&cnt[ROW_COV * TILE + rel + hi], 0xFFFFFFFFu);
            }
            if (__ballot(live && (cur.pack & LN_SHARED))) {  // a depth share other than 1 (rare jobs: all-hits reads): class and file index from the source
                if (live && (cur.pack & LN_SHARED)) {
                    u32 kc, rec;
                    if (pass < pm) {
                        const u32 a = entry_of(pass, lane);
                        kc = kclass_of(A.wo[2ull * a].z);
                        rec = A.wo[2ull * a + 1].w;
                    } else {
                        const uint4 it = items[x0 + 64u * (pass - pm) + lane];
                        kc = (it.y >> 8) & 0xFFu;
                        rec = it.w;
                    }
                    share_range(cnt, ndbits, S, rel + lo, rel + hi, kc, rec);
                }
            }
            // nibble x of the loaded chunks <-> window position relc + x; the nibbles [xlo, xhi) count (none of a lane that is not live)
            const int relc = live ? rel - (int)adj : 0;  // > -ASM4_PAD
            const u32 xlo = live ? (u32)lo + adj : 0u, xhi = live ? (u32)hi + adj : 0u;
            const u32 *const mlo = S.pmask + xlo, *const mhi = S.pmask + xhi;
            const u32 sh = (u32)(relc + ASM4_PAD) << 2;  // (v_alignbit takes it modulo 32)
            const u32 *const ap = asm4 + ((relc + ASM4_PAD) >> 3);  // the dword of asm4 with chunk 0's nibble 0; 4 NCH + 1 dwords from here
#pragma unroll
            for (int c = 0; c < NCH; c++) {
                const int P0 = relc + 32 * c;  // window position of the chunk's nibble 0
                const u32 a0 = ap[4 * c], a1 = ap[4 * c + 1], a2 = ap[4 * c + 2], a3 = ap[4 * c + 3], a4 = ap[4 * c + 4];
                const uint4 Wc = W[c];
                u32 F = nz_perm(Wc.x ^ __builtin_amdgcn_alignbit(a1, a0, sh), Wc.y ^ __builtin_amdgcn_alignbit(a2, a1, sh),
                                Wc.z ^ __builtin_amdgcn_alignbit(a3, a2, sh), Wc.w ^ __builtin_amdgcn_alignbit(a4, a3, sh));
                F &= mhi[PMASK_BASE - 32 * c] & ~mlo[PMASK_BASE - 32 * c];
#ifdef PP_EXP_NOCOMPARE
                F = (Wc.x + Wc.y + Wc.z + Wc.w == 0x12345u && F) ? 1u : 0u;  // (experiment: the loads are waited for, nothing is tallied)
#endif
#ifdef PP_EXP_NOLOAD
                F = (lane == 13u * c && live) ? 1u : (F & 0u);  // (experiment: one trip of the loop below per chunk, as the real data make it)
#endif
                while (F) {  // one trip per differing base: bit t = 4k + d <=> nibble k of dword d
                    const u32 t = (u32)__ffs((int)F) - 1u;
                    F &= F - 1u;
                    // (selects, not masks: two compares and three v_cndmask where the mask form was ten instructions)
                    const bool b0 = (t & 1u) != 0, b1 = (t & 2u) != 0;
                    const u32 wlo = b0 ? Wc.y : Wc.x, whi = b0 ? Wc.w : Wc.z;
                    const u32 code = __builtin_amdgcn_ubfe(b1 ? whi : wlo, t & 28u, 4u);
                    const int p = P0 + (int)(((t & 3u) << 3) | (t >> 2));
                    atomicAdd(&cnt[row_of_code8(code) * TILE + p], 1u);
                    atomicAdd(&cnt[ROW_MIS * TILE + p], 1u);
                }
#if PP_GROUP_SPLIT > 0
                // The next pass's chunks in two groups, each back to back -- chunks 0 .. PP_GROUP_SPLIT - 1 when this pass's chunk
                // PP_GROUP_SPLIT - 1 is through, the others at the end: one by one (each into the registers its predecessor had just
                // left) every load was an L1 miss of its own, the loads of a read's neighbouring 16 bytes far apart in time (17.8 M
                // L1 -> L2 read requests per launch for 6.9 M L2 -> HBM ones, profiles/r6i_*; k_tile_direct 0.236 -> 0.217 ms).
                if (c == PP_GROUP_SPLIT - 1) {
#pragma unroll
                    for (int cc = 0; cc < PP_GROUP_SPLIT; cc++) W[cc] = PP_EXP_LOAD16(PP_EXP_CHUNK(q_next, cc, lane));
                }
                if (c == NCH - 1 && PP_GROUP_SPLIT < NCH) {
#pragma unroll
                    for (int cc = PP_GROUP_SPLIT; cc < NCH; cc++) W[cc] = PP_EXP_LOAD16(PP_EXP_CHUNK(q_next, cc, lane));
                }
                if (c == NCH - 1) tail = load4_unaligned(t_next);  // (its tail bytes with them: the line its last chunk is in)
#else
                W[c] = PP_EXP_LOAD16(PP_EXP_CHUNK(q_next, c, lane));  // (experiment: the next pass's chunk c, into the registers this pass's has just left)
                if (c == NCH - 1) tail = load4_unaligned(t_next);
#endif
            }
            cur = nxt;
        }
    }
};

constexpr u32 HSLAB_WORDS = (u32)(N_ROWS * TILE + TILE / 32);  // the nine counter rows + the bitmap of order-dependent positions
static_assert(HSLAB_WORDS % 4 == 0 && (N_ROWS * TILE) % 4 == 0 && TILE % 128 == 0, "the partial tallies move as 16-byte words");
constexpr u32 HEAVY_BLOCKS = HEAVY_SLOTS * HEAVY_PARTS;      // helper blocks at the front of k_tile's grid (a multiple of 8)

#ifndef PP_TILE_LAZY_ARGS
#define PP_TILE_LAZY_ARGS 1
#endif
// Profiling builds only (make variant NAME=stopK DEFS=-DPP_TILE_STOP=K, tools/exp_tile_phases.sh): an ordinary window's
// workgroup ends behind phase K -- 9 at its start, 8 the first entries asked for, 1 prologue, 2 items, 3 prefix sums, 4 vote pass 1,
// 5 vote pass 2 -- so that the counters
// and the duration of the truncated kernels give every phase's instructions and its share of the time (the window then
// emits nothing: the results are wrong by design).
#ifndef PP_TILE_STOP
#define PP_TILE_STOP 0
#endif
#define PP_STOP_AFTER(K)                                                          \
    if (PP_TILE_STOP == (K) && !heavy) {                                          \
        if (tid == 0) { A.win_len[w] = 0; A.win_nflag[w] = 0; }                   \
        return;                                                                   \
    }
// DIRECT (pp_k_direct.h): the window's bulk comes straight from the window-order mirror -- its entries in every run, run
// after run, through the plain class with their work items made up in registers -- and only its extras (the reads that
// reach in from the window before, the pieces of records with indels) are items in memory.
// One instance per lane-group width GW of the plain class and per kind of read fetch P4 (the 4-bit mirror or the bytes): the
// host launches the one that fits the job (run_pipeline; a job whose longest fast-class read the instance does not take
// raises DE_GW_HINT and is rerun with the one that does).  As ONE kernel that picked its item loop at run time -- six loops
// in a 22,000-line kernel -- k_tile_direct had 206 SGPR spills, a v_readlane / v_writelane each.
template <bool DIRECT, int GW, bool P4>
__device__ __forceinline__ void tile_window(const TileArgs &A) {
    __shared__ __attribute__((aligned(16))) u32 cnt[N_ROWS * TILE];
    __shared__ __attribute__((aligned(16))) u32 asm_w[ASM_WORDS];  // the window's assembly bytes at byte offset ASM_PAD
    __shared__ u32 asm4[ASM4_WORDS];  // the same as 4-bit codes, position p in nibble p + ASM4_PAD (only with TileArgs::seq4)
    __shared__ u32 s_len, s_changed, s_zero, s_c0, s_c1, s_fbits[TILE / 32], s_nflag, s_ticket, s_ndirty, s_shared;
    __shared__ __attribute__((aligned(16))) unsigned short s_dirty[TILE];  // the positions that need the vote proper (see below); while the items are tallied its first PMASK_WORDS words hold TileShare::pmask
    __shared__ __attribute__((aligned(16))) u32 s_ndbits[TILE / 32];
    __shared__ u64 s_depth;
    __shared__ u32 s_pt[PT_SLOTS * 3], s_ptover;
    __shared__ u32 s_need;  // DIRECT: a position of this window goes to an exact replay (its items have to be written out)
    __shared__ u32 s_nslow;  // BULK: the extras left for the slow round (ItemList::defer)
    // DIRECT: the window's stretches of the mirror, in s_dirty's space behind the range table (both are only needed while the
    // items are tallied): [0, R] prefix sums of the stretches' lengths, [32, 32 + R) where each begins
    u32 *const s_run = (u32 *)s_dirty + 512;
    // ... [48, 52) where the window's first contig starts and its length, [52] the window's extras, [53] its entries that are not bulk
    // ... and the prefix sums' per-wave totals (between the items and the vote: neither table is needed then, the list not yet)
    u32 *const s_wsum = (u32 *)s_dirty + 640;
    // ... and (BULK) the list of the window's slow extras (ItemList::defer), while the items are tallied
    u32 *const s_slow = (u32 *)s_dirty + 704;
    constexpr u32 SLOW_CAP = TILE / 2 - 704;
    static_assert(640 + TILE_THREADS / 64 <= 704 && SLOW_CAP <= 64 * (TILE_THREADS / 64), "the slow list");
    static_assert(PMASK_WORDS <= 512 && 512 + 54 <= 640 && 640 + TILE_THREADS / 64 <= TILE / 2 && PP_WO_MAX_RUNS <= 16, "the run table shares s_dirty with the range table");

    // The first HEAVY_BLOCKS blocks are helpers: block HEAVY_PARTS * slot + part tallies one part of the items of the
    // heavy window in that slot of the list.  They are dispatched first, so the longest windows start at time zero, and
    // the live ones are the LOWEST block numbers: the dispatcher deals consecutive workgroups round the CUs of an XCD
    // in order and waits when the next CU in turn is full, so two long-lived blocks on one CU would hold up the
    // dispatch for the whole XCD (measured: half of the chip's slots empty until the first helper finished).  Numbered
    // like this every CU gets at most one helper, next to an ordinary window.
    // Then the windows in XCD-aware order: consecutive windows (which share boundary-crossing reads) stay on one XCD.
    u32 w, part = 0, hslot = 0;
    const bool heavy = blockIdx.x < HEAVY_BLOCKS;
    // (asked for before anything is waited for: with a branch between them every one of these was a round trip of its own
    // at the start of every workgroup)
    const u64 status_word = *A.status;
    const u32 longest = *A.maxlen;
    if (heavy) {
        hslot = blockIdx.x / HEAVY_PARTS;
        part = blockIdx.x % HEAVY_PARTS;
        if (hslot >= min(A.heavy[0], HEAVY_SLOTS)) return;
        w = A.heavy[1 + hslot];
    } else {
        const u32 b = blockIdx.x - HEAVY_BLOCKS, per = (gridDim.x - HEAVY_BLOCKS) >> 3;
        w = (b & 7u) * per + (b >> 3);
        if (A.own_win) {
            // Sharded job: the grid only spans the windows this context works on; w is the number of one of them.
            // (Dealt out like this the XCDs share them evenly; a grid over ALL windows gave a rank's one stretch of a
            // window-tiled contig to a single XCD -- 7.1 ms per rank where the whole 250 Mbp job takes 6.9.)
            const u32 nr = A.own_win[0];
            const u32 *first = A.own_win + 1, *before = A.own_win + 1 + nr;
            if (w >= before[nr]) return;
            u32 lo = 0, hi = nr;  // before[lo] <= w < before[hi]
            while (hi - lo > 1) {
                const u32 step = (hi - lo + 63u) / 64u, idx = lo + ((threadIdx.x & 63u) + 1u) * step;
                const u32 k = (u32)__popcll(__ballot(idx < hi && before[idx] <= w));
                lo += k * step;
                hi = min(hi, lo + step);
            }
            w = first[lo] + (w - before[lo]);
        }
        if (w >= A.nwin) return;
    }
    if (PP_TILE_STOP == 9 && !heavy) {  // (profiling build: what a workgroup costs that does nothing)
        if (threadIdx.x == 0) { A.win_len[w] = 0; A.win_nflag[w] = 0; }
        return;
    }
    constexpr bool BULK = DIRECT && P4 && GW == 5;
    // (BULK: where the window's entries lie in every run and how many extras it has -- asked for HERE, with the three words the
    // next lines wait for: behind them it was a round trip of its own at the start of every workgroup)
    const u32 lane_early = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const u32 listed = (u32)A.win_heavy[w];  // (read by a helper too: a load under a condition is waited for where the branches meet)
    BulkRunsRaw raw_runs{0u, 0u, 0u};
    BulkRuns runs_early{0u, 0u, 0u};
    if constexpr (BULK) {
        // (... and scanned here, in front of the early returns: loads nobody uses on the returning path are moved behind the branch)
        raw_runs = bulk_runs_load(A, w, lane_early);
        runs_early = bulk_runs_scan(A, raw_runs, lane_early);
    }
    if (listed && !heavy) return;  // listed windows belong to the helpers
    if (status_word != ~0ull && (status_word & 0xFFu) != DE_CAPACITY_LATE) return;  // (job_state == 2: aborted)
    if (longest > PlainCfg<GW>::MAXL) {  // not this instance's job (the host's hint was off: it reruns with the right one)
        if (threadIdx.x == 0) report(A.status, (1ull << 40) - 1ull, DE_GW_HINT);
        return;
    }
    // (the wave's number in a scalar register, the lane's from v_mbcnt, the thread's from the two: three values that live across
    // the whole kernel, none of them in a vector register that has to be kept -- or spilled: the item loop of the direct path
    // left four registers of them in scratch memory, 80 MB of traffic per launch on configs[1])
    const u32 wave = (u32)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const u32 lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const u32 tid = (wave << 6) | lane;
    const u64 w0 = (u64)w * TILE;
#ifdef PP_TILE_STAMPS
    if (tid == 0) {
        A.stamps[8ull * blockIdx.x] = wall_clock64();
        A.stamps[8ull * blockIdx.x + 3] = w | ((u64)part << 32) | ((u64)heavy << 40);
        A.stamps[8ull * blockIdx.x + 4] = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));   // HW_ID
        A.stamps[8ull * blockIdx.x + 5] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));   // XCC_ID
    }
#endif

    if (A.own && !A.own_win) {
        // Sharded job: a window that lies inside ONE contig and outside the range of it this context emits is
        // somebody else's (k_prep gave it no work items): nothing to tally, nothing to vote.
        // (Not with the list of the windows this context works on -- own_win, what the host always brings along with the
        // ranges: every window of the grid then touches a range it emits, and the question -- the window's contig, its
        // offset, its range: four dependent round trips at the top of every workgroup -- made a rank's share of configs[3]
        // 0.175 ms of k_tile where its windows take 0.134 at the whole job's rate.)
        const u64 last = min(w0 + TILE, A.G) - 1;
        const u32 c0 = find_contig_wave(A.contig_off, A.n_contigs, w0, lane);
        const u64 cb = A.contig_off[c0];
        if (last < A.contig_off[c0 + 1] && (last - cb < A.own[2 * c0] || w0 - cb >= A.own[2 * c0 + 1])) {
            for (u32 p = tid; p < (u32)TILE && w0 + p < A.G; p += TILE_THREADS) A.code[w0 + p] = 0;
            if (tid < (u32)(TILE / 32)) A.flag_bits[(u64)w * (TILE / 32) + tid] = 0;
            if (tid == 0) {
                A.win_nflag[w] = 0;
                A.win_len[w] = 0;
            }
            return;
        }
    }
    // The bulk of a direct window, one lane per mirror entry (DirectBulk): every wave works the window's stretches out for
    // itself (2 n_runs loads) and asks for its first pass's entries right away -- the rest of the prologue (zeroing the
    // counters, the assembly's bytes and codes, the tables) runs while they are on their way.
    DirectBulk<5> D;
    if constexpr (BULK) {
        const BulkRuns runs = runs_early;
        const u32 n_x = min(raw_runs.n_extras, A.xcap);  // the window's extras (asked for with the stretches)
        const u32 n_all = (u32)__builtin_amdgcn_readlane((int)runs.pre_v, (int)runs.R);  // the window's mirror entries
        u32 v0 = 0, v1 = n_all, x0 = 0, x1 = n_x;
        if (heavy) {  // this helper's share (as below)
            const u32 vchunk = ((n_all + HEAVY_PARTS - 1u) / HEAVY_PARTS + 63u) & ~63u;
            v0 = min(n_all, part * vchunk);
            v1 = min(n_all, v0 + vchunk);
            const u32 chunk = ((n_x + HEAVY_PARTS - 1u) / HEAVY_PARTS + 63u) & ~63u;
            x0 = min(n_x, part * chunk);
            x1 = min(n_x, x0 + chunk);
        }
        D.begin(A, runs, v0, v1 - v0, A.xent + (u64)w * A.xcap, (u32)__builtin_amdgcn_readfirstlane((int)x0),
                (u32)__builtin_amdgcn_readfirstlane((int)(x1 - x0)), (u32)__builtin_amdgcn_readfirstlane((int)wave), lane);
    }
    PP_STOP_AFTER(8)  // (profiling build: the window's stretches known, the first pass's entries asked for)
    for (u32 i = tid; i < (u32)(N_ROWS * TILE / 4); i += TILE_THREADS) ((uint4 *)cnt)[i] = make_uint4(0, 0, 0, 0);  // (16 bytes a store)
    if (tid < (u32)(TILE / 32)) { s_fbits[tid] = 0; s_ndbits[tid] = 0; }
    {
        u8 *ab = (u8 *)asm_w;
        for (u32 i = tid; i < (u32)TILE; i += TILE_THREADS) ab[ASM_PAD + i] = (w0 + i < A.G) ? A.bases[w0 + i] : (u8)0;
        if (tid < (u32)ASM_PAD) ab[tid] = 0;
        if (tid < (u32)(ASM_WORDS * 4 - ASM_PAD - TILE)) ab[ASM_PAD + TILE + tid] = 0;
    }
    if (P4 && tid >= TILE_THREADS - (u32)ASM4_WORDS) {  // (the last waves: the first two search the contig table)
        const u32 t = tid - (TILE_THREADS - (u32)ASM4_WORDS);
        const int p0 = 8 * (int)t - ASM4_PAD;  // dword t holds positions p0 .. p0 + 7
        u32 lo8 = 0, hi8 = 0;                  // their bytes; 0 where there is no position (code "other": never compared)
        if (p0 >= 0 && p0 + 8 <= TILE && w0 + (u64)p0 + 8u <= A.G) {
            uint2 two;
            __builtin_memcpy(&two, A.bases + w0 + (u64)p0, 8);  // one load
            lo8 = two.x; hi8 = two.y;
        } else {
            for (int jj = 0; jj < 8; jj++) {
                const int p = p0 + jj;
                if (p >= 0 && p < TILE && w0 + (u64)p < A.G) {
                    const u32 c = A.bases[w0 + (u64)p];
                    if (jj < 4) lo8 |= c << (8 * jj); else hi8 |= c << (8 * (jj - 4));
                }
            }
        }
        u32 v = 0;
#pragma unroll
        for (int jj = 0; jj < 8; jj++) {
            const u32 c = ((jj < 4 ? lo8 : hi8) >> (8 * (jj & 3))) & 0xFFu;
            const u32 tt = (c >> 1) & 3u, expect = (0x47544341u >> (tt * 8u)) & 0xFFu;
            const u32 code = c == expect ? tt : (c == (u32)'N' ? (u32)PP_SEQ4_N : (c == (u32)'-' ? (u32)PP_SEQ4_DASH : SEQ4_ASM_OTHER));
            v |= code << (4 * jj);
        }
        asm4[t] = v;
    }
    if (tid == 0) { s_len = 0; s_changed = 0; s_zero = 0; s_depth = 0; s_nflag = 0; s_ndirty = 0; s_shared = 0; s_ptover = heavy ? 1u : 0u; s_need = 0; s_nslow = 0; }
    if (DIRECT && tid >= 192u && tid < 192u + 64u) {  // one wave: the window's stretch of every run, and their prefix sums
        const u32 r = tid - 192u, R = A.n_runs;
        u32 len = 0, f0 = 0;
        if (r < R) {
            f0 = A.first[(u64)r * (A.nwin + 1u) + w];
            len = A.first[(u64)r * (A.nwin + 1u) + w + 1u] - f0;
        }
        u32 inc = len;
        for (int o = 1; o < 32; o <<= 1) {
            const u32 t = (u32)__shfl_up((int)inc, o, 64);
            if ((int)r >= o) inc += t;
        }
        if (r <= R && r < 32u) s_run[r] = inc - len;  // (lane R holds the total)
        if (r < R) s_run[32u + r] = f0;
        if (r == 63u) s_run[52] = min(A.x_cnt[w], A.xcap);  // the window's extras (asked for here, with everything else of the prologue)
        if (r == 62u) s_run[53] = A.x_cnt[A.nwin + w];       // ... and its entries that are not bulk (x_nb: k_tile passes them over)
    }
    if (tid >= 128u && tid < 128u + (u32)PMASK_WORDS) ((u32 *)s_dirty)[tid - 128u] = g_pmask_tab.v[tid - 128u];  // (one load: ~40 instructions to work it out)
    if (tid >= 64u && tid < 64u + PT_SLOTS * 3u) s_pt[tid - 64u] = 0;  // (a heavy window's helpers would each have their own table: listed as before)
    if (wave < 2u) {  // the contigs of the window's first and last position, one wave each
        const u32 cw = find_contig_wave(A.contig_off, A.n_contigs, wave == 0 ? w0 : min(w0 + TILE, A.G) - 1, lane);
        if (lane == 0) { if (wave == 0) s_c0 = cw; else s_c1 = cw; }
        if (DIRECT && wave == 0 && lane == 0) {  // where the window's (first) contig starts and how long it is: for its mirror entries
            const u64 lo64 = A.contig_off[cw], len64 = A.contig_off[cw + 1] - lo64;
            s_run[48] = (u32)lo64; s_run[49] = (u32)(lo64 >> 32); s_run[50] = (u32)len64; s_run[51] = (u32)(len64 >> 32);
        }
    }
#ifndef PP_EXP_STAGE_LATE
    if constexpr (BULK) D.stage(A, lane);  // the first pass's chunks and the second's entries: asked for in front of the barrier
#endif
    __syncthreads();
    PP_STOP_AFTER(1)

#ifdef PP_TILE_STAMPS
    if (tid == 0) A.stamps[8ull * blockIdx.x + 6] = wall_clock64();
#endif
    // The window's items: [e0, e1) of the work items in memory -- the bucketing's (k_fill), or (DIRECT) the window's extras --
    // and, DIRECT, its n_rec mirror entries in front of them.
    u32 e0, e1, n_rec = 0, n_notbulk = 0;
    const uint4 *items;
    if (DIRECT) {
        e0 = 0;
        e1 = s_run[52];
        items = A.xent + (u64)w * A.xcap;
        n_rec = s_run[A.n_runs];
        n_notbulk = min(s_run[53], n_rec);  // (their pieces are among the extras: what is left is what the bucketing counts, as in k_winplan)
    } else {
        e0 = A.win_off[w];
        e1 = A.win_off[w + 1];
        items = A.entA;
    }
    const TileShare S{win_fx_bits(e1 - e0 + n_rec - n_notbulk), &s_shared, A.kk, s_pt, &s_ptover, (const u32 *)s_dirty};  // (a heavy window's helpers: the same bits, their deficits add up)
    {
        u32 i0 = e0, i1 = e1, v0 = 0, v1 = n_rec;
        if (heavy) {  // this helper's share of the window's items
            const u32 chunk = ((e1 - e0 + HEAVY_PARTS - 1u) / HEAVY_PARTS + 63u) & ~63u;
            i0 = min(e1, e0 + part * chunk);
            i1 = min(e1, i0 + chunk);
            const u32 vchunk = ((n_rec + HEAVY_PARTS - 1u) / HEAVY_PARTS + 63u) & ~63u;
            v0 = min(n_rec, part * vchunk);
            v1 = min(n_rec, v0 + vchunk);
        }
        // (wave-uniform values out of LDS and memory: moved to scalar registers by hand, the compiler cannot know)
        const u32 c0 = (u32)__builtin_amdgcn_readfirstlane((int)s_c0);
        RecMap M{s_run, s_run + 32, A.n_runs, (u32)__builtin_amdgcn_readfirstlane((int)v0),
                 (u32)__builtin_amdgcn_readfirstlane((int)(v1 - v0)), w, s_c0 == s_c1, c0, 0, 0, 0};
        if (DIRECT) {
            M.c_lo = (u64)(u32)__builtin_amdgcn_readfirstlane((int)s_run[48]) | ((u64)(u32)__builtin_amdgcn_readfirstlane((int)s_run[49]) << 32);
            M.clen = (u64)(u32)__builtin_amdgcn_readfirstlane((int)s_run[50]) | ((u64)(u32)__builtin_amdgcn_readfirstlane((int)s_run[51]) << 32);
            M.w_in_contig = (u32)(w0 - M.c_lo);
        }
        constexpr u32 WAVES = TILE_THREADS / 64;
        if constexpr (BULK) {
            // The window's work as PASSES of 64, dealt to the waves one by one (DirectBulk): the passes over its mirror entries,
            // then its extras [i0, i1) in the same pipeline.  Every pass but the last of either kind is full: at 50x a window
            // is 11 + 1 passes where a sixteenth of one list per wave made 16 of them, two thirds full.  The extras' SLOW items
            // are only listed there (the list lives behind the tables in s_dirty's space) and tallied in a round of their own
            // below, a sixteenth per wave, through tile_items: k_prepg's pieces sit at the tail of a window's extras, hundreds
            // of them in a window whose reads cross two planted indels, a memory round trip each.
#ifdef PP_EXP_STAGE_LATE
            D.stage(A, lane);  // (experiment: behind the barrier)
#endif
            D.run(A, cnt, s_ndbits, S, asm_w, asm4, M, s_slow, &s_nslow, SLOW_CAP, lane);
            __syncthreads();
            const u32 n_slow = s_nslow;
            if (n_slow > SLOW_CAP) {  // more slow items than the list holds: every wave looks through a sixteenth of the extras for them
                const u32 n = i1 - i0;
                if (wave < n)
                    tile_items<GW, P4, false>(A, cnt, s_ndbits, S, asm_w, asm4, items, M, i0, i1,
                                              ItemList{i0 + wave, (n - wave - 1u) / WAVES + 1u, WAVES, nullptr, nullptr, nullptr, 0u, true}, wave, lane);
            } else if (wave < n_slow)  // (every sixteenth of the list, one batch of at most SLOW_CAP / 16 <= 64 items per wave)
                tile_items<GW, P4, false>(A, cnt, s_ndbits, S, asm_w, asm4, items, M, i0, i1,
                                          ItemList{wave, (n_slow - wave - 1u) / WAVES + 1u, WAVES, s_slow, nullptr, nullptr, 0u, true}, wave, lane);
        } else {
            constexpr u32 IPP = (P4 && GW == 5) ? 1u : PlainCfg<GW>::IPP;  // (as tile_items: whole passes of the plain class)
            tile_items<GW, P4, DIRECT>(A, cnt, s_ndbits, S, asm_w, asm4, items, M, i0, i1, wave_slice<IPP>(i0, i1, wave), wave, lane);
        }
    }
    const u32 n_items = e1 - e0 + n_rec - n_notbulk;
    if (!DIRECT && n_items >= MAX_BUCKET && tid == 0 && part == 0) report(A.status, w, DE_TOO_DEEP);  // (DIRECT: k_winplan)
    __syncthreads();
    PP_STOP_AFTER(2)
#ifdef PP_TILE_STAMPS
    if (tid == 0) A.stamps[8ull * blockIdx.x + 1] = wall_clock64();
#endif

    if (heavy) {
        // The helpers of one window meet in its slab: every part stores its partial tallies in its own region and takes
        // a ticket; the last one to arrive adds all parts up -- integers, the order does not matter; the deficit row
        // carries a flag in bit 31, which is OR-ed -- and goes on to vote like the workgroup of an ordinary window.
        // The partials travel as agent-scope atomic stores and loads (write-through / L2-bypassing dwords), ordered
        // against the ticket by waiting for the stores' acknowledgements: a __threadfence() here means a write-back
        // and an invalidation of the XCD's whole L2 per helper, which slowed every block of the chip's first round to
        // half its speed (measured: k_tile 0.49 -> 0.64 ms on BASELINE configs[2]).
        // Memory-model note: relaxed agent-scope operations on different addresses are unordered in the HIP/LLVM model;
        // what orders them here is the gfx942/gfx950 implementation of agent scope (LLVM AMDGPU "memory model gfx942":
        // an agent-scope atomic store is a write-through sc1 store, complete when vmcnt says so; an agent-scope atomic
        // load bypasses the non-coherent L2 lines) -- i.e. this is the hardware's release sequence minus the L2
        // write-back, which has nothing to write back because every datum travels as an sc1 atomic.  The guard below
        // keeps the code from being built for anything else; test_heavy_windows_* stress the rendezvous.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__)
#error "k_tile's heavy-window rendezvous relies on gfx942/gfx950 agent-scope store/load semantics"
#endif
        u32 *mine = A.hslab + ((u64)hslot * HEAVY_PARTS + part) * HSLAB_WORDS;
        for (u32 i = tid; i < (u32)(N_ROWS * TILE); i += TILE_THREADS)
            __hip_atomic_store(&mine[i], cnt[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid < (u32)(TILE / 32))
            __hip_atomic_store(&mine[N_ROWS * TILE + tid], s_ndbits[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_s_waitcnt(0);  // vmcnt(0): this lane's stores have been acknowledged
        __syncthreads();
        if (tid == 0)
            s_ticket = __hip_atomic_fetch_add(&A.heavy[1 + HEAVY_SLOTS + hslot], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
#ifdef PP_TILE_STAMPS
        if (tid == 0) A.stamps[8ull * blockIdx.x + 2] = wall_clock64();
#endif
        if (s_ticket != HEAVY_PARTS - 1u) return;
        const u32 *base = A.hslab + (u64)hslot * HEAVY_PARTS * HSLAB_WORDS;
        for (u32 i = tid; i < (u32)(N_ROWS * TILE); i += TILE_THREADS) {
            u32 o[HEAVY_PARTS];
#pragma unroll
            for (u32 q = 0; q < HEAVY_PARTS; q++)  // all parts' loads in flight (this part's own tallies are there as well)
                o[q] = __hip_atomic_load(&base[(u64)q * HSLAB_WORDS + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            u32 sum = 0;
#pragma unroll
            for (u32 q = 0; q < HEAVY_PARTS; q++) sum += o[q];  // (the deficit row holds differences: they add up like counts)
            cnt[i] = sum;
            if (sum && i / (u32)TILE == (u32)ROW_DEF) s_shared = 1u;  // some part saw a depth share other than 1
        }
        if (tid < (u32)(TILE / 32)) {
            u32 bits = 0;
#pragma unroll
            for (u32 q = 0; q < HEAVY_PARTS; q++)
                bits |= __hip_atomic_load(&base[(u64)q * HSLAB_WORDS + N_ROWS * TILE + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_ndbits[tid] = bits;
        }
        __syncthreads();
    }

    // ---- coverage of the fast class, and the depth deficits: prefix sums of the two difference arrays, in place ----
    auto scan_row = [&](u32 *row) {
        const u32 d0 = row[2 * tid], d1 = row[2 * tid + 1];
        const u32 sum = d0 + d1;
        const u32 inc = wave_scan_incl(sum);
        if (lane == 63) s_wsum[wave] = inc;
        __syncthreads();
        // the waves in front: their totals (one per lane), summed
        const u32 base = wave_sum_dpp(lane < wave ? s_wsum[lane & 15u] : 0u);
        const u32 ex = base + inc - sum;
        row[2 * tid] = ex + d0;
        row[2 * tid + 1] = ex + d0 + d1;
        __syncthreads();
    };
    scan_row(cnt + ROW_COV * TILE);
    if (s_shared) scan_row(cnt + ROW_DEF * TILE);  // (uniform: written before the barrier in front of this)
    PP_STOP_AFTER(3)

    // ---- vote ----
    // Pass 1, one lane per position: a position where NOTHING was tallied explicitly -- every read that covers it shows
    // the assembly's own A/C/G/T there (no mismatch, no indel, no shared or odd depth share) -- needs no vote: the
    // assembly's base is the only key with a count, its count is the depth, and whatever pileup.rs:67-134 makes of that
    // (kept, low depth, or with a zero invalid threshold "too close" / "multiple") the base stays.  That is two thirds of
    // the positions at 200x and nine tenths at 50x.  The others are listed in LDS and voted in pass 2, densely: the vote
    // proper is ~200 instructions with two f64 multiplies per position, and a wave runs it for all of its 64 lanes or
    // for none.
    u32 my_len = 0, my_changed = 0, my_zero = 0;
    // (depths: whole reads are counted in 32 bits -- a lane sees two positions of at most 2^21 items each --, only the positions of
    // a window with shared reads carry fractions: 64-bit sums, and their 64-bit wave total, only there)
    u32 my_cov = 0;
    u64 my_dfx = 0;
    const bool win_shared = s_shared != 0;
    const bool one_contig = (s_c0 == s_c1);
    for (u32 p0 = tid; p0 < (u32)TILE; p0 += TILE_THREADS) {
        const u64 gp = w0 + p0;
        bool dirty = false;
        if (gp < A.G) {
            bool mine = true;
            if (A.own) {  // window tiling: halo positions are voted by the rank that owns them
                const u32 c = one_contig ? s_c0 : find_contig(A.contig_off, A.n_contigs, gp);
                const u32 rel = (u32)(gp - A.contig_off[c]);
                if (rel < A.own[2 * c] || rel >= A.own[2 * c + 1]) {
                    A.code[gp] = 0;
                    mine = false;
                }
            }
            if (mine) {
                const u8 orig = ((const u8 *)asm_w)[ASM_PAD + p0];
                const u32 any = cnt[ROW_A * TILE + p0] | cnt[ROW_C * TILE + p0] | cnt[ROW_T * TILE + p0] | cnt[ROW_G * TILE + p0] |
                                cnt[ROW_DEL * TILE + p0] | cnt[ROW_OTH * TILE + p0] | cnt[ROW_DEF * TILE + p0] | cnt[ROW_MIS * TILE + p0] |
                                ((s_ndbits[p0 >> 5] >> (p0 & 31u)) & 1u);
                if (any == 0 && row_of(orig) < 4 && one_contig && !A.dbg) {
                    const u32 cov = cnt[ROW_COV * TILE + p0];
                    A.code[gp] = orig;
                    my_len += 1u;
                    my_zero += cov == 0;
                    my_cov += cov;
                } else dirty = true;
            }
        }
        const u64 dm = __ballot(dirty);
        if (dm) {
            u32 base = 0;
            if (lane == 0) base = atomicAdd(&s_ndirty, (u32)__popcll(dm));
            base = (u32)__builtin_amdgcn_readfirstlane((int)base);
            if (dirty) s_dirty[base + (u32)__popcll(dm & ((1ull << lane) - 1ull))] = (unsigned short)p0;
        }
    }
    __syncthreads();
    PP_STOP_AFTER(4)
#ifdef PP_TILE_STAMPS
    if (tid == 0) A.stamps[8ull * blockIdx.x + 7] = wall_clock64();
#endif
    // Pass 2: the vote proper (pileup.rs:67-134) for the listed positions
    const u32 n_dirty = s_ndirty;
    for (u32 di = tid; di < n_dirty; di += TILE_THREADS) {
        const u32 p = s_dirty[di];
        const u64 gp = w0 + p;
        u32 nA, nC, nG, nT, nDel, nOth;
        const u32 deficit = cnt[ROW_DEF * TILE + p];  // in units of 2^-S.b (prefix-summed above)
        const u8 orig = ((const u8 *)asm_w)[ASM_PAD + p];
        position_tallies(cnt, orig, p, nA, nC, nG, nT, nDel, nOth);
        const bool nd = ((s_ndbits[p >> 5] >> (p & 31u)) & 1u) != 0;  // some share here is not a multiple of the unit
        const u32 ntot = nA + nC + nG + nT + nDel + nOth;
        if (orig >= 0x80u) report(A.status, gp, DE_NON_ASCII);
        const u64 dfx_b = ((u64)ntot << S.b) - deficit;
        const double depth = (double)dfx_b * (1.0 / (double)(1u << S.b));  // exact unless nd
        const u64 dfx = dfx_b << ((u32)DEPTH_FX_BITS - S.b);               // the same in the statistics' unit
        bool flag = false, for_keys = false, decided = true, low;
        u32 vthr, ithr;
        VoteOut v;
        v.out = (orig == (u8)'-') ? 0 : orig;
        v.status = PP_ST_LOW_DEPTH;
        v.vthr = 0; v.ithr = 0;
        if (nd) {
            // The depth is an order-dependent f64 sum (pileup.rs:64: depth += 1.0 / k, in file order) that the fixed-point
            // tally only BOUNDS: every inexact share is off by at most half a unit, the f64 sum itself by < 1e-9.  The vote
            // looks at the depth through three monotone step functions -- bankers_rounding(depth * fraction_valid),
            // bankers_rounding(depth * fraction_invalid), depth < min_depth (pileup.rs:70-72,114) -- so where both ends of
            // the interval give the same three values every depth inside it does, and the position is decided here with
            // those.  Where ONE of the three differs between the ends, by one step, the exact depth gives one of the two
            // sets of values: if the vote comes out the same with both (no tally sits exactly on the threshold that
            // moved -- shares like 1/3 put depth * fraction on an exact .5 at one position in six, where the order of the
            // additions decides the rounding, but a tally on that very threshold is rare) it is decided as well.  Only
            // the rest (a few positions in 10^4), and wherever the depth itself is printed (--debug), is replayed in file
            // order.  (debug level 3 records the thresholds it voted with: there the strict rule only.)
            if (A.dbg == 1 || A.dbg == 2) decided = false;
            else {
                const double eps = (double)ntot * (0.5 / (double)(1u << S.b)) + 1e-9;
                const double lo = fmax(depth - eps, 0.0), hi = depth + eps;
                const u32 v0 = d_bankers(__dmul_rn(lo, A.fv)), v1 = d_bankers(__dmul_rn(hi, A.fv));
                const u32 i0 = d_bankers(__dmul_rn(lo, A.fi)), i1 = d_bankers(__dmul_rn(hi, A.fi));
                const bool l0 = lo < (double)A.min_depth, l1 = hi < (double)A.min_depth;
                const u32 ndiff = (u32)(v0 != v1) + (u32)(i0 != i1) + (u32)(l0 != l1);
                vthr = max(A.min_depth, v0); ithr = i0; low = l0;
                decided = ndiff == 0;
                if (ndiff == 1 && A.dbg == 0 && v1 - v0 <= 1u && i1 - i0 <= 1u) {
                    const bool keys0 = !l0 && nOth > 0 && nOth >= i0, keys1 = !l1 && nOth > 0 && nOth >= i1;
                    if (!keys0 && !keys1) {
                        const VoteOut va = vote5_thr(nA, nC, nG, nT, nDel, vthr, ithr, low, orig);
                        const VoteOut vb = vote5_thr(nA, nC, nG, nT, nDel, max(A.min_depth, v1), i1, l1, orig);
                        decided = va.out == vb.out && va.status == vb.status;
                    }
                }
            }
            // (depth <= ntot always, so ntot < min_depth would decide DepthTooLow -- but the depth itself feeds the contig's
            // mean read depth, polish.rs:173-180: an undecided position is replayed whenever anything covers it)
            if (!decided && (ntot > 0 || A.dbg)) flag = true;
        } else if (deficit == 0 && ntot < VOTE_TAB_N) {
            // an integer depth (no shared read here): both thresholds from the job's table -- the same two multiplications
            // and roundings, done once per depth instead of once per position
            const uint2 th = ((const uint2 *)A.vote_tab)[ntot];
            vthr = max(A.min_depth, th.x);
            ithr = th.y;
            low = ntot < A.min_depth;
        } else {
            vthr = max(A.min_depth, d_bankers(__dmul_rn(depth, A.fv)));
            ithr = d_bankers(__dmul_rn(depth, A.fi));
            low = depth < (double)A.min_depth;
        }
        u32 multi_eff = 0;   // > 0: a two-byte key won and leaves this many bytes (polish.rs:188 drops '-')
        u64 multi_off = 0;
        if (decided) {
            if (!low && nOth > 0 && nOth >= ithr) {
                // A string-keyed tally could reach a threshold.  If the window's table of two-byte keys accounts for the whole
                // string-keyed row of this position, the vote of pileup.rs:77-134 runs here over A C G T, "-" and those keys
                // (when exactly one key is valid it wins whatever the order of the keys, otherwise nothing changes);
                // else the position is listed for k_exact.
                KeyVote kv{0u, 0u, 0u, 0u, 0ull};
                if (!s_ptover && A.dbg != 1 && A.dbg != 2) kv = vote_with_keys(s_pt, p, nA, nC, nG, nT, nDel, nOth, vthr, ithr, (u32)orig);
                if (kv.handled) {
                    v.vthr = vthr; v.ithr = ithr;
                    v.out = (u8)kv.out;
                    v.status = (u8)kv.status;
                    multi_eff = kv.multi_eff;
                    multi_off = kv.multi_off;
                } else {
                    flag = true;
                    for_keys = true;
                }
            } else v = vote5_thr(nA, nC, nG, nT, nDel, vthr, ithr, low, orig);
        }
        if ((A.dbg == 1 || A.dbg == 2) && nDel + nOth > 0) flag = true;  // --debug lists every key: k_exact writes the records
        if (flag) {
            // dbg 2: test hook, see run_pipeline; a listed heavy window is replayed by k_exact2's sub-range instance.
            // A position that is only here for its string-keyed tallies (an insertion that may win: its depth is exact
            // already) goes straight to the list of k_exact's wave-per-position replay -- k_exact2 would sort the whole
            // window for the ordered depth it does not need, and then hand it over all the same.
            const bool to_list = A.dbg == 1 || (n_items > SORT_MAX && !heavy) || !nd || for_keys;
            if (DIRECT) s_need = 1u;
            if (!to_list) {
                atomicOr(&s_fbits[p >> 5], 1u << (p & 31u));
                atomicAdd(&s_nflag, 1u);
            } else {
                atomicAdd(&A.counters[2], 1u);
            }
            if (to_list) {
                // bucket too large for the wave-per-position replay: global list for k_exact
                const u32 slot = atomicAdd(&A.counters[0], 1u);
                const u64 scr_at = atomicAdd(A.scr_need, (u64)ntot);  // its stretch of the replay scratch
                if (slot < A.cap_flag) {
                    A.flag_pos[slot] = (u32)gp;
                    A.flag_cov[slot] = ntot;
                    A.flag_scr[slot] = scr_at;
                } else {
                    report(A.status, slot, DE_CAPACITY_LATE);
                }
            }
            A.code[gp] = 0;
            continue;
        }
        u32 l = v.out ? 1u : 0u;
        if (multi_eff) {  // the winner has two bytes: k_emit takes them from the seq array (as for k_exact's multi-byte winners)
            l = multi_eff;
            A.code[gp] = (u8)(0x80u | multi_eff);
            const u32 slot = atomicAdd(&A.counters[1], 1u);
            if (slot < A.cap_multi) {
                MultiEnt m;
                m.off = multi_off; m.pos = (u32)gp; m.len = 2u; m.eff = multi_eff; m.pad = 0;
                A.multi[slot] = m;
            } else {
                report(A.status, slot, DE_CAPACITY_LATE);
            }
        } else A.code[gp] = v.out;
        const u32 ch = (v.status == PP_ST_CHANGED), z = (ntot == 0);
        if (one_contig) {
            my_len += l; my_changed += ch; my_zero += z;
            if (win_shared) my_dfx += dfx; else my_cov += ntot;  // (no shared read in the window: dfx = ntot << DEPTH_FX_BITS)
        } else {
            my_len += l;
            const u32 c = find_contig(A.contig_off, A.n_contigs, gp);
            if (ch) atomicAdd(&A.stats[c].changed, 1ull);
            if (z) atomicAdd(&A.stats[c].zero_depth, 1ull);
            if (dfx) atomicAdd(&A.stats[c].depth_fx, dfx);
        }
        if (A.dbg) {
            A.dbg_depth[gp] = depth;
            A.dbg_counts[0 * A.G + gp] = nA;
            A.dbg_counts[1 * A.G + gp] = nC;
            A.dbg_counts[2 * A.G + gp] = nG;
            A.dbg_counts[3 * A.G + gp] = nT;
            A.dbg_counts[4 * A.G + gp] = nDel + nOth;
            A.dbg_counts[5 * A.G + gp] = v.vthr;
            A.dbg_counts[6 * A.G + gp] = v.ithr;
            A.dbg_status[gp] = v.status;
        }
    }
    PP_STOP_AFTER(5)
    my_len = wave_sum_dpp(my_len);
    my_changed = wave_sum_dpp(my_changed);
    my_zero = wave_sum_dpp(my_zero);
    u64 my_depth = (u64)wave_sum_dpp(my_cov) << DEPTH_FX_BITS;
    if (win_shared) my_depth += wave_sum64(my_dfx);
    if (lane == 0) {
        if (my_len) atomicAdd(&s_len, my_len);
        if (my_changed) atomicAdd(&s_changed, my_changed);
        if (my_zero) atomicAdd(&s_zero, my_zero);
        if (my_depth) atomicAdd(&s_depth, my_depth);
    }
    __syncthreads();
    // A handful of flagged positions (since the interval vote settles nearly everything, a window rarely has more than one
    // or two left) is not worth the window's ordered replay -- k_exact2 sorts ALL of the window's items for them: they go
    // to the global list instead, where k_exact replays each with a workgroup of its own (its depth in file order too).
    if (s_nflag && s_nflag <= FEW_FLAGGED && A.dbg != 2 && !heavy) {
        if (tid < (u32)(TILE / 32)) {
            u32 bits = s_fbits[tid];
            while (bits) {
                const u32 p = 32u * tid + (u32)__ffs((int)bits) - 1u;
                bits &= bits - 1u;
                u32 nA, nC, nG, nT, nDel, nOth;
                position_tallies(cnt, ((const u8 *)asm_w)[ASM_PAD + p], p, nA, nC, nG, nT, nDel, nOth);
                const u32 ntot = nA + nC + nG + nT + nDel + nOth;
                const u32 slot = atomicAdd(&A.counters[0], 1u);
                const u64 scr_at = atomicAdd(A.scr_need, (u64)ntot);
                if (slot < A.cap_flag) {
                    A.flag_pos[slot] = (u32)(w0 + p);
                    A.flag_cov[slot] = ntot;
                    A.flag_scr[slot] = scr_at;
                } else {
                    report(A.status, slot, DE_CAPACITY_LATE);
                }
            }
            s_fbits[tid] = 0;
        }
        __syncthreads();
        if (tid == 0) { atomicAdd(&A.counters[2], s_nflag); s_nflag = 0; }
        __syncthreads();
    }
    if (tid < (u32)(TILE / 32)) A.flag_bits[(u64)w * (TILE / 32) + tid] = s_fbits[tid];
    if (s_nflag && (n_items <= SORT_MAX || heavy)) {
        // the ordered-depth replay needs this window's integer tallies: save them (rare windows only)
        if (tid == 0) {
            const u32 slab = atomicAdd(&A.counters[3], 1u);
            if (slab >= A.cap_slabs) report(A.status, slab, DE_CAPACITY_LATE);
            s_c1 = slab;
            A.win_slab[w] = slab;
            if (slab < A.cap_slabs) A.slab_win[slab] = w;
        }
        __syncthreads();
        const u32 slab = s_c1;
        if (slab < A.cap_slabs) {
            u32 *dst = A.slabs + (u64)slab * 6u * TILE;
            for (u32 p = tid; p < (u32)TILE; p += TILE_THREADS) {
                u32 nA, nC, nG, nT, nDel, nOth;
                position_tallies(cnt, ((const u8 *)asm_w)[ASM_PAD + p], p, nA, nC, nG, nT, nDel, nOth);
                dst[0 * TILE + p] = nA; dst[1 * TILE + p] = nC; dst[2 * TILE + p] = nG;
                dst[3 * TILE + p] = nT; dst[4 * TILE + p] = nDel; dst[5 * TILE + p] = nOth;
            }
        }
    }
#ifdef PP_TILE_STAMPS
    if (tid == 0) A.stamps[8ull * blockIdx.x + 2] = wall_clock64();
#endif
    if (tid == 0) {
        if (DIRECT && s_need) A.need_win[atomicAdd(A.n_need, 1ull)] = w;  // (every window at most once: room for all of them)
        A.win_nflag[w] = s_nflag;
        if (s_nflag) atomicAdd(&A.counters[2], s_nflag);
        A.win_len[w] = s_len;
        if (s_len) note_out_len(A.win_coarse, A.win_coarse2, w, s_len);
        if (s_changed) atomicAdd(&A.stats[s_c0].changed, (u64)s_changed);
        if (s_zero) atomicAdd(&A.stats[s_c0].zero_depth, (u64)s_zero);
        if (s_depth) atomicAdd(&A.stats[s_c0].depth_fx, s_depth);
    }
}

// The arguments are read where they are used, from the kernel-argument segment itself (scalar loads): taken by value
// the ~45 fields are all loaded at entry, as 16-dword tuples that the register allocator can only spill whole -- 222
// SGPR spills, and a v_readlane per spilled dword in front of every use: a fifth of the item loop's VALU issue.
template <int GW, bool P4>
__global__ __launch_bounds__(TILE_THREADS, 8) void k_tile(TileArgs A_in_kernarg) {
#if PP_TILE_LAZY_ARGS
    tile_window<false, GW, P4>(*(const TileArgs *)__builtin_amdgcn_kernarg_segment_ptr());
#else
    tile_window<false, GW, P4>(A_in_kernarg);
#endif
}
template <int GW, bool P4>
__global__ __launch_bounds__(TILE_THREADS, 8) void k_tile_direct(TileArgs A_in_kernarg) {
#if PP_TILE_LAZY_ARGS
    tile_window<true, GW, P4>(*(const TileArgs *)__builtin_amdgcn_kernarg_segment_ptr());
#else
    tile_window<true, GW, P4>(A_in_kernarg);
#endif
}

}  // namespace pp
