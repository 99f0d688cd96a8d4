// pp_k_exact.h -- k_exact / k_exact2: exact replay of the flagged positions (string-keyed tallies, ordered f64 depth).
// Part of pp_kernels.hip (included there, in this order, and nowhere else: it defines __global__ kernels).
#pragma once

namespace pp {

// =============================================================================================
// k_exact: exact replay of flagged positions (one thread per position)
// =============================================================================================
// Read slice (offset relative to the read, length) of entry q of an alignment with indels:
// get_read_bases_for_each_target_base, alignment.rs:175-201.
__device__ void entry_slice(const u32 *cg, u32 nc, u32 q, u64 *s_rel, u32 *len) {
    u32 ent = 0;
    u64 ro = 0;
    for (u32 r = 0; r < nc; r++) {
        u32 op = cg[r], l = op >> 4, o = op & 15u;
        if (o == PP_OP_I) { ro += l; continue; }
        if (q < ent + l) {
            u32 ins = 0;
            if (q == ent + l - 1)
                for (u32 r2 = r + 1; r2 < nc && (cg[r2] & 15u) == PP_OP_I; r2++) ins += cg[r2] >> 4;
            if (o == PP_OP_D) { *s_rel = ro; *len = ins; }
            else { *s_rel = ro + (q - ent); *len = 1u + ins; }
            return;
        }
        ent += l;
        if (o != PP_OP_D) ro += l;
    }
    *s_rel = 0;
    *len = 0;
}

__device__ void sift_down(ulonglong2 *a, u32 start, u32 n) {
    u32 root = start;
    for (;;) {
        u32 child = 2 * root + 1;
        if (child >= n) break;
        if (child + 1 < n && a[child].x < a[child + 1].x) child++;
        if (a[root].x >= a[child].x) break;
        ulonglong2 t = a[root]; a[root] = a[child]; a[child] = t;
        root = child;
    }
}
__device__ void heapsort_by_x(ulonglong2 *a, u32 n) {
    if (n < 2) return;
    for (u32 s = n / 2; s-- > 0;) sift_down(a, s, n);
    for (u32 end = n - 1; end > 0; end--) {
        ulonglong2 t = a[0]; a[0] = a[end]; a[end] = t;
        sift_down(a, 0, end);
    }
}

struct ExactArgs {
    u32 cap_multi;
    u32 cap_flag;
    u32 *flag_pos_w;         // global replay list (k_exact2 appends key-table overflows)
    u32 *flag_cov_w;
    u64 *flag_scr_w;
    u64 *scr_need;       // replay scratch handed out so far (a listed position takes its stretch as it is listed)
    u64 cap_scr;
    const u32 *flag_bits;
    const u32 *win_nflag;
    const u32 *win_slab;
    const u32 *slab_win;  // the window of each tally slab: the list of the windows with flagged positions
    u32 cap_slabs;
    const u32 *slabs;
    KeyRec *keys;       // debug only
    u64 cap_keys;
    u64 *n_keys;
    ulonglong2 *ents;   // per replayed window: (start | extent << 32, 1/k as f64 bits) in file order
    u64 cap_ents;
    u64 *ents_cursor;
    const u32 *flag_pos;
    const u32 *flag_cov;
    const u64 *flag_scr;
    const uint4 *entA;
    const u32 *win_lo, *win_hi;  // the items of window w: entA[win_lo[w] .. win_hi[w]) -- the bucketing's offsets of w and w + 1, or (direct path) what k_xmat wrote out for the windows k_tile listed
    const u8 *seq;
    const u64 *seq_off;
    const u64 *cig_off;
    const u32 *n_cig;
    const u32 *cigar;
    const u32 *kk;
    const u8 *bases;
    u64 G;
    const u64 *contig_off;
    u32 n_contigs;
    u32 min_depth;
    double fv, fi;
    ulonglong2 *scratch;
    u8 *code;
    u32 *win_len, *win_coarse, *win_coarse2;
    u32 *counters;
    MultiEnt *multi;
    ContigStatsDev *stats;
    double *dbg_depth;
    u32 *dbg_counts;
    u8 *dbg_status;
    u64 *status;
    int dbg;
    u64 seq_bytes;        // size of the seq array (unaligned word loads stay inside it)
    const u32 *heavy;     // list of the heavy windows (HEAVY_WORDS) ...
    const u8 *win_heavy;  // ... and, per window, 0 or 1 + its slot in it
};

constexpr u64 SL_OFF_MASK = (1ull << 40) - 1;
constexpr u64 SL_DONE = 1ull << 63;

__device__ void exact_one(const ExactArgs &A, u32 f);
__device__ void exact_block(const ExactArgs &A, u32 f, u32 cap, ulonglong2 *cov, double *rcp, u32 *kw, u32 *s_n);

// One WORKGROUP per listed position (a position whose string-keyed tallies -- an insertion, N ... -- could reach a
// threshold, every flagged position of a window too deep for k_exact2, and with --debug everything that has such a key):
// all its threads scan the window's work items together, then its first wave sorts the covering alignments by file
// index in LDS, tallies and groups the keys.  An assembly that lacks a base makes every read over that spot vote for a
// two-byte key (src/alignment.rs:175-201: an I run extends the entry before it; src/pileup.rs:56-63 counts it by
// string), so these positions are what polishing is about -- one THREAD per position, walking the window's ~3,000 items
// and heap-sorting 200 of them in global memory, took 10 ms per job at 200x; one wave per position 0.2 ms (the scan is
// a chain of 45 round trips); the workgroup's scan is six.  Positions covered by more than EXW_MAX alignments keep
// the thread-serial path.
constexpr u32 EXW_MAX = 1024;
constexpr u32 EXW_THREADS = 512;
constexpr u32 EXW_BLOCKS = 4096;

// one wave working on LDS by itself: its DS operations execute in program order, the compiler must keep them there
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__global__ __launch_bounds__(EXW_THREADS) void k_exact(ExactArgs A) {
    __shared__ ulonglong2 cov[EXW_MAX];
    __shared__ double rcp[EXW_MAX];
    __shared__ u32 kw[EXW_MAX];  // the first four bytes of every entry's key
    __shared__ u32 s_n;
    if (*A.status != ~0ull) return;
    if (*A.scr_need > A.cap_scr) {  // the scratch is too small: the host grows it and reruns
        if (blockIdx.x == 0 && threadIdx.x == 0) report(A.status, *A.scr_need, DE_CAPACITY);
        return;
    }
    const u32 n_flagged = A.counters[0];
    for (u32 f = blockIdx.x; f < n_flagged; f += gridDim.x) {
        const u32 cap = A.flag_cov[f];
        if (cap <= EXW_MAX) exact_block(A, f, cap, cov, rcp, kw, &s_n);
        else if (threadIdx.x == 0) exact_one(A, f);
        __syncthreads();
    }
}

// (the kernel is a chain of dependent memory round trips: whatever is known early -- the assembly's base, the position's
// contig, an item's share next to its trim bytes, a key's bytes next to its row -- is asked for early)
__device__ void exact_block(const ExactArgs &A, u32 f, u32 cap, ulonglong2 *cov, double *rcp, u32 *kw, u32 *s_n) {
    const u32 tid = threadIdx.x, lane = tid & 63u;
    const u32 gp = A.flag_pos[f];
    const u32 w = gp / (u32)TILE;
    const int pr = (int)(gp - w * (u32)TILE);
    const u32 e0 = A.win_lo[w], e1 = A.win_hi[w];
    const u8 orig = A.bases[gp];
    const u32 c = find_contig_wave(A.contig_off, A.n_contigs, gp, lane);
    // ---- the covering alignments: x = (file index << 32 | k), y = slice (offset | len << 40); any order, sorted below ----
    if (tid == 0) *s_n = 0;
    __syncthreads();
    constexpr u32 EXW_UNROLL = 4;  // items a thread asks for before it looks at the first (their loads are in flight together)
    for (u32 eb = e0 + tid; eb < e1; eb += EXW_UNROLL * EXW_THREADS) {
        uint4 ents[EXW_UNROLL];
#pragma unroll
        for (u32 u = 0; u < EXW_UNROLL; u++) ents[u] = A.entA[min(eb + u * EXW_THREADS, e1 - 1u)];
#pragma unroll
        for (u32 u = 0; u < EXW_UNROLL; u++) {
            if (eb + u * EXW_THREADS >= e1) break;
            const uint4 ent = ents[u];
            const int q = pr - item_rel(ent.z);
            const u32 fl = (ent.y >> 16) & 0xFFu, idx = ent.w;
            const bool point = (ent.z >> 31) != 0, notrim = ((ent.z >> 30) & 1u) != 0;
            if (q < 0 || q >= (int)item_extent(ent.x, ent.y, ent.z)) continue;
            const u64 so = fl ? A.seq_off[idx] : ((u64)ent.x | ((u64)(ent.y & 0xFFu) << 32));
            const u32 kq = A.kk[idx];
            // fast-class items carry their untrimmed length: apply the trim here (from the last four bases in one load,
            // as k_tile's plain class does; a trailing homopolymer of four or more walks byte by byte)
            if (fl == 0 && !point && !notrim) {
                const u8 *rp = A.seq + so;
                const u32 L = ent.y >> 24;
                u32 tf = 0;
                if (L >= 4u) {
                    const u32 tail = load4_unaligned(rp + (L - 4u));
                    tf = nz_flags(tail ^ splat8(tail >> 24));
                }
                const u32 lim = tf ? L - 4u + (u32)((31 - __clz((int)tf)) >> 3) : simple_nkeep(rp, L);
                if ((u32)q >= lim) continue;
            }
            u64 s_rel;
            u32 len;
            if (point) { s_rel = 0; len = ent.y >> 24; }  // the entry at a read's single indel: its key bytes, or none
            else if (!(fl & ENT_COMPLEX)) { s_rel = (u64)q; len = 1; }
            else entry_slice(A.cigar + A.cig_off[idx], A.n_cig[idx], (u32)q, &s_rel, &len);
            const u32 slot = atomicAdd(s_n, 1u);
            if (slot < EXW_MAX) {
                ulonglong2 v;
                v.x = ((u64)idx << 32) | (u64)kq;
                v.y = ((so + s_rel) & SL_OFF_MASK) | ((u64)(len & 0x7FFFFFu) << 40);
                cov[slot] = v;
            }
        }
    }
    __syncthreads();
    const u32 n = *s_n;
    if (tid >= 64u) return;  // the rest is the first wave's (no workgroup barrier from here on: wave_sync)
    if (n != cap) { if (lane == 0) report(A.status, gp, DE_INTERNAL); return; }
    // ---- file order: bitonic network over the next power of two (the padding sorts to the end) ----
    u32 np2 = 2;
    while (np2 < n) np2 <<= 1;
    for (u32 i = n + lane; i < np2; i += 64) cov[i] = make_ulonglong2(~0ull, 0);
    wave_sync();
    for (u32 k = 2; k <= np2; k <<= 1)
        for (u32 lj = 31u - (u32)__clz((int)k); lj-- > 0;) {
            const u32 j = 1u << lj;
            for (u32 t = lane; t < (np2 >> 1); t += 64) {
                const u32 i = ((t >> lj) << (lj + 1u)) | (t & (j - 1u)), o = i + j;
                const bool asc = (i & k) == 0;
                const ulonglong2 x = cov[i], y = cov[o];
                if ((x.x > y.x) == asc) { cov[i] = y; cov[o] = x; }
            }
            wave_sync();
        }
    // ---- tallies (by the wave) and the depth (sequential f64 adds of 1.0/k in file order: pileup.rs:64, alignment.rs:288) ----
    u32 nA = 0, nC = 0, nG = 0, nT = 0, nDel = 0, nOth = 0;
    u64 fx = 0;      // the depth in 2^-DEPTH_FX_BITS units while every share is a power of two (exact in any order)
    bool odd = false;
    for (u32 i = lane; i < n; i += 64) {
        const u32 kq = (u32)(cov[i].x & 0xFFFFFFFFull), kc = kclass_of(kq);
        rcp[i] = 1.0 / (double)kq;
        if (kc > KCLASS_DYADIC_MAX) odd = true; else fx += (u64)(1u << DEPTH_FX_BITS) >> kc;
        const u64 y = cov[i].y;
        const u32 len = (u32)((y >> 40) & 0x7FFFFFu);
        // the key's first (up to four) bytes: one load where four bytes lie inside the seq array, else byte by byte
        u32 word = 0;
        if (len) {
            const u64 ko = y & SL_OFF_MASK;
            if (ko + 4 <= A.seq_bytes) word = load4_unaligned(A.seq + ko);
            else for (u32 b = 0; b < min(len, 4u); b++) word |= (u32)A.seq[ko + b] << (8u * b);
            if (len < 4u) word &= (1u << (8u * len)) - 1u;
        }
        kw[i] = word;
        int row = ROW_OTH;
        if (len == 0) row = ROW_DEL;
        else if (len == 1) row = row_of(word);
        if (row != ROW_OTH) {
            if (row == ROW_A) nA++; else if (row == ROW_C) nC++; else if (row == ROW_G) nG++;
            else if (row == ROW_T) nT++; else nDel++;
            cov[i].y = y | SL_DONE;
        } else nOth++;
    }
    nA = wave_sum(nA); nC = wave_sum(nC); nG = wave_sum(nG); nT = wave_sum(nT); nDel = wave_sum(nDel); nOth = wave_sum(nOth);
    wave_sync();
    double depth;
    if (__ballot(odd)) {  // some share is not a power of two: the sum depends on the order -- one lane adds them up in file order
        depth = 0.0;
        if (lane == 0)
            for (u32 i = 0; i < n; i++) depth += rcp[i];
        depth = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(depth)), __builtin_amdgcn_readfirstlane(__double2loint(depth)));
    } else {
        depth = (double)wave_sum64(fx) * (1.0 / (double)(1u << DEPTH_FX_BITS));
    }
    VoteOut v = vote5(nA, nC, nG, nT, nDel, depth, orig, A.min_depth, A.fv, A.fi);
    u64 win_off = 0;
    u32 win_len = 0;  // winning string-keyed sequence, if any
    const bool low = v.status == PP_ST_LOW_DEPTH;
    if (A.dbg && nDel > 0 && lane == 0) {  // --debug lists the deletion key like any other
        const u64 slot = atomicAdd(A.n_keys, 1ull);
        if (slot < A.cap_keys) {
            KeyRec kr;
            kr.off = 0; kr.pos = gp; kr.len = 0; kr.count = nDel; kr.pad = 0;
            A.keys[slot] = kr;
        } else report(A.status, slot, DE_CAPACITY);
    }
    if (nOth > 0 && (!low || A.dbg)) {
        // redo the tally of pileup.rs:77-109 with the remaining keys added
        int nv = 0, ni = 0;
        u8 win = 0;
        const u32 c5[5] = {nA, nC, nG, nT, nDel};
        const u8 k5[5] = {'A', 'C', 'G', 'T', '-'};
        for (int j = 0; j < 5; j++) {
            if (j == 4 && nDel == 0) break;
            if (c5[j] >= v.vthr) { if (!nv) win = k5[j]; nv++; } else if (c5[j] >= v.ithr) ni++;
        }
        // one distinct key after the other, in the order of their first appearance: the lanes compare theirs with it.
        // The next key is the first entry that is still open: every lane looks through its own entries (a few), the
        // wave takes the minimum -- walking all n entries one broadcast read at a time was a third of the kernel.
        u32 from = 0;
        for (;;) {
            u32 mine_first = 0xFFFFFFFFu;
            for (u32 j = from + lane; j < n; j += 64)
                if (!(cov[j].y & SL_DONE)) { mine_first = j; break; }
            for (int o = 32; o > 0; o >>= 1) mine_first = min(mine_first, (u32)__shfl_xor((int)mine_first, o, 64));
            if (mine_first == 0xFFFFFFFFu) break;
            const u32 i = mine_first;
            from = i + 1;
            const u64 yi = cov[i].y;  // (the same word for every lane: one broadcast read)
            const u32 li = (u32)((yi >> 40) & 0x7FFFFFu);
            const u8 *si = A.seq + (yi & SL_OFF_MASK);
            const u32 wi = kw[i];
            u32 mine = 0;
            for (u32 j = i + 1 + lane; j < n; j += 64) {
                const u64 yj = cov[j].y;
                if ((yj & SL_DONE) || (u32)((yj >> 40) & 0x7FFFFFu) != li || kw[j] != wi) continue;
                bool same = true;
                if (li > 4u) {  // (keys of up to four bytes are compared in their words: the usual case)
                    const u8 *sj = A.seq + (yj & SL_OFF_MASK);
                    for (u32 b = 4; b < li; b++) if (si[b] != sj[b]) { same = false; break; }
                }
                if (same) { mine++; cov[j].y = yj | SL_DONE; }
            }
            const u32 count = 1u + wave_sum(mine);
            wave_sync();
            if (count >= v.vthr) { if (!nv) { win = 0; win_off = yi & SL_OFF_MASK; win_len = li; } nv++; }
            else if (count >= v.ithr) ni++;
            if (A.dbg && lane == 0) {
                const u64 slot = atomicAdd(A.n_keys, 1ull);
                if (slot < A.cap_keys) {
                    KeyRec kr;
                    kr.off = yi & SL_OFF_MASK; kr.pos = gp; kr.len = li; kr.count = count; kr.pad = 0;
                    A.keys[slot] = kr;
                } else report(A.status, slot, DE_CAPACITY);
            }
        }
        if (low) {
            win_len = 0;  // the keys were only walked for the --debug records
        } else {
            v.out = (orig == (u8)'-') ? 0 : orig;
            v.status = PP_ST_KEPT;
            if (nv == 1) {
                if (ni > 0) v.status = PP_ST_TOO_CLOSE;
                else if (win_len == 0) {
                    v.out = (win == (u8)'-') ? 0 : win;
                    if (win != orig) v.status = PP_ST_CHANGED;
                } else {
                    v.status = (win_len == 1 && A.seq[win_off] == orig) ? PP_ST_KEPT : PP_ST_CHANGED;
                }
            } else {
                win_len = 0;
                v.status = (nv == 0) ? PP_ST_NONE : PP_ST_MULTIPLE;
            }
            if (v.status == PP_ST_TOO_CLOSE) win_len = 0;
        }
    }
    if (lane != 0) return;
    u32 emit;
    if (win_len > 0) {
        u32 eff = 0;
        u8 only = 0;
        for (u32 b = 0; b < win_len; b++) {
            const u8 ch = A.seq[win_off + b];
            if (ch != (u8)'-') { eff++; only = ch; }
        }
        if (eff == 0) { A.code[gp] = 0; }
        else if (eff == 1 && only < 0x80u) { A.code[gp] = only; }
        else {
            A.code[gp] = (eff <= 126u) ? (u8)(0x80u | eff) : (u8)0xFFu;
            const u32 slot = atomicAdd(&A.counters[1], 1u);
            if (slot < A.cap_multi) {
                MultiEnt m;
                m.off = win_off; m.pos = gp; m.len = win_len; m.eff = eff; m.pad = 0;
                A.multi[slot] = m;
            } else {
                report(A.status, slot, DE_CAPACITY);
            }
        }
        emit = eff;
    } else {
        A.code[gp] = v.out;
        emit = v.out ? 1u : 0u;
    }
    if (emit) { atomicAdd(&A.win_len[w], emit); note_out_len(A.win_coarse, A.win_coarse2, w, emit); }
    if (v.status == PP_ST_CHANGED) atomicAdd(&A.stats[c].changed, 1ull);
    if (n == 0) atomicAdd(&A.stats[c].zero_depth, 1ull);
    atomicAdd(&A.stats[c].depth_fx, (u64)llrint(depth * (double)(1u << DEPTH_FX_BITS)));
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

__device__ void exact_one(const ExactArgs &A, u32 f) {
    const u32 gp = A.flag_pos[f], cap = A.flag_cov[f];
    const u32 w = gp / (u32)TILE;
    const int pr = (int)(gp - w * (u32)TILE);
    ulonglong2 *scr = A.scratch + A.flag_scr[f];

    // collect the covering alignments: x = (file index << 32 | k), y = slice (offset | len << 40)
    u32 n = 0;
    for (u32 e = A.win_lo[w]; e < A.win_hi[w]; e++) {
        const uint4 ent = A.entA[e];
        const int q = pr - item_rel(ent.z);
        const u32 fl = (ent.y >> 16) & 0xFFu, idx = ent.w;
        const bool point = (ent.z >> 31) != 0, notrim = ((ent.z >> 30) & 1u) != 0;
        if (q < 0 || q >= (int)item_extent(ent.x, ent.y, ent.z)) continue;
        const u64 so = fl ? A.seq_off[idx] : ((u64)ent.x | ((u64)(ent.y & 0xFFu) << 32));
        // fast-class items carry their untrimmed length: apply the trim here
        if (fl == 0 && !point && !notrim && (u32)q >= simple_nkeep(A.seq + so, ent.y >> 24)) continue;
        u64 s_rel;
        u32 len;
        if (point) { s_rel = 0; len = ent.y >> 24; }
        else if (!(fl & ENT_COMPLEX)) { s_rel = (u64)q; len = 1; }
        else entry_slice(A.cigar + A.cig_off[idx], A.n_cig[idx], (u32)q, &s_rel, &len);
        if (n < cap) {
            ulonglong2 v;
            v.x = ((u64)idx << 32) | (u64)A.kk[idx];
            v.y = ((so + s_rel) & SL_OFF_MASK) | ((u64)(len & 0x7FFFFFu) << 40);
            scr[n] = v;
        }
        n++;
    }
    if (n != cap) { report(A.status, gp, DE_INTERNAL); return; }
    heapsort_by_x(scr, n);

    // depth: sequential f64 adds of 1.0/k in file order (pileup.rs:64, alignment.rs:288)
    double depth = 0.0;
    u32 nA = 0, nC = 0, nG = 0, nT = 0, nDel = 0, nOth = 0;
    for (u32 i = 0; i < n; i++) {
        depth += 1.0 / (double)(u32)(scr[i].x & 0xFFFFFFFFull);
        const u64 y = scr[i].y;
        const u32 len = (u32)((y >> 40) & 0x7FFFFFu);
        if (len == 0) { nDel++; scr[i].y = y | SL_DONE; continue; }
        if (len == 1) {
            const int row = row_of(A.seq[y & SL_OFF_MASK]);
            if (row != ROW_OTH) {
                if (row == ROW_A) nA++; else if (row == ROW_C) nC++; else if (row == ROW_G) nG++;
                else if (row == ROW_T) nT++; else nDel++;
                scr[i].y = y | SL_DONE;
                continue;
            }
        }
        nOth++;
    }
    const u8 orig = A.bases[gp];
    VoteOut v = vote5(nA, nC, nG, nT, nDel, depth, orig, A.min_depth, A.fv, A.fi);
    u64 win_off = 0;
    u32 win_len = 0;  // winning string-keyed sequence, if any
    const bool low = v.status == PP_ST_LOW_DEPTH;
    if (A.dbg && nDel > 0) {  // --debug lists the deletion key like any other
        const u64 slot = atomicAdd(A.n_keys, 1ull);
        if (slot < A.cap_keys) {
            KeyRec kr;
            kr.off = 0; kr.pos = gp; kr.len = 0; kr.count = nDel; kr.pad = 0;
            A.keys[slot] = kr;
        } else report(A.status, slot, DE_CAPACITY);
    }
    if (nOth > 0 && (!low || A.dbg)) {
        // redo the tally of pileup.rs:77-109 with the remaining keys added
        int nv = 0, ni = 0;
        u8 win = 0;
        const u32 c5[5] = {nA, nC, nG, nT, nDel};
        const u8 k5[5] = {'A', 'C', 'G', 'T', '-'};
        for (int j = 0; j < 5; j++) {
            if (j == 4 && nDel == 0) break;
            if (c5[j] >= v.vthr) { if (!nv) win = k5[j]; nv++; } else if (c5[j] >= v.ithr) ni++;
        }
        for (u32 i = 0; i < n; i++) {
            const u64 yi = scr[i].y;
            if (yi & SL_DONE) continue;
            const u32 li = (u32)((yi >> 40) & 0x7FFFFFu);
            const u8 *si = A.seq + (yi & SL_OFF_MASK);
            u32 count = 1;
            for (u32 j = i + 1; j < n; j++) {
                const u64 yj = scr[j].y;
                if (yj & SL_DONE) continue;
                if ((u32)((yj >> 40) & 0x7FFFFFu) != li) continue;
                const u8 *sj = A.seq + (yj & SL_OFF_MASK);
                bool same = true;
                for (u32 b = 0; b < li; b++) if (si[b] != sj[b]) { same = false; break; }
                if (same) { count++; scr[j].y = yj | SL_DONE; }
            }
            if (count >= v.vthr) { if (!nv) { win = 0; win_off = yi & SL_OFF_MASK; win_len = li; } nv++; }
            else if (count >= v.ithr) ni++;
            if (A.dbg) {
                const u64 slot = atomicAdd(A.n_keys, 1ull);
                if (slot < A.cap_keys) {
                    KeyRec kr;
                    kr.off = yi & SL_OFF_MASK; kr.pos = gp; kr.len = li; kr.count = count; kr.pad = 0;
                    A.keys[slot] = kr;
                } else report(A.status, slot, DE_CAPACITY);
            }
        }
        if (low) {
            win_len = 0;  // the keys were only walked for the --debug records
        } else {
            v.out = (orig == (u8)'-') ? 0 : orig;
            v.status = PP_ST_KEPT;
            if (nv == 1) {
                if (ni > 0) v.status = PP_ST_TOO_CLOSE;
                else if (win_len == 0) {
                    v.out = (win == (u8)'-') ? 0 : win;
                    if (win != orig) v.status = PP_ST_CHANGED;
                } else {
                    v.status = (win_len == 1 && A.seq[win_off] == orig) ? PP_ST_KEPT : PP_ST_CHANGED;
                }
            } else {
                win_len = 0;
                v.status = (nv == 0) ? PP_ST_NONE : PP_ST_MULTIPLE;
            }
            if (v.status == PP_ST_TOO_CLOSE) win_len = 0;
        }
    }

    u32 emit;
    if (win_len > 0) {
        u32 eff = 0;
        u8 only = 0;
        for (u32 b = 0; b < win_len; b++) {
            const u8 ch = A.seq[win_off + b];
            if (ch != (u8)'-') { eff++; only = ch; }
        }
        if (eff == 0) { A.code[gp] = 0; }
        else if (eff == 1 && only < 0x80u) { A.code[gp] = only; }
        else {
            A.code[gp] = (eff <= 126u) ? (u8)(0x80u | eff) : (u8)0xFFu;
            const u32 slot = atomicAdd(&A.counters[1], 1u);
            if (slot < A.cap_multi) {
                MultiEnt m;
                m.off = win_off; m.pos = gp; m.len = win_len; m.eff = eff; m.pad = 0;
                A.multi[slot] = m;
            } else {
                report(A.status, slot, DE_CAPACITY);
            }
        }
        emit = eff;
    } else {
        A.code[gp] = v.out;
        emit = v.out ? 1u : 0u;
    }
    if (emit) { atomicAdd(&A.win_len[w], emit); note_out_len(A.win_coarse, A.win_coarse2, w, emit); }
    const u32 c = find_contig(A.contig_off, A.n_contigs, gp);
    if (v.status == PP_ST_CHANGED) atomicAdd(&A.stats[c].changed, 1ull);
    if (n == 0) atomicAdd(&A.stats[c].zero_depth, 1ull);
    atomicAdd(&A.stats[c].depth_fx, (u64)llrint(depth * (double)(1u << DEPTH_FX_BITS)));
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

// =============================================================================================
// k_exact2: ordered-depth replay for windows of up to SORT_MAX work items
// =============================================================================================
// Only the f64 depth depends on the order of the additions (pileup.rs:64); the integer tallies do
// not, and k_tile saved them.  One workgroup per window that has flagged positions:
//  (1) bitonic sort of the window's work items by record index (= SAM file order) in LDS;
//  (2) per item, in that order: window-relative start, trimmed extent and 1.0/k, to a global slab;
//  (3) ONE sequential pass over the items with all 2048 positions in parallel lanes: scalar loads of
//      the item, `depth += 1/k` in the lanes it covers -- every position sees its additions in file order;
//  (4) vote per flagged position; the few whose string-keyed tallies could reach a threshold are
//      handed to the thread-serial k_exact through the global list.
// Two instances share the windows by their item count: (0, SORT_SMALL] with 80 KiB of LDS -- two workgroups per CU,
// the usual case -- and (SORT_SMALL, SORT_MAX] with 128 KiB.  A third one (SUB > 1) takes the listed heavy windows,
// whatever their size: one block per TILE / SUB positions, which first picks the items that reach its positions out of
// the window's list (sel[]) and then does the same as the others with those -- the pass over a window of five times
// the usual depth is spread over SUB CUs instead of holding one for five times as long.
template <u32 SMAX, u32 NLOW, u32 SUB>
__global__ __launch_bounds__(1024, (SUB == 1 && SMAX <= SORT_SMALL) ? 8 : 4) void k_exact2(ExactArgs A, u32 nwin) {
    __shared__ u64 pk[SMAX];  // bitonic sort keys (record index << 16 | slot), or the counting sort's arrays
    __shared__ u32 sel[SUB > 1 ? SMAX : 1];  // SUB > 1: index (in the window's list) of the j-th item picked
    __shared__ u64 s_base;
    __shared__ u32 s_npick;
    constexpr u32 LDS_LIST_MAX = SMAX * 3u / 8u;  // ordered (start, extent, share) records that fit below ord[]
    constexpr int PSPAN = TILE / (int)SUB;        // positions of this block
    static_assert(PSPAN % 128 == 0 && SMAX <= 65536, "a wave owns 64 or 128 positions; slots are 16-bit");
    const u32 tid = threadIdx.x;
    u32 w;
    int plo = 0;
    const int state = job_state(A.status);
    if (state == 2) return;
    if (SUB == 1) {
        // Block f replays the f-th window that k_tile found flagged positions in (the windows with a tally slab are
        // listed in slab_win: a grid over ALL windows costs a launch of 122 K empty blocks per instance on a 250 Mbp
        // job).  After a late capacity overflow the list may be incomplete: the needs are then added up over all windows.
        if (state == 1) {
            if (tid == 0)
                for (u32 ww = blockIdx.x; ww < nwin; ww += gridDim.x) {
                    if (!A.win_nflag[ww] || A.win_heavy[ww]) continue;
                    const u32 nn = A.win_hi[ww] - A.win_lo[ww];
                    if (nn <= SMAX && nn > NLOW && nn > LDS_LIST_MAX) atomicAdd(A.ents_cursor, (u64)nn);  // smaller lists stay in LDS
                }
            return;
        }
        if (blockIdx.x >= min(A.counters[3], A.cap_slabs)) return;
        w = A.slab_win[blockIdx.x];
    } else {
        const u32 hs = blockIdx.x / SUB;
        if (hs >= min(A.heavy[0], HEAVY_SLOTS)) return;
        w = A.heavy[1 + hs];
        plo = (int)(blockIdx.x % SUB) * PSPAN;
    }
    if (A.win_nflag[w] == 0) return;
    const u32 e0 = A.win_lo[w], n_all = A.win_hi[w] - e0;
    if (SUB == 1) {
        if (A.win_heavy[w]) return;                  // the sub-range instance's
        if (n_all > SMAX || n_all <= NLOW) return;  // the other instance's window, or (n > SORT_MAX) replayed by k_exact
    } else {
        u32 any = 0;  // nothing flagged among this block's positions?
        for (int q = 0; q < PSPAN / 32; q++) any |= A.flag_bits[(u64)w * (TILE / 32) + (u32)(plo / 32 + q)];
        if (!any) return;
    }
    if (state == 1) {  // (SUB > 1) a buffer was too small: only add up the replay scratch the rerun will need
        if (tid == 0) atomicAdd(A.ents_cursor, (u64)n_all);
        return;
    }
    const u32 slab = A.win_slab[w];

    // ---- (1) order the window's items by record index (= SAM file order) ----
    // Record indices of a window's items are spread over the file, so a counting sort on their leading bits
    // (up to SORT_BUCKETS buckets between the window's smallest and largest index) leaves buckets of a few
    // items, finished by one thread each with an insertion sort: ~10x fewer LDS passes than a bitonic network
    // over 16 K keys.  Clustered indices (a bucket above SORT_BUCKET_MAX items) take the bitonic sort instead.
    // The arrays of the counting sort live inside pk[].
    u32 *rec = (u32 *)pk;                                              // [SMAX] record index of slot i
    u32 *bkt = (u32 *)(pk + SMAX / 2);                                  // [SORT_BUCKETS + 1] counts -> cursors
    unsigned short *ord = (unsigned short *)(pk + SMAX / 2 + SMAX / 4);  // [SMAX] slots in file order (last quarter)
    static_assert((SORT_BUCKETS + 1u) * 4u <= SMAX * 2u, "bkt[] must end before ord[]");
    __shared__ u32 s_lo, s_hi, s_big, s_wtot[16];
    if (tid == 0) { s_lo = 0xFFFFFFFFu; s_hi = 0; s_big = 0; s_npick = 0; }
    for (u32 i = tid; i <= SORT_BUCKETS; i += 1024) bkt[i] = 0;
    __syncthreads();
    u32 n = n_all;
    if (SUB > 1) {
        // the items that can reach [plo, plo + PSPAN): by their untrimmed extent (the trim only shortens it)
        for (u32 i0 = tid; i0 < n_all; i0 += 4096) {
            uint4 ent[4];
#pragma unroll
            for (u32 u = 0; u < 4; u++) ent[u] = A.entA[e0 + min(i0 + 1024u * u, n_all - 1u)];  // four loads in flight
#pragma unroll
            for (u32 u = 0; u < 4; u++) {
                const u32 i = i0 + 1024u * u;
                const u32 ext = item_extent(ent[u].x, ent[u].y, ent[u].z);
                const int z = item_rel(ent[u].z);
                if (i < n_all && z < plo + PSPAN && (long long)z + (long long)ext > (long long)plo) {
                    const u32 j = atomicAdd(&s_npick, 1u);
                    if (j < SMAX) sel[j] = i;
                }
            }
        }
        __syncthreads();
        n = s_npick;
        if (n > SMAX || n == 0) {
            // More than one block sorts (a repeat far deeper than the list is meant for): this block's flagged
            // positions go to the thread-serial k_exact through the global list, like a window above SORT_MAX.
            // (n == 0 cannot be, a flagged position is covered; it takes the same exit rather than a special case.)
            const u32 *tal0 = A.slabs + (u64)slab * 6u * TILE;
            for (u32 p = (u32)plo + tid; p < (u32)(plo + PSPAN); p += 1024) {
                if (!((A.flag_bits[(u64)w * (TILE / 32) + (p >> 5)] >> (p & 31u)) & 1u)) continue;
                const u32 ntot = tal0[0 * TILE + p] + tal0[1 * TILE + p] + tal0[2 * TILE + p] + tal0[3 * TILE + p] +
                                 tal0[4 * TILE + p] + tal0[5 * TILE + p];
                const u32 slot = atomicAdd(&A.counters[0], 1u);
                const u64 scr_at = atomicAdd(A.scr_need, (u64)ntot);
                if (slot < A.cap_flag) { A.flag_pos_w[slot] = w * (u32)TILE + p; A.flag_cov_w[slot] = ntot; A.flag_scr_w[slot] = scr_at; }
                else report(A.status, slot, DE_CAPACITY_LATE);
            }
            return;
        }
    }
    auto item_of_slot = [&](u32 j) -> u32 { return SUB > 1 ? sel[j] : j; };
    {
        u32 lo = 0xFFFFFFFFu, hi = 0;
        for (u32 i = tid; i < n; i += 1024) {
            const u32 r = A.entA[e0 + item_of_slot(i)].w;
            rec[i] = r;
            lo = min(lo, r); hi = max(hi, r);
        }
        for (int o = 32; o > 0; o >>= 1) {
            lo = min(lo, (u32)__shfl_xor((int)lo, o, 64));
            hi = max(hi, (u32)__shfl_xor((int)hi, o, 64));
        }
        if ((tid & 63u) == 0) { atomicMin(&s_lo, lo); atomicMax(&s_hi, hi); }
    }
    __syncthreads();
    const u32 r_lo = s_lo;
    u32 sh = 0;  // bucket of r = (r - r_lo) >> sh, below SORT_BUCKETS
    while (((s_hi - r_lo) >> sh) >= SORT_BUCKETS) sh++;
    for (u32 i = tid; i < n; i += 1024) atomicAdd(&bkt[(rec[i] - r_lo) >> sh], 1u);
    __syncthreads();
    {   // exclusive scan of the SORT_BUCKETS counts: SORT_BUCKETS / 1024 per thread, wave scan, wave totals
        constexpr u32 PER = SORT_BUCKETS / 1024;
        u32 c[PER], sum = 0, big = 0;
#pragma unroll
        for (u32 q = 0; q < PER; q++) { c[q] = bkt[tid * PER + q]; sum += c[q]; big = max(big, c[q]); }
        u32 inc = sum;
        for (int o = 1; o < 64; o <<= 1) {
            const u32 v = (u32)__shfl_up((int)inc, o, 64);
            if ((int)(tid & 63u) >= o) inc += v;
        }
        if ((tid & 63u) == 63u) s_wtot[tid >> 6] = inc;
        if (big > SORT_BUCKET_MAX) atomicOr(&s_big, 1u);
        __syncthreads();
        u32 before = inc - sum;
        for (u32 v = 0; v < (tid >> 6); v++) before += s_wtot[v];
#pragma unroll
        for (u32 q = 0; q < PER; q++) { bkt[tid * PER + q] = before; before += c[q]; }
        if (tid == 1023) bkt[SORT_BUCKETS] = before;
    }
    __syncthreads();
    const bool bitonic = s_big != 0;
    // The ordered list of (start, extent, share) records of step (2) stays in LDS when it fits the part of pk[]
    // that is free by then (everything below ord[]; the bitonic keys occupy it), else it goes to a global slab.
    const bool in_lds = !bitonic && n <= LDS_LIST_MAX;
    if (tid == 0) {
        u64 base = 0;
        if (!in_lds) {
            base = atomicAdd(A.ents_cursor, (u64)n);
            if (base + n > A.cap_ents) report(A.status, base + n, DE_CAPACITY_LATE);
        }
        s_base = base;
    }
    if (!bitonic) {
        // scatter the slots into their buckets (the cursor of bucket b ends at the start of bucket b+1) ...
        for (u32 i = tid; i < n; i += 1024) ord[atomicAdd(&bkt[(rec[i] - r_lo) >> sh], 1u)] = (unsigned short)i;
        __syncthreads();
        // ... and finish every bucket: buckets tid*PER .. tid*PER+PER-1 are one contiguous stretch of ord[]
        constexpr u32 PER = SORT_BUCKETS / 1024;
        u32 beg = tid ? bkt[tid * PER - 1] : 0u;
        for (u32 q = 0; q < PER; q++) {
            const u32 end = bkt[tid * PER + q];
            for (u32 a2 = beg + 1; a2 < end; a2++) {
                const unsigned short v = ord[a2];
                const u32 key = rec[v];
                u32 c2 = a2;
                while (c2 > beg && rec[ord[c2 - 1]] > key) { ord[c2] = ord[c2 - 1]; c2--; }
                ord[c2] = v;
            }
            beg = end;
        }
        __syncthreads();
    } else {
        __syncthreads();
        u32 np2 = 2;
        while (np2 < n) np2 <<= 1;
        for (u32 i = tid; i < np2; i += 1024) pk[i] = i < n ? (((u64)A.entA[e0 + item_of_slot(i)].w << 16) | (u64)i) : ~0ull;
        __syncthreads();
        for (u32 k = 2; k <= np2; k <<= 1) {
            for (u32 lj = 31u - (u32)__clz((int)k); lj-- > 0;) {  // partner distance j = 2^lj = k/2 ... 1
                const u32 j = 1u << lj;
                for (u32 t = tid; t < (np2 >> 1); t += 1024) {
                    const u32 i = ((t >> lj) << (lj + 1u)) | (t & (j - 1u)), o = i + j;
                    const bool asc = (i & k) == 0;
                    const u64 x = pk[i], y = pk[o];
                    if ((x > y) == asc) { pk[i] = y; pk[o] = x; }
                }
                __syncthreads();
            }
        }
    }
    if (!in_lds && s_base + n > A.cap_ents) return;  // the host grows the buffer and reruns
    ulonglong2 *ents = A.ents + s_base;
    ulonglong2 *ents_lds = (ulonglong2 *)pk;
    // ---- (2) start, trimmed extent and depth share of every item, in file order ----
    for (u32 i = tid; i < n; i += 1024) {
        const uint4 ent = A.entA[e0 + item_of_slot(bitonic ? (u32)(pk[i] & 0xFFFFu) : (u32)ord[i])];
        const u32 fl = (ent.y >> 16) & 0xFFu, kc = (ent.y >> 8) & 0xFFu;
        u32 lim;
        if (fl) lim = ent.x;
        else if (ent.z >> 31) lim = 1u;                    // the entry at a read's single indel
        else if ((ent.z >> 30) & 1u) lim = ent.y >> 24;    // the flank in front of it: no trim
        else {
            // the trim from the read's last four bases, one load (as k_tile's plain class does it); a trailing
            // homopolymer of four or more, or a read shorter than that, walks byte by byte
            const u8 *rp = A.seq + ((u64)ent.x | ((u64)(ent.y & 0xFFu) << 32));
            const u32 L = ent.y >> 24;
            u32 tf = 0;
            if (L >= 4u) {
                const u32 tail = load4_unaligned(rp + (L - 4u));
                tf = nz_flags(tail ^ splat8(tail >> 24));
            }
            lim = tf ? L - 4u + (u32)((31 - __clz((int)tf)) >> 3) : simple_nkeep(rp, L);
        }
        const u32 k = k_of_class(kc, A.kk, ent.w);
        ulonglong2 r;
        r.x = (u64)(u32)item_rel(ent.z) | ((u64)lim << 32);
        r.y = (u64)__double_as_longlong(1.0 / (double)k);
        if (in_lds) ents_lds[i] = r;
        else ents[i] = r;
    }
    __threadfence_block();
    __syncthreads();

    // ---- (3) the sequential pass: lanes are positions ----
    // 64 items per vector load (one per lane), then v_readlane turns each item into scalars
    // Wave v owns the 128 consecutive positions [128v, 128v+128): an item overlaps ~2 of the 16 waves, the
    // others never enter the scalar loop (ballot of a per-lane overlap test).
    const u32 lane = tid & 63u;
    // SUB > 1: a lane owns ONE position and a wave 64 -- the blocks of a heavy window run alone on their CUs, so a
    // wave's chain of dependent additions is what takes the time, and half as many positions per wave means fewer
    // items to visit and one fma per visit; the block's positions take only its first waves.
    constexpr int PPL = SUB > 1 ? 1 : 2, WSPAN = 64 * PPL;
    const bool pos_wave = (int)(tid >> 6) * WSPAN < PSPAN;
    const int wlo = plo + (int)(tid >> 6) * WSPAN;
    const int p0 = wlo + (int)lane, p1 = p0 + 64;
    double d0 = 0.0, d1 = 0.0;
    // The pass runs over (start, extent, share) records in LDS: 64 records per vector read (one per lane) and a ballot
    // pick the records that reach the wave's positions, each of those is then read once more at a wave-uniform address
    // (ONE broadcast read for its three words) and added where it applies: fma(1.0, share, depth) is the rounded sum
    // depth + share, fma(0.0, share, depth) leaves depth as it is -- one select + one fma per position instead of a
    // branch.  Two batch registers in turn, so that the read of the batch after the current one is in flight while
    // the current one is visited.  A list that does not fit LDS is taken from the global slab in chunks of CHUNK
    // records (picking a record's words out of registers with v_readlane instead costs 22 instead of 14 instructions
    // per visit, and the visits are what the pass consists of).
    auto visit = [&](const ulonglong2 &mine, bool have, u32 bbase) {
        const int xl = (int)(u32)mine.x, xh = (int)(u32)(mine.x >> 32);
        // one vector compare picks the items of this batch that reach the wave's positions; only those are
        // visited one by one, in ascending order = file order
        u64 hits = __ballot(have && xl < wlo + WSPAN && (long long)xl + (long long)(u32)xh > (long long)wlo);
        if (SUB == 1) {
            // sixteen waves share the CU: the other waves hide the read's latency, the instruction count is what matters
            while (hits) {
                const int j = __ffsll((long long)hits) - 1;
                hits &= hits - 1;
                const ulonglong2 it = ents_lds[bbase + (u32)j];
                const int rel = (int)(u32)it.x;
                const u32 lim = (u32)(it.x >> 32);
                const double dc = __longlong_as_double((long long)it.y);
                d0 = fma(__hiloint2double((u32)(p0 - rel) < lim ? 0x3FF00000 : 0, 0), dc, d0);
                d1 = fma(__hiloint2double((u32)(p1 - rel) < lim ? 0x3FF00000 : 0, 0), dc, d1);
            }
            return;
        }
        if (!hits) return;
        // a few waves alone on their CU: the item after the current one is fetched before the current one is applied
        // (a lone wave would otherwise wait out one LDS round trip per item)
        int j = __ffsll((long long)hits) - 1;
        hits &= hits - 1;
        ulonglong2 it = ents_lds[bbase + (u32)j];
        for (;;) {
            const bool more = hits != 0;
            ulonglong2 nx = it;
            if (more) {
                j = __ffsll((long long)hits) - 1;
                hits &= hits - 1;
                nx = ents_lds[bbase + (u32)j];
            }
            const int rel = (int)(u32)it.x;
            const u32 lim = (u32)(it.x >> 32);
            const double dc = __longlong_as_double((long long)it.y);
            d0 = fma(__hiloint2double((u32)(p0 - rel) < lim ? 0x3FF00000 : 0, 0), dc, d0);
            if (!more) break;
            it = nx;
        }
    };
    auto ordered_pass = [&](u32 m) {  // over ents_lds[0, m)
        ulonglong2 ba = ents_lds[min(lane, m - 1u)], bb;
        for (u32 base = 0; base < m; base += 128) {
            bb = ents_lds[min(base + 64u + lane, m - 1u)];
            visit(ba, base + lane < m, base);
            ba = ents_lds[min(base + 128u + lane, m - 1u)];
            visit(bb, base + 64u + lane < m, base + 64u);
        }
    };
    if (in_lds) {
        if (pos_wave) ordered_pass(n);
    } else {
        constexpr u32 CHUNK = SMAX / 2u;  // 16-byte records in pk[], which nobody needs any more
        for (u32 c0 = 0; c0 < n; c0 += CHUNK) {
            const u32 m = min(CHUNK, n - c0);
            __syncthreads();
            for (u32 i = tid; i < m; i += 1024) ents_lds[i] = ents[c0 + i];
            __syncthreads();
            if (pos_wave) ordered_pass(m);
        }
    }

    // ---- (4) vote for the flagged positions; per-window sums are reduced in the block first ----
    const u32 *tal = A.slabs + (u64)slab * 6u * TILE;
    const u64 gw0 = (u64)w * TILE;
    const u32 c_first = find_contig_wave(A.contig_off, A.n_contigs, gw0, lane);
    const bool one_contig = c_first == find_contig_wave(A.contig_off, A.n_contigs, min(gw0 + TILE, A.G) - 1, lane);
    u32 my_len = 0, my_changed = 0, my_zero = 0;
    u64 my_depth = 0;
    for (int h = 0; h < (pos_wave ? PPL : 0); h++) {
        const u32 p = h ? (u32)p1 : (u32)p0;
        const double depth = h ? d1 : d0;
        if (!((A.flag_bits[(u64)w * (TILE / 32) + (p >> 5)] >> (p & 31u)) & 1u)) continue;
        const u32 gp = w * (u32)TILE + p;
        const u32 nA = tal[0 * TILE + p], nC = tal[1 * TILE + p], nG = tal[2 * TILE + p], nT = tal[3 * TILE + p],
                  nDel = tal[4 * TILE + p], nOth = tal[5 * TILE + p];
        const u32 ntot = nA + nC + nG + nT + nDel + nOth;
        const u8 orig = A.bases[gp];
        const VoteOut vo = vote5(nA, nC, nG, nT, nDel, depth, orig, A.min_depth, A.fv, A.fi);
        if (vo.status != PP_ST_LOW_DEPTH && nOth > 0 && nOth >= vo.ithr) {
            // a string-keyed tally could reach a threshold: full replay by the thread-serial kernel
            const u32 slot = atomicAdd(&A.counters[0], 1u);
            const u64 scr_at = atomicAdd(A.scr_need, (u64)ntot);
            if (slot < A.cap_flag) { A.flag_pos_w[slot] = gp; A.flag_cov_w[slot] = ntot; A.flag_scr_w[slot] = scr_at; }
            else report(A.status, slot, DE_CAPACITY_LATE);
            continue;
        }
        A.code[gp] = vo.out;
        const u64 dfx = (u64)llrint(depth * (double)(1u << DEPTH_FX_BITS));
        if (one_contig) {
            my_len += vo.out ? 1u : 0u;
            my_changed += vo.status == PP_ST_CHANGED;
            my_zero += ntot == 0;
            my_depth += dfx;
        } else {
            my_len += vo.out ? 1u : 0u;
            const u32 cg = find_contig(A.contig_off, A.n_contigs, gp);
            if (vo.status == PP_ST_CHANGED) atomicAdd(&A.stats[cg].changed, 1ull);
            if (ntot == 0) atomicAdd(&A.stats[cg].zero_depth, 1ull);
            atomicAdd(&A.stats[cg].depth_fx, dfx);
        }
        if (A.dbg) {
            A.dbg_depth[gp] = depth;
            A.dbg_counts[0 * A.G + gp] = nA;
            A.dbg_counts[1 * A.G + gp] = nC;
            A.dbg_counts[2 * A.G + gp] = nG;
            A.dbg_counts[3 * A.G + gp] = nT;
            A.dbg_counts[4 * A.G + gp] = nDel + nOth;
            A.dbg_counts[5 * A.G + gp] = vo.vthr;
            A.dbg_counts[6 * A.G + gp] = vo.ithr;
            A.dbg_status[gp] = vo.status;
        }
    }
    __syncthreads();  // pk is free again: reuse its first words for the block reduction
    if (tid < 4) pk[tid] = 0;
    __syncthreads();
    my_len = wave_sum(my_len); my_changed = wave_sum(my_changed); my_zero = wave_sum(my_zero);
    my_depth = wave_sum64(my_depth);
    if (lane == 0) {
        if (my_len) atomicAdd(&pk[0], (u64)my_len);
        if (my_changed) atomicAdd(&pk[1], (u64)my_changed);
        if (my_zero) atomicAdd(&pk[2], (u64)my_zero);
        if (my_depth) atomicAdd(&pk[3], my_depth);
    }
    __syncthreads();
    if (tid == 0) {
        if (pk[0]) { atomicAdd(&A.win_len[w], (u32)pk[0]); note_out_len(A.win_coarse, A.win_coarse2, w, (u32)pk[0]); }
        if (pk[1]) atomicAdd(&A.stats[c_first].changed, pk[1]);
        if (pk[2]) atomicAdd(&A.stats[c_first].zero_depth, pk[2]);
        if (pk[3]) atomicAdd(&A.stats[c_first].depth_fx, pk[3]);
    }
}

}  // namespace pp
