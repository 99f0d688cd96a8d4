// pp_k_stream.h -- k_stream: ONE pass over the alignment records in file order -- validation, CIGAR-walk spans, the
// right-end homopolymer trim, and the comparison of every read with the assembly -- that turns each record into a few
// 8-byte "units" for the window(s) it overlaps.  Part of pp_kernels.hip (included there and nowhere else).
//
// Why: in file order the SEQ bytes are ONE coalesced stream (every 128-byte line is fetched once, 1.0 GB for
// configs[1]); gathering them window by window (round 1) touched 2.2 lines per 150-byte read.  The assembly bytes a
// read is compared with come from L2 / Infinity Cache (5 MB for configs[1]).  What a window then needs from a read is
// tiny: where its kept bases start and end (a coverage difference pair) and the few bases that differ from the assembly.
//
//   PLAIN unit  a read without indels, 8..252 bases, inside its contig, depth share 1 (or non-dyadic): window-relative
//               start, kept entries after the trim (alignment.rs:364-378), record index
//   EVENT unit  one base of a PLAIN read that differs from the assembly: window position + counter row
//   SLOW  unit  every other record (indels, long reads, dyadic shares != 1): record index + class; k_tile walks it
//
// Units are staged in LDS, sorted by coarse bucket (2^shift windows) and flushed as a SEGMENT (contiguous in HBM,
// full-line writes) with its row of bucket offsets; k_regroup gathers a bucket's pieces from all segments.
#pragma once

namespace pp {

constexpr u32 UNIT_PLAIN = 0, UNIT_EVENT = 1, UNIT_SLOW = 2, UNIT_NOP = 3;
// Geometry of k_stream (compile-time; the defaults are the measured best, the macros exist for experiments)
#ifndef PP_STREAM_THREADS
#define PP_STREAM_THREADS 1024
#endif
#ifndef PP_STREAM_MINW
#define PP_STREAM_MINW 8      // waves per SIMD the register allocation must allow (8 -> 64 VGPRs, 4 -> 128)
#endif
#ifndef PP_STREAM_LDS_KB
#define PP_STREAM_LDS_KB 80   // LDS per workgroup: units staged per segment ~ this / 10 bytes
#endif
#ifndef PP_STREAM_NT
#define PP_STREAM_NT 0     // 1: non-temporal loads for the SEQ stream (experiment)
#endif
constexpr u32 STREAM_THREADS = PP_STREAM_THREADS;
constexpr u32 STREAM_WAVES = STREAM_THREADS / 64;
constexpr u32 STREAM_BLOCKS_PER_CU = (PP_STREAM_MINW * 256) / PP_STREAM_THREADS < 160 / PP_STREAM_LDS_KB
                                         ? (PP_STREAM_MINW * 256) / PP_STREAM_THREADS : 160 / PP_STREAM_LDS_KB;
constexpr u32 STREAM_BATCH = 60;          // records per wave and block iteration (a multiple of every group count)
constexpr u32 STREAM_HEADROOM = STREAM_WAVES * STREAM_BATCH * 2;  // units the waves may add before they all notice that a segment is due
constexpr u32 STREAM_DIRECT_WINDOWS = 8;  // a SLOW record spanning more windows goes to the late list as ONE entry
constexpr int UNIT_REL_BIAS = 256;

struct LateEnt {  // a unit for the windows [w0, w1] that did not go through the LDS staging (rare)
    u32 w0, w1;
    u64 unit;
};

// unit = (record index << 32) | low word
//   PLAIN  [1:0] 0 | [13:2] start - window start + 256 | [21:14] kept entries | [22] non-dyadic share | [31:24] window in bucket
//   EVENT  [1:0] 1 | [12:2] window position | [15:13] counter row | [31:24] window in bucket
//   SLOW   [1:0] 2 | [3:2] ENT_COMPLEX / ENT_PRETRIM | [31:24] window in bucket
__device__ __forceinline__ u64 unit_plain(u32 idx, int rel, u32 nkeep, bool nd, u32 sub) {
    return ((u64)idx << 32) | (u64)(UNIT_PLAIN | ((u32)(rel + UNIT_REL_BIAS) << 2) | (nkeep << 14) | ((u32)nd << 22) | (sub << 24));
}
__device__ __forceinline__ u64 unit_event(u32 idx, u32 pos, u32 row, u32 sub) {
    return ((u64)idx << 32) | (u64)(UNIT_EVENT | (pos << 2) | (row << 13) | (sub << 24));
}
__device__ __forceinline__ u64 unit_slow(u32 idx, u32 flags, u32 sub) {
    return ((u64)idx << 32) | (u64)(UNIT_SLOW | (flags << 2) | (sub << 24));
}
__device__ __forceinline__ u64 unit_with_sub(u64 unit, u32 sub) {
    return (unit & ~(0xFFull << 24)) | ((u64)sub << 24);
}

struct StreamArgs {
    u64 n, chunk;
    const u32 *contig, *ref_start, *kk;
    const u64 *seq_off;
    const u32 *seq_len;
    const u64 *cig_off;
    const u32 *n_cig, *cigar;
    const u8 *seq;
    u64 seq_bytes;
    const u8 *bases;
    u64 G;
    const u64 *contig_off;
    u32 n_contigs;
    u32 shift, nbk;        // bucket = window >> shift; nbk buckets
    u32 stage_cap;         // staged units per segment (LDS)
    u64 *units;            // segments, back to back
    u64 cap_units;
    u64 *unit_cursor;
    unsigned short *seg_off;  // per segment: nbk + 1 offsets relative to its base
    u64 *seg_base;
    u32 *seg_cursor;
    u32 cap_segs;
    LateEnt *late;
    u64 *late_cursor;
    u64 cap_late;
    u32 *nkeep_arr;        // kept entries of SLOW records (written here, read by k_tile / the replay kernels)
    u64 *status;
    u64 *prof;             // PP_STREAM_PROFILE builds: cycles per phase, summed over the waves
};

struct Stage {
    u32 *hist;             // [nbk] counts, then (during a flush) exclusive offsets / cursors
    u64 *units;            // [cap]
    unsigned short *bkt;   // [cap]
    u32 *n;                // units offered so far (may run past cap: those went to the late list)
    u32 cap;
};

__device__ __forceinline__ void late_put(const StreamArgs &A, u32 w0, u32 w1, u64 unit) {
    const u64 slot = atomicAdd(A.late_cursor, 1ull);
    if (slot < A.cap_late) {
        LateEnt e;
        e.w0 = w0; e.w1 = w1; e.unit = unit;
        A.late[slot] = e;
    } else {
        report(A.status, slot, DE_CAPACITY);
    }
}

__device__ __forceinline__ void stage_at(const StreamArgs &A, const Stage &S, u32 pos, u64 unit, u32 w) {
    const u32 b = w >> A.shift, sub = w & ((1u << A.shift) - 1u);
    if (pos < S.cap) {
        S.units[pos] = unit_with_sub(unit, sub);
        S.bkt[pos] = (unsigned short)b;
        atomicAdd(&S.hist[b], 1u);
    } else {
        late_put(A, w, w, unit);
    }
}

// Sort the staged units by bucket and write them out as one segment.  Called by every thread of the block after the
// rendezvous barrier (nobody appends).  Barriers are what a flush costs, so there are three: the first wave scans the
// bucket counts into cursors and fetches the segment's place in HBM while the others wait; every thread then sends
// its units to their place in the segment (LDS cursors; the segment's footprint is small and written within
// microseconds, L2 merges the 8-byte stores).  The bucket counts for the NEXT segment live in the other half of a
// double array that this flush clears on the way, so nothing is cleared between the last barrier and the appends.
__device__ void stream_flush(const StreamArgs &A, const Stage &S, u32 *hist_next, u64 *s_base, u32 *s_seg) {
    const u32 tid = threadIdx.x, lane = tid & 63u;
    const u32 n = min(*(volatile u32 *)S.n, S.cap);
    if (n == 0) {  // uniform: everything offered went to the late list, or nothing was offered
        if (tid == 0) *S.n = 0;
        __syncthreads();
        return;
    }
#ifdef PP_STREAM_PROFILE
    u64 ft0 = clock64(), ft1;
#define PP_FSTAMP(k) do { ft1 = clock64(); if (lane == 0) atomicAdd(&A.prof[k], ft1 - ft0); ft0 = ft1; } while (0)
#else
#define PP_FSTAMP(k) do { } while (0)
#endif
    if (tid < 64u) {
        u64 base = 0;
        u32 seg = 0;
        if (lane == 0) {
            base = atomicAdd(A.unit_cursor, (u64)n);
            seg = atomicAdd(A.seg_cursor, 1u);
        }
        // exclusive scan of the bucket counts: consecutive buckets per lane, wave scan; counts become cursors
        const u32 per = (A.nbk + 63u) / 64u;
        const u32 b0 = min(A.nbk, lane * per), b1 = min(A.nbk, b0 + per);
        u32 sum = 0;
        for (u32 b = b0; b < b1; b++) sum += S.hist[b];
        u32 inc = sum;
        for (int o = 1; o < 64; o <<= 1) {
            const u32 v = (u32)__shfl_up((int)inc, o, 64);
            if ((int)lane >= o) inc += v;
        }
        base = ((u64)(u32)__builtin_amdgcn_readfirstlane((int)(u32)(base >> 32)) << 32) |
               (u64)(u32)__builtin_amdgcn_readfirstlane((int)(u32)base);
        seg = (u32)__builtin_amdgcn_readfirstlane((int)seg);
        const bool ok = base + n <= A.cap_units && seg < A.cap_segs;
        if (!ok && lane == 0) report(A.status, base + n, DE_CAPACITY);  // the cursors keep counting what a rerun needs
        u32 run = inc - sum;
        unsigned short *row = A.seg_off + (u64)seg * (A.nbk + 1u);
        for (u32 b = b0; b < b1; b++) {
            const u32 c = S.hist[b];
            S.hist[b] = run;
            if (ok) row[b] = (unsigned short)run;
            run += c;
        }
        if (lane == 0) {
            if (ok) {
                row[A.nbk] = (unsigned short)n;
                A.seg_base[seg] = base;
            }
            *s_base = ok ? base : ~0ull;
            *s_seg = seg;
        }
    } else {
        for (u32 b = tid - 64u; b < A.nbk; b += STREAM_THREADS - 64u) hist_next[b] = 0;
    }
    PP_FSTAMP(6);
    __syncthreads();
    PP_FSTAMP(7);
    const u64 base = *s_base;
    if (base != ~0ull) {
        for (u32 i = tid; i < n; i += STREAM_THREADS) {
            const u32 p = atomicAdd(&S.hist[S.bkt[i]], 1u);
            A.units[base + p] = S.units[i];
        }
    }
    PP_FSTAMP(0);
    if (tid == 0) *S.n = 0;
    __syncthreads();
}

// ---- classification of one record (one lane each): validation, spans, class ---------------------------------
struct RecInfo {
    u32 g;      // global start (assembly position of the first entry)
    u32 len;    // PLAIN: read length; SLOW: kept entries after the trim
    u32 kind;   // 0 contributes nothing (or an error was reported), 1 PLAIN, 2 SLOW
    u32 flags;  // SLOW: ENT_COMPLEX / ENT_PRETRIM
    u32 nd;     // PLAIN: depth share is not a power of two
    u64 so;     // PLAIN: offset of the read in the seq array
};

// the SoA fields of one record (first round trip; prefetched one block iteration ahead)
struct RecFields {
    u32 c, nc, sl, rs, k;
    u64 co, so;
};
__device__ __forceinline__ RecFields stream_fields(const StreamArgs &A, u64 a) {
    const u64 ai = min(a, A.n - 1ull);  // unconditional loads, clamped
    RecFields F;
    F.c = A.contig[ai]; F.nc = A.n_cig[ai]; F.sl = A.seq_len[ai]; F.rs = A.ref_start[ai]; F.k = A.kk[ai];
    F.co = A.cig_off[ai]; F.so = A.seq_off[ai];
    return F;
}

// c_lo / c_hi: bounds of the record's contig (clamped index), op0: its first CIGAR run -- the second round trip
__device__ __forceinline__ RecInfo stream_classify(const StreamArgs &A, u64 a, bool valid, const RecFields &F, u64 c_lo,
                                                   u64 c_hi, u32 op0) {
    RecInfo R;
    R.g = 0; R.len = 0; R.kind = 0; R.flags = 0; R.nd = 0; R.so = 0;
    if (!valid) return R;
    const u32 c = F.c, nc = F.nc, sl = F.sl, rs = F.rs, k = F.k;
    const u64 so = F.so;
    const u32 *cg = A.cigar + F.co;
    bool bad = false;
    if (k == 0) { report(A.status, a, DE_BAD_K); bad = true; }
    if (so + sl > (1ull << 40)) { report(A.status, a, DE_OVERFLOW); bad = true; }
    if (c >= A.n_contigs) { report(A.status, a, DE_BAD_CONTIG); return R; }
    if (nc == 0) { report(A.status, a, DE_BAD_RUN); return R; }
    const u64 clen = c_hi - c_lo;
    u32 g = 0, nk = 0;
    u8 fl = 0;
    bool fastlike;
    if (nc == 1 && (op0 & 15u) == PP_OP_M && (op0 >> 4) == sl && sl > 0 && sl <= FAST_MAX_LEN && (u64)rs + sl <= clen) {
        fastlike = true;  // the bulk: one short M run inside its contig
        g = (u32)(c_lo + rs);
    } else {
        prep_general(a, rs, sl, so, cg, nc, A.seq, c_lo, c_hi, &g, &nk, &fl, A.status);
        fastlike = fl == 0 && nk > 0;  // no indel, short, inside the contig: not trimmed yet
    }
    if (bad) return R;
    const u32 kc = kclass_of(k);
    if (fastlike) {
        const u64 span32 = ((u64)sl + 31ull) & ~31ull;
        const bool plain_ok = sl >= PLAIN_MIN_LEN && (kc == 0 || kc == KCLASS_NONDYADIC) && so + span32 <= A.seq_bytes &&
                              (u64)g + span32 <= A.G;
        if (plain_ok) {
            R.g = g; R.len = sl; R.kind = 1; R.nd = kc == KCLASS_NONDYADIC; R.so = so;
            return R;
        }
        const u32 run = sl - simple_trim_start(A.seq + so, sl);  // alignment.rs:364-378: the run, then one more
        nk = sl > run ? sl - run - 1u : 0u;
        fl = (u8)ENT_PRETRIM;
    }
    if (nk == 0) return R;
    R.g = g; R.len = nk; R.kind = 2; R.flags = fl;
    return R;
}

// ---- PLAIN passes: GW lanes per read, 32 read bytes per lane, compared with the assembly bytes at the same positions.
// Everything that is per READ (trim, units) was done with one record per lane before the passes; a pass only loads,
// compares and stages one EVENT per differing base.  (Measured and dropped: pipelining the loads of the next pass
// behind this one, and a cheap any-difference test with the byte-level work handed to eight lanes through LDS --
// both slower here: the passes are bound by the rate at which the memory pipeline takes the four 16-byte loads.)
template <int GW>
__device__ __forceinline__ void stream_plain_passes(const StreamArgs &A, const Stage &S, u64 base, u32 nb, u32 lane,
                                                    u32 my_g, u64 my_so, u32 my_nkeep) {
    typedef PlainCfg<GW> C;
    const u32 g = C::group(lane), s = lane - (u32)GW * g;
    for (u32 first = 0; first < nb; first += C::IPP) {
        const u32 j = first + g;
        const int src = (int)(min(j, nb - 1u) << 2);
        const u32 rg = (u32)__builtin_amdgcn_ds_bpermute(src, (int)my_g);
        const u32 nk = (u32)__builtin_amdgcn_ds_bpermute(src, (int)my_nkeep);  // 0 unless PLAIN with kept entries
        const u32 solo = (u32)__builtin_amdgcn_ds_bpermute(src, (int)(u32)my_so);
        const u32 sohi = (u32)__builtin_amdgcn_ds_bpermute(src, (int)(u32)(my_so >> 32));
        const bool active = g < C::IPP && j < nb && 32u * s < nk;
        if (!active) continue;
        const u8 *lp = A.seq + ((u64)solo | ((u64)sohi << 32)) + 32u * s;
        const u8 *ap = A.bases + ((u64)rg + 32u * s);
        const uint4 Wa = load16_unaligned(lp), Wb = load16_unaligned(lp + 16);
        const uint4 Aa = load16_unaligned(ap), Ab = load16_unaligned(ap + 16);
        // bit i <=> byte i of this lane differs from the assembly; bytes past the kept entries drop out
        u32 D = nz_mask4(Wa.x ^ Aa.x) | (nz_mask4(Wa.y ^ Aa.y) << 4) | (nz_mask4(Wa.z ^ Aa.z) << 8) | (nz_mask4(Wa.w ^ Aa.w) << 12) |
                (nz_mask4(Wb.x ^ Ab.x) << 16) | (nz_mask4(Wb.y ^ Ab.y) << 20) | (nz_mask4(Wb.z ^ Ab.z) << 24) |
                (nz_mask4(Wb.w ^ Ab.w) << 28);
        const u32 b1 = min(nk - 32u * s, 32u);  // >= 1 here
        D &= 0xFFFFFFFFu >> (32u - b1);
        if (D) {  // one EVENT per differing base (about one lane in sixteen has any)
            u32 slot = atomicAdd(S.n, (u32)__popc(D));
            const u32 P0 = rg + 32u * s, idx = (u32)(base + j);
            do {
                const u32 i = (u32)__ffs((int)D) - 1u;
                D &= D - 1u;
                // byte i of the lane's eight dwords, by a select tree on the bits of i (no memory round trip)
                const u32 m4 = (u32)(((int)(i << 29)) >> 31), m8 = (u32)(((int)(i << 28)) >> 31), m16 = (u32)(((int)(i << 27)) >> 31);
#define PP_SEL(m, b, a) (((m) & (b)) | (~(m) & (a)))
                const u32 w01 = PP_SEL(m4, Wa.y, Wa.x), w23 = PP_SEL(m4, Wa.w, Wa.z);
                const u32 w45 = PP_SEL(m4, Wb.y, Wb.x), w67 = PP_SEL(m4, Wb.w, Wb.z);
                const u32 wlo = PP_SEL(m8, w23, w01), whi = PP_SEL(m8, w67, w45);
                const u32 c = (PP_SEL(m16, whi, wlo) >> (8u * (i & 3u))) & 0xFFu;
#undef PP_SEL
                const u32 p = P0 + i;
                stage_at(A, S, slot++, unit_event(idx, p & (u32)(TILE - 1), (u32)row_of(c), 0), p >> 11);
            } while (D);
        }
    }
}

__global__ __launch_bounds__(STREAM_THREADS, PP_STREAM_MINW) void k_stream(StreamArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char stream_smem[];
    __shared__ u32 s_n, s_next, s_seg;
    __shared__ u64 s_base;
    Stage S;
    S.units = (u64 *)stream_smem;
    u32 *const hist2 = (u32 *)(S.units + A.stage_cap);   // two arrays of bucket counts, used in turn
    S.hist = hist2;
    S.bkt = (unsigned short *)(hist2 + 2u * A.nbk);
    S.n = &s_n;
    S.cap = A.stage_cap;
    const u32 tid = threadIdx.x, lane = tid & 63u;
    for (u32 b = tid; b < 2u * A.nbk; b += STREAM_THREADS) hist2[b] = 0;
    const u64 lo = (u64)blockIdx.x * A.chunk, hi = min(A.n, lo + A.chunk);
    const u32 flush_at = A.stage_cap > STREAM_HEADROOM ? A.stage_cap - STREAM_HEADROOM : 0u;
    if (tid == 0) {
        s_n = 0;
        s_next = 0;
    }
    __syncthreads();
#ifdef PP_STREAM_PROFILE
    u64 pacc[6] = {0, 0, 0, 0, 0, 0}, plast = 0;
#define PP_STAMP(k) do { const u64 t_ = clock64(); if (k) pacc[k] += t_ - plast; plast = t_; } while (0)
#else
#define PP_STAMP(k) do { } while (0)
#endif
    // The waves of the block take batches of STREAM_BATCH records from a shared counter and only meet (barrier) when a
    // segment is due or the chunk is used up: between two rendezvous they run freely, and at a rendezvous they are at
    // most one batch apart.
    const u32 nbatch = (u32)((hi > lo ? hi - lo : 0ull) + STREAM_BATCH - 1u) / STREAM_BATCH;
    u32 cur = 0;
    for (;;) {
        for (;;) {
            PP_STAMP(0);
            if (*(volatile u32 *)S.n >= flush_at) break;  // segment due (a stale read only delays the rendezvous by a batch)
            u32 bi = 0;
            if (lane == 0) bi = atomicAdd(&s_next, 1u);
            bi = (u32)__builtin_amdgcn_readfirstlane((int)bi);
            if (bi >= nbatch) break;
            const u64 base = lo + (u64)bi * STREAM_BATCH;
            const u32 nb = (u32)min((u64)STREAM_BATCH, hi - base);
            const RecFields F = stream_fields(A, base + lane);
            // ---- one record per lane: class, trim, units ----
            const u32 cc = min(F.c, A.n_contigs - 1u);
            const u64 c_lo = A.contig_off[cc], c_hi = A.contig_off[cc + 1];
            const bool valid = lane < nb;
            const u32 op0 = (valid && F.nc) ? A.cigar[F.co] : 0u;
            const bool tail_ok = valid && F.sl >= 4u && F.so + F.sl <= A.seq_bytes;
#ifdef PP_EXP_NOTAIL
            const u32 tail = tail_ok ? 0x41434754u : 0u;
#else
            const u32 tail = tail_ok ? load4_unaligned(A.seq + F.so + (F.sl - 4u)) : 0u;
#endif
            const RecInfo my = stream_classify(A, base + lane, valid, F, c_lo, c_hi, op0);
            const u32 idx = (u32)(base + lane);
            u32 nkeep = 0;  // PLAIN: kept entries after the trim (alignment.rs:364-378)
            if (my.kind == 1u) {
                // index of the last base that differs from the last base, read off the last four bases; a trailing
                // homopolymer of four or more walks left byte by byte (rare)
                const u8 *rp = A.seq + my.so;
                const u32 last = tail >> 24;
                const u32 tf = nz_flags(tail ^ splat8(last));
                if (tf) {
                    nkeep = my.len - 4u + (u32)((31 - __clz((int)tf)) >> 3);
                } else {
                    u32 i = my.len - 4u;
                    while (i > 0 && rp[i - 1] == (u8)last) i--;
                    nkeep = i > 0 ? i - 1u : 0u;
                }
            }
            PP_STAMP(1);
            const u32 span = my.kind == 1u ? nkeep : (my.kind == 2u ? my.len : 0u);  // entries that reach the pileup
            const u32 w0 = my.g >> 11, w1 = span ? (u32)(((u64)my.g + span - 1ull) >> 11) : w0;
            if (my.kind == 2u) A.nkeep_arr[idx] = my.len;
            if (span) {
                if (w1 - w0 < STREAM_DIRECT_WINDOWS) {
                    u32 slot = atomicAdd(S.n, w1 - w0 + 1u);
                    for (u32 w = w0; w <= w1; w++)  // a PLAIN unit carries its start relative to the window it goes to
                        stage_at(A, S, slot++, my.kind == 1u ? unit_plain(idx, (int)(my.g - (w << 11)), nkeep, my.nd != 0, 0)
                                                              : unit_slow(idx, my.flags, 0), w);
                } else {
                    late_put(A, w0, w1, unit_slow(idx, my.flags, 0));  // SLOW only: a PLAIN read spans two windows at most
                }
            }
            // ---- the comparison with the assembly: the longest kept stretch of the batch picks the lane-group width:
            // 5 lanes x 32 B up to 160 bases (12 reads per pass), 6 up to 192 (10), 8 up to 252 (8)
            PP_STAMP(2);
            u32 longest = nkeep;
            for (int o = 32; o > 0; o >>= 1) longest = max(longest, (u32)__shfl_xor((int)longest, o, 64));
            if (longest) {
                if (longest <= PlainCfg<5>::MAXL) stream_plain_passes<5>(A, S, base, nb, lane, my.g, my.so, nkeep);
                else if (longest <= PlainCfg<6>::MAXL) stream_plain_passes<6>(A, S, base, nb, lane, my.g, my.so, nkeep);
                else stream_plain_passes<8>(A, S, base, nb, lane, my.g, my.so, nkeep);
            }
            PP_STAMP(3);
        }
        __syncthreads();  // rendezvous: nobody is inside a batch
        PP_STAMP(4);
        const bool last = *(volatile u32 *)&s_next >= nbatch;
        cur ^= 1u;
        stream_flush(A, S, hist2 + cur * A.nbk, &s_base, &s_seg);
        S.hist = hist2 + cur * A.nbk;
        PP_STAMP(5);
        if (last) break;
    }
#ifdef PP_STREAM_PROFILE
    if (lane == 0) for (int q = 1; q < 6; q++) atomicAdd(&A.prof[q], pacc[q]);
#endif
}

}  // namespace pp
