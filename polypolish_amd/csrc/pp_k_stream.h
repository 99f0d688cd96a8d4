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
constexpr u32 STREAM_THREADS = 1024;
constexpr u32 STREAM_WAVES = STREAM_THREADS / 64;
constexpr u32 STREAM_BATCH = 60;          // records per wave and block iteration (a multiple of every group count)
constexpr u32 STREAM_HEADROOM = 2304;     // staged units a block iteration may add before the segment is flushed
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

// Sort the staged units by bucket and write them out as one segment.  Called by every thread of the block
// (after a __syncthreads()); leaves the staging empty.
__device__ void stream_flush(const StreamArgs &A, const Stage &S, u32 *s_wtot, u64 *s_base, u32 *s_seg) {
    const u32 tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const u32 n = min(*S.n, S.cap);
    if (n == 0) {
        __syncthreads();
        if (tid == 0) *S.n = 0;  // everything offered went to the late list
        __syncthreads();
        return;
    }
    // exclusive scan of the bucket counts: PER consecutive buckets per thread, wave scan, wave totals
    const u32 per = (A.nbk + STREAM_THREADS - 1u) / STREAM_THREADS;
    const u32 b0 = min(A.nbk, tid * per), b1 = min(A.nbk, b0 + per);
    u32 sum = 0;
    for (u32 b = b0; b < b1; b++) sum += S.hist[b];
    u32 inc = sum;
    for (int o = 1; o < 64; o <<= 1) {
        const u32 v = (u32)__shfl_up((int)inc, o, 64);
        if ((int)lane >= o) inc += v;
    }
    if (lane == 63u) s_wtot[wave] = inc;
    if (tid == 0) {
        const u64 base = atomicAdd(A.unit_cursor, (u64)n);
        const u32 seg = atomicAdd(A.seg_cursor, 1u);
        const bool ok = base + n <= A.cap_units && seg < A.cap_segs;
        if (!ok) report(A.status, base + n, DE_CAPACITY);  // the cursors keep counting what a rerun needs
        *s_base = ok ? base : ~0ull;
        *s_seg = seg;
    }
    __syncthreads();
    u32 run = inc - sum;
    for (u32 v = 0; v < wave; v++) run += s_wtot[v];
    const u64 base = *s_base;
    const bool ok = base != ~0ull;
    unsigned short *row = A.seg_off + (u64)(*s_seg) * (A.nbk + 1u);
    for (u32 b = b0; b < b1; b++) {
        const u32 c = S.hist[b];
        S.hist[b] = run;  // cursor of the bucket
        if (ok) row[b] = (unsigned short)run;
        run += c;
    }
    if (ok && tid == 0) {
        row[A.nbk] = (unsigned short)n;
        A.seg_base[*s_seg] = base;
    }
    __syncthreads();
    if (ok) {
        for (u32 i = tid; i < n; i += STREAM_THREADS) {
            const u32 p = atomicAdd(&S.hist[S.bkt[i]], 1u);
            A.units[base + p] = S.units[i];
        }
    }
    __syncthreads();
    for (u32 b = tid; b < A.nbk; b += STREAM_THREADS) S.hist[b] = 0;
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

__device__ __forceinline__ RecInfo stream_classify(const StreamArgs &A, u64 a, bool valid) {
    RecInfo R;
    R.g = 0; R.len = 0; R.kind = 0; R.flags = 0; R.nd = 0; R.so = 0;
    const u64 ai = min(a, A.n - 1ull);  // unconditional loads, clamped
    const u32 c = A.contig[ai], nc = A.n_cig[ai], sl = A.seq_len[ai], rs = A.ref_start[ai], k = A.kk[ai];
    const u64 co = A.cig_off[ai], so = A.seq_off[ai];
    const u32 cc = min(c, A.n_contigs - 1u);
    const u64 c_lo = A.contig_off[cc], c_hi = A.contig_off[cc + 1];
    const u32 *cg = A.cigar + co;
    const u32 op0 = (valid && nc) ? cg[0] : 0u;
    if (!valid) return R;
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

// ---- PLAIN passes: GW lanes per read, 32 read bytes per lane, compared with the assembly bytes at the same positions
template <int GW>
__device__ __forceinline__ void stream_plain_passes(const StreamArgs &A, const Stage &S, u64 base, u32 nb, u32 lane,
                                                    const RecInfo &my) {
    typedef PlainCfg<GW> C;
    const u32 g = C::group(lane), s = lane - (u32)GW * g;
    const u32 my_w1 = my.len | (my.nd << 8) | ((my.kind == 1u ? 1u : 0u) << 9);
    const u64 below = (1ull << lane) - 1ull;
    for (u32 first = 0; first < nb; first += C::IPP) {
        const u32 j = first + g;
        const int src = (int)(min(j, nb - 1u) << 2);
        const u32 rg = (u32)__builtin_amdgcn_ds_bpermute(src, (int)my.g);
        const u32 w1 = (u32)__builtin_amdgcn_ds_bpermute(src, (int)my_w1);
        const u32 solo = (u32)__builtin_amdgcn_ds_bpermute(src, (int)(u32)my.so);
        const u32 sohi = (u32)__builtin_amdgcn_ds_bpermute(src, (int)(u32)(my.so >> 32));
        const u32 L = w1 & 0xFFu;
        const bool plain = g < C::IPP && j < nb && ((w1 >> 9) & 1u);
        const bool active = plain && 32u * s < L;
        const u8 *rp = A.seq + ((u64)solo | ((u64)sohi << 32));
        uint4 Wa = make_uint4(0, 0, 0, 0), Wb = Wa, Aa = Wa, Ab = Wa;
        u32 tail = 0;
        if (plain) tail = load4_unaligned(rp + (L - 4u));
        if (active) {
            const u8 *lp = rp + 32u * s, *ap = A.bases + ((u64)rg + 32u * s);
            Wa = load16_unaligned(lp);
            Wb = load16_unaligned(lp + 16);
            Aa = load16_unaligned(ap);
            Ab = load16_unaligned(ap + 16);
        }
        // ---- trim (alignment.rs:364-378): kept entries = index of the last base that differs from the last base
        const u32 last = tail >> 24;
        const u32 tf = nz_flags(tail ^ splat8(last));
        int nkeep = (int)L - 4 + ((31 - __clz((int)tf)) >> 3);
        if (plain && tf == 0) {  // rare: a homopolymer of four or more at the end, walk left
            u32 i = L - 4u;
            while (i > 0 && rp[i - 1] == (u8)last) i--;
            nkeep = i > 0 ? (int)i - 1 : 0;
        }
        const bool live = plain && nkeep > 0;
        // ---- PLAIN units: one per overlapped window (one or two), written by the group's first lane
        const u32 wA = rg >> 11, wB = (rg + (u32)max(nkeep, 1) - 1u) >> 11;
        const bool e1 = live && s == 0, e2 = e1 && wB != wA;
        const u64 m1 = __ballot(e1), m2 = __ballot(e2);
        // ---- compare: bit i <=> byte i of this lane differs from the assembly; bytes past the kept entries drop out
        u32 D = 0;
        if (live && active) {
            D = nz_mask4(Wa.x ^ Aa.x) | (nz_mask4(Wa.y ^ Aa.y) << 4) | (nz_mask4(Wa.z ^ Aa.z) << 8) | (nz_mask4(Wa.w ^ Aa.w) << 12) |
                (nz_mask4(Wb.x ^ Ab.x) << 16) | (nz_mask4(Wb.y ^ Ab.y) << 20) | (nz_mask4(Wb.z ^ Ab.z) << 24) |
                (nz_mask4(Wb.w ^ Ab.w) << 28);
            const int b1 = min(max(nkeep - (int)(32u * s), 0), 32);
            D = b1 > 0 ? (D & (0xFFFFFFFFu >> (32 - b1))) : 0u;
        }
        const u32 ne = (u32)__popc(D);
        const u32 idx = (u32)(base + j);
        if (m1) {
            u32 slot0 = 0;
            if (lane == 0) slot0 = atomicAdd(S.n, (u32)__popcll(m1) + (u32)__popcll(m2));
            slot0 = (u32)__builtin_amdgcn_readfirstlane((int)slot0);
            if (e1) stage_at(A, S, slot0 + (u32)__popcll(m1 & below), unit_plain(idx, (int)(rg - (wA << 11)), (u32)nkeep, (w1 >> 8) & 1u, 0), wA);
            if (e2) stage_at(A, S, slot0 + (u32)__popcll(m1) + (u32)__popcll(m2 & below),
                             unit_plain(idx, (int)rg - (int)(wB << 11), (u32)nkeep, (w1 >> 8) & 1u, 0), wB);
        }
        if (ne) {  // one EVENT per differing base (about one lane in sixteen has any)
            u32 slot = atomicAdd(S.n, ne);
            const u32 P0 = rg + 32u * s;
            while (D) {
                const int i = __ffs((int)D) - 1;
                D &= D - 1u;
                // byte i of the lane's eight dwords, by a select tree on the bits of i (no memory access)
                const u32 m4 = (u32)(((int)((u32)i << 29)) >> 31), m8 = (u32)(((int)((u32)i << 28)) >> 31),
                          m16 = (u32)(((int)((u32)i << 27)) >> 31);
#define PP_SEL(m, b, a) (((m) & (b)) | (~(m) & (a)))
                const u32 w01 = PP_SEL(m4, Wa.y, Wa.x), w23 = PP_SEL(m4, Wa.w, Wa.z);
                const u32 w45 = PP_SEL(m4, Wb.y, Wb.x), w67 = PP_SEL(m4, Wb.w, Wb.z);
                const u32 wlo = PP_SEL(m8, w23, w01), whi = PP_SEL(m8, w67, w45);
                const u32 c = (PP_SEL(m16, whi, wlo) >> (8 * (i & 3))) & 0xFFu;
#undef PP_SEL
                const u32 p = P0 + (u32)i;
                stage_at(A, S, slot++, unit_event(idx, p & (u32)(TILE - 1), (u32)row_of(c), 0), p >> 11);
            }
        }
    }
}

__global__ __launch_bounds__(STREAM_THREADS, 8) void k_stream(StreamArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char stream_smem[];
    __shared__ u32 s_n, s_wtot[STREAM_WAVES], s_seg;
    __shared__ u64 s_base;
    Stage S;
    S.units = (u64 *)stream_smem;
    S.hist = (u32 *)(S.units + A.stage_cap);
    S.bkt = (unsigned short *)(S.hist + A.nbk);
    S.n = &s_n;
    S.cap = A.stage_cap;
    const u32 tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    for (u32 b = tid; b < A.nbk; b += STREAM_THREADS) S.hist[b] = 0;
    if (tid == 0) s_n = 0;
    __syncthreads();
    const u64 lo = (u64)blockIdx.x * A.chunk, hi = min(A.n, lo + A.chunk);
    const u32 flush_at = S.cap > STREAM_HEADROOM ? S.cap - STREAM_HEADROOM : 0u;
    for (u64 it = lo; it < hi; it += (u64)STREAM_WAVES * STREAM_BATCH) {
        const u64 base = it + (u64)wave * STREAM_BATCH;
        const u32 nb = base < hi ? (u32)min((u64)STREAM_BATCH, hi - base) : 0u;
        if (nb) {
            const RecInfo my = stream_classify(A, base + lane, lane < nb);
            // the longest PLAIN read of the batch picks the lane-group width: 5 lanes x 32 B up to 160 bases
            // (12 reads per pass), 6 up to 192 (10), 8 up to 252 (8)
            u32 longest = my.kind == 1u ? my.len : 0u;
            for (int o = 32; o > 0; o >>= 1) longest = max(longest, (u32)__shfl_xor((int)longest, o, 64));
            if (longest) {
                if (longest <= PlainCfg<5>::MAXL) stream_plain_passes<5>(A, S, base, nb, lane, my);
                else if (longest <= PlainCfg<6>::MAXL) stream_plain_passes<6>(A, S, base, nb, lane, my);
                else stream_plain_passes<8>(A, S, base, nb, lane, my);
            }
            if (my.kind == 2u) {  // SLOW: one unit per overlapped window, or one late entry for a long span
                const u32 idx = (u32)(base + lane);
                A.nkeep_arr[idx] = my.len;
                const u32 w0 = my.g >> 11, w1 = (u32)(((u64)my.g + my.len - 1ull) >> 11);
                const u64 unit = unit_slow(idx, my.flags, 0);
                if (w1 - w0 < STREAM_DIRECT_WINDOWS) {
                    u32 slot = atomicAdd(S.n, w1 - w0 + 1u);
                    for (u32 w = w0; w <= w1; w++) stage_at(A, S, slot++, unit, w);
                } else {
                    late_put(A, w0, w1, unit);
                }
            }
        }
        __syncthreads();
        if (s_n > flush_at || it + (u64)STREAM_WAVES * STREAM_BATCH >= hi) stream_flush(A, S, s_wtot, &s_base, &s_seg);
    }
}

}  // namespace pp
