// pp_k_bucket.h -- k_count / k_scan_cols / k_scan / k_fill / k_regroup: multisplit of the records into 2048-position windows.
// Part of pp_kernels.hip (included there, in this order, and nowhere else: it defines __global__ kernels).
#pragma once

namespace pp {

// =============================================================================================
// bucketing: count -> scan -> fill (no global atomics; LDS histograms per block and window range)
// =============================================================================================
// Two-level multisplit of the (alignment, window) items.  Level 1 scatters the items into COARSE buckets of
// COARSE_WINDOWS windows: a block's items for one coarse bucket form one contiguous run (full-line writes),
// where a direct scatter into the windows would be 16-byte writes all over HBM (and, on a 250 Mbp job, all over
// more pages than the TLB holds).  Level 2 (k_regroup) sorts a coarse bucket into its windows inside a region
// small enough to stay in L2, and that is also where the windows' counts and offsets come from.
//   k_count     per-block LDS histogram over the windows -> the block's per-coarse-bucket counts
//   k_scan_cols column scan over the blocks of the coarse counts; k_scan: offsets of the coarse buckets
//   k_fill      items -> coarse buckets (LDS cursors);  k_regroup  coarse bucket -> windows (+ win_off)
template <int CW>  // windows per coarse bucket; 1 = single level (k_fill writes the windows directly)
__global__ __launch_bounds__(1024) void k_count(u64 n, u64 chunk, const u32 *__restrict__ gstart,
                                                const u32 *__restrict__ nkeep, u32 nwin, u32 ncoarse,
                                                u32 *__restrict__ hist_c) {
    __shared__ u32 h[COUNT_RANGE];
    u32 range_lo = blockIdx.y * (u32)COUNT_RANGE;
    u32 range_n = min((u32)COUNT_RANGE, nwin - range_lo);
    for (u32 i = threadIdx.x; i < (u32)COUNT_RANGE; i += blockDim.x) h[i] = 0;
    __syncthreads();
    u64 lo = (u64)blockIdx.x * chunk, hi = min(n, lo + chunk);
    for (u64 a0 = lo + threadIdx.x; a0 < hi; a0 += 4ull * blockDim.x) {
        u32 nk[4], g[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {  // four records per trip: their loads are in flight together
            const u64 a = a0 + (u64)u * blockDim.x;
            nk[u] = a < hi ? nkeep[a] : 0u;
            g[u] = a < hi ? gstart[a] : 0u;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (!nk[u]) continue;
            for_each_piece(g[u], nk[u], [&](u32, u32 gp, u32 sp) {
                if (!sp) return;
                u32 w0 = gp / (u32)TILE, w1 = (gp + sp - 1u) / (u32)TILE;
                u32 wa = max(w0, range_lo), wb = min(w1, range_lo + range_n - 1u);
                for (u32 w = wa; w <= wb && w >= wa; w++) atomicAdd(&h[w - range_lo], 1u);
            });
        }
    }
    __syncthreads();
    const u32 c_lo = range_lo / (u32)CW, c_n = (range_n + CW - 1u) / (u32)CW;
    for (u32 c = threadIdx.x; c < c_n; c += blockDim.x) {
        u32 sum = 0;
#pragma unroll
        for (int j = 0; j < CW; j++) sum += h[c * CW + j];  // rows past range_n are zero
        hist_c[(u64)blockIdx.x * ncoarse + c_lo + c] = sum;
    }
}

// heavy-window list (see HEAVY_SLOTS): heavy[0] counts the windows of at least `heavy_min` items, the first HEAVY_SLOTS
// of them are listed in heavy[1..] and marked in win_heavy (slot + 1)
__device__ __forceinline__ void note_heavy(u32 w, u32 cnt, u32 heavy_min, u32 *heavy, u8 *win_heavy) {
    u32 mark = 0;
    if (cnt >= heavy_min) {
        const u32 slot = atomicAdd(&heavy[0], 1u);
        if (slot < HEAVY_SLOTS) { heavy[1 + slot] = w; mark = slot + 1u; }
    }
    win_heavy[w] = (u8)mark;
}

// two-level path: the windows' offsets come from k_regroup
__global__ __launch_bounds__(256) void k_heavy(u32 nwin, const u32 *__restrict__ win_off, u32 heavy_min,
                                               u32 *__restrict__ heavy, u8 *__restrict__ win_heavy) {
    const u32 w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w < nwin) note_heavy(w, win_off[w + 1] - win_off[w], heavy_min, heavy, win_heavy);
}

// win_heavy != nullptr (single-level path, the columns ARE the windows): also lists the heavy windows.
// One wave per column, eight rows per lane (<= 512 rows: PP_NB_MAX), all of a lane's loads in flight at once.  (Round 4 tried
// the row-wise variant -- a wave reads 64 adjacent words of a row, the rows in sequence: 5 MB fetched instead of 40, but 32
// dependent trips per wave: 26 us instead of 18.)
__global__ __launch_bounds__(256) void k_scan_cols(u32 nwin, u32 nblocks, u32 *__restrict__ hist,
                                                   u32 *__restrict__ win_cnt, u32 heavy_min, u32 *__restrict__ heavy,
                                                   u8 *__restrict__ win_heavy) {
    const u32 w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63u;
    if (w >= nwin) return;
    u32 v[8];
    u32 sum = 0;
#pragma unroll
    for (u32 i = 0; i < 8; i++) {
        const u32 b = 8u * lane + i;
        v[i] = (b < nblocks) ? hist[(u64)b * nwin + w] : 0u;
        sum += v[i];
    }
    u32 inc = sum;
    for (int o = 1; o < 64; o <<= 1) {
        const u32 t = __shfl_up(inc, o, 64);
        if ((int)lane >= o) inc += t;
    }
    u32 run = inc - sum;
#pragma unroll
    for (u32 i = 0; i < 8; i++) {
        const u32 b = 8u * lane + i;
        if (b < nblocks) hist[(u64)b * nwin + w] = run;
        run += v[i];
    }
    if (lane == 63) {
        win_cnt[w] = inc;
        if (win_heavy) note_heavy(w, inc, heavy_min, heavy, win_heavy);
    }
}

// single-block exclusive scan: out[i] = sum(in[0..i)), out[n] = total
// n_ptr (optional) overrides n with a count held on the device; the total is also stored to *total_out;
// a total above `limit` (capacity of the buffer the offsets index into) aborts the job with DE_CAPACITY.
// Tiles of 16384 values, sixteen per thread in four 16-byte loads that are all out before the first is used: coalesced,
// and 8 trips for the 122 K windows of a 250 Mbp job (a trip is a round trip to memory, the wave scans and two barriers:
// ~2 us; with four values per thread the same job took 30 of them, and a thread walking its own 120-value stretch 114 us).
template <typename T>
__global__ __launch_bounds__(1024) void k_scan(const u32 *__restrict__ in, u64 n, const u32 *__restrict__ n_ptr,
                                               T *__restrict__ out, u64 *__restrict__ total_out, u64 limit,
                                               u64 *status) {
    __shared__ u64 wsum[16];
    if (*status != ~0ull) return;
    if (n_ptr) n = *n_ptr;
    const u32 t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    constexpr int Q = 4;  // 16-byte loads per thread and trip
    u64 carry = 0;
    for (u64 base = 0; base < n; base += 4096ull * Q) {
        const u64 i0 = base + 4ull * Q * t;
        u32 v[4 * Q];
#pragma unroll
        for (int q = 0; q < Q; q++) {
            const u64 i = i0 + 4ull * q;
            if (i + 4 <= n) {
                const uint4 x = *(const uint4 *)(in + i);
                v[4 * q] = x.x; v[4 * q + 1] = x.y; v[4 * q + 2] = x.z; v[4 * q + 3] = x.w;
            } else {
#pragma unroll
                for (int k = 0; k < 4; k++) v[4 * q + k] = i + k < n ? in[i + k] : 0u;
            }
        }
        u64 s = 0;
#pragma unroll
        for (int k = 0; k < 4 * Q; k++) s += v[k];
        u64 inc = s;
        for (int o = 1; o < 64; o <<= 1) {
            const u64 x = (u64)__shfl_up((long long)inc, o, 64);
            if ((int)lane >= o) inc += x;
        }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        u64 before = carry + inc - s, tile = 0;
        for (u32 i = 0; i < 16; i++) {
            const u64 ws = wsum[i];
            if (i < wave) before += ws;
            tile += ws;
        }
#pragma unroll
        for (int k = 0; k < 4 * Q; k++) {
            if (i0 + k < n) out[i0 + k] = (T)before;
            before += v[k];
        }
        carry += tile;
        __syncthreads();
    }
    if (t == 0) {
        const u64 total = carry;
        out[n] = (T)total;
        if (total_out) *total_out = total;
        if (sizeof(T) == 4 && total > 0xFFFFFFFFull) report(status, 0, DE_OVERFLOW);
        else if (total > limit) report(status, total, DE_CAPACITY);
    }
}

#ifndef PP_FILL_NT
#define PP_FILL_NT 0
#endif
// WO: the records come through the window-order mirror (gstart / nkeep are in mirror order, see k_prep): k, seq_off,
// seq_len and the file index -- the item's `w` -- are read from it, and the slot of a record's FIRST item is taken once per
// wave and column (ballot, one returning LDS atomic by the leader): a wave's records mostly go to one window.
template <int CW, bool WO>
__global__ __launch_bounds__(1024) void k_fill(u64 n, u64 chunk, const pp_wo_rec *__restrict__ wo, const u32 *__restrict__ gstart,
                                               const u32 *__restrict__ nkeep,
                                               const u32 *__restrict__ kk,
                                               const u64 *__restrict__ seq_off,
                                               const u32 *__restrict__ seq_len, u32 nwin, u32 ncoarse,
                                               const u32 *__restrict__ hist_c,
                                               const u32 *__restrict__ coarse_off,
                                               uint4 *__restrict__ entB, u32 frange, u64 seq_bytes, u64 *__restrict__ status) {
    __shared__ u32 cur[COUNT_RANGE];  // cursors of the coarse buckets of this pass
    PP_STAMP(1, 0);
    if (*status != ~0ull) return;  // a record error, or the work-item buffer is too small (host reruns)
    // blockIdx.y: the pass -- frange (<= COUNT_RANGE) columns each.  The blocks of one pass are dispatched together (x runs
    // fastest), so the lines a pass writes to are few enough to stay in the L2s until they are full (see run_pipeline).
    const u32 crange_lo = blockIdx.y * frange;
    const u32 crange_n = min(frange, ncoarse - crange_lo);
    for (u32 i = threadIdx.x; i < crange_n; i += blockDim.x)
        cur[i] = coarse_off[crange_lo + i] + hist_c[(u64)blockIdx.x * ncoarse + crange_lo + i];
    __syncthreads();
    PP_STAMP(1, 1);
    const u32 range_lo = crange_lo * (u32)CW;                       // the same range, in windows
    const u32 range_n = min(crange_n * (u32)CW, nwin - range_lo);
    u64 lo = (u64)blockIdx.x * chunk, hi = min(n, lo + chunk);
    const u64 a_end = WO ? lo + (hi - lo + 4ull * blockDim.x - 1) / (4ull * blockDim.x) * (4ull * blockDim.x) : hi;  // (WO: whole waves, for the ballots)
    for (u64 a0 = lo + threadIdx.x; a0 < a_end; a0 += 4ull * blockDim.x) {
        u32 nk4[4], g4[4], k4[4], sl4[4], fi4[4];
        u64 so4[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {  // four records per trip: their loads are in flight together
            const u64 a = a0 + (u64)u * blockDim.x;
            const bool ok = a < hi;
            nk4[u] = ok ? nkeep[a] : 0u;
            g4[u] = ok ? gstart[a] : 0u;
            if (WO) {  // the mirror's 32-byte record as two 16-byte loads
                uint4 qa = make_uint4(0, 0, 1, 0), qb = make_uint4(0, 0, 0, 0);
                if (ok) { qa = ((const uint4 *)wo)[2 * a]; qb = ((const uint4 *)wo)[2 * a + 1]; }
                k4[u] = qa.z; sl4[u] = qa.w; so4[u] = (u64)qb.x | ((u64)qb.y << 32); fi4[u] = qb.w;
            } else {
                k4[u] = ok ? kk[a] : 1u;
                so4[u] = ok ? seq_off[a] : 0ull;
                sl4[u] = ok ? seq_len[a] : 0u;
                fi4[u] = (u32)a;
            }
            if (ok && blockIdx.y == 0) {  // checks that need k / seq_off (not read by k_prep's fast path)
                if (k4[u] == 0) report(status, fi4[u], DE_BAD_K);
                else if (so4[u] + sl4[u] > (1ull << 40)) report(status, fi4[u], DE_OVERFLOW);
                else if (so4[u] + sl4[u] > seq_bytes) report(status, fi4[u], DE_SEQ_RANGE);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const u32 word = nk4[u];
            // WO: the slots of the first two items of the record's first piece (its windows in this pass: one, or two for
            // a read across a window boundary), one cursor update per wave and column each
            u32 pre_slot[2] = {0, 0}, pre_n = 0, pre_used = 0;
            if (WO) {
                u32 key[2] = {0xFFFFFFFFu, 0xFFFFFFFFu};
                if (word) {
                    const u32 sp0 = (word >> 30) == NKW_INDEL1 ? (((word >> 17) & 1u) ? ((word >> 9) & 0xFFu) : ((word >> 9) & 0xFFu) - 1u)
                                                               : (word & 0x3FFFFFFFu);  // span of the first piece (for_each_piece)
                    const u32 w0 = g4[u] / (u32)TILE, w1 = (g4[u] + sp0 - 1u) / (u32)TILE;
                    const u32 wa = max(w0, range_lo), wb = min(w1, range_lo + range_n - 1u);
                    if (sp0 && wa <= wb) {
                        key[0] = wa / (u32)CW - crange_lo;
                        if (wa < wb) key[1] = (wa + 1u) / (u32)CW - crange_lo;
                    }
                }
                const u32 lane = threadIdx.x & 63u;
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    u32 mine = key[q];
                    for (;;) {
                        const u64 todo = __ballot(mine != 0xFFFFFFFFu);
                        if (!todo) break;
                        const int lead = __ffsll((long long)todo) - 1;
                        const u32 kl = (u32)__builtin_amdgcn_readlane((int)mine, lead);
                        const u64 same = __ballot(mine == kl);
                        u32 base = 0;
                        if ((int)lane == lead) base = atomicAdd(&cur[kl], (u32)__popcll(same));
                        base = (u32)__builtin_amdgcn_readlane((int)base, lead);
                        if (mine == kl) {
                            pre_slot[q] = base + (u32)__popcll(same & ((1ull << lane) - 1ull));
                            pre_n = (u32)q + 1u;
                            mine = 0xFFFFFFFFu;
                        }
                    }
                }
            }
            if (!word) continue;
            const u64 a = fi4[u];  // the record's index in file order: the item's `w`
            const u32 kc = kclass_of(k4[u]);
            const u32 cls = word >> 30;
            const u32 ia = (word >> 9) & 0xFFu, idel = (word >> 17) & 1u;  // a one-indel read: run length in front, kind
            for_each_piece(g4[u], word, [&](u32 piece, u32 g, u32 sp) {
                if (!sp) return;
                u32 w0 = g / (u32)TILE, w1 = (g + sp - 1u) / (u32)TILE;
                u32 wa = max(w0, range_lo), wb = min(w1, range_lo + range_n - 1u);
                if (wa > wb) return;
                // work item, 16 bytes (bits 18..23 of y carry the window's index inside its coarse bucket until
                // k_regroup has used it):
                //   x  fast class: seq offset bits 0..31          otherwise: kept entries (trim done by k_prep)
                //   y  [7:0] seq offset bits 32..39 (fast) | [15:8] depth-share class | [17:16] class flags | [31:24] length (fast)
                //   z  [29:0] global start of the piece minus the window start (signed) | [30] ENT_NOTRIM | [31] ENT_POINT
                //   w  record index (file order)
                // "fast" = a read without indels (its bytes are the entries, trimmed in k_tile) and the pieces of a
                // one-indel read: the flank in front (its end is not the read's: ENT_NOTRIM), the entry at the indel
                // (ENT_POINT: x/y point at its key bytes, the length field holds their number -- 2, or 0 for a deletion)
                // and the flank behind.
                u64 so = so4[u];
                u32 len = sp, fl = 0, zf = 0;
                if (cls == NKW_INDEL1) {
                    if (piece == 0u) zf = 1u;                                   // ENT_NOTRIM
                    else if (piece == 1u) { zf = 2u; so += ia - 1u + idel; len = idel ? 0u : 2u; }  // ENT_POINT
                    else so += idel ? ia : ia + 1u;
                } else fl = cls;
                for (u32 w = wa; w <= wb && w >= wa; w++) {
                    u32 slot;
                    if (WO && piece == 0u && pre_used < pre_n) slot = pre_slot[pre_used++];  // (taken with the wave, above)
                    else slot = atomicAdd(&cur[w / (u32)CW - crange_lo], 1u);
                    uint4 e;
                    e.x = fl ? sp : (u32)so;
                    e.y = (fl ? 0u : (((u32)(so >> 32) & 0xFFu) | (len << 24))) | (kc << 8) | (fl << 16) |
                          ((w % (u32)CW) << 18);
                    e.z = ((u32)(int)((long long)g - (long long)w * TILE) & 0x3FFFFFFFu) | (zf << 30);
                    e.w = (u32)a;
#if PP_FILL_NT
                    // (experiment) streaming store: the 16-byte items land all over the window buckets, a line is rarely
                    // completed while it is in the L2
                    typedef unsigned v4u __attribute__((ext_vector_type(4)));
                    __builtin_nontemporal_store((v4u){e.x, e.y, e.z, e.w}, (v4u *)&entB[slot]);
#else
                    entB[slot] = e;
#endif
                }
            });
        }
    }
    PP_STAMP(1, 2);
}

// Level 2 of the multisplit: one workgroup per coarse bucket counts its items per window, turns the counts into the
// windows' offsets (win_off: nobody else computes them on this path) and moves the items into their windows.  The
// destination region (COARSE_WINDOWS windows) is small, so the 16-byte writes merge into full lines in L2; the second
// read of the bucket comes from L2 as well.
template <int CW>  // 8: cursors advanced once per wave and window (ballot + popcount); 64: one LDS atomic per item
__global__ __launch_bounds__(1024) void k_regroup(u32 nwin, u32 ncoarse, const u32 *__restrict__ coarse_off,
                                                  u32 *__restrict__ win_off, const uint4 *__restrict__ entB,
                                                  uint4 *__restrict__ entA, u64 *__restrict__ status) {
    static_assert(CW == 8 || CW == 64, "the sub-index has six bits; one wave scans the windows of a coarse bucket");
    __shared__ u32 cnt[64], cur[64];
    if (*status != ~0ull) return;
    const u32 c = blockIdx.x, tid = threadIdx.x, lane = tid & 63u;
    const u32 lo = coarse_off[c], hi = coarse_off[c + 1];
    if (tid < 64u) cnt[tid] = 0;
    __syncthreads();
    if (CW == 64) {
        for (u32 i = lo + tid; i < hi; i += 1024) atomicAdd(&cnt[(entB[i].y >> 18) & 63u], 1u);
    } else {
        for (u32 i0 = lo + (tid & ~63u); i0 < hi; i0 += 1024) {
            const u32 i = i0 + lane;
            const bool valid = i < hi;
            const u32 sub = valid ? (entB[i].y >> 18) & 63u : 0u;
#pragma unroll
            for (u32 t = 0; t < (u32)CW; t++) {
                const u64 m = __ballot(valid && sub == t);
                if (m && lane == 0) atomicAdd(&cnt[t], (u32)__popcll(m));
            }
        }
    }
    __syncthreads();
    if (tid < 64u) {
        const u32 v = cnt[tid];
        u32 inc = v;
        for (int o = 1; o < 64; o <<= 1) {
            const u32 t = (u32)__shfl_up((int)inc, o, 64);
            if ((int)tid >= o) inc += t;
        }
        const u32 first = lo + inc - v, w = c * (u32)CW + tid;
        cur[tid] = first;
        if (tid < (u32)CW && w < nwin) {
            win_off[w] = first;
            if (v >= MAX_BUCKET) report(status, w, DE_TOO_DEEP);
        }
        if (c == ncoarse - 1u && tid == 0) win_off[nwin] = hi;
    }
    __syncthreads();
    if (CW == 64) {
        for (u32 i = lo + tid; i < hi; i += 1024) {
            uint4 e = entB[i];
            const u32 slot = atomicAdd(&cur[(e.y >> 18) & 63u], 1u);
            e.y &= ~(63u << 18);
            entA[slot] = e;
        }
    } else {
        for (u32 i0 = lo + (tid & ~63u); i0 < hi; i0 += 1024) {
            const u32 i = i0 + lane;
            const bool valid = i < hi;
            uint4 e = valid ? entB[i] : make_uint4(0, 0, 0, 0);
            const u32 sub = (e.y >> 18) & 63u;
            u32 slot = 0;
#pragma unroll
            for (u32 t = 0; t < (u32)CW; t++) {
                const u64 m = __ballot(valid && sub == t);
                if (!m) continue;
                u32 base = 0;
                if (lane == (u32)__ffsll((long long)m) - 1u) base = atomicAdd(&cur[t], (u32)__popcll(m));
                base = (u32)__builtin_amdgcn_readlane((int)base, __ffsll((long long)m) - 1);
                if (valid && sub == t) slot = base + (u32)__popcll(m & ((1ull << lane) - 1ull));
            }
            if (valid) {
                e.y &= ~(63u << 18);
                entA[slot] = e;
            }
        }
    }
}

}  // namespace pp
