// pp_kernels.hip -- gfx950 kernels of the polish hot path (seam B of include/polypolish_hip.h).
//
// Pipeline (one pp_polish_finish):
//   k_prep     one thread per alignment: CIGAR walk validation, reference span and the right-end
//              homopolymer trim (alignment.rs:175-201,364-378) -> (global start, kept entries)
//   k_count    per-block LDS histogram of (alignment, window) items   \  atomics-free multisplit
//   k_scan_cols / k_scan   column scan over blocks + scan over windows  > of the alignments into
//   k_fill     scatter 16-byte work items into their window's bucket   /  2048-position windows
//   k_tile     one workgroup per window: counters for 2048 positions live in LDS, one wave per
//              work item streams the read bases (coalesced byte loads) and does one LDS atomic per
//              base (pileup.rs:56-65,189-200); then one lane per position votes
//              (pileup.rs:67-134) and writes a 1-byte emit code
//   k_exact    the rare positions whose outcome depends on string-keyed counts (insertions, N...)
//              or on the ORDER of f64 depth additions (non-power-of-two 1/k shares) are replayed
//              exactly: covering alignments sorted by file order, sequential f64 adds, byte-exact
//              key grouping
//   k_compact / k_finalize  drop '-' (polish.rs:188), prefix-sum emit lengths, write polished bytes
//
// Integer counting, HBM/LDS bound: no MFMA anywhere by design.
#include "pp_internal.h"

#include <algorithm>
#include <cstring>

namespace pp {

typedef unsigned long long u64;
typedef uint32_t u32;
typedef uint8_t u8;

// LDS counter rows of one window.  A..OTH hold EXPLICIT tallies (every base of slow-class items, and
// the mismatching bases of fast-class items); COV is the coverage difference array of the fast
// class (+1 at the first kept position of a read, -1 one past the last; prefix-summed before the
// vote) and MIS the number of fast-class bases that differ from the assembly, so that the tally of
// the assembly's own base is  explicit + COV - MIS  without touching LDS once per matching base.
enum { ROW_A = 0, ROW_C = 1, ROW_T = 2, ROW_G = 3, ROW_DEL = 4, ROW_OTH = 5, ROW_DEF = 6, ROW_COV = 7,
       ROW_MIS = 8, N_ROWS = 9 };

struct KeyRec {    // debug only: one distinct non-ACGT key of a position (len 0 = the deletion key "-")
    u64 off;
    u32 pos, len, count, pad;
};

struct MultiEnt {  // a position whose polished string has 2+ bytes (an insertion won the vote)
    u64 off;       // absolute offset of the winning string in the seq array
    u32 pos;       // global assembly position
    u32 len;       // raw byte length of the string
    u32 eff;       // bytes left after removing '-' (polish.rs:188)
    u32 pad;
};

__device__ __forceinline__ void report(u64 *status, u64 idx, u32 code) {
    atomicMin(status, (idx << 8) | (u64)code);
}
// Job state as k_tile / k_exact2 see it: 0 running, 1 only a late capacity overflow so far (keep counting the
// needs, every write is guarded by its capacity), 2 aborted.
__device__ __forceinline__ int job_state(const u64 *status) {
    const u64 s = *status;
    return s == ~0ull ? 0 : ((s & 0xFFu) == DE_CAPACITY_LATE ? 1 : 2);
}

// misc.rs:208-215 for x >= 0
__device__ __forceinline__ u32 d_bankers(double x) {
    u32 r = (x >= 4294967295.0) ? 0xFFFFFFFFu : (u32)x;
    double f = x - trunc(x);
    if (f < 0.5) return r;
    if (f > 0.5) return r + 1u;
    return r + (r & 1u);
}

__device__ __forceinline__ u32 kclass_of(u32 k) {
    if (k == 1) return 0;
    if ((k & (k - 1)) == 0) {
        u32 j = 31u - (u32)__clz((int)k);
        if (j <= (u32)DEPTH_FX_BITS) return j;
    }
    return KCLASS_NONDYADIC;
}

// counter row of one read byte: exact "A"/"C"/"G"/"T" (pileup.rs:58-61), "-" shares the
// deletion key, everything else goes to the string-keyed table
__device__ __forceinline__ int row_of(u32 c) {
    u32 t = (c >> 1) & 3u;  // A->0 C->1 T->2 G->3
    u32 expect = (0x47544341u >> (t * 8u)) & 0xFFu;
    return (c == expect) ? (int)t : (c == (u32)'-' ? ROW_DEL : ROW_OTH);
}

__device__ __forceinline__ u32 wave_sum(u32 v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ u64 wave_sum64(u64 v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// trim of a read without indels: index of the first base of the trailing homopolymer; the kept
// entries are [0, start-1) (alignment.rs:364-378: pop the run, then one more)
__device__ __forceinline__ u32 simple_trim_start(const u8 *s, u32 sl) {
    const u8 last = s[sl - 1];
    u32 i = sl - 1;
    while (i > 0 && s[i - 1] == last) i--;
    return i;
}
__device__ __forceinline__ u32 simple_nkeep(const u8 *s, u32 sl) {
    const u32 i = simple_trim_start(s, sl);
    return i > 0 ? i - 1u : 0u;
}

// =============================================================================================
// k_prep
// =============================================================================================
// every record that is not a single short M run inside its contig
__device__ __noinline__ void prep_general(u64 a, u32 rs, u32 sl, u64 so, const u32 *cg, u32 nc, const u8 *seq,
                                               u64 c_lo, u64 c_hi, u32 *g_out, u32 *nk_out, u8 *fl_out, u64 *status) {
    // walk the runs (alignment.rs:178-194): spans and validity
    u64 ref_span = 0, read_span = 0;
    bool indel = false;
    for (u32 r = 0; r < nc; r++) {
        u32 op = cg[r], len = op >> 4, o = op & 15u;
        if (len == 0 || o > 8u) { report(status, a, DE_BAD_RUN); return; }
        if (o == PP_OP_M || o == PP_OP_EQ || o == PP_OP_X) { ref_span += len; read_span += len; }
        else if (o == PP_OP_I) { read_span += len; indel = true; }
        else if (o == PP_OP_D) { ref_span += len; indel = true; }
        else { report(status, a, DE_UNEXPECTED_OP); return; }
    }
    u32 o_first = cg[0] & 15u, o_last = cg[nc - 1] & 15u;
    if (!((o_first == PP_OP_M || o_first == PP_OP_EQ) && (o_last == PP_OP_M || o_last == PP_OP_EQ))) {
        report(status, a, DE_BAD_ENDS);
        return;
    }
    if (read_span != (u64)sl) { report(status, a, DE_LEN_MISMATCH); return; }
    if (ref_span >= 0x3FFFFFFFull) { report(status, a, DE_OVERFLOW); return; }

    const u64 clen = c_hi - c_lo;
    const u8 *s = seq + so;
    u32 n_entries = (u32)ref_span;
    if (!indel && sl <= FAST_MAX_LEN && (u64)rs + ref_span <= clen) {
        // fast class (=/X runs): k_tile loads the whole read anyway and trims it there, so the read
        // bytes are not touched here; bucketed by its untrimmed span
        *g_out = (u32)(c_lo + rs);
        *nk_out = n_entries;
        return;
    }
    // trim_bases_for_homopolymers (alignment.rs:364-378).  The last entry is the single base
    // seq[sl-1] (the last run is M/=).  `run` = number of trailing entries equal to it.
    u8 last = s[sl - 1];
    u32 run = 0;
    if (!indel) {
        run = sl - simple_trim_start(s, sl);
    } else {
        // walk the entries from the right end and stop at the first one that differs from the
        // last base (typically after 2-3 steps): runs in reverse; `pend` = bases inserted right
        // after the run being visited (they extend its last entry)
        u64 ro = sl;
        u32 pend = 0;
        bool stop = false;
        for (u32 r = nc; r-- > 0 && !stop;) {
            const u32 op = cg[r], len = op >> 4, o = op & 15u;
            if (o == PP_OP_I) { ro -= len; pend += len; continue; }
            if (o == PP_OP_D) {
                // last slot of the run: empty, or rewritten to the inserted bases; the others are empty
                if (pend == 1 && s[ro] == last) { run += 1; if (len > 1) stop = true; }
                else stop = true;
            } else {
                for (u32 t = 0; t < len; t++) {
                    const bool extended = (t == 0) && pend > 0;
                    if (!extended && s[ro - 1 - t] == last) run += 1; else { stop = true; break; }
                }
                ro -= len;
            }
            pend = 0;
        }
    }
    u32 nk = (n_entries > run) ? n_entries - run - 1u : 0u;
    if (nk == 0) return;  // contributes nothing; the reference never indexes the pileup for it
    if ((u64)rs + nk > clen) { report(status, a, DE_OUT_OF_BOUNDS); return; }
    *g_out = (u32)(c_lo + rs);
    *nk_out = nk;
    *fl_out = indel ? (u8)ENT_COMPLEX : (u8)ENT_PRETRIM;
}

#ifndef PP_PLAIN_ALIGNED
#define PP_PLAIN_ALIGNED 0
#endif
constexpr u32 PLAIN_NARROW_MAX = PP_PLAIN_ALIGNED ? 129u : 160u;  // = PlainCfg<5>::MAXL below

__device__ __forceinline__ void prep_one(u64 a, u64 n, const u32 *__restrict__ contig,
                                         const u32 *__restrict__ ref_start, const u32 *__restrict__ kk,
                                         const u64 *__restrict__ seq_off, const u32 *__restrict__ seq_len,
                                         const u64 *__restrict__ cig_off, const u32 *__restrict__ n_cig,
                                         const u32 *__restrict__ cigar, const u8 *__restrict__ seq,
                                         const u64 *__restrict__ contig_off, u32 n_contigs,
                                         u32 *__restrict__ gstart, u32 *__restrict__ nkeep, u32 *fast_len, u64 *status) {
    // independent loads first, then the dependent ones (clamped so that they are unconditional):
    // two memory round trips per record.  The bulk (one short M run inside its contig) touches 28
    // bytes of input per record; k and seq_off are only validated later, by k_fill, which reads them anyway.
    const u32 c = contig[a], nc = n_cig[a], sl = seq_len[a], rs = ref_start[a];
    const u64 co = cig_off[a];
    const u32 cc = min(c, n_contigs - 1u);
    const u64 c_lo = contig_off[cc], c_hi = contig_off[cc + 1];
    const u32 *cg = cigar + co;
    const u32 op0 = nc ? cg[0] : 0u;
    u32 g_out = 0, nk_out = 0;
    u8 fl_out = 0;
    if (c >= n_contigs) { report(status, a, DE_BAD_CONTIG); }
    else if (nc == 0) { report(status, a, DE_BAD_RUN); }
    else if (nc == 1 && (op0 & 15u) == PP_OP_M && (op0 >> 4) == sl && sl > 0 && sl <= FAST_MAX_LEN &&
             (u64)rs + sl <= c_hi - c_lo) {
        // the bulk: one M run, short, inside its contig -> fast class, trimmed later by k_tile
        g_out = (u32)(c_lo + rs);
        nk_out = sl;
        *fast_len = sl;  // the longest fast-class read picks the lane-group width of k_tile's plain class
    } else {
        prep_general(a, rs, sl, seq_off[a], cg, nc, seq, c_lo, c_hi, &g_out, &nk_out, &fl_out, status);
    }
    gstart[a] = g_out;
    nkeep[a] = nk_out | ((u32)fl_out << 30);  // kept entries (< 2^30) | class flags
}

__global__ __launch_bounds__(256) void k_prep(u64 n, const u32 *__restrict__ contig,
                                              const u32 *__restrict__ ref_start,
                                              const u32 *__restrict__ kk,
                                              const u64 *__restrict__ seq_off,
                                              const u32 *__restrict__ seq_len,
                                              const u64 *__restrict__ cig_off,
                                              const u32 *__restrict__ n_cig,
                                              const u32 *__restrict__ cigar,
                                              const u8 *__restrict__ seq,
                                              const u64 *__restrict__ contig_off, u32 n_contigs,
                                              u32 *__restrict__ gstart, u32 *__restrict__ nkeep,
                                              u32 *__restrict__ maxlen, u64 *status) {
    u64 a = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    u32 fast_len = 0;
    if (a < n) prep_one(a, n, contig, ref_start, kk, seq_off, seq_len, cig_off, n_cig, cigar, seq, contig_off, n_contigs,
                        gstart, nkeep, &fast_len, status);
    // Only every 64th block looks (a sample: the word merely picks the lane-group width that suits the bulk of the
    // reads -- a longer read than the sample saw simply takes the non-plain path), once per wave, and only for reads
    // beyond the narrowest group (<= 160 bases); the word is read from L2, not from a possibly stale CU-local copy.
    // A per-record look at that one address costs a millisecond on a 250-base job.
    if ((blockIdx.x & 63u) == 0 && __ballot(fast_len > PLAIN_NARROW_MAX)) {
        for (int o = 32; o > 0; o >>= 1) fast_len = max(fast_len, (u32)__shfl_xor((int)fast_len, o, 64));
        if ((threadIdx.x & 63u) == 0 && fast_len > __hip_atomic_load(maxlen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            atomicMax(maxlen, fast_len);
    }
}

// =============================================================================================
// bucketing: count -> scan -> fill (no global atomics; LDS histograms per block and window range)
// =============================================================================================
// Two-level multisplit of the (alignment, window) items.  Level 1 scatters the items into COARSE buckets of
// COARSE_WINDOWS windows: a block's items for one coarse bucket form one contiguous run (full-line writes),
// where a direct scatter into the windows would be 16-byte writes all over HBM.  Level 2 (k_regroup) sorts a
// coarse bucket into its windows inside a region small enough to stay in L2.
//   k_count     per-block LDS histogram over the windows -> global per-window counts (atomics) and the
//               block's per-coarse-bucket counts
//   k_scan_cols column scan over the blocks of the coarse counts; k_scan: offsets of coarse buckets and windows
//   k_fill      items -> coarse buckets (LDS cursors);  k_regroup  coarse bucket -> windows
template <int CW>  // windows per coarse bucket; 1 = single level (k_fill writes the windows directly)
__global__ __launch_bounds__(1024) void k_count(u64 n, u64 chunk, const u32 *__restrict__ gstart,
                                                const u32 *__restrict__ nkeep, u32 nwin, u32 ncoarse,
                                                u32 *__restrict__ hist_c, u32 *__restrict__ win_cnt) {
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
            nk[u] = a < hi ? (nkeep[a] & 0x3FFFFFFFu) : 0u;
            g[u] = a < hi ? gstart[a] : 0u;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (!nk[u]) continue;
            u32 w0 = g[u] / (u32)TILE, w1 = (g[u] + nk[u] - 1u) / (u32)TILE;
            u32 wa = max(w0, range_lo), wb = min(w1, range_lo + range_n - 1u);
            for (u32 w = wa; w <= wb && w >= wa; w++) atomicAdd(&h[w - range_lo], 1u);
        }
    }
    __syncthreads();
    if (CW > 1) {  // the windows' own totals are needed as well (k_scan_cols only sees the coarse buckets)
        for (u32 i = threadIdx.x; i < range_n; i += blockDim.x)
            if (h[i]) atomicAdd(&win_cnt[range_lo + i], h[i]);
    }
    const u32 c_lo = range_lo / (u32)CW, c_n = (range_n + CW - 1u) / (u32)CW;
    for (u32 c = threadIdx.x; c < c_n; c += blockDim.x) {
        u32 sum = 0;
#pragma unroll
        for (int j = 0; j < CW; j++) sum += h[c * CW + j];  // rows past range_n are zero
        hist_c[(u64)blockIdx.x * ncoarse + c_lo + c] = sum;
    }
}

__global__ __launch_bounds__(256) void k_scan_cols(u32 nwin, u32 nblocks, u32 *__restrict__ hist,
                                                   u32 *__restrict__ win_cnt) {
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
    if (lane == 63) win_cnt[w] = inc;
}

// single-block exclusive scan: out[i] = sum(in[0..i)), out[n] = total
// n_ptr (optional) overrides n with a count held on the device; the total is also stored to *total_out;
// a total above `limit` (capacity of the buffer the offsets index into) aborts the job with DE_CAPACITY
template <typename T>
__global__ __launch_bounds__(1024) void k_scan(const u32 *__restrict__ in, u64 n, const u32 *__restrict__ n_ptr,
                                               T *__restrict__ out, u64 *__restrict__ total_out, u64 limit,
                                               u64 *status) {
    __shared__ u64 part[1024];
    if (*status != ~0ull) return;
    if (n_ptr) n = *n_ptr;
    u32 t = threadIdx.x;
    u64 per = (n + 1023) / 1024;
    u64 lo = min(n, (u64)t * per), hi = min(n, lo + per);
    u64 s = 0;
    for (u64 i = lo; i < hi; i++) s += in[i];
    part[t] = s;
    __syncthreads();
    for (u32 off = 1; off < 1024; off <<= 1) {
        u64 v = (t >= off) ? part[t - off] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    u64 run = part[t] - s;
    for (u64 i = lo; i < hi; i++) {
        out[i] = (T)run;
        run += in[i];
    }
    if (t == 1023) {
        const u64 total = part[1023];
        out[n] = (T)total;
        if (total_out) *total_out = total;
        if (sizeof(T) == 4 && total > 0xFFFFFFFFull) report(status, 0, DE_OVERFLOW);
        else if (total > limit) report(status, total, DE_CAPACITY);
    }
}

template <int CW>
__global__ __launch_bounds__(1024) void k_fill(u64 n, u64 chunk, const u32 *__restrict__ gstart,
                                               const u32 *__restrict__ nkeep,
                                               const u32 *__restrict__ kk,
                                               const u64 *__restrict__ seq_off,
                                               const u32 *__restrict__ seq_len, u32 nwin, u32 ncoarse,
                                               const u32 *__restrict__ hist_c,
                                               const u32 *__restrict__ coarse_off,
                                               uint4 *__restrict__ entB, u64 *__restrict__ status) {
    __shared__ u32 cur[COUNT_RANGE];  // cursors of the coarse buckets of this pass
    if (*status != ~0ull) return;  // a record error, or the work-item buffer is too small (host reruns)
    const u32 crange_lo = blockIdx.y * (u32)COUNT_RANGE;
    const u32 crange_n = min((u32)COUNT_RANGE, ncoarse - crange_lo);
    for (u32 i = threadIdx.x; i < crange_n; i += blockDim.x)
        cur[i] = coarse_off[crange_lo + i] + hist_c[(u64)blockIdx.x * ncoarse + crange_lo + i];
    __syncthreads();
    const u32 range_lo = crange_lo * (u32)CW;                       // the same range, in windows
    const u32 range_n = min(crange_n * (u32)CW, nwin - range_lo);
    u64 lo = (u64)blockIdx.x * chunk, hi = min(n, lo + chunk);
    for (u64 a0 = lo + threadIdx.x; a0 < hi; a0 += 4ull * blockDim.x) {
        u32 nk4[4], g4[4], k4[4], fl4[4];
        u64 so4[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {  // four records per trip: their loads are in flight together
            const u64 a = a0 + (u64)u * blockDim.x;
            const bool ok = a < hi;
            const u32 nkw = ok ? nkeep[a] : 0u;
            nk4[u] = nkw & 0x3FFFFFFFu;
            fl4[u] = nkw >> 30;
            g4[u] = ok ? gstart[a] : 0u;
            k4[u] = ok ? kk[a] : 1u;
            so4[u] = ok ? seq_off[a] : 0ull;
            if (ok && blockIdx.y == 0) {  // checks that need k / seq_off (not read by k_prep's fast path)
                if (k4[u] == 0) report(status, a, DE_BAD_K);
                else if (so4[u] + seq_len[a] > (1ull << 40)) report(status, a, DE_OVERFLOW);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const u32 nk = nk4[u];
            if (!nk) continue;
            const u64 a = a0 + (u64)u * blockDim.x;
            const u32 g = g4[u];
            u32 w0 = g / (u32)TILE, w1 = (g + nk - 1u) / (u32)TILE;
            u32 wa = max(w0, range_lo), wb = min(w1, range_lo + range_n - 1u);
            if (wa > wb) continue;
            const u64 so = so4[u];
            const u32 kc = kclass_of(k4[u]);
            const u32 fl = fl4[u];
            for (u32 w = wa; w <= wb && w >= wa; w++) {
                u32 slot = atomicAdd(&cur[w / (u32)CW - crange_lo], 1u);
                // work item, 16 bytes (bits 20..22 of y carry the window's index inside its coarse bucket until
                // k_regroup has used it):
                //   x  fast class: seq offset bits 0..31          otherwise: kept entries (trim done by k_prep)
                //   y  [7:0] seq offset bits 32..39 (fast) | [15:8] depth-share class | [23:16] flags | [31:24] read length (fast)
                //   z  global start of the read minus the window start (signed)      w  record index (file order)
                uint4 e;
                e.x = fl ? nk : (u32)so;
                e.y = (fl ? 0u : (((u32)(so >> 32) & 0xFFu) | (nk << 24))) | (kc << 8) | (fl << 16) |
                      ((w % (u32)CW) << 20);
                e.z = (u32)(int)((long long)g - (long long)w * TILE);
                e.w = (u32)a;
                entB[slot] = e;
            }
        }
    }
}

// Level 2 of the multisplit: one workgroup per coarse bucket moves its items into their windows.  The
// destination region (COARSE_WINDOWS windows) is small, so the 16-byte writes merge into full lines in L2.
// Cursors are advanced once per wave and window (ballot + popcount), not once per item.
__global__ __launch_bounds__(1024) void k_regroup(u32 nwin, const u32 *__restrict__ coarse_off,
                                                  const u32 *__restrict__ win_off, const uint4 *__restrict__ entB,
                                                  uint4 *__restrict__ entA, u64 *__restrict__ status) {
    __shared__ u32 cur[COARSE_WINDOWS];
    if (*status != ~0ull) return;
    const u32 c = blockIdx.x, lane = threadIdx.x & 63u;
    if (threadIdx.x < (u32)COARSE_WINDOWS) {
        const u32 w = c * (u32)COARSE_WINDOWS + threadIdx.x;
        cur[threadIdx.x] = w < nwin ? win_off[w] : 0u;
        if (w < nwin && win_off[w + 1] - win_off[w] >= MAX_BUCKET) report(status, w, DE_TOO_DEEP);
    }
    __syncthreads();
    const u32 lo = coarse_off[c], hi = coarse_off[c + 1];
    for (u32 i0 = lo + (threadIdx.x & ~63u); i0 < hi; i0 += blockDim.x) {
        const u32 i = i0 + lane;
        const bool valid = i < hi;
        uint4 e = valid ? entB[i] : make_uint4(0, 0, 0, 0);
        const u32 sub = (e.y >> 20) & 7u;
        u32 slot = 0;
#pragma unroll
        for (u32 t = 0; t < (u32)COARSE_WINDOWS; t++) {
            const u64 m = __ballot(valid && sub == t);
            if (!m) continue;
            u32 base = 0;
            if (lane == (u32)__ffsll((long long)m) - 1u) base = atomicAdd(&cur[t], (u32)__popcll(m));
            base = (u32)__builtin_amdgcn_readlane((int)base, __ffsll((long long)m) - 1);
            if (valid && sub == t) slot = base + (u32)__popcll(m & ((1ull << lane) - 1ull));
        }
        if (valid) {
            e.y &= ~(7u << 20);
            entA[slot] = e;
        }
    }
}

// =============================================================================================
// k_tile: pileup accumulate + vote for one 2048-position window
// =============================================================================================
struct TileArgs {
    const uint4 *entA;
    const u32 *win_off;
    u32 nwin;
    const u8 *seq;
    const u64 *seq_off;
    const u64 *cig_off;
    const u32 *n_cig;
    const u32 *cigar;
    const u8 *bases;
    u64 G;
    const u64 *contig_off;
    u32 n_contigs;
    u32 min_depth;
    double fv, fi;
    u8 *code;
    u32 *win_len;
    u32 *counters;  // [0] positions on the global replay list, [1] n_multi, [2] all flagged positions
    u32 cap_flag;
    u32 *flag_bits;   // per window: 2048-bit map of flagged positions (64 words)
    u32 *win_nflag;   // per window: number of flagged positions
    u32 *win_slab;    // per window: index of its tally slab (6 x 2048 u32), or ~0
    u32 *slabs;
    u32 cap_slabs;
    u32 *flag_pos;
    u32 *flag_cov;
    u64 *scr_need;  // replay scratch the listed positions will need (sum of their coverage), counted past cap_flag too
    ContigStatsDev *stats;
    const u32 *maxlen;  // longest fast-class read (written by k_prep)
    u64 seq_bytes;
    const u32 *own;   // optional (lo, hi) emit range per contig, relative to the contig (pp_polish_set_emit)
    double *dbg_depth;
    u32 *dbg_counts;  // 7 planes of G: a, c, g, t, other, valid_thr, invalid_thr
    u8 *dbg_status;
    u64 *status;
    int dbg;
};

__device__ __forceinline__ void tile_add(u32 *cnt, int row, int p, u32 kc) {
    atomicAdd(&cnt[row * TILE + p], 1u);
    if (kc) {
        if (kc == KCLASS_NONDYADIC) atomicOr(&cnt[ROW_DEF * TILE + p], 0x80000000u);
        else atomicAdd(&cnt[ROW_DEF * TILE + p], (1u << DEPTH_FX_BITS) - (1u << (DEPTH_FX_BITS - kc)));
    }
}

__device__ __forceinline__ u32 find_contig(const u64 *contig_off, u32 n_contigs, u64 p) {
    u32 lo = 0, hi = n_contigs;  // contig_off[lo] <= p < contig_off[hi]
    while (hi - lo > 1) {
        u32 mid = (lo + hi) >> 1;
        if (contig_off[mid] <= p) lo = mid; else hi = mid;
    }
    return lo;
}

struct VoteOut {
    u8 out;     // byte to emit (0 = nothing)
    u8 status;  // PP_ST_*
    u32 vthr, ithr;
};

// pileup.rs:67-134 restricted to the keys A,C,G,T and "-"; callers guarantee that no other key
// can reach either threshold.
__device__ __forceinline__ VoteOut vote5(u32 nA, u32 nC, u32 nG, u32 nT, u32 nDel, double depth,
                                         u8 orig, u32 min_depth, double fv, double fi) {
    VoteOut v;
    u32 vt = d_bankers(__dmul_rn(depth, fv));
    v.vthr = max(min_depth, vt);
    v.ithr = d_bankers(__dmul_rn(depth, fi));
    v.out = orig;
    v.status = PP_ST_KEPT;
    if (depth < (double)min_depth) {
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
__device__ __forceinline__ uint4 load16_unaligned(const u8 *p) {
    uint4 v;
    __builtin_memcpy(&v, p, 16);
    return v;
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
    // work-item words x, y: no flags, share class 0 (k = 1) or non-dyadic (depth replayed exactly anyway),
    // length in range, and every 32-byte chunk of the read inside the seq array
    __device__ static __forceinline__ bool ok(u32 ex, u32 ey, u64 seq_bytes) {
        const u32 L = ey >> 24, kc = (ey >> 8) & 0xFFu;
        const u64 so = (u64)ex | ((u64)(ey & 0xFFu) << 32);
        return (ey & 0x00FF0000u) == 0 && (kc == 0 || kc == KCLASS_NONDYADIC) && L >= PLAIN_MIN_LEN && L <= MAXL &&
               so + ((L + 31u) & ~31u) <= seq_bytes;
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
    bool nd;           // depth share is not a power of two: its positions are replayed by k_exact2
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
    it.rel = __builtin_amdgcn_ds_bpermute(src, (int)my.z);
    it.L = ey >> 24;
    it.nd = ((ey >> 8) & 0xFFu) == KCLASS_NONDYADIC;
    it.plain = g < C::IPP && j < nb && C::ok(ex, ey, seq_bytes);
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

__device__ __forceinline__ void plain_apply(u32 *cnt, u32 *ndbits, const u32 *asm_w, const PlainItem &it, u32 lane) {
    const int rel = it.rel;
    const u32 L = it.L;

    // ---- trim (alignment.rs:364-378): nkeep = index of the last base that differs from the last base,
    // read off the last four bases; a trailing homopolymer of four or more takes the byte loop
    const u32 last = it.tail >> 24;
    const u32 tf = nz_flags(it.tail ^ splat8(last));
    int nkeep = (int)L - 4 + ((31 - __clz((int)tf)) >> 3);
    if (it.plain && tf == 0) {  // rare: walk left over the homopolymer
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
        if (it.nd) {  // mark [rel+lo, rel+hi) in the window's bitmap of order-dependent positions
            const u32 a = (u32)(rel + lo), b = (u32)(rel + hi);
            for (u32 wd = a >> 5; wd <= (b - 1u) >> 5; wd++) {
                const u32 from = wd == (a >> 5) ? (a & 31u) : 0u, to = wd == ((b - 1u) >> 5) ? ((b - 1u) & 31u) : 31u;
                atomicOr(&ndbits[wd], (0xFFFFFFFFu >> (31u - to)) & (0xFFFFFFFFu << from));
            }
        }
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

// ---- fast class of work items: a read without indels, <= FAST_MAX_LEN bases, inside its contig ----
struct FastItem {  // wave-uniform (built from v_readlane results)
    u64 so;    // offset of the read in the seq array
    int rel;   // global start of the read minus the window start
    u32 L;     // read length == number of entries before the trim
    u32 kc;    // depth-share class of 1/k
    u32 mis;   // (address of the read) & 3
    bool on;
};

__device__ __forceinline__ FastItem fast_fetch(const uint4 &my, u32 j, u32 nb, const u8 *seq) {
    const int jj = (int)min(j, nb - 1u);
    const u32 x = (u32)__builtin_amdgcn_readlane((int)my.x, jj), y = (u32)__builtin_amdgcn_readlane((int)my.y, jj);
    FastItem f;
    f.so = (u64)x | ((u64)(y & 0xFFu) << 32);
    f.rel = __builtin_amdgcn_readlane((int)my.z, jj);
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
__device__ __forceinline__ void fast_apply(u32 *cnt, const u32 *asm_w, const FastItem &f, u32 word, u32 lane) {
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
    const int lo = max(0, -f.rel), hi = min(nkeep, TILE - f.rel);
    if (hi <= lo) return;
    if (f.kc != 0) {
        // byte order rotated by lane/8 so that the 32 lanes of an LDS group hit 32 different banks
#pragma unroll
        for (int jj = 0; jj < 4; jj++) {
            const int b = (jj + (int)(lane >> 3)) & 3;
            const int i = ib + b;
            if (i >= lo && i < hi) tile_add(cnt, row_of((word >> (8 * b)) & 0xFFu), f.rel + i, f.kc);
        }
        return;
    }
    if (lane == 0) {
        atomicAdd(&cnt[ROW_COV * TILE + f.rel + lo], 1u);
        if (f.rel + hi < TILE) atomicAdd(&cnt[ROW_COV * TILE + f.rel + hi], 0xFFFFFFFFu);
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

// two 1024-thread workgroups per CU (8 waves per SIMD): at most 64 VGPRs
// The work items of one window, one batch per wave at a time: one coalesced load of the batch's 16-byte
// records, then the plain class IPP items per pass, then the other classes one item per pass.  Latency is
// hidden by the other 7 waves of the SIMD, not by software pipelining (which measured slower).
template <int GW>
__device__ __forceinline__ void tile_items(const TileArgs &A, u32 *cnt, u32 *s_ndbits, const u32 *asm_w, u32 e0, u32 e1,
                                           u32 wave, u32 lane) {
    typedef PlainCfg<GW> C;
    // every wave takes one contiguous slice of the window's items, equal to within one pass (the order
    // of the items does not matter: the counters are integers)
    constexpr u32 WAVES = TILE_THREADS / 64;
    const u32 per_wave = ((e1 - e0 + WAVES - 1u) / WAVES + C::IPP - 1u) / C::IPP * C::IPP;
    const u32 lo_w = min(e1, e0 + wave * per_wave), hi_w = min(e1, lo_w + per_wave);
    if (lo_w >= hi_w) return;
    for (u32 eb = lo_w; eb < hi_w; eb += C::BATCH) {
        const u32 nb = min(C::BATCH, hi_w - eb);
        const uint4 my = A.entA[eb + min(lane, nb - 1u)];
        const u32 my_flags = (my.y >> 16) & 0xFFu;
        const bool my_slow = lane < nb && my_flags != 0;
        const bool my_plain = lane < nb && C::ok(my.x, my.y, A.seq_bytes);
        for (u32 first = 0; first < nb; first += C::IPP)
            plain_apply(cnt, s_ndbits, asm_w, plain_fetch<GW>(A.seq, A.seq_bytes, my, nb, first, lane), lane);
        // fast class that is not plain (shared depth, or 242..252 bases): one item per pass
        u64 rest = __ballot(lane < nb && !my_slow && !my_plain);
        while (rest) {
            const u32 j = (u32)__ffsll((long long)rest) - 1u;
            rest &= rest - 1;
            const FastItem f = fast_fetch(my, j, nb, A.seq);
            fast_apply(cnt, asm_w, f, fast_load(A.seq, f, lane), lane);
        }
        u64 slow = __ballot(my_slow);
        while (slow) {
            const int j = __ffsll((long long)slow) - 1;
            slow &= slow - 1;
            const u32 ey = (u32)__builtin_amdgcn_readlane((int)my.y, j), idx = (u32)__builtin_amdgcn_readlane((int)my.w, j);
            const int rel = __builtin_amdgcn_readlane((int)my.z, j), nkeep = __builtin_amdgcn_readlane((int)my.x, j);
            const u32 kc = (ey >> 8) & 0xFFu;
            const u8 *s = A.seq + A.seq_off[idx];
            if (!((ey >> 16) & ENT_COMPLEX)) {
                // no indels, trim precomputed by k_prep (long read or contig overhang): entry i is base i
                const int lo = max(0, -rel), hi = min(nkeep, TILE - rel);
                for (int i = lo + (int)lane; i < hi; i += 64) tile_add(cnt, row_of(s[i]), rel + i, kc);
            } else {
                const u32 *cg = A.cigar + A.cig_off[idx];
                const u32 nc = A.n_cig[idx];
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
                        tile_add(cnt, row, rel + q, kc);
                    }
                    ent0 += (int)len;
                    if (o != PP_OP_D) ro += len;
                }
            }
        }
    }
}

__global__ __launch_bounds__(TILE_THREADS, 8) void k_tile(TileArgs A) {
    __shared__ u32 cnt[N_ROWS * TILE];
    __shared__ __attribute__((aligned(16))) u32 asm_w[ASM_WORDS];  // the window's assembly bytes at byte offset ASM_PAD
    __shared__ u32 s_len, s_changed, s_zero, s_c0, s_c1, s_wsum[TILE_THREADS / 64], s_fbits[TILE / 32], s_ndbits[TILE / 32], s_nflag;
    __shared__ u64 s_depth;

    // XCD-aware order: consecutive windows (which share boundary-crossing reads) stay on one XCD
    u32 per = gridDim.x >> 3;
    u32 w = (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
    if (w >= A.nwin || job_state(A.status) == 2) return;
    const u32 tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const u64 w0 = (u64)w * TILE;

    for (u32 i = tid; i < (u32)(N_ROWS * TILE); i += TILE_THREADS) cnt[i] = 0;
    if (tid < (u32)(TILE / 32)) { s_fbits[tid] = 0; s_ndbits[tid] = 0; }
    {
        u8 *ab = (u8 *)asm_w;
        for (u32 i = tid; i < (u32)TILE; i += TILE_THREADS) ab[ASM_PAD + i] = (w0 + i < A.G) ? A.bases[w0 + i] : (u8)0;
        if (tid < (u32)ASM_PAD) ab[tid] = 0;
        if (tid < (u32)(ASM_WORDS * 4 - ASM_PAD - TILE)) ab[ASM_PAD + TILE + tid] = 0;
    }
    if (tid == 0) {
        s_len = 0; s_changed = 0; s_zero = 0; s_depth = 0; s_nflag = 0;
        s_c0 = find_contig(A.contig_off, A.n_contigs, w0);
        u64 last = min(w0 + TILE, A.G) - 1;
        s_c1 = find_contig(A.contig_off, A.n_contigs, last);
    }
    __syncthreads();

    const u32 e0 = A.win_off[w], e1 = A.win_off[w + 1];
    {
        const u32 longest = *A.maxlen;  // longest fast-class read of the job (k_prep)
        if (longest <= PlainCfg<5>::MAXL) tile_items<5>(A, cnt, s_ndbits, asm_w, e0, e1, wave, lane);
        else if (longest <= PlainCfg<6>::MAXL) tile_items<6>(A, cnt, s_ndbits, asm_w, e0, e1, wave, lane);
        else tile_items<8>(A, cnt, s_ndbits, asm_w, e0, e1, wave, lane);
    }
    if (e1 - e0 >= MAX_BUCKET && tid == 0) report(A.status, w, DE_TOO_DEEP);
    __syncthreads();

    // ---- coverage of the fast class: prefix sum of the difference array, in place ----
    {
        u32 *cov = cnt + ROW_COV * TILE;
        const u32 d0 = cov[2 * tid], d1 = cov[2 * tid + 1];
        const u32 sum = d0 + d1;
        u32 inc = sum;
        for (int o = 1; o < 64; o <<= 1) {
            const u32 v = __shfl_up(inc, o, 64);
            if ((int)lane >= o) inc += v;
        }
        if (lane == 63) s_wsum[wave] = inc;
        __syncthreads();
        u32 base = 0;
        for (u32 i = 0; i < wave; i++) base += s_wsum[i];
        const u32 ex = base + inc - sum;
        cov[2 * tid] = ex + d0;
        cov[2 * tid + 1] = ex + d0 + d1;
    }
    __syncthreads();

    // ---- vote: one lane per position ----
    u32 my_len = 0, my_changed = 0, my_zero = 0;
    u64 my_depth = 0;
    const bool one_contig = (s_c0 == s_c1);
    for (u32 p = tid; p < (u32)TILE; p += TILE_THREADS) {
        const u64 gp = w0 + p;
        if (gp >= A.G) break;
        if (A.own) {  // window tiling: halo positions are voted by the rank that owns them
            const u32 c = one_contig ? s_c0 : find_contig(A.contig_off, A.n_contigs, gp);
            const u32 rel = (u32)(gp - A.contig_off[c]);
            if (rel < A.own[2 * c] || rel >= A.own[2 * c + 1]) {
                A.code[gp] = 0;
                continue;
            }
        }
        u32 nA, nC, nG, nT, nDel, nOth;
        const u32 defw = cnt[ROW_DEF * TILE + p];
        const u8 orig = ((const u8 *)asm_w)[ASM_PAD + p];
        position_tallies(cnt, orig, p, nA, nC, nG, nT, nDel, nOth);
        const bool nd = (defw >> 31) != 0 || ((s_ndbits[p >> 5] >> (p & 31u)) & 1u) != 0;
        const u32 deficit = defw & 0x7FFFFFFFu;
        const u32 ntot = nA + nC + nG + nT + nDel + nOth;
        if (orig >= 0x80u) report(A.status, gp, DE_NON_ASCII);
        const u64 dfx = ((u64)ntot << DEPTH_FX_BITS) - deficit;
        const double depth = (double)dfx * (1.0 / (double)(1u << DEPTH_FX_BITS));  // exact
        bool flag = false;
        VoteOut v;
        v.out = (orig == (u8)'-') ? 0 : orig;
        v.status = PP_ST_LOW_DEPTH;
        v.vthr = 0; v.ithr = 0;
        if (nd) {
            // depth is an order-dependent f64 sum: exact only in k_exact.  depth <= ntot always,
            // so ntot < min_depth already decides DepthTooLow.
            if (ntot >= A.min_depth || A.dbg) flag = true;
        } else {
            const u32 ithr = d_bankers(__dmul_rn(depth, A.fi));
            if (!(depth < (double)A.min_depth) && nOth > 0 && nOth >= ithr) flag = true;
            else v = vote5(nA, nC, nG, nT, nDel, depth, orig, A.min_depth, A.fv, A.fi);
        }
        if (A.dbg && nDel + nOth > 0) flag = true;  // --debug lists every key: k_exact writes the records
        if (flag) {
            const bool to_list = A.dbg == 1 || e1 - e0 > SORT_MAX;  // dbg 2: test hook, see run_pipeline
            if (!to_list) {
                atomicOr(&s_fbits[p >> 5], 1u << (p & 31u));
                atomicAdd(&s_nflag, 1u);
            } else {
                atomicAdd(&A.counters[2], 1u);
            }
            if (to_list) {
                // bucket too large for the wave-per-position replay: global list for k_exact
                const u32 slot = atomicAdd(&A.counters[0], 1u);
                atomicAdd(A.scr_need, (u64)ntot);
                if (slot < A.cap_flag) {
                    A.flag_pos[slot] = (u32)gp;
                    A.flag_cov[slot] = ntot;
                } else {
                    report(A.status, slot, DE_CAPACITY_LATE);
                }
            }
            A.code[gp] = 0;
            continue;
        }
        A.code[gp] = v.out;
        const u32 l = v.out ? 1u : 0u, ch = (v.status == PP_ST_CHANGED), z = (ntot == 0);
        if (one_contig) {
            my_len += l; my_changed += ch; my_zero += z; my_depth += dfx;
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
    my_len = wave_sum(my_len);
    my_changed = wave_sum(my_changed);
    my_zero = wave_sum(my_zero);
    my_depth = wave_sum64(my_depth);
    if (lane == 0) {
        if (my_len) atomicAdd(&s_len, my_len);
        if (my_changed) atomicAdd(&s_changed, my_changed);
        if (my_zero) atomicAdd(&s_zero, my_zero);
        if (my_depth) atomicAdd(&s_depth, my_depth);
    }
    __syncthreads();
    if (tid < (u32)(TILE / 32)) A.flag_bits[(u64)w * (TILE / 32) + tid] = s_fbits[tid];
    if (s_nflag && e1 - e0 <= SORT_MAX) {
        // the ordered-depth replay needs this window's integer tallies: save them (rare windows only)
        if (tid == 0) {
            const u32 slab = atomicAdd(&A.counters[3], 1u);
            if (slab >= A.cap_slabs) report(A.status, slab, DE_CAPACITY_LATE);
            s_c1 = slab;
            A.win_slab[w] = slab;
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
    if (tid == 0) {
        A.win_nflag[w] = s_nflag;
        if (s_nflag) atomicAdd(&A.counters[2], s_nflag);
        A.win_len[w] = s_len;
        if (s_changed) atomicAdd(&A.stats[s_c0].changed, (u64)s_changed);
        if (s_zero) atomicAdd(&A.stats[s_c0].zero_depth, (u64)s_zero);
        if (s_depth) atomicAdd(&A.stats[s_c0].depth_fx, s_depth);
    }
}

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
    u64 *scr_need;
    const u32 *flag_bits;
    const u32 *win_nflag;
    const u32 *win_slab;
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
    const u32 *win_off;
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
    u32 *win_len;
    u32 *counters;
    MultiEnt *multi;
    ContigStatsDev *stats;
    double *dbg_depth;
    u32 *dbg_counts;
    u8 *dbg_status;
    u64 *status;
    int dbg;
};

constexpr u64 SL_OFF_MASK = (1ull << 40) - 1;
constexpr u64 SL_DONE = 1ull << 63;

__device__ void exact_one(const ExactArgs &A, u32 f);

__global__ __launch_bounds__(64) void k_exact(ExactArgs A) {
    if (*A.status != ~0ull) return;
    const u32 n_flagged = A.counters[0];
    for (u32 f = blockIdx.x * blockDim.x + threadIdx.x; f < n_flagged; f += gridDim.x * blockDim.x) exact_one(A, f);
}

__device__ void exact_one(const ExactArgs &A, u32 f) {
    const u32 gp = A.flag_pos[f], cap = A.flag_cov[f];
    const u32 w = gp / (u32)TILE;
    const int pr = (int)(gp - w * (u32)TILE);
    ulonglong2 *scr = A.scratch + A.flag_scr[f];

    // collect the covering alignments: x = (file index << 32 | k), y = slice (offset | len << 40)
    u32 n = 0;
    for (u32 e = A.win_off[w]; e < A.win_off[w + 1]; e++) {
        const uint4 ent = A.entA[e];
        const int q = pr - (int)ent.z;
        const u32 fl = (ent.y >> 16) & 0xFFu, idx = ent.w;
        if (q < 0 || q >= (int)(fl ? ent.x : (ent.y >> 24))) continue;
        const u64 so = fl ? A.seq_off[idx] : ((u64)ent.x | ((u64)(ent.y & 0xFFu) << 32));
        // fast-class items carry their untrimmed length: apply the trim here
        if (fl == 0 && (u32)q >= simple_nkeep(A.seq + so, ent.y >> 24)) continue;
        u64 s_rel;
        u32 len;
        if (!(fl & ENT_COMPLEX)) { s_rel = (u64)q; len = 1; }
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
    if (emit) atomicAdd(&A.win_len[w], emit);
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
__global__ __launch_bounds__(1024) void k_exact2(ExactArgs A, u32 nwin) {
    __shared__ u64 pk[SORT_MAX];  // bitonic sort keys (record index << 16 | slot), or the counting sort's arrays
    __shared__ u64 s_base;
    const u32 w = blockIdx.x, tid = threadIdx.x;
    const int state = w < nwin ? job_state(A.status) : 2;
    if (state == 2) return;
    if (A.win_nflag[w] == 0) return;
    const u32 e0 = A.win_off[w], n = A.win_off[w + 1] - e0;
    if (n > SORT_MAX || n == 0) return;  // large buckets are replayed by k_exact
    if (state == 1) {  // a buffer was too small: only add up the replay scratch the rerun will need
        if (tid == 0 && n > SORT_MAX / 4) atomicAdd(A.ents_cursor, (u64)n);  // smaller lists stay in LDS
        return;
    }
    const u32 slab = A.win_slab[w];

    // ---- (1) order the window's items by record index (= SAM file order) ----
    // Record indices of a window's items are spread over the file, so a counting sort on their leading bits
    // (up to SORT_BUCKETS buckets between the window's smallest and largest index) leaves buckets of a few
    // items, finished by one thread each with an insertion sort: ~10x fewer LDS passes than a bitonic network
    // over 16 K keys.  Clustered indices (a bucket above SORT_BUCKET_MAX items) take the bitonic sort instead.
    // The arrays of the counting sort live inside pk[] (112 of its 128 KiB).
    u32 *rec = (u32 *)pk;                           // [SORT_MAX] record index of slot i
    unsigned short *ord = (unsigned short *)(pk + SORT_MAX / 2);       // [SORT_MAX] slots in file order
    u32 *bkt = (u32 *)(pk + SORT_MAX / 2 + SORT_MAX / 4);              // [SORT_BUCKETS + 1] counts -> cursors
    __shared__ u32 s_lo, s_hi, s_big, s_wtot[16];
    if (tid == 0) { s_lo = 0xFFFFFFFFu; s_hi = 0; s_big = 0; }
    for (u32 i = tid; i <= SORT_BUCKETS; i += 1024) bkt[i] = 0;
    __syncthreads();
    {
        u32 lo = 0xFFFFFFFFu, hi = 0;
        for (u32 i = tid; i < n; i += 1024) {
            const u32 r = A.entA[e0 + i].w;
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
    // that is free by then (rec[]: 4096 records; the bitonic keys occupy it), else it goes to a global slab.
    const bool in_lds = !bitonic && n <= SORT_MAX / 4;
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
        for (u32 i = tid; i < np2; i += 1024) pk[i] = i < n ? (((u64)A.entA[e0 + i].w << 16) | (u64)i) : ~0ull;
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
        const uint4 ent = A.entA[e0 + (bitonic ? (u32)(pk[i] & 0xFFFFu) : (u32)ord[i])];
        const u32 fl = (ent.y >> 16) & 0xFFu, kc = (ent.y >> 8) & 0xFFu;
        u32 lim;
        if (fl) lim = ent.x;
        else lim = simple_nkeep(A.seq + ((u64)ent.x | ((u64)(ent.y & 0xFFu) << 32)), ent.y >> 24);
        const u32 k = kc == 0 ? 1u : (kc != KCLASS_NONDYADIC ? (1u << kc) : A.kk[ent.w]);
        ulonglong2 r;
        r.x = (u64)ent.z | ((u64)lim << 32);
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
    const int wlo = (int)(tid >> 6) * 128;
    const int p0 = wlo + (int)lane, p1 = p0 + 64;
    double d0 = 0.0, d1 = 0.0;
    // Two instances of the loop, one per address space (through a generic pointer the loads would be flat loads,
    // whose counters force a full wait), and two batch registers in turn, so that the load of the batch after the
    // current one is in flight while the current one is visited.
    auto visit = [&](const ulonglong2 &mine, bool have) {
        const int xl = (int)(u32)mine.x, xh = (int)(u32)(mine.x >> 32), yl = (int)(u32)mine.y, yh = (int)(u32)(mine.y >> 32);
        // one vector compare picks the items of this batch that reach the wave's positions; only those are
        // visited one by one, in ascending order = file order
        u64 hits = __ballot(have && xl < wlo + 128 && (long long)xl + (long long)(u32)xh > (long long)wlo);
        while (hits) {
            const int j = __ffsll((long long)hits) - 1;
            hits &= hits - 1;
            const int rel = __builtin_amdgcn_readlane(xl, j);
            const u32 lim = (u32)__builtin_amdgcn_readlane(xh, j);
            const double dc = __hiloint2double(__builtin_amdgcn_readlane(yh, j), __builtin_amdgcn_readlane(yl, j));
            if ((u32)(p0 - rel) < lim) d0 += dc;
            if ((u32)(p1 - rel) < lim) d1 += dc;
        }
    };
    auto ordered_pass = [&](auto load) {
        // unconditional loads from clamped indices (a load under a branch would make the wait for the older
        // batch a wait for everything); `have` masks the lanes past the end
        ulonglong2 ba = load(min(lane, n - 1u)), bb;
        for (u32 base = 0; base < n; base += 128) {
            bb = load(min(base + 64u + lane, n - 1u));
            visit(ba, base + lane < n);
            ba = load(min(base + 128u + lane, n - 1u));
            visit(bb, base + 64u + lane < n);
        }
    };
    if (in_lds) ordered_pass([&](u32 i) { return ents_lds[i]; });
    else ordered_pass([&](u32 i) { return ents[i]; });

    // ---- (4) vote for the flagged positions; per-window sums are reduced in the block first ----
    const u32 *tal = A.slabs + (u64)slab * 6u * TILE;
    const u64 gw0 = (u64)w * TILE;
    const u32 c_first = find_contig(A.contig_off, A.n_contigs, gw0);
    const bool one_contig = c_first == find_contig(A.contig_off, A.n_contigs, min(gw0 + TILE, A.G) - 1);
    u32 my_len = 0, my_changed = 0, my_zero = 0;
    u64 my_depth = 0;
    for (int h = 0; h < 2; h++) {
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
            atomicAdd(A.scr_need, (u64)ntot);
            if (slot < A.cap_flag) { A.flag_pos_w[slot] = gp; A.flag_cov_w[slot] = ntot; }
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
        if (pk[0]) atomicAdd(&A.win_len[w], (u32)pk[0]);
        if (pk[1]) atomicAdd(&A.stats[c_first].changed, pk[1]);
        if (pk[2]) atomicAdd(&A.stats[c_first].zero_depth, pk[2]);
        if (pk[3]) atomicAdd(&A.stats[c_first].depth_fx, pk[3]);
    }
}

// =============================================================================================
// emission: code bytes -> polished bytes
// =============================================================================================
__device__ __forceinline__ u32 code_len(u8 c, u32 gp, const MultiEnt *multi, u32 n_multi) {
    if (c == 0) return 0;
    if (c < 0x80u) return 1;
    if (c != 0xFFu) return c & 0x7Fu;
    for (u32 i = 0; i < n_multi; i++)
        if (multi[i].pos == gp) return multi[i].eff;
    return 0;
}

__global__ __launch_bounds__(TILE_THREADS) void k_compact(const u8 *__restrict__ code, u64 G,
                                                          const u64 *__restrict__ win_out,
                                                          const MultiEnt *__restrict__ multi,
                                                          const u32 *__restrict__ counters,
                                                          u8 *__restrict__ out, const u64 *__restrict__ status) {
    __shared__ u32 wsum[TILE_THREADS / 64];
    if (*status != ~0ull) return;
    const u32 w = blockIdx.x, t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    const u64 p0 = (u64)w * TILE + 2ull * t;
    const u32 n_multi = counters[1];
    const u8 c0 = (p0 < G) ? code[p0] : 0, c1 = (p0 + 1 < G) ? code[p0 + 1] : 0;
    const u32 l0 = code_len(c0, (u32)p0, multi, n_multi), l1 = code_len(c1, (u32)(p0 + 1), multi, n_multi);
    const u32 s = l0 + l1;
    u32 inc = s;  // inclusive scan within the wave
    for (int o = 1; o < 64; o <<= 1) {
        u32 v = __shfl_up(inc, o, 64);
        if ((int)lane >= o) inc += v;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    u32 base = 0;
    for (u32 i = 0; i < wave; i++) base += wsum[i];
    const u64 off = win_out[w] + base + (inc - s);
    if (c0 && c0 < 0x80u) out[off] = c0;
    if (c1 && c1 < 0x80u) out[off + l0] = c1;
}

// threads [0, n_multi): copy a multi-byte winner into its reserved gap;
// threads [n_multi, n_multi + n_contigs]: output offset of each contig start (and the total)
__global__ __launch_bounds__(64) void k_finalize(const u8 *__restrict__ code, u64 G,
                                                 const u64 *__restrict__ win_out, u32 nwin,
                                                 const MultiEnt *__restrict__ multi,
                                                 const u32 *__restrict__ counters,
                                                 const u8 *__restrict__ seq,
                                                 const u64 *__restrict__ contig_off, u32 n_contigs,
                                                 u8 *__restrict__ out, u64 *__restrict__ ctg_out,
                                                 const u64 *__restrict__ status) {
    if (*status != ~0ull) return;
    const u32 n_multi = counters[1];
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_multi + n_contigs + 1u) return;
    u64 gp;
    if (t < n_multi) gp = multi[t].pos; else gp = contig_off[t - n_multi];
    u64 off;
    if (gp >= G) {
        off = win_out[nwin];
    } else {
        const u32 w = (u32)(gp / TILE);
        off = win_out[w];
        for (u64 q = (u64)w * TILE; q < gp; q++) off += code_len(code[q], (u32)q, multi, n_multi);
    }
    if (t < n_multi) {
        const u8 *s = seq + multi[t].off;
        for (u32 b = 0; b < multi[t].len; b++)
            if (s[b] != (u8)'-') out[off++] = s[b];
    } else {
        ctg_out[t - n_multi] = off;
    }
}

}  // namespace pp

// =============================================================================================
// host side of the polish pipeline
// =============================================================================================
using namespace pp;

namespace pp {

int dev_ensure(pp_ctx *ctx, DevBuf &b, size_t bytes) {
    if (bytes == 0) bytes = 16;
    if (b.cap >= bytes) return PP_OK;
    if (b.p) {
        PP_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        PP_HIPCHK(ctx, hipFree(b.p));
        b.p = nullptr;
        b.cap = 0;
    }
    size_t want = bytes + bytes / 8 + 256;  // slack so steady-state jobs of similar size do not realloc
    PP_HIPCHK(ctx, hipMalloc(&b.p, want));
    b.cap = want;
    return PP_OK;
}

void dev_free(DevBuf &b) {
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
}

// Per-kernel-group timing with HIP events on the context's stream.  profiling == 1 times every group,
// profiling == 2 only the dominant kernel ("tile": one event pair per job, for the bench's timed region).
// Events come from a pool that lives as long as the context.
void timer_begin(pp_ctx *ctx, const char *name) {
    if (!ctx->profiling || (ctx->profiling == 2 && strcmp(name, "tile") != 0)) return;
    KernelTimer t;
    t.name = name;
    if (ctx->event_pool.size() >= 2) {
        t.stop = ctx->event_pool.back(); ctx->event_pool.pop_back();
        t.start = ctx->event_pool.back(); ctx->event_pool.pop_back();
    } else if (hipEventCreate(&t.start) != hipSuccess || hipEventCreate(&t.stop) != hipSuccess) {
        return;
    }
    (void)hipEventRecord(t.start, ctx->stream);
    ctx->timers.push_back(t);
    ctx->timer_open = true;
}
void timer_end(pp_ctx *ctx) {
    if (!ctx->timer_open) return;
    (void)hipEventRecord(ctx->timers.back().stop, ctx->stream);
    ctx->timer_open = false;
}
void timers_release(pp_ctx *ctx) {
    for (auto &t : ctx->timers) {
        ctx->event_pool.push_back(t.start);
        ctx->event_pool.push_back(t.stop);
    }
    ctx->timers.clear();
    ctx->timer_open = false;
}
int timers_collect(pp_ctx *ctx, pp_kernel_times *out) {
    out->n = 0;
    for (auto &t : ctx->timers) {
        float ms = 0.f;
        (void)hipEventSynchronize(t.stop);
        (void)hipEventElapsedTime(&ms, t.start, t.stop);
        int k;
        for (k = 0; k < out->n; k++)
            if (out->name[k] == t.name) break;
        if (k == out->n) {
            if (out->n == PP_MAX_KERNELS) continue;
            out->name[k] = t.name;
            out->ms[k] = 0.f;
            out->n++;
        }
        out->ms[k] += ms;
    }
    timers_release(ctx);
    return PP_OK;
}

}  // namespace pp

static int upload(pp_ctx *ctx, DevBuf &b, const void *src, size_t bytes, const void **dev) {
    int rc = dev_ensure(ctx, b, bytes);
    if (rc) return rc;
    if (bytes) PP_HIPCHK(ctx, hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    *dev = b.p;
    return PP_OK;
}

extern "C" int pp_polish_begin(pp_ctx *ctx, uint32_t n_contigs, const uint64_t *contig_off,
                               const uint8_t *bases, int bases_mem, const pp_params *params) {
    if (!ctx) return PP_ERR_ARG;
    if (int rdy = pp_ctx_wait(ctx)) return rdy;
    if (!contig_off || !bases || !params || n_contigs == 0)
        return ctx->fail(PP_ERR_ARG, "pp_polish_begin: null argument or no contigs");
    /* check_option_values, polish.rs:277-287 */
    if (params->fraction_valid <= 0.0 || params->fraction_valid >= 1.0)
        return ctx->fail(PP_ERR_QUIT, "--fraction_valid must be between 0 and 1 (exclusive)");
    if (params->fraction_invalid <= 0.0 || params->fraction_invalid >= 1.0)
        return ctx->fail(PP_ERR_QUIT, "--fraction_invalid must be between 0 and 1 (exclusive)");
    if (params->fraction_invalid >= params->fraction_valid)
        return ctx->fail(PP_ERR_QUIT, "--fraction_invalid must be less than --fraction_valid");
    for (uint32_t c = 0; c < n_contigs; c++)
        if (contig_off[c + 1] <= contig_off[c])
            return ctx->fail(PP_ERR_ARG, "pp_polish_begin: contig %u is empty or offsets decrease", c);
    if (contig_off[0] != 0) return ctx->fail(PP_ERR_ARG, "pp_polish_begin: contig_off[0] must be 0");
    uint64_t G = contig_off[n_contigs];
    if (G >= 0xFFFFFFFFull - 4096ull)
        return ctx->fail(PP_ERR_LIMIT, "assembly of %llu bp exceeds the 2^32-4096 bp limit of this version",
                         (unsigned long long)G);
    PP_HIPCHK(ctx, hipSetDevice(ctx->device));
    ctx->n_contigs = n_contigs;
    ctx->contig_off.assign(contig_off, contig_off + n_contigs + 1);
    ctx->G = G;
    ctx->params = *params;
    const void *d;
    int rc = upload(ctx, ctx->b_contig_off, contig_off, (n_contigs + 1) * sizeof(uint64_t), &d);
    if (rc) return rc;
    if (bases_mem == PP_MEM_DEVICE) {
        ctx->d_bases = bases;
    } else {
        rc = upload(ctx, ctx->b_bases, bases, G, &d);
        if (rc) return rc;
        ctx->d_bases = (const uint8_t *)d;
    }
    ctx->job_open = true;
    ctx->job_done = false;
    ctx->have_batch = false;
    ctx->emit.clear();
    memset(&ctx->dbatch, 0, sizeof ctx->dbatch);
    return PP_OK;
}

extern "C" int pp_polish_set_emit(pp_ctx *ctx, const uint64_t *emit_lo, const uint64_t *emit_hi) {
    if (!ctx) return PP_ERR_ARG;
    if (!ctx->job_open) return ctx->fail(PP_ERR_ARG, "pp_polish_set_emit without pp_polish_begin");
    ctx->emit.clear();
    if (!emit_lo && !emit_hi) return PP_OK;
    if (!emit_lo || !emit_hi) return ctx->fail(PP_ERR_ARG, "pp_polish_set_emit: both arrays or neither");
    for (uint32_t c = 0; c < ctx->n_contigs; c++) {
        const uint64_t len = ctx->contig_off[c + 1] - ctx->contig_off[c];
        if (emit_lo[c] > emit_hi[c] || emit_hi[c] > len) {
            ctx->emit.clear();
            return ctx->fail(PP_ERR_ARG, "pp_polish_set_emit: range of contig %u is not inside the contig", c);
        }
        ctx->emit.push_back((uint32_t)emit_lo[c]);
        ctx->emit.push_back((uint32_t)emit_hi[c]);
    }
    return PP_OK;
}

extern "C" int pp_polish_add(pp_ctx *ctx, const pp_aln_batch *b, int mem) {
    if (!ctx) return PP_ERR_ARG;
    if (int rdy = pp_ctx_wait(ctx)) return rdy;
    if (!ctx->job_open) return ctx->fail(PP_ERR_ARG, "pp_polish_add without pp_polish_begin");
    if (ctx->have_batch)
        return ctx->fail(PP_ERR_LIMIT, "this version takes one alignment batch per polish job");
    if (!b) return ctx->fail(PP_ERR_ARG, "pp_polish_add: null batch");
    if (b->n_aln >= 0xFFFFFFFFull)
        return ctx->fail(PP_ERR_LIMIT, "more than 2^32-1 alignments in one batch");
    if (b->n_aln && (!b->contig || !b->ref_start || !b->k || !b->seq_off || !b->seq_len ||
                     !b->cig_off || !b->n_cig || !b->seq || !b->cigar))
        return ctx->fail(PP_ERR_ARG, "pp_polish_add: null array in a non-empty batch");
    PP_HIPCHK(ctx, hipSetDevice(ctx->device));
    if (mem == PP_MEM_DEVICE) {
        ctx->dbatch = *b;
    } else {
        pp_aln_batch d = *b;
        const void *p;
        int rc;
        size_t n = b->n_aln;
#define UP(i, field, T, count)                                                    \
    rc = upload(ctx, ctx->b_in[i], b->field, (size_t)(count) * sizeof(T), &p);   \
    if (rc) return rc;                                                            \
    d.field = (const T *)p;
        UP(0, contig, uint32_t, n)
        UP(1, ref_start, uint32_t, n)
        UP(2, k, uint32_t, n)
        UP(3, seq_off, uint64_t, n)
        UP(4, seq_len, uint32_t, n)
        UP(5, cig_off, uint64_t, n)
        UP(6, n_cig, uint32_t, n)
        UP(7, seq, uint8_t, b->seq_bytes)
        UP(8, cigar, uint32_t, b->n_cig_total)
#undef UP
        ctx->dbatch = d;
    }
    ctx->have_batch = true;
    return PP_OK;
}

static int map_device_error(pp_ctx *ctx, uint64_t key) {
    uint32_t code = (uint32_t)(key & 0xFF);
    unsigned long long idx = (unsigned long long)(key >> 8);
    switch (code) {
    case DE_UNEXPECTED_OP:
        return ctx->fail(PP_ERR_QUIT, "unexpected character (other than M, =, X, I or D) in CIGAR string "
                                      "for alignment record %llu - did you use BWA MEM to generate your alignments?", idx);
    case DE_LEN_MISMATCH:
        return ctx->fail(PP_ERR_QUIT, "CIGAR string for alignment record %llu does not match read sequence", idx);
    case DE_OUT_OF_BOUNDS:
        return ctx->fail(PP_ERR_PANIC, "alignment record %llu runs past the end of its contig", idx);
    case DE_BAD_CONTIG:
        return ctx->fail(PP_ERR_QUIT, "alignment record %llu refers to a contig that is not in the assembly", idx);
    case DE_BAD_K: return ctx->fail(PP_ERR_ARG, "alignment record %llu has k = 0", idx);
    case DE_BAD_RUN: return ctx->fail(PP_ERR_ARG, "alignment record %llu has an empty CIGAR, a zero-length run or an unknown op code", idx);
    case DE_BAD_ENDS: return ctx->fail(PP_ERR_ARG, "alignment record %llu does not start and end with M/= (gate of alignment.rs:155-159 not applied)", idx);
    case DE_NON_ASCII: return ctx->fail(PP_ERR_LIMIT, "assembly position %llu holds a non-ASCII byte", idx);
    case DE_TOO_DEEP: return ctx->fail(PP_ERR_LIMIT, "window %llu has more than 2^21 overlapping alignments", idx);
    case DE_OVERFLOW: return ctx->fail(PP_ERR_LIMIT, "32-bit work-item count or reference span overflow (record/window %llu)", idx);
    default: return ctx->fail(PP_ERR_HIP, "internal device inconsistency %u at %llu", code, idx);
    }
}

// One pass over the whole pipeline with the current buffer capacities.  Everything is enqueued on
// the context's stream without an intermediate host round trip; the sizes that are only known on
// the device (work items, flagged positions, replay scratch, polished bytes) are bounded by
// optimistic capacities, a kernel that would overflow one raises DE_CAPACITY and every later
// kernel then returns at once.  A single read-back of the metadata block ends the pass.
static int run_pipeline(pp_ctx *ctx, std::vector<uint64_t> &meta, uint32_t *n_entries_out) {
    hipStream_t st = ctx->stream;
    const pp_aln_batch &B = ctx->dbatch;
    const uint64_t n = ctx->have_batch ? B.n_aln : 0;
    const uint64_t G = ctx->G;
    const uint32_t nc = ctx->n_contigs;
    const uint32_t nwin = (uint32_t)((G + TILE - 1) / TILE);
    const uint32_t NB = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(512, (n + 4095) / 4096));  // k_scan_cols: <= 512
    const uint64_t chunk = (n + NB - 1) / NB;
    const uint32_t nranges = (nwin + COUNT_RANGE - 1) / COUNT_RANGE;
    int rc;
#define ENS(buf, bytes) if ((rc = dev_ensure(ctx, ctx->buf, (size_t)(bytes)))) return rc
    // metadata block (u64 words): 0 status | 1-2 counters | 3 work items | 4 scratch elements (10: the same, counted
    // as the positions are listed) |
    // 5 polished bytes | 6 ordered replay items | 8 key records | 9 longest fast read | 16.. contig output offsets
    // (nc+1) | then per-contig stats (3 words each)
    const size_t meta_words = 16 + (size_t)nc + 1 + 3 * (size_t)nc;
    ENS(b_meta, meta_words * 8);
    ENS(b_gstart, n * 4); ENS(b_nkeep, n * 4);
    // One level (items straight into their windows) while all windows fit one LDS pass of k_fill; two levels
    // (coarse buckets of COARSE_WINDOWS windows, then k_regroup) beyond that: there the single-level k_fill
    // would re-read its records once per range of 16384 windows.  Measured on MI355X: 5 Mbp: one level 0.19 ms
    // vs two 0.21 ms (+ k_tile 4 % slower on the regrouped order); 250 Mbp: one level 6.8 ms vs two 4.1 ms.
    static const int forced_levels = getenv("PP_BUCKET_LEVELS") ? atoi(getenv("PP_BUCKET_LEVELS")) : 0;  // tuning / tests
    const bool two_level = forced_levels ? forced_levels == 2 : nranges > 1;
    const uint32_t cw = two_level ? COARSE_WINDOWS : 1;
    const uint32_t ncoarse = (nwin + cw - 1) / cw;
    const uint32_t ncranges = (ncoarse + COUNT_RANGE - 1) / COUNT_RANGE;
    ENS(b_hist, (uint64_t)NB * ncoarse * 4); ENS(b_wincnt, (uint64_t)nwin * 4); ENS(b_winoff, ((uint64_t)nwin + 1) * 4);
    ENS(b_ccnt, (uint64_t)ncoarse * 4); ENS(b_coff, ((uint64_t)ncoarse + 1) * 4);
    if (two_level) ENS(b_entB, ctx->cap_ent * 16);
    ENS(b_code, G); ENS(b_winlen, (uint64_t)nwin * 4); ENS(b_winout, ((uint64_t)nwin + 1) * 8);
    ENS(b_entA, ctx->cap_ent * 16);
    ENS(b_flag_pos, ctx->cap_flag * 4); ENS(b_flag_cov, ctx->cap_flag * 4); ENS(b_flag_scr, (ctx->cap_flag + 1) * 8);
    ENS(b_flag_bits, (uint64_t)nwin * (TILE / 8)); ENS(b_win_nflag, (uint64_t)nwin * 4);
    ENS(b_win_slab, (uint64_t)nwin * 4); ENS(b_slabs, (uint64_t)ctx->cap_slabs * 6 * TILE * 4); ENS(b_ents, (uint64_t)ctx->cap_ents * 16);
    if (ctx->debug) ENS(b_keys, (uint64_t)ctx->cap_keys * sizeof(KeyRec));
    ENS(b_scratch, ctx->cap_scr * 16); ENS(b_multi, ctx->cap_multi * sizeof(MultiEnt)); ENS(b_out, ctx->cap_out);
    if (ctx->debug) { ENS(b_dbg_depth, G * 8); ENS(b_dbg_counts, G * 28); ENS(b_dbg_status, G); }
#undef ENS
    u64 *d_meta = (u64 *)ctx->b_meta.p;
    u64 *d_status = d_meta;
    u32 *d_counters = (u32 *)(d_meta + 1);
    u64 *d_ctg_out = d_meta + 16;
    ContigStatsDev *d_stats = (ContigStatsDev *)(d_meta + 17 + nc);
    PP_HIPCHK(ctx, hipMemsetAsync(d_meta, 0, meta_words * 8, st));
    PP_HIPCHK(ctx, hipMemsetAsync(d_status, 0xFF, 8, st));

    u32 *d_gstart = (u32 *)ctx->b_gstart.p, *d_nkeep = (u32 *)ctx->b_nkeep.p;
    u32 *d_hist = (u32 *)ctx->b_hist.p, *d_wincnt = (u32 *)ctx->b_wincnt.p, *d_winoff = (u32 *)ctx->b_winoff.p;
    const u64 *d_ctg = (const u64 *)ctx->b_contig_off.p;
    uint4 *d_entA = (uint4 *)ctx->b_entA.p, *d_entB = (uint4 *)ctx->b_entB.p;
    u32 *d_ccnt = (u32 *)ctx->b_ccnt.p, *d_coff = (u32 *)ctx->b_coff.p;

    if (n) {
        timer_begin(ctx, "prep");
        hipLaunchKernelGGL(k_prep, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (u64)n, B.contig,
                           B.ref_start, B.k, (const u64 *)B.seq_off, B.seq_len, (const u64 *)B.cig_off,
                           B.n_cig, B.cigar, B.seq, d_ctg, nc, d_gstart, d_nkeep, (u32 *)(d_meta + 9), d_status);
        timer_end(ctx);
    }
    timer_begin(ctx, "bucket");
    if (two_level) {
        PP_HIPCHK(ctx, hipMemsetAsync(d_wincnt, 0, (size_t)nwin * 4, st));
        hipLaunchKernelGGL(k_count<COARSE_WINDOWS>, dim3(NB, nranges), dim3(1024), 0, st, (u64)n, (u64)chunk, d_gstart,
                           d_nkeep, nwin, ncoarse, d_hist, d_wincnt);
        hipLaunchKernelGGL(k_scan_cols, dim3((ncoarse + 3) / 4), dim3(256), 0, st, ncoarse, NB, d_hist, d_ccnt);
        hipLaunchKernelGGL(k_scan<u32>, dim3(1), dim3(1024), 0, st, (const u32 *)d_ccnt, (u64)ncoarse, (const u32 *)nullptr,
                           d_coff, d_meta + 3, (u64)ctx->cap_ent, d_status);
        hipLaunchKernelGGL(k_scan<u32>, dim3(1), dim3(1024), 0, st, (const u32 *)d_wincnt, (u64)nwin, (const u32 *)nullptr,
                           d_winoff, (u64 *)nullptr, ~0ull, d_status);
        if (n) {
            hipLaunchKernelGGL(k_fill<COARSE_WINDOWS>, dim3(NB, ncranges), dim3(1024), 0, st, (u64)n, (u64)chunk, d_gstart,
                               d_nkeep, B.k, (const u64 *)B.seq_off, B.seq_len, nwin, ncoarse, (const u32 *)d_hist,
                               (const u32 *)d_coff, d_entB, d_status);
            hipLaunchKernelGGL(k_regroup, dim3(ncoarse), dim3(1024), 0, st, nwin, (const u32 *)d_coff,
                               (const u32 *)d_winoff, (const uint4 *)d_entB, d_entA, d_status);
        }
    } else {
        hipLaunchKernelGGL(k_count<1>, dim3(NB, nranges), dim3(1024), 0, st, (u64)n, (u64)chunk, d_gstart, d_nkeep, nwin,
                           nwin, d_hist, d_wincnt);
        hipLaunchKernelGGL(k_scan_cols, dim3((nwin + 3) / 4), dim3(256), 0, st, nwin, NB, d_hist, d_wincnt);
        hipLaunchKernelGGL(k_scan<u32>, dim3(1), dim3(1024), 0, st, (const u32 *)d_wincnt, (u64)nwin, (const u32 *)nullptr,
                           d_winoff, d_meta + 3, (u64)ctx->cap_ent, d_status);
        if (n)
            hipLaunchKernelGGL(k_fill<1>, dim3(NB, ncranges), dim3(1024), 0, st, (u64)n, (u64)chunk, d_gstart, d_nkeep,
                               B.k, (const u64 *)B.seq_off, B.seq_len, nwin, nwin, (const u32 *)d_hist,
                               (const u32 *)d_winoff, d_entA, d_status);
    }
    timer_end(ctx);

    TileArgs T;
    T.entA = d_entA; T.win_off = d_winoff; T.nwin = nwin;
    T.seq = B.seq; T.seq_off = (const u64 *)B.seq_off; T.cig_off = (const u64 *)B.cig_off;
    T.n_cig = B.n_cig; T.cigar = B.cigar;
    T.bases = ctx->d_bases; T.G = G; T.contig_off = d_ctg; T.n_contigs = nc;
    T.min_depth = ctx->params.min_depth; T.fv = ctx->params.fraction_valid; T.fi = ctx->params.fraction_invalid;
    T.code = (u8 *)ctx->b_code.p; T.win_len = (u32 *)ctx->b_winlen.p;
    T.counters = d_counters; T.cap_flag = (u32)ctx->cap_flag;
    T.flag_pos = (u32 *)ctx->b_flag_pos.p; T.flag_cov = (u32 *)ctx->b_flag_cov.p;
    T.flag_bits = (u32 *)ctx->b_flag_bits.p; T.win_nflag = (u32 *)ctx->b_win_nflag.p;
    T.win_slab = (u32 *)ctx->b_win_slab.p; T.slabs = (u32 *)ctx->b_slabs.p; T.cap_slabs = (u32)ctx->cap_slabs;
    T.stats = d_stats;
    T.maxlen = (const u32 *)(d_meta + 9);
    T.scr_need = d_meta + 10;
    T.seq_bytes = B.seq_bytes;
    T.own = nullptr;
    if (!ctx->emit.empty()) {
        const void *d_own = nullptr;
        if (int rc = upload(ctx, ctx->b_own, ctx->emit.data(), ctx->emit.size() * sizeof(uint32_t), &d_own)) return rc;
        T.own = (const u32 *)d_own;
    }
    T.dbg_depth = (double *)ctx->b_dbg_depth.p; T.dbg_counts = (u32 *)ctx->b_dbg_counts.p;
    T.dbg_status = (u8 *)ctx->b_dbg_status.p; T.status = d_status;
    // PP_DEBUG_REPLAY2=1 (tests): per-position records while order-dependent positions still go through k_exact2,
    // so that its f64 depths can be compared bit for bit (the key records of the TSV are then incomplete)
    static const bool dbg_replay2 = getenv("PP_DEBUG_REPLAY2") && atoi(getenv("PP_DEBUG_REPLAY2")) != 0;
    T.dbg = ctx->debug ? (dbg_replay2 ? 2 : 1) : 0;
    const uint32_t per = (nwin + 7) / 8;
    timer_begin(ctx, "tile");
    hipLaunchKernelGGL(k_tile, dim3(per * 8), dim3(TILE_THREADS), 0, st, T);
    timer_end(ctx);

    u64 *d_scr = (u64 *)ctx->b_flag_scr.p;
    timer_begin(ctx, "exact");
    ExactArgs E;
    E.cap_multi = (u32)ctx->cap_multi; E.cap_flag = (u32)ctx->cap_flag;
    E.flag_pos_w = T.flag_pos; E.flag_cov_w = T.flag_cov; E.flag_bits = T.flag_bits; E.win_nflag = T.win_nflag;
    E.win_slab = T.win_slab; E.slabs = T.slabs; E.ents = (ulonglong2 *)ctx->b_ents.p; E.cap_ents = ctx->cap_ents;
    E.ents_cursor = d_meta + 6;
    E.scr_need = d_meta + 10;
    E.keys = (KeyRec *)ctx->b_keys.p; E.cap_keys = ctx->debug ? ctx->cap_keys : 0; E.n_keys = d_meta + 8;
    E.flag_pos = T.flag_pos; E.flag_cov = T.flag_cov; E.flag_scr = d_scr;
    E.entA = d_entA; E.win_off = d_winoff; E.seq = B.seq; E.seq_off = (const u64 *)B.seq_off;
    E.cig_off = (const u64 *)B.cig_off; E.n_cig = B.n_cig; E.cigar = B.cigar; E.kk = B.k;
    E.bases = ctx->d_bases; E.G = G; E.contig_off = d_ctg; E.n_contigs = nc;
    E.min_depth = T.min_depth; E.fv = T.fv; E.fi = T.fi;
    E.scratch = (ulonglong2 *)ctx->b_scratch.p; E.code = T.code; E.win_len = T.win_len;
    E.counters = d_counters; E.multi = (MultiEnt *)ctx->b_multi.p; E.stats = d_stats;
    E.dbg_depth = T.dbg_depth; E.dbg_counts = T.dbg_counts; E.dbg_status = T.dbg_status;
    E.status = d_status; E.dbg = T.dbg;
    // windows of up to SORT_MAX items: wave-per-position replay; the rest (and key-table overflows)
    // go through the global list to the thread-serial k_exact
    hipLaunchKernelGGL(k_exact2, dim3(nwin), dim3(1024), 0, st, E, nwin);
    hipLaunchKernelGGL(k_scan<u64>, dim3(1), dim3(1024), 0, st, (const u32 *)T.flag_cov, (u64)0, (const u32 *)d_counters,
                       d_scr, d_meta + 4, (u64)ctx->cap_scr, d_status);
    hipLaunchKernelGGL(k_exact, dim3(1024), dim3(64), 0, st, E);
    timer_end(ctx);

    u64 *d_winout = (u64 *)ctx->b_winout.p;
    timer_begin(ctx, "emit");
    hipLaunchKernelGGL(k_scan<u64>, dim3(1), dim3(1024), 0, st, (const u32 *)T.win_len, (u64)nwin, (const u32 *)nullptr,
                       d_winout, d_meta + 5, (u64)ctx->cap_out, d_status);
    hipLaunchKernelGGL(k_compact, dim3(nwin), dim3(TILE_THREADS), 0, st, (const u8 *)T.code, (u64)G, (const u64 *)d_winout,
                       (const MultiEnt *)ctx->b_multi.p, (const u32 *)d_counters, (u8 *)ctx->b_out.p, (const u64 *)d_status);
    const uint64_t nfin = ctx->cap_multi + nc + 1;
    hipLaunchKernelGGL(k_finalize, dim3((unsigned)((nfin + 63) / 64)), dim3(64), 0, st, (const u8 *)T.code, (u64)G,
                       (const u64 *)d_winout, nwin, (const MultiEnt *)ctx->b_multi.p, (const u32 *)d_counters,
                       B.seq, d_ctg, nc, (u8 *)ctx->b_out.p, d_ctg_out, (const u64 *)d_status);
    timer_end(ctx);
    PP_HIPCHK(ctx, hipGetLastError());

    meta.resize(meta_words);
    PP_HIPCHK(ctx, hipMemcpyAsync(meta.data(), d_meta, meta_words * 8, hipMemcpyDeviceToHost, st));
    PP_HIPCHK(ctx, hipStreamSynchronize(st));
    *n_entries_out = (uint32_t)meta[3];
    return PP_OK;
}

extern "C" int pp_polish_finish(pp_ctx *ctx) {
    if (!ctx) return PP_ERR_ARG;
    if (int rdy = pp_ctx_wait(ctx)) return rdy;
    if (!ctx->job_open) return ctx->fail(PP_ERR_ARG, "pp_polish_finish without pp_polish_begin");
    PP_HIPCHK(ctx, hipSetDevice(ctx->device));
    const uint64_t n = ctx->have_batch ? ctx->dbatch.n_aln : 0;
    const uint64_t G = ctx->G;
    const uint32_t nc = ctx->n_contigs;
    // optimistic capacities (grow-only across jobs)
    ctx->cap_ent = std::max<size_t>(ctx->cap_ent, (size_t)(n + n / 4 + 4096));
    ctx->cap_flag = std::max<size_t>(ctx->cap_flag, std::min<size_t>((size_t)G, std::max<size_t>(65536, (size_t)(G / 64))));
    ctx->cap_scr = std::max<size_t>(ctx->cap_scr, (size_t)1 << 20);
    ctx->cap_multi = std::max<size_t>(ctx->cap_multi, 65536);
    ctx->cap_slabs = std::max<size_t>(ctx->cap_slabs, 64);
    ctx->cap_ents = std::max<size_t>(ctx->cap_ents, (size_t)1 << 20);
    if (ctx->debug) ctx->cap_keys = std::max<size_t>(ctx->cap_keys, (size_t)1 << 20);
    ctx->cap_out = std::max<size_t>(ctx->cap_out, (size_t)(G + G / 16 + 65536));

    std::vector<uint64_t> meta;
    uint32_t n_entries = 0;
    int attempt = 0;
    for (;; attempt++) {
        timers_release(ctx);
        int rc = run_pipeline(ctx, meta, &n_entries);
        if (rc) return rc;
        const uint64_t key = meta[0];
        if (key == ~0ull) break;
        if ((key & 0xFF) != DE_CAPACITY && (key & 0xFF) != DE_CAPACITY_LATE) return map_device_error(ctx, key);
        if (attempt >= 6) return ctx->fail(PP_ERR_HIP, "device buffers kept overflowing after %d attempts", attempt);
        // grow whatever was too small (sizes the device got to before it stopped), then rerun
        const uint32_t *cnt = (const uint32_t *)&meta[1];
        bool grew = false;
        static const bool trace = getenv("PP_TIMING") != nullptr;
        auto grow = [&](size_t &cap, uint64_t need, const char *what) {
            if (need <= cap) return;
            if (trace) fprintf(stderr, "[timing] pass %d: %s %zu -> need %llu\n", attempt + 1, what, cap, (unsigned long long)need);
            cap = (size_t)(need + need / 8 + 1024);
            grew = true;
        };
        grow(ctx->cap_ent, meta[3], "work items");
        grow(ctx->cap_flag, std::min<uint64_t>(cnt[0], G), "listed positions");
        grow(ctx->cap_scr, std::max(meta[4], meta[10]), "replay scratch");
        grow(ctx->cap_multi, cnt[1], "multi-byte winners");
        grow(ctx->cap_slabs, cnt[3], "tally slabs");
        grow(ctx->cap_ents, meta[6], "ordered replay items");
        if (ctx->debug) grow(ctx->cap_keys, meta[8], "key records");
        grow(ctx->cap_out, meta[5], "polished bytes");
        if (!grew) return ctx->fail(PP_ERR_HIP, "device reported a capacity overflow that the host cannot locate");
    }
    const uint32_t *cnt = (const uint32_t *)&meta[1];
    ctx->total_out = meta[5];
    ctx->contig_out_off.assign(meta.begin() + 16, meta.begin() + 16 + nc + 1);
    const ContigStatsDev *hs = (const ContigStatsDev *)&meta[17 + nc];
    ctx->n_multi = cnt[1];
    ctx->n_keys = meta[8];
    ctx->stats.resize(nc);
    for (uint32_t c = 0; c < nc; c++) {
        ctx->stats[c].polished_len = ctx->contig_out_off[c + 1] - ctx->contig_out_off[c];
        ctx->stats[c].changed = hs[c].changed;
        ctx->stats[c].zero_depth = hs[c].zero_depth;
        ctx->stats[c].depth_sum = (double)hs[c].depth_fx / (double)(1u << DEPTH_FX_BITS);
    }
    if (ctx->profiling) {
        timers_collect(ctx, &ctx->last_times);
        ctx->last_times.n_entries = n_entries;
        ctx->last_times.n_flagged = cnt[2];
        ctx->last_times.n_passes = (uint64_t)attempt + 1;
    }
    ctx->job_done = true;
    ctx->job_open = false;
    return PP_OK;
}

extern "C" int pp_polish_result_size(pp_ctx *ctx, uint64_t *total_bytes) {
    if (!ctx || !total_bytes) return PP_ERR_ARG;
    if (!ctx->job_done) return ctx->fail(PP_ERR_ARG, "no finished polish job");
    *total_bytes = ctx->total_out;
    return PP_OK;
}

extern "C" int pp_polish_result(pp_ctx *ctx, uint8_t *out, int out_mem, uint64_t *contig_out_off,
                                pp_contig_stats *stats) {
    if (!ctx) return PP_ERR_ARG;
    if (!ctx->job_done) return ctx->fail(PP_ERR_ARG, "no finished polish job");
    if (out && ctx->total_out) {
        PP_HIPCHK(ctx, hipMemcpyAsync(out, ctx->b_out.p, ctx->total_out,
                                      out_mem == PP_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost,
                                      ctx->stream));
        PP_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    if (contig_out_off) memcpy(contig_out_off, ctx->contig_out_off.data(), (ctx->n_contigs + 1) * 8ull);
    if (stats) memcpy(stats, ctx->stats.data(), ctx->n_contigs * sizeof(pp_contig_stats));
    return PP_OK;
}

extern "C" const uint8_t *pp_polish_result_device(pp_ctx *ctx) {
    return (ctx && ctx->job_done) ? (const uint8_t *)ctx->b_out.p : nullptr;
}

extern "C" int pp_polish_set_debug(pp_ctx *ctx, int enable) {
    if (!ctx) return PP_ERR_ARG;
    ctx->debug = enable != 0;
    return PP_OK;
}

extern "C" int pp_polish_positions(pp_ctx *ctx, const pp_positions *o) {
    if (!ctx || !o) return PP_ERR_ARG;
    if (!ctx->job_done || !ctx->debug || !ctx->b_dbg_depth.p)
        return ctx->fail(PP_ERR_ARG, "per-position records need pp_polish_set_debug(1) before pp_polish_finish");
    const uint64_t G = ctx->G;
    hipStream_t st = ctx->stream;
    const u32 *cnt = (const u32 *)ctx->b_dbg_counts.p;
    uint32_t *dst[7] = {o->count_a, o->count_c, o->count_g, o->count_t, o->count_other, o->valid_thr, o->invalid_thr};
    if (o->depth) PP_HIPCHK(ctx, hipMemcpyAsync(o->depth, ctx->b_dbg_depth.p, G * 8, hipMemcpyDeviceToHost, st));
    for (int i = 0; i < 7; i++)
        if (dst[i]) PP_HIPCHK(ctx, hipMemcpyAsync(dst[i], cnt + (uint64_t)i * G, G * 4, hipMemcpyDeviceToHost, st));
    if (o->status) PP_HIPCHK(ctx, hipMemcpyAsync(o->status, ctx->b_dbg_status.p, G, hipMemcpyDeviceToHost, st));
    PP_HIPCHK(ctx, hipStreamSynchronize(st));
    return PP_OK;
}

extern "C" int pp_polish_debug_extra(pp_ctx *ctx, pp_debug_extra *o) {
    if (!ctx || !o) return PP_ERR_ARG;
    memset(o, 0, sizeof *o);
    if (!ctx->job_done || !ctx->debug)
        return ctx->fail(PP_ERR_ARG, "debug records need pp_polish_set_debug(1) before pp_polish_finish");
    hipStream_t st = ctx->stream;
    const uint64_t G = ctx->G, nm = ctx->n_multi, nk = ctx->n_keys;
    std::vector<MultiEnt> hm(nm ? nm : 1);
    std::vector<KeyRec> hk(nk ? nk : 1);
    o->emit = (uint8_t *)malloc(G ? G : 1);
    PP_HIPCHK(ctx, hipMemcpyAsync(o->emit, ctx->b_code.p, G, hipMemcpyDeviceToHost, st));
    if (nm) PP_HIPCHK(ctx, hipMemcpyAsync(hm.data(), ctx->b_multi.p, nm * sizeof(MultiEnt), hipMemcpyDeviceToHost, st));
    if (nk) PP_HIPCHK(ctx, hipMemcpyAsync(hk.data(), ctx->b_keys.p, nk * sizeof(KeyRec), hipMemcpyDeviceToHost, st));
    PP_HIPCHK(ctx, hipStreamSynchronize(st));
    o->n_multi = nm;
    o->multi_pos = (uint32_t *)malloc((nm ? nm : 1) * 4);
    o->multi_len = (uint32_t *)malloc((nm ? nm : 1) * 4);
    o->multi_off = (uint64_t *)malloc((nm ? nm : 1) * 8);
    for (uint64_t i = 0; i < nm; i++) { o->multi_pos[i] = hm[i].pos; o->multi_len[i] = hm[i].len; o->multi_off[i] = hm[i].off; }
    o->n_keys = nk;
    o->key_pos = (uint32_t *)malloc((nk ? nk : 1) * 4);
    o->key_len = (uint32_t *)malloc((nk ? nk : 1) * 4);
    o->key_count = (uint32_t *)malloc((nk ? nk : 1) * 4);
    o->key_off = (uint64_t *)malloc((nk ? nk : 1) * 8);
    for (uint64_t i = 0; i < nk; i++) {
        o->key_pos[i] = hk[i].pos; o->key_len[i] = hk[i].len; o->key_count[i] = hk[i].count; o->key_off[i] = hk[i].off;
    }
    return PP_OK;
}

extern "C" void pp_debug_extra_free(pp_debug_extra *d) {
    if (!d) return;
    free(d->emit); free(d->multi_pos); free(d->multi_len); free(d->multi_off);
    free(d->key_pos); free(d->key_len); free(d->key_count); free(d->key_off);
    memset(d, 0, sizeof *d);
}

extern "C" int pp_ctx_set_profiling(pp_ctx *ctx, int enable) {
    if (!ctx) return PP_ERR_ARG;
    if (int rdy = pp_ctx_wait(ctx)) return rdy;
    ctx->profiling = enable == 2 ? 2 : (enable != 0);
    return PP_OK;
}

extern "C" int pp_polish_kernel_times(pp_ctx *ctx, pp_kernel_times *out) {
    if (!ctx || !out) return PP_ERR_ARG;
    *out = ctx->last_times;
    return PP_OK;
}

// ---- context -----------------------------------------------------------------------------------
__global__ void k_warm(uint32_t *p) {
    if (p) p[threadIdx.x] = 0;
}

static int device_init(pp_ctx *ctx) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return PP_ERR_HIP;
    if (ctx->device < 0 || ctx->device >= n) return PP_ERR_ARG;
    if (hipSetDevice(ctx->device) != hipSuccess) return PP_ERR_HIP;
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) return PP_ERR_HIP;
    return PP_OK;
}

extern "C" int pp_ctx_create(int device, pp_ctx **out) {
    if (!out) return PP_ERR_ARG;
    *out = nullptr;
    pp_ctx *ctx = new pp_ctx();
    ctx->device = device;
    const int rc = device_init(ctx);
    if (rc) {
        delete ctx;
        return rc;
    }
    *out = ctx;
    return PP_OK;
}

extern "C" int pp_ctx_create_async(int device, pp_ctx **out) {
    if (!out) return PP_ERR_ARG;
    pp_ctx *ctx = new pp_ctx();
    ctx->device = device;
    ctx->init_pending = true;
    ctx->init_thread = std::thread([ctx] {
        ctx->init_rc = device_init(ctx);
        if (ctx->init_rc == PP_OK) {  // load the code object and spin up the queue while the host parses
            hipLaunchKernelGGL(k_warm, dim3(1), dim3(64), 0, ctx->stream, (uint32_t *)nullptr);
            (void)hipStreamSynchronize(ctx->stream);
        }
    });
    *out = ctx;
    return PP_OK;
}

extern "C" int pp_ctx_wait(pp_ctx *ctx) {
    if (!ctx) return PP_ERR_ARG;
    if (ctx->init_pending) {
        ctx->init_thread.join();
        ctx->init_pending = false;
        if (ctx->init_rc == PP_OK) (void)hipSetDevice(ctx->device);  // the device is a per-thread setting
    }
    if (ctx->init_rc) {
        ctx->err = "no usable MI355X (HIP) device " + std::to_string(ctx->device) + " -- this build has no CPU path";
        return ctx->init_rc;
    }
    return PP_OK;
}

extern "C" void pp_ctx_destroy(pp_ctx *ctx) {
    if (!ctx) return;
    if (pp_ctx_wait(ctx) != PP_OK) {  // never initialised: nothing on the device to release
        delete ctx;
        return;
    }
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    DevBuf *all[] = {&ctx->b_bases, &ctx->b_contig_off, &ctx->b_status, &ctx->b_gstart, &ctx->b_nkeep,
                     &ctx->b_aflag, &ctx->b_hist, &ctx->b_wincnt, &ctx->b_winoff, &ctx->b_entA, &ctx->b_entB, &ctx->b_ccnt, &ctx->b_coff,
                     &ctx->b_code, &ctx->b_winlen, &ctx->b_winout, &ctx->b_flag_pos, &ctx->b_flag_cov,
                     &ctx->b_flag_scr, &ctx->b_scratch, &ctx->b_multi, &ctx->b_meta, &ctx->b_flag_bits, &ctx->b_win_nflag, &ctx->b_win_slab, &ctx->b_slabs, &ctx->b_ents, &ctx->b_keys, &ctx->b_own,
                     &ctx->b_out, &ctx->b_dbg_depth, &ctx->b_dbg_counts, &ctx->b_dbg_status,
                     &ctx->f_refend[0], &ctx->f_refend[1], &ctx->f_pass[0], &ctx->f_pass[1], &ctx->f_orient,
                     &ctx->f_insert};
    for (DevBuf *b : all) dev_free(*b);
    for (auto &b : ctx->b_in) dev_free(b);
    for (auto &f : ctx->f_in) for (auto &b : f) dev_free(b);
    timers_release(ctx);
    for (hipEvent_t e : ctx->event_pool) (void)hipEventDestroy(e);
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

extern "C" int pp_ctx_set_error_(pp_ctx *ctx, int code, const char *msg) {
    if (ctx) ctx->err = msg ? msg : "";
    return code;
}
extern "C" const char *pp_last_error(const pp_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }
extern "C" int pp_ctx_sync(pp_ctx *ctx) {
    if (!ctx) return PP_ERR_ARG;
    if (int rdy = pp_ctx_wait(ctx)) return rdy;
    PP_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return PP_OK;
}
extern "C" int pp_ctx_download(pp_ctx *ctx, void *host_dst, const void *dev_src, uint64_t bytes) {
    if (!ctx || (bytes && (!host_dst || !dev_src))) return PP_ERR_ARG;
    if (int rdy = pp_ctx_wait(ctx)) return rdy;
    if (!bytes) return PP_OK;
    PP_HIPCHK(ctx, hipMemcpyAsync(host_dst, dev_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    PP_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return PP_OK;
}

extern "C" void *pp_ctx_stream(pp_ctx *ctx) { return ctx && pp_ctx_wait(ctx) == PP_OK ? (void *)ctx->stream : nullptr; }
extern "C" const char *pp_version(void) { return "polypolish-mi355x 0.1.0 (parity target v0.6.1)"; }
