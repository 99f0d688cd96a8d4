// pp_k_common.h -- types, LDS row layout and the small device helpers every kernel of the polish path shares.
// Part of pp_kernels.hip (included there, in this order, and nowhere else: it defines __global__ kernels).
#pragma once

namespace pp {

typedef unsigned long long u64;
typedef uint32_t u32;
typedef uint8_t u8;

// LDS counter rows of one window.  A..OTH hold EXPLICIT tallies (every base of slow-class items, and
// the mismatching bases of fast-class items); COV is the coverage difference array of the fast
// class (+1 at the first kept position of a read, -1 one past the last; prefix-summed before the
// vote) and MIS the number of fast-class bases that differ from the assembly, so that the tally of
// the assembly's own base is  explicit + COV - MIS  without touching LDS once per matching base.
// (MIS and COV before DEF: their rows start below 64 KiB, so a tally's second atomic and a read's two coverage atomics take
// the row as the instruction's 16-bit offset; the deficit row is touched by the rare jobs with shared reads only)
enum { ROW_A = 0, ROW_C = 1, ROW_T = 2, ROW_G = 3, ROW_DEL = 4, ROW_OTH = 5, ROW_MIS = 6, ROW_COV = 7,
       ROW_DEF = 8, N_ROWS = 9 };

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

// the job's metadata block: word 0 is the status ("no error" = all ones: errors are combined with atomicMin), the
// counters and sizes behind it start at zero -- one launch where two memsets left a gap between them
// ... and the job's table of the vote's two thresholds for every INTEGER depth below VOTE_TAB_N (d_bankers is defined below):
// thr[2n] = bankers_rounding(n * fraction_valid), thr[2n + 1] = bankers_rounding(n * fraction_invalid) (pileup.rs:70-72),
// computed with the very operations the vote uses on an f64 depth, so that a position whose depth is an integer -- every
// position no shared read touches -- reads its thresholds instead of multiplying and rounding twice.
constexpr u32 VOTE_TAB_N = 4096;
// Output bytes are counted per window (win_len) and per WIN_COARSE consecutive windows (win_coarse, added to with atomics by whoever
// adds to win_len): a workgroup of k_emit finds where its window's bytes begin from the coarse sums in front of its group and the
// windows of the group in front of it -- no scan kernel, whatever the job's size (pp_k_emit.h).
constexpr u32 WIN_COARSE = 64, WIN_COARSE2 = 64 * 64;  // (and per WIN_COARSE2 windows on top: a 250 Mbp job has 122 k windows)
__device__ __forceinline__ void note_out_len(u32 *win_coarse, u32 *win_coarse2, u32 w, u32 len) {
    atomicAdd(&win_coarse[w / WIN_COARSE], len);
    atomicAdd(&win_coarse2[w / WIN_COARSE2], len);
}
__device__ __forceinline__ u32 d_bankers(double x);
__global__ void k_meta_init(u64 *meta, u32 words, u32 *zero_a, u32 *zero_b, u32 *zero_c, u32 *zero_d, u32 n_zero, u32 *zero_e, u32 n_zero_e,
                            u32 *thr, double fv, double fi) {
    if (blockIdx.x == 0) {
        for (u32 i = threadIdx.x; i < words; i += blockDim.x) meta[i] = i == 0 ? ~0ull : 0ull;
        for (u32 i = threadIdx.x; i < n_zero_e; i += blockDim.x) zero_e[i] = 0;  // (the coarse sums of the output lengths)
        for (u32 n = threadIdx.x; n < VOTE_TAB_N; n += blockDim.x) {
            thr[2 * n] = d_bankers(__dmul_rn((double)n, fv));
            thr[2 * n + 1] = d_bankers(__dmul_rn((double)n, fi));
        }
        return;
    }
    // blocks 1..: 4096 elements of the arrays each (a sharded job's win_len / win_nflag; the direct path's counts of extras)
    const u32 lo = (blockIdx.x - 1u) * 4096u, hi = min(n_zero, lo + 4096u);
    if (zero_a)
        for (u32 i = lo + threadIdx.x; i < hi; i += blockDim.x) { zero_a[i] = 0; zero_b[i] = 0; }
    if (zero_c)
        for (u32 i = lo + threadIdx.x; i < hi; i += blockDim.x) { zero_c[i] = 0; zero_d[i] = 0; }
}

__device__ __forceinline__ void report(u64 *status, u64 idx, u32 code) {
    atomicMin(status, (idx << 8) | (u64)code);
}
// Job state as k_tile / k_exact2 see it: 0 running, 1 only a late capacity overflow so far (keep counting the
// needs, every write is guarded by its capacity), 2 aborted.
__device__ __forceinline__ int job_state(const u64 *status) {
    const u64 s = *status;
    return s == ~0ull ? 0 : ((s & 0xFFu) == DE_CAPACITY_LATE ? 1 : 2);
}

// ---- work items (16 bytes, written by k_fill, see there) ----
__device__ __forceinline__ int item_rel(u32 z) { return (int)(z << 2) >> 2; }          // start minus window start
__device__ __forceinline__ u32 item_flags(u32 y, u32 z) { return ((y >> 16) & 3u) | ((z >> 28) & 0xCu); }
// positions an item can reach before its trim: kept entries (trim done by k_prep), 1 (a point), or the read / piece length
__device__ __forceinline__ u32 item_extent(u32 x, u32 y, u32 z) {
    return ((y >> 16) & 3u) ? x : ((z >> 31) ? 1u : (y >> 24));
}
// the nkeep word of a record (k_prep -> k_count / k_fill): reference span of the record, for the windows it reaches
__device__ __forceinline__ u32 nkw_span(u32 word) { return (word >> 30) == NKW_INDEL1 ? (word & 0x1FFu) : (word & 0x3FFFFFFFu); }
// The pieces of a record, as (global start, positions) pairs: one for most records, three for a one-indel read (the
// flank in front, the entry at the indel, the flank behind).  f(piece, g, span) is called for pieces 0 [, 1, 2].
template <typename F>
__device__ __forceinline__ void for_each_piece(u32 g, u32 word, F f) {
    if ((word >> 30) != NKW_INDEL1) { f(0u, g, word & 0x3FFFFFFFu); return; }
    const u32 span = word & 0x1FFu, a = (word >> 9) & 0xFFu, del = (word >> 17) & 1u;
    // insertion (aM1IbM, span = a + b):   flank [0, a-1) | entry a-1 (two-byte key) | flank from entry a on
    // deletion  (aM1DbM, span = a + 1 + b): flank [0, a)   | entry a (empty)          | flank from entry a+1 on
    const u32 l1 = del ? a : a - 1u;
    f(0u, g, l1);
    f(1u, g + l1, 1u);
    f(2u, g + l1 + 1u, span - l1 - 1u);
}

// misc.rs:208-215 for x >= 0
__device__ __forceinline__ u32 d_bankers(double x) {
    u32 r = (x >= 4294967295.0) ? 0xFFFFFFFFu : (u32)x;
    double f = x - trunc(x);
    if (f < 0.5) return r;
    if (f > 0.5) return r + 1u;
    return r + (r & 1u);
}

// depth-share class of k good alignments per read (pp_internal.h: KCLASS_*) and back
__device__ __forceinline__ u32 kclass_of(u32 k) {
    if (k == 1) return 0;
    if ((k & (k - 1)) == 0) {
        const u32 j = 31u - (u32)__clz((int)k);
        if (j <= KCLASS_DYADIC_MAX) return j;
        return KCLASS_OTHER;
    }
    return k <= KCLASS_SMALL_MAX_K ? k + KCLASS_SMALL_BASE : KCLASS_OTHER;
}
__device__ __forceinline__ u32 k_of_class(u32 kc, const u32 *kk, u32 rec) {
    return kc == 0 ? 1u : (kc <= KCLASS_DYADIC_MAX ? (1u << kc) : (kc != KCLASS_OTHER ? kc - KCLASS_SMALL_BASE : kk[rec]));
}
// fixed-point bits of a window with n_items work items: a position's deficit (< items covering it * 2^b) stays below 2^31
__device__ __forceinline__ u32 win_fx_bits(u32 n_items) {
    const u32 bits = 32u - (u32)__clz((int)(n_items | 1u));
    return min((u32)DEPTH_FX_BITS, 31u - bits);
}
// The deficit (2^b - share) of a read of class kc in units of 2^-b, the share rounded to the nearest unit; *inexact = the
// share is not a multiple of the unit (its positions' depths are then only bounded, see DEPTH_FX_BITS).
__device__ __forceinline__ u32 share_deficit(u32 kc, u32 b, const u32 *kk, u32 rec, bool *inexact) {
    if (kc <= KCLASS_DYADIC_MAX && kc <= b) { *inexact = false; return (1u << b) - (1u << (b - kc)); }
    const u32 k = k_of_class(kc, kk, rec), one = 1u << b;
    const u32 s = (one + (k >> 1)) / k;
    *inexact = s * k != one;
    return one - s;
}

// counter row of one read byte: exact "A"/"C"/"G"/"T" (pileup.rs:58-61), "-" shares the
// deletion key, everything else goes to the string-keyed table
__device__ __forceinline__ int row_of(u32 c) {
    u32 t = (c >> 1) & 3u;  // A->0 C->1 T->2 G->3
    u32 expect = (0x47544341u >> (t * 8u)) & 0xFFu;
    return (c == expect) ? (int)t : (c == (u32)'-' ? ROW_DEL : ROW_OTH);
}

// Inclusive prefix sum over the wave's 64 lanes with data-parallel primitives (gfx9 DPP: shifts inside a row of 16 lanes, then
// lane 15 of a row to the next row, lane 31 to the upper half): six v_add with a DPP operand and no address registers, where
// six __shfl_up are six ds_bpermute with an index register each -- indices the compiler kept alive from k_tile's prologue to
// its prefix sums, in scratch memory across the item loop (round 6).
__device__ __forceinline__ u32 wave_scan_incl(u32 v) {
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);   // row_shr:1
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);   // row_shr:2
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);   // row_shr:4
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);   // row_shr:8
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);  // row_bcast:15 into rows 1 and 3
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);  // row_bcast:31 into rows 2 and 3
    return v;
}
// (every lane of the wave has to be active where these two are called: an inactive lane's register is what a DPP operand reads)
__device__ __forceinline__ u32 wave_sum_dpp(u32 v) {  // (the same value in every lane)
    return (u32)__builtin_amdgcn_readlane((int)wave_scan_incl(v), 63);
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

}  // namespace pp
