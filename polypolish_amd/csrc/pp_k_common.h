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

// ---- byte-parallel helpers of the read / assembly comparison --------------------------------------------------
constexpr u32 PLAIN_MIN_LEN = 8;    // the trim reads the last four bases; shorter reads take the SLOW class
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
// the same through a non-temporal load (a byte-aligned vector type keeps it one unaligned 16-byte instruction)
typedef u32 u32x4_unaligned __attribute__((ext_vector_type(4), aligned(1)));
__device__ __forceinline__ uint4 load16_stream(const u8 *p) {
    const u32x4_unaligned v = __builtin_nontemporal_load((const u32x4_unaligned *)p);
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ u32 load4_unaligned(const u8 *p) {
    u32 v;
    __builtin_memcpy(&v, p, 4);
    return v;
}

// PLAIN class geometry: a group of GW lanes owns one read; lane s of the group owns read bytes [32s, 32s+32), fetched
// with two 16-byte global loads at the read's own (arbitrary) byte offset -- gfx950 global loads need no alignment.
// GW is picked per batch of records from its longest PLAIN read: 5 lanes (12 reads per wave pass) up to 160 bases,
// 6 (10 reads) up to 192, 8 (8 reads) up to 252.
template <int GW>
struct PlainCfg {
    static constexpr u32 IPP = 64 / GW;                    // reads per wave pass
    static constexpr u32 SPAN = 32 * GW;
    static constexpr u32 MAXL = SPAN < FAST_MAX_LEN ? SPAN : FAST_MAX_LEN;
    __device__ static __forceinline__ u32 group(u32 lane) {
        return GW == 8 ? lane >> 3 : (GW == 5 ? (lane * 52u) >> 8 : (lane * 43u) >> 8);
    }
};

}  // namespace pp
