// pp_k_emit.h -- k_compact / k_finalize: emit codes -> polished bytes.
// Part of pp_kernels.hip (included there, in this order, and nowhere else: it defines __global__ kernels).
#pragma once

namespace pp {

// The assembly bytes of a compact run (run_pipeline): contig j of the run is the stretch of the job's assembly that starts
// at src_start[j]; eight bytes per thread, one 8-byte load and store where the stretch lies inside one contig.
__global__ __launch_bounds__(256) void k_sub_bases(const u8 *__restrict__ bases, const u64 *__restrict__ sub_off, u32 n_sub,
                                                   const u64 *__restrict__ src_start, u8 *__restrict__ out, u64 g_sub) {
    const u64 p0 = ((u64)blockIdx.x * blockDim.x + threadIdx.x) * 8ull;
    if (p0 >= g_sub) return;
    u32 lo = 0, hi = n_sub;  // sub_off[lo] <= p0 < sub_off[hi]
    while (hi - lo > 1) {
        const u32 mid = (lo + hi) >> 1;
        if (sub_off[mid] <= p0) lo = mid; else hi = mid;
    }
    if (p0 + 8 <= sub_off[lo + 1]) {
        u64 v;
        __builtin_memcpy(&v, bases + src_start[lo] + (p0 - sub_off[lo]), 8);
        __builtin_memcpy(out + p0, &v, 8);
        return;
    }
    for (u64 p = p0; p < min(p0 + 8, g_sub); p++) {  // a contig boundary inside the stretch (once per contig)
        while (p >= sub_off[lo + 1]) lo++;
        out[p] = bases[src_start[lo] + (p - sub_off[lo])];
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

// One workgroup per window, PP_EMIT_PPT consecutive positions per thread (one load of their codes, and, where every position
// emits exactly one byte -- nearly always -- one store).  Sixteen positions a thread since round 6 (eight before): two waves
// per window instead of four -- the 2442 windows of a 5 Mbp job are then 0.6 rounds of the chip's wave slots, not 1.2.
#ifndef PP_EMIT_PPT
#define PP_EMIT_PPT 16
#endif
constexpr int EMIT_PPT = PP_EMIT_PPT;
static_assert(EMIT_PPT == 8 || EMIT_PPT == 16, "an 8- or 16-byte load of codes per thread");
constexpr int COMPACT_THREADS = TILE / EMIT_PPT;
// NO scan kernel in front of the emission (round 6): a workgroup adds up the output lengths in front of its window itself, from
// sums kept at two levels by whoever writes win_len (win_coarse: per WIN_COARSE = 64 windows, win_coarse2: per 4,096) -- a few
// sixteen-byte loads, asked for with everything else it needs, summed through the barrier it has anyway; the wave that writes
// the job's total checks it against the room.  k_scan over 2,442 windows was 5 us of kernel and 5 us of waiting for its launch in
// a 300 us job, over the 122 k windows of a 250 Mbp job 80 us.  (First cut: every workgroup reading ALL lengths in front of it --
// 16 KB at 4,096 windows, growing with the square of the job: jobs of up to EMIT_FUSE_MAX windows only.  Second: one level of
// coarse sums -- 7.6 KB a workgroup at 122 k windows: k_emit there as slow as k_scan + k_emit had been.)
constexpr u32 EMIT_FUSE_MAX = 4096;  // (what is left of it: up to here k_emit's last workgroup re-initialises the job's metadata alone)
// what this THREAD adds to sum(a[0 .. n)): the elements 4 t + 4 nthreads j + {0..3} below n
template <u32 NTHREADS>
__device__ __forceinline__ u64 sum_part(const u32 *__restrict__ a, u32 n, u32 t) {
    // (four loads in flight at a time; all eight at once made the kernel 112 registers a lane)
    u64 p = 0;
#pragma unroll 1
    for (u32 i0 = 4u * t; i0 < n; i0 += 16u * NTHREADS) {
        uint4 v[4];
#pragma unroll
        for (u32 j = 0; j < 4u; j++) {
            const u32 i = i0 + 4u * NTHREADS * j;
            v[j] = make_uint4(0u, 0u, 0u, 0u);
            if (i < n) v[j] = *(const uint4 *)(a + i);  // (the arrays have room for a multiple of four elements; what lies at or behind n is masked below)
        }
#pragma unroll
        for (u32 j = 0; j < 4u; j++) {
            const u32 i = i0 + 4u * NTHREADS * j;
            p += (u64)(i < n ? v[j].x : 0u) + (i + 1u < n ? v[j].y : 0u) + (i + 2u < n ? v[j].z : 0u) + (i + 3u < n ? v[j].w : 0u);
        }
    }
    return p;
}
// what this THREAD adds to sum(win_len[0 .. w)) = the second-level sums in front of the window's second-level group + the coarse
// sums of that group in front of the window's coarse group + the windows of the coarse group in front of it: at most three
// 16-byte loads a thread up to 262 k windows, in flight together
template <u32 NTHREADS>
__device__ __forceinline__ u64 prefix_part(const u32 *__restrict__ win_coarse, const u32 *__restrict__ win_coarse2,
                                           const u32 *__restrict__ win_len, u32 w, u32 t) {
    static_assert(WIN_COARSE == 64 && WIN_COARSE2 == 64 * WIN_COARSE && NTHREADS >= 32, "two stretches of up to 64 elements: sixteen threads each");
    const u32 g = w / WIN_COARSE, g2 = w / WIN_COARSE2;
    // threads 0-15: the windows [64 g, w); threads 16-31: the coarse sums [64 g2, g)
    const bool second = t >= 16u;
    const u32 *const a = second ? win_coarse + (u64)g2 * 64u : win_len + (u64)g * WIN_COARSE;
    const u32 n = second ? g - g2 * 64u : w - g * WIN_COARSE, i = 4u * (t & 15u);
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (t < 32u && i < n) v = *(const uint4 *)(a + i);  // (asked for in front of the loop below: in flight with its first batch)
    u64 p = sum_part<NTHREADS>(win_coarse2, g2, t);
    if (t < 32u) p += (u64)(i < n ? v.x : 0u) + (i + 1u < n ? v.y : 0u) + (i + 2u < n ? v.z : 0u) + (i + 3u < n ? v.w : 0u);
    return p;
}

// What a workgroup needs before it can do anything -- the job's status, the window's two output offsets, the number of
// multi-byte winners, the thread's codes -- is asked for AT ONCE and looked at afterwards: written as a chain of early
// returns (status, then the offsets, then the codes) it was four memory round trips, one after the other, for a workgroup
// that computes for a few hundred nanoseconds (k_emit 14.6 us for the 2442 windows of a 5 Mbp job, two rounds of the chip's
// wave slots).  The empty asm keeps the compiler from moving the codes' load back down behind the returns.
// FUSED: no scan in front -- win_len instead of win_out; cap_out: the room of `out`
template <bool FUSED>
__device__ __forceinline__ void compact_window(u32 w, const u8 *__restrict__ code, u64 G,
                                               const u64 *__restrict__ win_out, const u32 *__restrict__ win_len, const u32 *__restrict__ win_coarse, const u32 *__restrict__ win_coarse2, u64 cap_out,
                                               const MultiEnt *__restrict__ multi,
                                               const u32 *__restrict__ counters,
                                               u8 *__restrict__ out, const u64 *__restrict__ status) {
    __shared__ u32 wsum[COMPACT_THREADS / 64];
    __shared__ u64 psum[COMPACT_THREADS / 64];
    const u32 t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    const u64 p0 = (u64)w * TILE + (u64)EMIT_PPT * t;
    const bool whole = p0 + EMIT_PPT <= G;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (whole) {  // p0 is a multiple of EMIT_PPT and the code array is 256-byte aligned
        if constexpr (EMIT_PPT == 16) v = *(const uint4 *)(code + p0);
        else { const uint2 h = *(const uint2 *)(code + p0); v.x = h.x; v.y = h.y; }
    }
    const u64 st = *status;
    u64 o0 = 0, o1 = 0, part = 0;
    if constexpr (FUSED) {
        o1 = win_len[w];  // (its own length: all that the early return below asks about)
        part = prefix_part<COMPACT_THREADS>(win_coarse, win_coarse2, win_len, w, t);
    } else {
        o0 = win_out[w];
        o1 = win_out[w + 1];
    }
    const u32 n_multi = counters[1];
    asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));
    if ((st != ~0ull) | (o1 == o0)) return;  // the job is off / nothing to emit (a window of another rank, or all deletions)
    u8 c[EMIT_PPT];
    if (whole) {
        __builtin_memcpy(c, &v, EMIT_PPT);
    } else {
#pragma unroll
        for (int i = 0; i < EMIT_PPT; i++) c[i] = (p0 + i < G) ? code[p0 + i] : (u8)0;
    }
    u32 len[EMIT_PPT], s = 0;
    bool all_one = true;
#pragma unroll
    for (int i = 0; i < EMIT_PPT; i++) {
        len[i] = code_len(c[i], (u32)(p0 + i), multi, n_multi);
        s += len[i];
        all_one = all_one && c[i] != 0 && c[i] < 0x80u;
    }
    u32 inc = s;  // inclusive scan within the wave
    for (int o = 1; o < 64; o <<= 1) {
        u32 v2 = __shfl_up(inc, o, 64);
        if ((int)lane >= o) inc += v2;
    }
    if (lane == 63) wsum[wave] = inc;
    if constexpr (FUSED) {
        const u64 pw = wave_sum64(part);
        if (lane == 0) psum[wave] = pw;
    }
    __syncthreads();
    u32 base = 0;
    for (u32 i = 0; i < wave; i++) base += wsum[i];
    if constexpr (FUSED) {
        const u64 len_w = o1;
        o0 = 0;
        for (u32 i = 0; i < (u32)(COMPACT_THREADS / 64); i++) o0 += psum[i];
        if (o0 + len_w > cap_out) return;  // (no room: the wave that writes the total raises DE_CAPACITY, the host grows the buffer and reruns)
    }
    u64 off = o0 + base + (inc - s);
    if (all_one) {
        __builtin_memcpy(out + off, c, EMIT_PPT);  // one unaligned store
    } else {
#pragma unroll
        for (int i = 0; i < EMIT_PPT; i++) {
            if (c[i] && c[i] < 0x80u) out[off] = c[i];
            off += len[i];
        }
    }
}

// wave [0, n_multi): copies a multi-byte winner into its reserved gap;
// waves [n_multi, n_multi + n_contigs]: output offset of each contig start (and the total).
// The bytes emitted between the window's start and the position are added up by the lanes of the wave (a thread on
// its own walked up to 2047 codes one dependent load after the other: 0.3 ms for the 100 contig starts of configs[3]).
template <bool FUSED>
__device__ __forceinline__ void finalize_entries(u32 first_wave, u32 n_waves, const u8 *__restrict__ code, u64 G,
                                                 const u64 *__restrict__ win_out, u32 nwin,
                                                 const u32 *__restrict__ win_len, const u32 *__restrict__ win_coarse, const u32 *__restrict__ win_coarse2, u64 cap_out, u64 *__restrict__ total_out, u64 *status,
                                                 const MultiEnt *__restrict__ multi,
                                                 const u32 *__restrict__ counters,
                                                 const u8 *__restrict__ seq,
                                                 const u64 *__restrict__ contig_off, u32 n_contigs,
                                                 u8 *__restrict__ out, u64 *__restrict__ ctg_out) {
    const u32 n_multi = counters[1];
    const u32 lane = threadIdx.x & 63u;
    const u32 n_todo = n_multi + n_contigs + 1u;
    // where window w's bytes begin: the scan's, or (fused form, win_len != nullptr) added up by the wave
    auto begin_of = [&](u32 w) -> u64 {
        if constexpr (FUSED) return wave_sum64(prefix_part<64>(win_coarse, win_coarse2, win_len, w, lane)); else return win_out[w];
    };
    for (u32 t = first_wave; t < n_todo; t += n_waves) {
        u64 gp;
        if (t < n_multi) gp = multi[t].pos; else gp = contig_off[t - n_multi];
        u64 off;
        if (gp >= G) {
            off = begin_of(nwin);
            if (FUSED && t == n_todo - 1u && lane == 0) {  // the job's total (the scan's part in the fused form)
                __hip_atomic_store(total_out, off, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (write-through: emit_tail)
                if (off > cap_out) report(status, off, DE_CAPACITY);
            }
        } else {
            const u32 w = (u32)(gp / TILE);
            u32 part = 0;
            bool emits;
            if constexpr (FUSED) emits = win_len[w] != 0u; else emits = win_out[w + 1] != win_out[w];
            if (emits)  // (a window that emits nothing may be one nobody worked on: no codes there)
                for (u64 q = (u64)w * TILE + lane; q < gp; q += 64) part += code_len(code[q], (u32)q, multi, n_multi);
            off = begin_of(w) + wave_sum(part);
        }
        if (t < n_multi) {
            if (lane == 0 && !(FUSED && off + multi[t].len > cap_out)) {
                const u8 *s = seq + multi[t].off;
                for (u32 b = 0; b < multi[t].len; b++)
                    if (s[b] != (u8)'-') out[off++] = s[b];
            }
        } else if (lane == 0) {
            __hip_atomic_store(&ctg_out[t - n_multi], off, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (write-through: emit_tail)
        }
    }
}

// The job's results go to the HOST from here (round 6): the workgroup that finishes last -- a counter in the metadata block --
// copies the block into pinned host memory, and, when the job is through, sets the block and the per-window counts up for the
// next job of the same shape (k_meta_init's part).  Until then a step ended with a copy kernel of the runtime (8 us for 2 KB:
// a launch and a PCIe round trip of its own) and k_meta_init (4.7 us and the host's turn-around in front of it) behind k_emit.
// "Last" is counted in two levels -- groups of EMIT_DONE_GROUP workgroups, then the groups --, every counter in a 64-byte stretch of
// its own: one counter for all of them made k_emit five times as long (0.017 -> 0.10 ms for 4,490 workgroups, 0.23 -> 1.6 ms
// for 26,500: agent-scope atomics on ONE address go through at 30-55 ns apiece).  The counters reset themselves.
constexpr u32 EMIT_DONE_GROUP = 64, EMIT_DONE_STRIDE = 8;  // (workgroups per first-level counter; u64 words between two counters)
__host__ __device__ constexpr u64 emit_done_words(u64 blocks) { return (1ull + (blocks + EMIT_DONE_GROUP - 1) / EMIT_DONE_GROUP) * EMIT_DONE_STRIDE; }
struct EmitTail {
    u64 *meta;      // the job's metadata block ...
    u32 words;
    u32 reinit;     // 0: leave the block as it is; 1: set it up for the next job if this one is through AND flagged nothing (the host
                    // would run the replays and the emission once more); 2: ... if this one is through
    u64 *host;      // ... its copy in pinned host memory: words + 2 (nullptr: the host copies, nothing to do here) --
                    // [words] = 1: the block is set up again; [words + 1] = serial, written last
    u64 serial;
    u64 *done;      // the counters: [0] groups that are through, [EMIT_DONE_STRIDE * (1 + g)] workgroups of group g (zero between launches)
    u32 *zero_a, *zero_b, *zero_c, *zero_d;  // what k_meta_init zeroes per window (pairs; nullptr: nothing)
    u32 n_zero;
    u32 *zero_e;    // ... and the coarse sums of the output lengths
    u32 n_zero_e;
    u32 ordered;    // 1: the host polls [words + 1] instead of waiting for the kernel's end
};
__device__ __forceinline__ void emit_tail(const EmitTail &Z) {
    __shared__ u32 s_last;
    if (!Z.host) return;
    // What k_emit writes into the block travels as agent-scope atomics (the status: atomicMin; the contig offsets and the total:
    // write-through stores, finalize_entries), ordered against the counter by waiting for their acknowledgements -- as k_tile's
    // heavy-window rendezvous, see there for the memory-model note.  (With a __threadfence() per workgroup -- a write-back of the
    // XCD's L2 each -- k_emit took 0.11 ms instead of 0.02.)
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__)
#error "k_emit's last-workgroup hand-over relies on gfx942/gfx950 agent-scope store/load semantics"
#endif
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (threadIdx.x == 0) {
        u64 *const mine = Z.done + (u64)EMIT_DONE_STRIDE * (1u + blockIdx.x / EMIT_DONE_GROUP);
        const u32 in_group = min(EMIT_DONE_GROUP, gridDim.x - blockIdx.x / EMIT_DONE_GROUP * EMIT_DONE_GROUP);
        bool last = false;
        if (__hip_atomic_fetch_add(mine, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (u64)in_group - 1ull) {
            __hip_atomic_store(mine, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last = __hip_atomic_fetch_add(Z.done, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (u64)((gridDim.x + EMIT_DONE_GROUP - 1u) / EMIT_DONE_GROUP) - 1ull;
            if (last) __hip_atomic_store(Z.done, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        s_last = last;
    }
    __syncthreads();
    if (!s_last) return;
    // (one round trip: what every thread has to know, and its first word of the block)
    const u32 i0 = threadIdx.x;
    const u64 st = __hip_atomic_load(Z.meta, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const u64 c01 = __hip_atomic_load(Z.meta + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const u64 c23 = __hip_atomic_load(Z.meta + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    u64 v = __hip_atomic_load(Z.meta + min(i0, Z.words - 1u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool flagged = (u32)c01 != 0u || (u32)c23 != 0u;  // (counters 0 and 2: listed for k_exact, flagged in all)
    const bool again = st == ~0ull && (Z.reinit == 2u || (Z.reinit == 1u && !flagged));
    __syncthreads();  // (every wave has looked at words 0-2 before their owner resets them)
    for (u32 i = i0; i < Z.words; i += blockDim.x) {
        if (i != i0) v = __hip_atomic_load(Z.meta + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(Z.host + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (again) Z.meta[i] = i == 0 ? ~0ull : 0ull;
    }
    if (!Z.ordered && threadIdx.x == 0) {  // (a host that waits for the kernel's end: nothing to order)
        __hip_atomic_store(Z.host + Z.words, again ? 1ull : 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(Z.host + Z.words + 1, Z.serial, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (again) {
        for (u32 i = threadIdx.x; i < Z.n_zero_e; i += blockDim.x) Z.zero_e[i] = 0;
        if (Z.zero_a)
            for (u32 i = threadIdx.x; i < Z.n_zero; i += blockDim.x) { Z.zero_a[i] = 0; Z.zero_b[i] = 0; }
        if (Z.zero_c)
            for (u32 i = threadIdx.x; i < Z.n_zero; i += blockDim.x) { Z.zero_c[i] = 0; Z.zero_d[i] = 0; }
    }
    if (Z.ordered) {  // a host that polls the serial: written when the copy's write-through stores are acknowledged
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_store(Z.host + Z.words, again ? 1ull : 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __builtin_amdgcn_s_waitcnt(0);
            __hip_atomic_store(Z.host + Z.words + 1, Z.serial, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// One launch for both: blocks [0, nwin) compact their window, the blocks behind them finalize (the two touch different
// bytes of the output and read the same inputs).
// n_work = windows to compact: all of them, or (sharded job, own_win: see k_tile) the ones this context works on.
// FUSED: the form for jobs of up to EMIT_FUSE_MAX windows -- no k_scan in front, win_len instead of win_out; cap_out / total_out /
// status: what the scan checked and wrote.  (An instance of its own: the lengths in flight cost registers a large job's
// hundred thousand workgroups do without.)
template <bool FUSED>
__global__ __launch_bounds__(COMPACT_THREADS) void k_emit(const u8 *__restrict__ code, u64 G, const u64 *__restrict__ win_out,
                                                          const u32 *__restrict__ win_len, const u32 *__restrict__ win_coarse, const u32 *__restrict__ win_coarse2, u64 cap_out, u64 *__restrict__ total_out,
                                                          u32 nwin, u32 n_work, const u32 *__restrict__ own_win,
                                                          const MultiEnt *__restrict__ multi,
                                                          const u32 *__restrict__ counters, const u8 *__restrict__ seq,
                                                          const u64 *__restrict__ contig_off, u32 n_contigs,
                                                          u8 *__restrict__ out, u64 *__restrict__ ctg_out,
                                                          u64 *__restrict__ status, EmitTail Z) {
    if (blockIdx.x < n_work) {
        u32 w = blockIdx.x;
        if (own_win) {
            const u32 nr = own_win[0];
            const u32 *first = own_win + 1, *before = own_win + 1 + nr;
            u32 lo = 0, hi = nr;  // before[lo] <= w < before[hi]
            while (hi - lo > 1) {
                const u32 mid = (lo + hi) >> 1;
                if (before[mid] <= w) lo = mid; else hi = mid;
            }
            w = first[lo] + (w - before[lo]);
        }
        compact_window<FUSED>(w, code, G, win_out, win_len, win_coarse, win_coarse2, cap_out, multi, counters, out, status);
    } else if (*status == ~0ull) {
        constexpr u32 WPB = COMPACT_THREADS / 64;
        finalize_entries<FUSED>((blockIdx.x - n_work) * WPB + (threadIdx.x >> 6), (gridDim.x - n_work) * WPB, code, G, win_out, nwin, win_len, win_coarse, win_coarse2, cap_out,
                         total_out, status, multi, counters, seq, contig_off, n_contigs, out, ctg_out);
    }
    emit_tail(Z);
}

}  // namespace pp
