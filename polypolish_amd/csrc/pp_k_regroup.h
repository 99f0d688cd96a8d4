// pp_k_regroup.h -- k_regroup: gathers the pieces of one coarse bucket from every segment k_stream wrote and lays its
// units out window by window (items first, then events), and k_scan, the single-block exclusive scan the replay and
// emit stages use.  Part of pp_kernels.hip (included there and nowhere else).
#pragma once

namespace pp {

constexpr u32 REGROUP_THREADS = 1024;
constexpr u32 MAX_SUB = 256;  // windows per coarse bucket (8 bits of a unit)

struct RegroupArgs {
    u32 nwin, shift, nbk;
    const u32 *n_seg;             // segments written (device)
    u32 cap_segs;
    const unsigned short *seg_off;
    const u64 *seg_base;
    const u64 *units_in;
    const LateEnt *late;
    const u64 *n_late;
    u64 cap_late;
    u64 *units_out;
    u64 cap_out;
    u64 *out_cursor;
    u64 *win_start;               // per window: first unit in units_out
    u32 *win_nitem, *win_nev;     // PLAIN + SLOW units, then EVENT units
    u64 *status;
};

// One workgroup per bucket.  Two passes over the bucket's pieces (the second one is served by L2): count per
// (window, class), reserve the bucket's stretch of the output with ONE global atomic, scatter.
__global__ __launch_bounds__(REGROUP_THREADS) void k_regroup(RegroupArgs A) {
    __shared__ u32 cnt[2][MAX_SUB];   // [class][window in bucket]: counts, then cursors
    __shared__ u64 s_base;
    if (*A.status != ~0ull) return;  // a record error or a capacity overflow: the host reruns or reports
    // consecutive buckets stay on one XCD (their pieces share 128-byte lines of every segment)
    const u32 per = gridDim.x >> 3;
    const u32 c = (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
    if (c >= A.nbk) return;
    const u32 tid = threadIdx.x;
    const u32 nsub = 1u << A.shift;
    const u32 w_lo = c << A.shift, w_hi = min(A.nwin, w_lo + nsub) - 1u;
    for (u32 i = tid; i < 2u * MAX_SUB; i += REGROUP_THREADS) (&cnt[0][0])[i] = 0;
    __syncthreads();
    const u32 nseg = min(*A.n_seg, A.cap_segs);
    const u64 nlate = min(*A.n_late, A.cap_late);
    const u32 stride = A.nbk + 1u;
    for (int pass = 0; pass < 2; pass++) {
        // One lane per (segment, bucket) piece; loads batched so that several are in flight per lane: the offset
        // pairs and bases of four segments at once, then four units of a piece at a time.
        for (u32 sg0 = tid; sg0 < nseg; sg0 += 4u * REGROUP_THREADS) {
            u32 o0[4], o1[4];
            const u64 *pp[4];
#pragma unroll
            for (u32 q = 0; q < 4; q++) {
                const u32 sg = sg0 + q * REGROUP_THREADS;
                const bool ok = sg < nseg;
                const unsigned short *row = A.seg_off + (u64)(ok ? sg : 0u) * stride;
                o0[q] = ok ? row[c] : 0u;
                o1[q] = ok ? row[c + 1] : 0u;
                pp[q] = A.units_in + A.seg_base[ok ? sg : 0u];
            }
#pragma unroll
            for (u32 q = 0; q < 4; q++) {
                const u64 *p = pp[q];
                for (u32 i = o0[q]; i < o1[q]; i += 4) {
                    u64 v[4];
#pragma unroll
                    for (u32 r = 0; r < 4; r++) v[r] = i + r < o1[q] ? p[i + r] : (u64)UNIT_NOP;
#pragma unroll
                    for (u32 r = 0; r < 4; r++) {
                        const u32 lo = (u32)v[r], tag = lo & 3u, sub = lo >> 24;
                        if (tag == UNIT_NOP) continue;
                        const u32 at = atomicAdd(&cnt[tag == UNIT_EVENT ? 1u : 0u][sub], 1u);
                        if (pass) A.units_out[s_base + at] = v[r];
                    }
                }
            }
        }
        for (u64 e = tid; e < nlate; e += REGROUP_THREADS) {
            const LateEnt le = A.late[e];
            if (le.w1 < w_lo || le.w0 > w_hi) continue;
            const u32 cls = ((u32)le.unit & 3u) == UNIT_EVENT ? 1u : 0u;
            for (u32 w = max(le.w0, w_lo); w <= min(le.w1, w_hi); w++) {
                const u32 at = atomicAdd(&cnt[cls][w - w_lo], 1u);
                if (pass) A.units_out[s_base + at] = unit_with_sub(le.unit, w - w_lo);
            }
        }
        __syncthreads();
        if (pass) break;
        // layout of the bucket's stretch: window by window, items then events; counts become cursors
        if (tid < 64u) {
            u32 ni[MAX_SUB / 64], ne[MAX_SUB / 64], sum = 0;
#pragma unroll
            for (u32 q = 0; q < MAX_SUB / 64; q++) {
                const u32 sub = tid * (MAX_SUB / 64) + q;
                ni[q] = cnt[0][sub]; ne[q] = cnt[1][sub];
                sum += ni[q] + ne[q];
            }
            u32 inc = sum;
            for (int o = 1; o < 64; o <<= 1) {
                const u32 v = (u32)__shfl_up((int)inc, o, 64);
                if ((int)tid >= o) inc += v;
            }
            const u32 total = (u32)__shfl((int)inc, 63, 64);
            u64 base = 0;
            if (tid == 0) {
                base = atomicAdd(A.out_cursor, (u64)total);
                if (base + total > A.cap_out) { report(A.status, base + total, DE_CAPACITY); base = ~0ull; }
                s_base = base;
            }
            base = ((u64)(u32)__shfl((int)(u32)(base >> 32), 0, 64) << 32) | (u64)(u32)__shfl((int)(u32)base, 0, 64);
            u32 run = inc - sum;
#pragma unroll
            for (u32 q = 0; q < MAX_SUB / 64; q++) {
                const u32 sub = tid * (MAX_SUB / 64) + q;
                cnt[0][sub] = run;
                cnt[1][sub] = run + ni[q];
                if (sub < nsub && w_lo + sub <= w_hi) {
                    A.win_start[w_lo + sub] = base + run;  // meaningless (and unused) after a capacity overflow
                    A.win_nitem[w_lo + sub] = ni[q];
                    A.win_nev[w_lo + sub] = ne[q];
                    if (ni[q] >= MAX_BUCKET) report(A.status, w_lo + sub, DE_TOO_DEEP);
                }
                run += ni[q] + ne[q];
            }
        }
        __syncthreads();
        if (s_base == ~0ull) return;
    }
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

}  // namespace pp
