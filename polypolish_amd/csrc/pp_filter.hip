// pp_filter.hip -- gfx950 kernels of the paired-read insert-size filter (seam A of
// include/polypolish_hip.h): ref_end from the CIGAR runs, orientation + insert size of the
// uniquely-aligned pairs, and the per-alignment pass/fail rule.  Pure integer work, one lane per
// read, structure-of-arrays loads; HBM-bound.  Since round 5 ONE pass over the reads (k_filter_reads) and a pass over
// the few reads whose verdicts need the thresholds (k_filter_listed).
#include "pp_internal.h"

#include <cstring>

namespace pp {

typedef unsigned long long u64;
typedef uint32_t u32;
typedef uint8_t u8;

// get_orientation, filter.rs:189-209 -> 0 fr, 1 rf, 2 ff, 3 rr.  Argument order matters.
__device__ __forceinline__ u32 orientation_of(u32 flags1, u64 start1, u64 end1, u32 flags2,
                                              u64 start2, u64 end2) {
    const bool f1 = (flags1 & 16u) == 0, f2 = (flags2 & 16u) == 0;
    const u64 p1 = f1 ? start1 : end1, p2 = f2 ? start2 : end2;
    if (f1 != f2) {
        const bool first_is_f = (p1 < p2) ? f1 : f2;
        return first_is_f ? 0u : 1u;
    }
    if (f1) return (p1 < p2) ? 2u : 3u;
    return (p2 < p1) ? 2u : 3u;
}

// get_insert_size, filter.rs:212-218 (`as u32` truncation included)
__device__ __forceinline__ u32 insert_of(u64 s1, u64 e1, u64 s2, u64 e2) {
    const u64 lo = min(min(s1, e1), min(s2, e2)), hi = max(max(s1, e1), max(s2, e2));
    return (u32)(hi - lo);
}

struct FileDev {
    const u32 *ref_id, *ref_start, *flags, *grp_off, *grp_idx, *read;
    const u64 *ref_end;   // precomputed (the device loader), or NULL: from the runs below
    const u64 *cig_off;
    const u32 *n_cig, *cigar;
    u64 n_aln;
};

// Alignment::get_ref_end, alignment.rs:138-149: ref_start + sum of M, D, N, =, X run lengths -- where it is needed, from
// the runs (rounds 1-4 had a kernel of its own write it to memory for the other two to read back: 28 bytes per alignment
// in, 8 out, 8 in again)
__device__ __forceinline__ u64 ref_end_of(const FileDev &F, u32 a, u64 start) {
    if (F.ref_end) return F.ref_end[a];
    u64 end = start;
    const u32 *cg = F.cigar + F.cig_off[a];
    const u32 nr = F.n_cig[a];
    for (u32 r = 0; r < nr; r++) {
        const u32 op = cg[r], o = op & 15u;
        if (o == PP_OP_M || o == PP_OP_D || o == PP_OP_N || o == PP_OP_EQ || o == PP_OP_X) end += op >> 4;
        else if (o == (u32)PP_OP_UNPARSEABLE) return PP_REF_END_UNPARSEABLE;
    }
    return end;
}

// alignment_pass_qc, filter.rs:352-377, for alignment a of `self` (n_this of its read there), mates [p0, p1) in `other`
__device__ __forceinline__ u8 pass_qc(const FileDev &self, u32 a, u32 n_this, const FileDev &other, u32 p0, u32 p1, u32 low, u32 high,
                                      u32 correct, u32 *poisoned) {
    if (p1 == p0 || n_this == 1u) return 1;
    const u32 fl = self.flags[a], ref = self.ref_id[a];
    const u64 s = self.ref_start[a], e = ref_end_of(self, a, s);
    for (u32 j = p0; j < p1; j++) {
        const u32 b = other.grp_idx[j];
        const u64 s2 = other.ref_start[b], e2 = ref_end_of(other, b, s2);
        // get_insert_size comes first in the loop and parses both ends (filter.rs:367-368); a mate that is never
        // reached -- an earlier one made a good pair -- is never parsed
        if (e == PP_REF_END_UNPARSEABLE || e2 == PP_REF_END_UNPARSEABLE) atomicOr(poisoned, 1u);
        const u32 ins = insert_of(s, e, s2, e2);
        if (ref == other.ref_id[b] && low <= ins && ins <= high && orientation_of(fl, s, e, other.flags[b], s2, e2) == correct) return 1;
    }
    return 0;
}

// ONE pass over the reads (round 5; rounds 1-4: k_ref_end, k_samples and k_pairs -- three passes, 557 MB for the 140 MB
// SURVEY 8d prices).  One lane per read: the sampling loop of get_insert_size_thresholds (filter.rs:155-167) for a read
// with one alignment in each file, both ends out of the runs in registers.  The verdict of alignment_pass_qc for every
// alignment that needs no thresholds for it -- its read has one alignment in this file, or none in the other: all of them
// in a job without multi-mapped reads -- is "pass": the host fills both verdict arrays with it before the launch.  The reads whose verdicts need the thresholds (several alignments here, at least one
// there) are only listed, for k_filter_listed.  `poisoned`: an end is needed that the reference could not have parsed.
// Two reads a lane since round 6 (FILTER_RPT), each stage's loads of both asked for together: the kernel is its chain of three
// dependent gathers (group offsets -> group index -> fields), and with one read a lane the 13,000 workgroups of a 3.3 M read job
// were six rounds of that chain.  The loads are unconditional (a read that is not a one-and-one pair looks at alignment 0 of
// either file: the host only launches the kernel when both files have alignments).
constexpr u32 FILTER_RPT = 2, FILTER_RPB = 256u * FILTER_RPT;  // reads per lane / per workgroup (and the workgroup's stretch of the list)
__global__ __launch_bounds__(256) void k_filter_reads(u32 n_reads, FileDev f1, FileDev f2, u8 *__restrict__ orient,
                                                      u32 *__restrict__ insert,
                                                      u32 *__restrict__ list, u32 *__restrict__ blk_cnt, u32 *__restrict__ any_listed,
                                                      u32 *__restrict__ poisoned) {
    __shared__ u32 s_cnt;
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    u32 r[FILTER_RPT], g1[FILTER_RPT], g2[FILTER_RPT], n1[FILTER_RPT], n2[FILTER_RPT], a[FILTER_RPT], b[FILTER_RPT];
    bool in[FILTER_RPT], one[FILTER_RPT], listed[FILTER_RPT];
#pragma unroll
    for (u32 u = 0; u < FILTER_RPT; u++) {
        r[u] = blockIdx.x * FILTER_RPB + u * 256u + threadIdx.x;
        in[u] = r[u] < n_reads;
        const u32 rr = in[u] ? r[u] : 0u;
        g1[u] = f1.grp_off[rr]; n1[u] = f1.grp_off[rr + 1];
        g2[u] = f2.grp_off[rr]; n2[u] = f2.grp_off[rr + 1];
    }
#pragma unroll
    for (u32 u = 0; u < FILTER_RPT; u++) {
        n1[u] -= g1[u];
        n2[u] -= g2[u];
        one[u] = in[u] && n1[u] == 1u && n2[u] == 1u;
        listed[u] = in[u] && ((n1[u] > 1u && n2[u] > 0u) || (n2[u] > 1u && n1[u] > 0u));
        // (else: one alignment here and none there, several here and none there, none at all -- everything passes)
        a[u] = f1.grp_idx[one[u] ? g1[u] : 0u];
        b[u] = f2.grp_idx[one[u] ? g2[u] : 0u];
    }
    u32 ra[FILTER_RPT], rb[FILTER_RPT], fa[FILTER_RPT], fb[FILTER_RPT];
    u64 s1[FILTER_RPT], s2[FILTER_RPT], e1[FILTER_RPT], e2[FILTER_RPT];
#pragma unroll
    for (u32 u = 0; u < FILTER_RPT; u++) {
        const u32 ia = one[u] ? a[u] : 0u, ib = one[u] ? b[u] : 0u;
        ra[u] = f1.ref_id[ia]; rb[u] = f2.ref_id[ib];
        s1[u] = f1.ref_start[ia]; s2[u] = f2.ref_start[ib];
        fa[u] = f1.flags[ia]; fb[u] = f2.flags[ib];
        e1[u] = f1.ref_end ? f1.ref_end[ia] : 0ull;  // (precomputed ends: asked for with the other fields)
        e2[u] = f2.ref_end ? f2.ref_end[ib] : 0ull;
    }
#pragma unroll
    for (u32 u = 0; u < FILTER_RPT; u++) {
        u8 o = 255;
        u32 ins = 0;
        if (one[u] && ra[u] == rb[u]) {
            if (!f1.ref_end) e1[u] = ref_end_of(f1, a[u], s1[u]);
            if (!f2.ref_end) e2[u] = ref_end_of(f2, b[u], s2[u]);
            if (e1[u] == PP_REF_END_UNPARSEABLE || e2[u] == PP_REF_END_UNPARSEABLE) atomicOr(poisoned, 1u);
            o = (u8)orientation_of(fa[u], s1[u], e1[u], fb[u], s2[u], e2[u]);
            ins = insert_of(s1[u], e1[u], s2[u], e2[u]);
        }
        // A verdict that needs no thresholds is "pass", and the verdict arrays were filled with it before the launch: nothing
        // to store here (a byte per alignment at its place in FILE order -- the reads come in whatever order the loader
        // grouped them -- would be millions of scattered read-modify-writes).
        if (in[u]) {
            orient[r[u]] = o;
            insert[r[u]] = ins;
        }
    }
    // The list, in the workgroup's own stretch of it: [FILTER_RPB * block, + blk_cnt[block]).  (One counter for all of it -- a
    // returning atomic per wave on ONE address -- is served every ~5 ns however few lanes ask: 0.25 ms for the 52,000 waves
    // of a 3.3 M read job, five times the rest of the kernel.)
#pragma unroll
    for (u32 u = 0; u < FILTER_RPT; u++) {
        const u64 m = __ballot(listed[u]);
        if (m) {
            const u32 lane = threadIdx.x & 63u, leader = (u32)__ffsll((long long)m) - 1u;
            u32 base = 0;
            if (lane == leader) base = atomicAdd(&s_cnt, (u32)__popcll(m));
            base = (u32)__shfl((int)base, (int)leader, 64);
            if (listed[u]) list[blockIdx.x * FILTER_RPB + base + (u32)__popcll(m & ((1ull << lane) - 1ull))] = r[u];
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        blk_cnt[blockIdx.x] = s_cnt;
        if (s_cnt) *any_listed = 1u;  // (every writer writes the same)
    }
}

// the listed reads: alignment_pass_qc with the thresholds, one lane per read, every alignment of it in either file
__global__ __launch_bounds__(FILTER_RPB) void k_filter_listed(const u32 *__restrict__ list, const u32 *__restrict__ blk_cnt, FileDev f1, FileDev f2,
                                                       u32 low, u32 high, u32 correct, u8 *__restrict__ pass1, u8 *__restrict__ pass2,
                                                       u32 *__restrict__ poisoned) {
    if (threadIdx.x >= blk_cnt[blockIdx.x]) return;  // (the grid and the workgroups are k_filter_reads')
    const u32 r = list[blockIdx.x * blockDim.x + threadIdx.x];
    const u32 g1 = f1.grp_off[r], g1e = f1.grp_off[r + 1], g2 = f2.grp_off[r], g2e = f2.grp_off[r + 1];
    for (u32 j = g1; j < g1e; j++) {
        const u32 a = f1.grp_idx[j];
        pass1[a] = pass_qc(f1, a, g1e - g1, f2, g2, g2e, low, high, correct, poisoned);
    }
    for (u32 j = g2; j < g2e; j++) {
        const u32 a = f2.grp_idx[j];
        pass2[a] = pass_qc(f2, a, g2e - g2, f1, g1, g1e, low, high, correct, poisoned);
    }
}

}  // namespace pp

using namespace pp;

static int up(pp_ctx *ctx, DevBuf &b, const void *src, size_t bytes, int mem, const void **dev) {
    if (mem == PP_MEM_DEVICE) {
        *dev = src;
        return PP_OK;
    }
    int rc = dev_ensure(ctx, b, bytes);
    if (rc) return rc;
    if (bytes) PP_HIPCHK(ctx, hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    *dev = b.p;
    return PP_OK;
}

static FileDev file_dev(const pp_ctx *ctx, int f) {
    const pp_filter_file &d = ctx->fdev.file[f];
    FileDev r;
    r.ref_id = d.ref_id; r.ref_start = d.ref_start; r.flags = d.flags; r.grp_off = d.grp_off;
    r.grp_idx = d.grp_idx; r.read = d.read; r.ref_end = (const u64 *)ctx->f_refend_ptr[f];
    r.cig_off = (const u64 *)d.cig_off; r.n_cig = d.n_cig; r.cigar = d.cigar; r.n_aln = d.n_aln;
    return r;
}

extern "C" int pp_filter_begin(pp_ctx *ctx, const pp_filter_input *in, int mem) {
    if (!ctx) return PP_ERR_ARG;
    if (int rdy = pp_ctx_wait(ctx)) return rdy;
    if (!in) return ctx->fail(PP_ERR_ARG, "pp_filter_begin: null input");
    PP_HIPCHK(ctx, hipSetDevice(ctx->device));
    timers_release(ctx);
    ctx->fdev = *in;
    ctx->filter_reads_done = false;
    ctx->filter_n_listed = -1;
    for (int f = 0; f < 2; f++) {
        const pp_filter_file &s = in->file[f];
        pp_filter_file &d = ctx->fdev.file[f];
        if (s.n_aln >= 0xFFFFFFFFull) return ctx->fail(PP_ERR_LIMIT, "more than 2^32-1 alignments in one SAM file");
        if (!s.grp_off) return ctx->fail(PP_ERR_ARG, "pp_filter_begin: null grp_off");
        const void *p;
        int rc;
        size_t n = s.n_aln;
#define UPF(i, field, T, count)                                                       \
    rc = up(ctx, ctx->f_in[f][i], s.field, (size_t)(count) * sizeof(T), mem, &p);     \
    if (rc) return rc;                                                                \
    d.field = (const T *)p;
        UPF(0, ref_id, uint32_t, n)
        UPF(1, ref_start, uint32_t, n)
        UPF(2, flags, uint32_t, n)
        if (!s.ref_end) {
            UPF(3, cig_off, uint64_t, n)
            UPF(4, n_cig, uint32_t, n)
            UPF(5, cigar, uint32_t, s.n_cig_total)
        }
        UPF(6, read, uint32_t, n)
        UPF(7, grp_idx, uint32_t, n)
        UPF(8, grp_off, uint32_t, (size_t)in->n_reads + 1)
#undef UPF
        ctx->f_refend_ptr[f] = nullptr;  // (the ends are worked out from the runs where they are needed: no pass of its own since round 5)
        if (s.ref_end) {  // precomputed (the device loader): adopt or upload
            rc = up(ctx, ctx->f_refend[f], s.ref_end, n * 8, mem, &p);
            if (rc) return rc;
            ctx->f_refend_ptr[f] = (const uint64_t *)p;
        }
    }
    int rcf = dev_ensure(ctx, ctx->f_poisoned, 8);  // [0] poisoned, [1] some read is listed for k_filter_listed
    if (rcf) return rcf;
    PP_HIPCHK(ctx, hipMemsetAsync(ctx->f_poisoned.p, 0, 8, ctx->stream));
    PP_HIPCHK(ctx, hipGetLastError());
    ctx->filter_open = true;
    return PP_OK;
}

// the reference's unwrap() on a run length that does not fit usize (alignment.rs:141)
static int check_poisoned(pp_ctx *ctx) {
    uint32_t w[2] = {0, 0};  // [1]: some read is listed (final once k_filter_reads has run)
    PP_HIPCHK(ctx, hipMemcpyAsync(w, ctx->f_poisoned.p, 8, hipMemcpyDeviceToHost, ctx->stream));
    PP_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    const uint32_t flag = w[0];
    if (ctx->filter_reads_done) ctx->filter_n_listed = (int64_t)w[1];
    if (flag) return ctx->fail(PP_ERR_PANIC, "a CIGAR run length that does not fit 64 bits belongs to an alignment whose end a pair comparison needs");
    return PP_OK;
}

// the pass over the reads: samples, the verdicts that need no thresholds, the list of the reads whose verdicts do
static int filter_reads(pp_ctx *ctx) {
    const uint32_t n = ctx->fdev.n_reads;
    int rc;
    if ((rc = dev_ensure(ctx, ctx->f_orient, n))) return rc;
    if ((rc = dev_ensure(ctx, ctx->f_insert, (size_t)n * 4))) return rc;
    const uint32_t n_blocks = (n + FILTER_RPB - 1u) / FILTER_RPB;
    if ((rc = dev_ensure(ctx, ctx->f_list, (size_t)n_blocks * FILTER_RPB * 4))) return rc;
    if ((rc = dev_ensure(ctx, ctx->f_blkcnt, (size_t)n_blocks * 4))) return rc;
    for (int f = 0; f < 2; f++) {
        const uint64_t na = ctx->fdev.file[f].n_aln;
        if ((rc = dev_ensure(ctx, ctx->f_pass[f], na))) return rc;
        // (an alignment that is in no read's group -- the input's contract has none -- passes, as an alignment without mates does)
        if (na) PP_HIPCHK(ctx, hipMemsetAsync(ctx->f_pass[f].p, 1, na, ctx->stream));
    }
    if (n && ctx->fdev.file[0].n_aln && ctx->fdev.file[1].n_aln) {
        timer_begin(ctx, "samples");
        hipLaunchKernelGGL(k_filter_reads, dim3(n_blocks), dim3(256), 0, ctx->stream, n, file_dev(ctx, 0), file_dev(ctx, 1),
                           (u8 *)ctx->f_orient.p, (u32 *)ctx->f_insert.p, (u32 *)ctx->f_list.p, (u32 *)ctx->f_blkcnt.p,
                           (u32 *)ctx->f_poisoned.p + 1, (u32 *)ctx->f_poisoned.p);
        timer_end(ctx);
        PP_HIPCHK(ctx, hipGetLastError());
    } else if (n) {
        // a file without alignments: no read is a pair, none is listed (the kernel's unconditional loads want an alignment 0 in
        // both files)
        PP_HIPCHK(ctx, hipMemsetAsync(ctx->f_orient.p, 255, n, ctx->stream));
        PP_HIPCHK(ctx, hipMemsetAsync(ctx->f_insert.p, 0, (size_t)n * 4, ctx->stream));
        PP_HIPCHK(ctx, hipMemsetAsync(ctx->f_blkcnt.p, 0, (size_t)n_blocks * 4, ctx->stream));
    }
    ctx->filter_reads_done = true;
    return PP_OK;
}

extern "C" int pp_filter_samples(pp_ctx *ctx, uint8_t *orient, uint32_t *insert) {
    if (!ctx) return PP_ERR_ARG;
    if (!ctx->filter_open) return ctx->fail(PP_ERR_ARG, "pp_filter_samples without pp_filter_begin");
    if (!orient || !insert) return ctx->fail(PP_ERR_ARG, "pp_filter_samples: null output");
    const uint32_t n = ctx->fdev.n_reads;
    if (!ctx->filter_reads_done)
        if (int rc = filter_reads(ctx)) return rc;
    if (n) {
        PP_HIPCHK(ctx, hipMemcpyAsync(orient, ctx->f_orient.p, n, hipMemcpyDeviceToHost, ctx->stream));
        PP_HIPCHK(ctx, hipMemcpyAsync(insert, ctx->f_insert.p, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    }
    return check_poisoned(ctx);
}

extern "C" int pp_filter_pairs(pp_ctx *ctx, uint32_t low, uint32_t high, uint8_t orientation,
                               uint8_t *pass1, uint8_t *pass2) {
    if (!ctx) return PP_ERR_ARG;
    if (!ctx->filter_open) return ctx->fail(PP_ERR_ARG, "pp_filter_pairs without pp_filter_begin");
    uint8_t *outs[2] = {pass1, pass2};
    for (int f = 0; f < 2; f++)
        if (ctx->fdev.file[f].n_aln && !outs[f]) return ctx->fail(PP_ERR_ARG, "pp_filter_pairs: null output");
    if (!ctx->filter_reads_done)  // (pairs without samples: the pass over the reads has not run yet)
        if (int rc = filter_reads(ctx)) return rc;
    if (ctx->fdev.n_reads && ctx->filter_n_listed != 0) {  // (0: pp_filter_samples read the count back -- a job without multi-mapped reads)
        timer_begin(ctx, "pairs");
        hipLaunchKernelGGL(k_filter_listed, dim3((ctx->fdev.n_reads + FILTER_RPB - 1u) / FILTER_RPB), dim3(FILTER_RPB), 0, ctx->stream, (const u32 *)ctx->f_list.p,
                           (const u32 *)ctx->f_blkcnt.p, file_dev(ctx, 0), file_dev(ctx, 1), low, high, (u32)orientation, (u8 *)ctx->f_pass[0].p,
                           (u8 *)ctx->f_pass[1].p, (u32 *)ctx->f_poisoned.p);
        timer_end(ctx);
        PP_HIPCHK(ctx, hipGetLastError());
    }
    for (int f = 0; f < 2; f++) {
        const uint64_t n = ctx->fdev.file[f].n_aln;
        if (n) PP_HIPCHK(ctx, hipMemcpyAsync(outs[f], ctx->f_pass[f].p, n, hipMemcpyDeviceToHost, ctx->stream));
    }
    if (int rcp = check_poisoned(ctx)) return rcp;
    if (ctx->profiling) timers_collect(ctx, &ctx->last_times);
    return PP_OK;
}

extern "C" int pp_filter_kernel_times(pp_ctx *ctx, pp_kernel_times *out) {
    if (!ctx || !out) return PP_ERR_ARG;
    *out = ctx->last_times;
    return PP_OK;
}

// (see pp_tokenize_warm_: the code object of this translation unit is loaded by its first launch)
__global__ void k_filter_warm(uint32_t *p) {
    if (p) p[threadIdx.x] = 0;
}
extern "C" void pp_filter_warm_(hipStream_t st) { hipLaunchKernelGGL(k_filter_warm, dim3(1), dim3(64), 0, st, (uint32_t *)nullptr); }
