// pp_filter.hip -- gfx950 kernels of the paired-read insert-size filter (seam A of
// include/polypolish_hip.h): ref_end from the CIGAR runs, orientation + insert size of the
// uniquely-aligned pairs, and the per-alignment pass/fail rule.  Pure integer work, one lane per
// alignment (or per read), structure-of-arrays loads; HBM-bound at ~17 B per alignment.
#include "pp_internal.h"

#include <cstring>

namespace pp {

typedef unsigned long long u64;
typedef uint32_t u32;
typedef uint8_t u8;

// Alignment::get_ref_end, alignment.rs:138-149: ref_start + sum of M, D, N, =, X run lengths
__global__ __launch_bounds__(256) void k_ref_end(u64 n, const u32 *__restrict__ ref_start,
                                                 const u64 *__restrict__ cig_off,
                                                 const u32 *__restrict__ n_cig,
                                                 const u32 *__restrict__ cigar,
                                                 u64 *__restrict__ ref_end) {
    u64 a = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= n) return;
    u64 end = ref_start[a];
    const u32 *cg = cigar + cig_off[a];
    for (u32 r = 0; r < n_cig[a]; r++) {
        u32 op = cg[r], o = op & 15u;
        if (o == PP_OP_M || o == PP_OP_D || o == PP_OP_N || o == PP_OP_EQ || o == PP_OP_X) end += op >> 4;
        else if (o == (u32)PP_OP_UNPARSEABLE) { end = PP_REF_END_UNPARSEABLE; break; }
    }
    ref_end[a] = end;
}

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
    const u64 *ref_end;
    u64 n_aln;
};

// sampling loop of get_insert_size_thresholds, filter.rs:155-167: one lane per read
// `poisoned`: set when an end is needed that the reference could not have parsed (PP_REF_END_UNPARSEABLE)
__global__ __launch_bounds__(256) void k_samples(u32 n_reads, FileDev f1, FileDev f2,
                                                 u8 *__restrict__ orient, u32 *__restrict__ insert,
                                                 u32 *__restrict__ poisoned) {
    u32 r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    u8 o = 255;
    u32 ins = 0;
    if (f1.grp_off[r + 1] - f1.grp_off[r] == 1u && f2.grp_off[r + 1] - f2.grp_off[r] == 1u) {
        const u32 a = f1.grp_idx[f1.grp_off[r]], b = f2.grp_idx[f2.grp_off[r]];
        if (f1.ref_id[a] == f2.ref_id[b]) {
            if (f1.ref_end[a] == PP_REF_END_UNPARSEABLE || f2.ref_end[b] == PP_REF_END_UNPARSEABLE) atomicOr(poisoned, 1u);
            o = (u8)orientation_of(f1.flags[a], f1.ref_start[a], f1.ref_end[a], f2.flags[b], f2.ref_start[b], f2.ref_end[b]);
            ins = insert_of(f1.ref_start[a], f1.ref_end[a], f2.ref_start[b], f2.ref_end[b]);
        }
    }
    orient[r] = o;
    insert[r] = ins;
}

// alignment_pass_qc, filter.rs:352-377: one lane per alignment of `self`, mates in `other`
__global__ __launch_bounds__(256) void k_pairs(FileDev self, FileDev other, u32 low, u32 high,
                                               u32 correct, u8 *__restrict__ pass, u32 *__restrict__ poisoned) {
    u64 a = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= self.n_aln) return;
    const u32 r = self.read[a];
    const u32 n_this = self.grp_off[r + 1] - self.grp_off[r];
    const u32 p0 = other.grp_off[r], p1 = other.grp_off[r + 1];
    u8 ok = 0;
    if (p1 == p0 || n_this == 1u) {
        ok = 1;
    } else {
        const u32 fl = self.flags[a], ref = self.ref_id[a];
        const u64 s = self.ref_start[a], e = self.ref_end[a];
        for (u32 j = p0; j < p1 && !ok; j++) {
            const u32 b = other.grp_idx[j];
            const u64 s2 = other.ref_start[b], e2 = other.ref_end[b];
            // get_insert_size comes first in the loop and parses both ends (filter.rs:367-368); a mate that is never
            // reached -- an earlier one made a good pair -- is never parsed
            if (e == PP_REF_END_UNPARSEABLE || e2 == PP_REF_END_UNPARSEABLE) atomicOr(poisoned, 1u);
            const u32 ins = insert_of(s, e, s2, e2);
            if (ref == other.ref_id[b] && low <= ins && ins <= high &&
                orientation_of(fl, s, e, other.flags[b], s2, e2) == correct)
                ok = 1;
        }
    }
    pass[a] = ok;
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
    r.grp_idx = d.grp_idx; r.read = d.read; r.ref_end = (const u64 *)ctx->f_refend_ptr[f]; r.n_aln = d.n_aln;
    return r;
}

extern "C" int pp_filter_begin(pp_ctx *ctx, const pp_filter_input *in, int mem) {
    if (!ctx) return PP_ERR_ARG;
    if (int rdy = pp_ctx_wait(ctx)) return rdy;
    if (!in) return ctx->fail(PP_ERR_ARG, "pp_filter_begin: null input");
    PP_HIPCHK(ctx, hipSetDevice(ctx->device));
    timers_release(ctx);
    ctx->fdev = *in;
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
        if (s.ref_end) {  // precomputed (the device loader): adopt or upload, no k_ref_end
            rc = up(ctx, ctx->f_refend[f], s.ref_end, n * 8, mem, &p);
            if (rc) return rc;
            ctx->f_refend_ptr[f] = (const uint64_t *)p;
            continue;
        }
        rc = dev_ensure(ctx, ctx->f_refend[f], n * 8);
        if (rc) return rc;
        ctx->f_refend_ptr[f] = (const uint64_t *)ctx->f_refend[f].p;
        if (n) {
            timer_begin(ctx, "ref_end");
            hipLaunchKernelGGL(k_ref_end, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (u64)n,
                               d.ref_start, (const u64 *)d.cig_off, d.n_cig, d.cigar, (u64 *)ctx->f_refend[f].p);
            timer_end(ctx);
        }
    }
    int rcf = dev_ensure(ctx, ctx->f_poisoned, 4);
    if (rcf) return rcf;
    PP_HIPCHK(ctx, hipMemsetAsync(ctx->f_poisoned.p, 0, 4, ctx->stream));
    PP_HIPCHK(ctx, hipGetLastError());
    ctx->filter_open = true;
    return PP_OK;
}

// the reference's unwrap() on a run length that does not fit usize (alignment.rs:141)
static int check_poisoned(pp_ctx *ctx) {
    uint32_t flag = 0;
    PP_HIPCHK(ctx, hipMemcpyAsync(&flag, ctx->f_poisoned.p, 4, hipMemcpyDeviceToHost, ctx->stream));
    PP_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (flag) return ctx->fail(PP_ERR_PANIC, "a CIGAR run length that does not fit 64 bits belongs to an alignment whose end a pair comparison needs");
    return PP_OK;
}

extern "C" int pp_filter_samples(pp_ctx *ctx, uint8_t *orient, uint32_t *insert) {
    if (!ctx) return PP_ERR_ARG;
    if (!ctx->filter_open) return ctx->fail(PP_ERR_ARG, "pp_filter_samples without pp_filter_begin");
    if (!orient || !insert) return ctx->fail(PP_ERR_ARG, "pp_filter_samples: null output");
    const uint32_t n = ctx->fdev.n_reads;
    int rc;
    if ((rc = dev_ensure(ctx, ctx->f_orient, n))) return rc;
    if ((rc = dev_ensure(ctx, ctx->f_insert, (size_t)n * 4))) return rc;
    if (n) {
        timer_begin(ctx, "samples");
        hipLaunchKernelGGL(k_samples, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, n, file_dev(ctx, 0),
                           file_dev(ctx, 1), (u8 *)ctx->f_orient.p, (u32 *)ctx->f_insert.p, (u32 *)ctx->f_poisoned.p);
        timer_end(ctx);
        PP_HIPCHK(ctx, hipGetLastError());
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
    DevBuf *pb[2] = {&ctx->f_pass[0], &ctx->f_pass[1]};
    for (int f = 0; f < 2; f++) {
        const uint64_t n = ctx->fdev.file[f].n_aln;
        if (n && !outs[f]) return ctx->fail(PP_ERR_ARG, "pp_filter_pairs: null output");
        int rc = dev_ensure(ctx, *pb[f], n);
        if (rc) return rc;
        if (!n) continue;
        timer_begin(ctx, "pairs");
        hipLaunchKernelGGL(k_pairs, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, file_dev(ctx, f),
                           file_dev(ctx, 1 - f), low, high, (u32)orientation, (u8 *)pb[f]->p, (u32 *)ctx->f_poisoned.p);
        timer_end(ctx);
        PP_HIPCHK(ctx, hipGetLastError());
        PP_HIPCHK(ctx, hipMemcpyAsync(outs[f], pb[f]->p, n, hipMemcpyDeviceToHost, ctx->stream));
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
