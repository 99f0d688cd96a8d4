// pp_kernels.hip -- gfx950 kernels of the polish hot path (seam B of include/polypolish_hip.h).
//
// Two pipelines, the same results.  A batch that brings its window-order mirror AND the mirror's run table -- what the
// library's own ingests hand over, and what pp_shard_split leaves a rank of a sharded job -- takes the DIRECT path (round 5,
// pp_k_direct.h): [k_meta_init: only when the job before did not leave the metadata ready] -> k_prepd -> k_prepg -> k_winplan ->
// k_tile_direct -> [k_xmat, k_exact2 x 3, k_exact: only when something was flagged for them] -> k_emit (no scan in front: output
// offsets from coarse sums of the windows' lengths; its last workgroup hands the job's metadata to the host and sets them up for
// the next job, pp_k_emit.h): five launches in the steady state, no work items for the bulk of the records.  Everything else takes
// the BUCKETING path of rounds 1-4:
// Pipeline (one pp_polish_finish, 13 stream operations):
//   k_meta_init  the job's metadata block (status, counters, heavy-window list)
//   k_prep     persistent blocks stream the alignment records: the bulk (one short M run inside its contig) on the
//              spot, the others (indels, long reads: CIGAR walk validation, reference span and the right-end homopolymer
//              trim, alignment.rs:175-201,364-378) after the loop, one per lane -> (global start, kept entries / class);
//              a read with ONE 1-base indel is only classified (its pieces are cut by k_fill); the same pass counts each
//              block's (piece, window) items per window / coarse bucket in LDS.  A sharded job's context runs over a
//              compact assembly of what it owns: k_prep places the records there (g_base) and drops the others'
//   k_scan_cols / k_scan   column scan over the blocks + scan over the windows \ atomics-free multisplit of the
//   k_fill     scatter of 16-byte work items into their window's bucket         > pieces into 2048-position windows
//   (k_regroup / k_heavy / k_count: the two-level path from 33.5 Mbp on)       /  (+ the list of heavy windows)
//   k_tile     one workgroup per window: counters for 2048 positions and the window's assembly bytes live in LDS;
//              groups of 5-8 lanes own one read or flank of a read (32 bytes per lane), compare it with the assembly and
//              tally only the differing bases (two LDS atomics into a coverage difference array per piece,
//              pileup.rs:56-65,189-200); the entry AT a read's single indel is one tally per lane.  Then the positions where
//              anything was tallied are voted, one lane each (pileup.rs:67-134), the others keep the assembly's base; a
//              1-byte emit code per position.  Heavy windows are tallied by eight helper blocks each; a sharded job only
//              launches the windows it works on
//   k_exact2 (three instances)   the positions whose depth depends on the ORDER of f64 additions (non-power-of-two 1/k
//              shares) are replayed exactly: the window's items sorted by file order, one sequential f64 pass
//   k_exact    the positions whose string-keyed counts (insertions, N...) could reach a threshold: one workgroup per
//              position scans the window's items, its first wave sorts the covering alignments, tallies and groups keys
//   k_emit     drop '-' (polish.rs:188), prefix sums of the emit lengths (in-kernel; PP_EMIT_FUSE=0: k_scan), polished bytes, contig offsets
//
// Integer counting, HBM/LDS bound: no MFMA anywhere by design.
#include "pp_internal.h"

#include <algorithm>
#include <cstring>

// The kernels, in pipeline order (one translation unit: everything below is static or inlined)
#include "pp_k_common.h"
#include "pp_k_prep.h"
#include "pp_k_bucket.h"
#include "pp_k_direct.h"
#include "pp_k_tile.h"
#include "pp_k_exact.h"
#include "pp_k_emit.h"
#include <hip/hip_ext.h>

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

// grow a buffer to `bytes` keeping its first `used` bytes (the accumulated alignment batches)
int dev_grow_keep(pp_ctx *ctx, DevBuf &b, size_t bytes, size_t used) {
    if (bytes == 0) bytes = 16;
    if (b.cap >= bytes) return PP_OK;
    void *np = nullptr;
    const size_t want = std::max(bytes + bytes / 2, b.cap + b.cap / 2) + 256;
    PP_HIPCHK(ctx, hipMalloc(&np, want));
    if (b.p && used) PP_HIPCHK(ctx, hipMemcpyAsync(np, b.p, used, hipMemcpyDeviceToDevice, ctx->stream));
    if (b.p) {
        PP_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        PP_HIPCHK(ctx, hipFree(b.p));
    }
    b.p = np;
    b.cap = want;
    return PP_OK;
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
    } else if (hipEventCreateWithFlags(&t.start, hipEventReleaseToDevice) != hipSuccess ||
               hipEventCreateWithFlags(&t.stop, hipEventReleaseToDevice) != hipSuccess) {
        return;  // (device-scope release: the record does not flush the caches to system scope -- HIP's flag for timing)
    }
    (void)hipEventRecord(t.start, ctx->stream);
    ctx->timers.push_back(t);
    ctx->timer_open = true;
}
// The same for ONE kernel launch: the two events go along with the launch itself (hipExtLaunchKernelGGL: they take the
// dispatch's own start and end) instead of being recorded in front of it and behind it -- a recorded event is a barrier
// packet of its own on the queue, ~3-6 us of idle time each (rocprofv3 trace of the bench's timed steps, round 5: 6.6 us in
// front of k_tile_direct and 6.6 us behind it, where kernels without events between them follow each other within 1 us).
bool timer_for_launch(pp_ctx *ctx, const char *name, hipEvent_t *start, hipEvent_t *stop) {
    if (!ctx->profiling || (ctx->profiling == 2 && strcmp(name, "tile") != 0)) return false;
    static const bool recorded = getenv("PP_TIMER_RECORD") && atoi(getenv("PP_TIMER_RECORD")) != 0;  // tuning: events recorded around the launch
    if (recorded) return false;
    KernelTimer t;
    t.name = name;
    if (ctx->event_pool.size() >= 2) {
        t.stop = ctx->event_pool.back(); ctx->event_pool.pop_back();
        t.start = ctx->event_pool.back(); ctx->event_pool.pop_back();
    } else if (hipEventCreateWithFlags(&t.start, hipEventReleaseToDevice) != hipSuccess ||
               hipEventCreateWithFlags(&t.stop, hipEventReleaseToDevice) != hipSuccess) {
        return false;
    }
    ctx->timers.push_back(t);
    *start = t.start;
    *stop = t.stop;
    return true;
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
    // the contig table goes to the device unless the device already holds this very table (the job before was on the
    // same assembly: rounds of polishing, the bench's steps)
    const bool same_table = ctx->b_contig_off.p && ctx->contig_off.size() == (size_t)n_contigs + 1 &&
                            memcmp(ctx->contig_off.data(), contig_off, ((size_t)n_contigs + 1) * sizeof(uint64_t)) == 0;
    ctx->n_contigs = n_contigs;
    ctx->contig_off.assign(contig_off, contig_off + n_contigs + 1);
    ctx->G = G;
    ctx->params = *params;
    const void *d;
    int rc = PP_OK;
    if (!same_table) rc = upload(ctx, ctx->b_contig_off, contig_off, (n_contigs + 1) * sizeof(uint64_t), &d);
    if (rc) { ctx->contig_off.clear(); return rc; }
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
    ctx->batch_borrowed = false;
    ctx->acc_n = ctx->acc_seq = ctx->acc_cig = 0;
    ctx->emit.clear();
    ctx->wo_runs.clear();
    ctx->wo_untrusted = false;
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

// seq_off / cig_off of an appended batch are relative to ITS seq / cigar arrays: rebase them onto the accumulated ones
__global__ void k_rebase(u64 *seq_off, u64 *cig_off, u64 n, u64 seq_base, u64 cig_base) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        seq_off[i] += seq_base;
        cig_off[i] += cig_base;
    }
}

__global__ void k_rebase_wo(pp_wo_rec *wo, u64 n, u64 seq_base, u32 idx_base) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        wo[i].seq_off += seq_base;
        wo[i].file_idx += idx_base;
    }
}

// Optional, between pp_polish_begin and the first pp_polish_add: room for what all the batches of the job will hold
// (a guess is fine: the arrays still grow when it was too small, at the price of a reallocation and a copy).
extern "C" int pp_polish_reserve(pp_ctx *ctx, uint64_t n_aln, uint64_t seq_bytes, uint64_t n_cig_total) {
    if (!ctx) return PP_ERR_ARG;
    if (int rdy = pp_ctx_wait(ctx)) return rdy;
    if (!ctx->job_open) return ctx->fail(PP_ERR_ARG, "pp_polish_reserve without pp_polish_begin");
    PP_HIPCHK(ctx, hipSetDevice(ctx->device));
    // (room for up to 31 unused bytes at every joint of the seq array, and for its 4-bit mirror)
    const size_t esz[11] = {4, 4, 4, 8, 4, 8, 4, 1, 4, 1, sizeof(pp_wo_rec)};
    const uint64_t cnt[11] = {n_aln, n_aln, n_aln, n_aln, n_aln, n_aln, n_aln, seq_bytes + 4096, n_cig_total, (seq_bytes + 4096) / 2 + 96, n_aln};
    const uint64_t used[11] = {ctx->acc_n, ctx->acc_n, ctx->acc_n, ctx->acc_n, ctx->acc_n, ctx->acc_n, ctx->acc_n, ctx->acc_seq, ctx->acc_cig, (ctx->acc_seq + 1) / 2,
                               ctx->acc_wo ? ctx->acc_n : 0};
    const bool owned = ctx->have_batch && !ctx->batch_borrowed;
    for (int i = 0; i < 11; i++)
        if (int rc = dev_grow_keep(ctx, ctx->b_in[i], (size_t)cnt[i] * esz[i], owned ? (size_t)used[i] * esz[i] : 0)) return rc;
    return PP_OK;
}

// The 4-bit mirror (pp_aln_batch.seq4) of a stretch of the accumulated seq array, for a batch that came without one:
// one thread turns 32 bytes into 16 (codes PP_SEQ4_*; A C T G = their counter rows).  [lo, hi): lo a multiple of 32.
__device__ __forceinline__ u32 seq4_code_of(u32 c) {
    const u32 t = (c >> 1) & 3u;
    const u32 expect = (0x47544341u >> (t * 8u)) & 0xFFu;
    return c == expect ? t : (c == (u32)'N' ? (u32)PP_SEQ4_N : (c == (u32)'-' ? (u32)PP_SEQ4_DASH : (u32)PP_SEQ4_OTHER));
}
__global__ __launch_bounds__(256) void k_pack4(const u8 *__restrict__ seq, u8 *__restrict__ seq4, u64 lo, u64 hi) {
    const u64 i0 = lo + ((u64)blockIdx.x * blockDim.x + threadIdx.x) * 32u;
    if (i0 >= hi) return;
    u32 w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (i0 + 32u <= hi) {
        uint4 a, b;
        __builtin_memcpy(&a, seq + i0, 16);
        __builtin_memcpy(&b, seq + i0 + 16, 16);
        w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w; w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
    } else {
        for (u32 j = 0; i0 + j < hi; j++) w[j >> 2] |= (u32)seq[i0 + j] << (8u * (j & 3u));
    }
    u32 o[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        u32 v = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) v |= seq4_code_of((w[2 * q + (j >> 2)] >> (8 * (j & 3))) & 0xFFu) << (4 * j);
        o[q] = v;
    }
    const uint4 out = make_uint4(o[0], o[1], o[2], o[3]);
    __builtin_memcpy(seq4 + (i0 >> 1), &out, 16);
}

// The run table of a batch's window-order mirror (pp_aln_batch.wo_run_end, HOST memory) joins the job's: ends rebased by
// the records in front (n0), empty runs dropped.  The job's table stays known only while every batch so far brought a
// mirror AND its runs; anything odd (descending ends, a last end that is not n_aln, too many runs) makes it unknown -- the
// job then takes the bucketing path, which needs no order.
static void runs_join(pp_ctx *ctx, const pp_aln_batch *b, uint64_t n0) {
    const bool known_so_far = n0 == 0 || !ctx->wo_runs.empty();
    if (n0 == 0) ctx->wo_runs.clear();
    if (b->n_aln == 0) return;
    bool ok = known_so_far && b->wo && b->wo_n_runs && b->wo_run_end && b->wo_run_end[b->wo_n_runs - 1] == b->n_aln;
    uint64_t prev = 0;
    for (uint32_t r = 0; ok && r < b->wo_n_runs; r++) {
        const uint64_t e = b->wo_run_end[r];
        if (e < prev || e > b->n_aln) ok = false;
        else if (e > prev) ctx->wo_runs.push_back(n0 + e);
        prev = e;
    }
    if (!ok || ctx->wo_runs.size() > PP_WO_MAX_RUNS) ctx->wo_runs.clear();
}

// Append one batch (host or device memory) to the library-owned accumulated arrays.  Every batch's SEQ bytes start on a
// multiple of PP_SEQ_ALIGN of the accumulated seq array (up to 31 unused bytes at a joint), so that the 4-bit mirror of the
// batch lands on whole bytes of the accumulated mirror: it is copied when the batch brings one (the library's ingests and
// pp_shard_split do) and packed from the bytes on the device when it does not -- a gathered job is polished from the
// mirror like a job of one in-place batch.
static int append_batch(pp_ctx *ctx, const pp_aln_batch *b, int mem) {
    static const bool no_seq4 = getenv("PP_SEQ4") && atoi(getenv("PP_SEQ4")) == 0;  // tuning / tests
    const uint64_t n0 = ctx->acc_n, c0 = ctx->acc_cig;
    const uint64_t s0 = (ctx->acc_seq + (uint64_t)PP_SEQ_ALIGN - 1) & ~((uint64_t)PP_SEQ_ALIGN - 1);
    const uint64_t n = b->n_aln;
    if (n0 + n >= 0xFFFFFFFFull) return ctx->fail(PP_ERR_LIMIT, "more than 2^32-1 alignments in one polish job");
    if (s0 + b->seq_bytes >= (1ull << 40)) return ctx->fail(PP_ERR_LIMIT, "more than 2^40 SEQ bytes in one polish job");
    // PP_MEM_PEER: the source is on another GPU of this process -- the runtime picks the route (xGMI peer copy)
    const hipMemcpyKind kind = mem == PP_MEM_DEVICE ? hipMemcpyDeviceToDevice : (mem == PP_MEM_PEER ? hipMemcpyDefault : hipMemcpyHostToDevice);
    const void *src[9] = {b->contig, b->ref_start, b->k, b->seq_off, b->seq_len, b->cig_off, b->n_cig, b->seq, b->cigar};
    const size_t esz[9] = {4, 4, 4, 8, 4, 8, 4, 1, 4};
    const uint64_t old_cnt[9] = {n0, n0, n0, n0, n0, n0, n0, s0, c0};
    const uint64_t kept[9] = {n0, n0, n0, n0, n0, n0, n0, ctx->acc_seq, c0};
    const uint64_t add_cnt[9] = {n, n, n, n, n, n, n, b->seq_bytes, b->n_cig_total};
    for (int i = 0; i < 9; i++) {
        int rc = dev_grow_keep(ctx, ctx->b_in[i], (size_t)(old_cnt[i] + add_cnt[i]) * esz[i] + (i == 7 ? 64 : 0), (size_t)kept[i] * esz[i]);
        if (rc) return rc;
        if (add_cnt[i])
            PP_HIPCHK(ctx, hipMemcpyAsync((char *)ctx->b_in[i].p + old_cnt[i] * esz[i], src[i], (size_t)add_cnt[i] * esz[i], kind,
                                          ctx->stream));
    }
    if (!no_seq4) {
        if (int rc = dev_grow_keep(ctx, ctx->b_in[9], (size_t)((s0 + b->seq_bytes + 1) / 2 + 96), (size_t)((ctx->acc_seq + 1) / 2))) return rc;
        if (b->seq_bytes && b->seq4)
            PP_HIPCHK(ctx, hipMemcpyAsync((char *)ctx->b_in[9].p + s0 / 2, b->seq4, (size_t)((b->seq_bytes + 1) / 2), kind, ctx->stream));
        else if (b->seq_bytes)
            hipLaunchKernelGGL(k_pack4, dim3((unsigned)(((b->seq_bytes + 31) / 32 + 255) / 256)), dim3(256), 0, ctx->stream,
                               (const u8 *)ctx->b_in[7].p, (u8 *)ctx->b_in[9].p, (u64)s0, (u64)(s0 + b->seq_bytes));
    }
    if (n && (s0 || c0))
        hipLaunchKernelGGL(k_rebase, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (u64 *)ctx->b_in[3].p + n0,
                           (u64 *)ctx->b_in[5].p + n0, (u64)n, (u64)s0, (u64)c0);
    // the window-order mirror of the records (pp_aln_batch.wo) goes along while every batch of the job brings one: a
    // batch's entries follow those of the batches before it (the windows then come once per batch, like the SEQ bytes
    // of the files), its seq offsets and file indices rebased like the arrays'
    static const bool no_wo = getenv("PP_WO") && atoi(getenv("PP_WO")) == 0;
    ctx->acc_wo = !no_wo && (n0 == 0 || ctx->acc_wo) && (b->wo != nullptr || n == 0);
    if (ctx->acc_wo && n) {
        if (int rc = dev_grow_keep(ctx, ctx->b_in[10], (size_t)(n0 + n) * sizeof(pp_wo_rec), (size_t)n0 * sizeof(pp_wo_rec))) return rc;
        PP_HIPCHK(ctx, hipMemcpyAsync((pp_wo_rec *)ctx->b_in[10].p + n0, b->wo, (size_t)n * sizeof(pp_wo_rec), kind, ctx->stream));
        if (s0 || n0)
            hipLaunchKernelGGL(k_rebase_wo, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (pp_wo_rec *)ctx->b_in[10].p + n0,
                               (u64)n, (u64)s0, (u32)n0);
    }
    if (mem != PP_MEM_DEVICE) PP_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));  // host / peer buffers are only borrowed for the call
    if (ctx->acc_wo) runs_join(ctx, b, n0); else ctx->wo_runs.clear();
    ctx->acc_n = n0 + n; ctx->acc_seq = s0 + b->seq_bytes; ctx->acc_cig = c0 + b->n_cig_total;
    pp_aln_batch &d = ctx->dbatch;
    d.n_aln = ctx->acc_n; d.seq_bytes = ctx->acc_seq; d.n_cig_total = ctx->acc_cig;
    d.contig = (const uint32_t *)ctx->b_in[0].p; d.ref_start = (const uint32_t *)ctx->b_in[1].p; d.k = (const uint32_t *)ctx->b_in[2].p;
    d.seq_off = (const uint64_t *)ctx->b_in[3].p; d.seq_len = (const uint32_t *)ctx->b_in[4].p;
    d.cig_off = (const uint64_t *)ctx->b_in[5].p; d.n_cig = (const uint32_t *)ctx->b_in[6].p;
    d.seq = (const uint8_t *)ctx->b_in[7].p; d.cigar = (const uint32_t *)ctx->b_in[8].p;
    d.seq4 = no_seq4 ? nullptr : (const uint8_t *)ctx->b_in[9].p;
    d.wo = ctx->acc_wo && ctx->acc_n ? (const pp_wo_rec *)ctx->b_in[10].p : nullptr;
    return PP_OK;
}

// Batches may be added one after the other (the reference streams its SAM files, src/alignment.rs:238-265): record i
// of a later batch follows every record of the earlier ones in file order.  A single PP_MEM_DEVICE batch is borrowed
// as it is; as soon as there is a second batch everything is gathered into library-owned arrays.
extern "C" int pp_polish_add(pp_ctx *ctx, const pp_aln_batch *b, int mem) {
    if (!ctx) return PP_ERR_ARG;
    if (int rdy = pp_ctx_wait(ctx)) return rdy;
    if (!ctx->job_open) return ctx->fail(PP_ERR_ARG, "pp_polish_add without pp_polish_begin");
    if (!b) return ctx->fail(PP_ERR_ARG, "pp_polish_add: null batch");
    if (mem != PP_MEM_HOST && mem != PP_MEM_DEVICE && mem != PP_MEM_PEER) return ctx->fail(PP_ERR_ARG, "pp_polish_add: unknown memory kind");
    if (b->n_aln >= 0xFFFFFFFFull)
        return ctx->fail(PP_ERR_LIMIT, "more than 2^32-1 alignments in one batch");
    if (b->n_aln && (!b->contig || !b->ref_start || !b->k || !b->seq_off || !b->seq_len ||
                     !b->cig_off || !b->n_cig || !b->seq || !b->cigar))
        return ctx->fail(PP_ERR_ARG, "pp_polish_add: null array in a non-empty batch");
    PP_HIPCHK(ctx, hipSetDevice(ctx->device));
    // a mirror that is not one of the library's own (or part of one) is compared with the arrays before it is used (run_pipeline)
    if (b->wo && b->n_aln && !pp_mirror_trusted_(b->wo, (size_t)b->n_aln * sizeof(pp_wo_rec))) ctx->wo_untrusted = true;
    if (!ctx->have_batch && mem == PP_MEM_DEVICE) {
        ctx->dbatch = *b;
        runs_join(ctx, b, 0);  // (copied: the caller's table is only borrowed for the call)
        ctx->dbatch.wo_n_runs = 0;
        ctx->dbatch.wo_run_end = nullptr;
        ctx->batch_borrowed = true;
        ctx->have_batch = true;
        return PP_OK;
    }
    if (ctx->have_batch && ctx->batch_borrowed) {  // a second batch: the borrowed one moves into the accumulated arrays
        pp_aln_batch first = ctx->dbatch;
        const std::vector<uint64_t> first_runs = ctx->wo_runs;  // (its run table as it was taken when the batch was added)
        first.wo_n_runs = (uint32_t)first_runs.size();
        first.wo_run_end = first_runs.empty() ? nullptr : first_runs.data();
        ctx->batch_borrowed = false;
        ctx->acc_n = ctx->acc_seq = ctx->acc_cig = 0;
        if (int rc = append_batch(ctx, &first, PP_MEM_DEVICE)) return rc;
    }
    if (!ctx->have_batch) ctx->acc_n = ctx->acc_seq = ctx->acc_cig = 0;
    if (int rc = append_batch(ctx, b, mem)) return rc;
    ctx->have_batch = true;
    return PP_OK;
}

static int map_device_error(pp_ctx *ctx, uint64_t key) {
    uint32_t code = (uint32_t)(key & 0xFF);
    unsigned long long idx = (unsigned long long)(key >> 8);
    ctx->last_dev_error = key;
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
    case DE_BAD_MIRROR: return ctx->fail(PP_ERR_ARG, "entry %llu of the window-order mirror (pp_aln_batch.wo) names a record the batch does not have, "
                                                     "or (PP_CHECK_WO=1) one that another entry names too, or does not carry that record's fields", idx);
    case DE_SEQ_RANGE: return ctx->fail(PP_ERR_ARG, "alignment record %llu: its SEQ bytes (seq_off + seq_len) lie outside the batch's seq array", idx);
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
    // ---- geometry of this run ----
    // A context of a sharded job runs over a COMPACT assembly of what it owns (pp_polish_set_emit: whole contigs, or one
    // stretch of a tiled contig plus a halo of HALO positions either side, where the reads that reach in from the
    // neighbours lie): k_prep puts a record at its contig's place in it (g_base) and drops the records of the other
    // contigs, everything behind k_prep only ever sees global positions, and the per-contig results are spread back over
    // the job's numbering at the end.  One eighth of a 50 Mbp metagenome is then a 6 Mbp job, one eighth of a 250 Mbp
    // contig a 31 Mbp job -- single-level bucketing, scans and grids over an eighth of the windows -- and not the whole
    // assembly with seven eighths missing.  A record that reaches an owned stretch but does not lie inside its slice
    // (a read longer than the halo) makes k_prep raise DE_HALO: the job is then rerun uncompacted.  Not with --debug
    // records (they are indexed by the job's positions), and not when little would be saved.
    constexpr uint64_t HALO = 16384;
    const uint32_t nc_full = ctx->n_contigs;
    std::vector<uint64_t> run_off(ctx->contig_off);  // the run's contig table (host copy)
    std::vector<uint32_t> run_emit(ctx->emit);       // (lo, hi) per contig of the run
    std::vector<uint64_t> g_base;                    // per contig of the JOB: where its position 0 falls in the run's coordinates, ~0 = not in it
    std::vector<uint64_t> src_start;                 // per contig of the RUN: where its bytes start in the job's assembly
    std::vector<uint32_t> slice;                     // per contig of the JOB: [lo, hi) of it that the run holds
    ctx->run_full_of.clear();
    // ---- the direct path (pp_k_direct.h): a mirror whose runs are known ----
    // A sharded job takes it over the job's own coordinates (the mirror's window order is the job's: a compact run has
    // other window boundaries), with the grids of k_tile / k_emit over the windows it works on as in any uncompacted
    // sharded run; what is proportional to ALL windows of the job is a few words per window in k_meta_init, k_prepd's
    // table, k_winplan and k_scan.
    static const bool env_no_direct = getenv("PP_DIRECT") && atoi(getenv("PP_DIRECT")) == 0;  // tuning / tests
    static const bool env_no_wo = getenv("PP_WO") && atoi(getenv("PP_WO")) == 0;
    const uint32_t n_runs = (uint32_t)ctx->wo_runs.size();
    const bool direct = !env_no_direct && !env_no_wo && !ctx->no_direct && !ctx->no_wo && n > 0 && B.wo && n_runs > 0 &&
                        n_runs <= PP_WO_MAX_RUNS && ctx->wo_runs.back() == n;
    ctx->last_direct = direct;
    if (!ctx->emit.empty() && !ctx->debug && !ctx->no_compact && !direct) {
        uint64_t g_sub = 0;
        std::vector<uint32_t> owned;
        slice.assign(2 * (size_t)nc_full, 0);
        for (uint32_t c = 0; c < nc_full; c++) {
            const uint64_t lo = ctx->emit[2 * c], hi = ctx->emit[2 * c + 1], len = ctx->contig_off[c + 1] - ctx->contig_off[c];
            if (hi <= lo) continue;
            const uint64_t slo = lo > HALO ? lo - HALO : 0, shi = std::min(len, hi + HALO);
            slice[2 * c] = (uint32_t)slo;
            slice[2 * c + 1] = (uint32_t)shi;
            owned.push_back(c);
            g_sub += shi - slo;
        }
        if (!owned.empty() && g_sub * 4 <= ctx->G * 3) {
            g_base.assign(nc_full, ~0ull);
            run_off.assign(1, 0);
            run_emit.clear();
            for (uint32_t c : owned) {
                const uint64_t slo = slice[2 * c], shi = slice[2 * c + 1];
                g_base[c] = run_off.back() - slo;  // (wraps below zero for a slice that does not start at the contig's start: only ever added to a start >= slo)
                src_start.push_back(ctx->contig_off[c] + slo);
                run_off.push_back(run_off.back() + (shi - slo));
                run_emit.push_back((uint32_t)(ctx->emit[2 * c] - slo));
                run_emit.push_back((uint32_t)(ctx->emit[2 * c + 1] - slo));
            }
            ctx->run_full_of = owned;
        }
    }
    const bool compact = !ctx->run_full_of.empty();
    const uint32_t nc = (uint32_t)run_off.size() - 1;
    const uint64_t G = run_off.back();
    ctx->run_nc = nc;
    const uint32_t nwin = (uint32_t)((G + TILE - 1) / TILE);
#ifndef PP_NB_MAX
#define PP_NB_MAX 512  // one resident wave of k_prep / k_fill workgroups (two per CU), and what k_scan_cols takes; 1024 measured slower (bucket 0.10 -> 0.13 ms), 256 the same
#endif
    const uint32_t NB = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(PP_NB_MAX, (n + 4095) / 4096));
    const uint64_t chunk = (n + NB - 1) / NB;
    const uint32_t nranges = (nwin + COUNT_RANGE - 1) / COUNT_RANGE;
    int rc;
    // (room for the mirror indices of the records that are not bulk -- k_prepd notes them for k_prepg: an eighth of the
    // records; a job with more of them has its workgroups handle what does not fit themselves)
    const uint64_t cap_later = std::max<uint64_t>(65536, n / 8);
    if (direct) {
        // room for a window's extras: a quarter of the average window's records (7 % reach in from the window before, a few per
        // cent have indels) and then some; a window that needs more says so (DE_CAPACITY, meta word 12) and the job is rerun
        // (a sharded job's records lie in the windows it works on)
        uint64_t want = 128, own_est = nwin;
        if (!ctx->emit.empty()) {
            own_est = 0;
            for (uint32_t c = 0; c < nc_full; c++)
                if (ctx->emit[2 * c + 1] > ctx->emit[2 * c]) own_est += (ctx->emit[2 * c + 1] - ctx->emit[2 * c]) / TILE + 1;
            own_est = std::max<uint64_t>(1, std::min<uint64_t>(own_est, nwin));
        }
        while (want < n / own_est / 4 + 64) want <<= 1;
        // Every window gets the same room, so ONE deep window (a collapsed repeat, a plasmid at 10,000x) sets it for all of them:
        // the room is capped at what keeps the windows' rooms together under PP_XENT_BUDGET bytes (default 8 GiB, never less than
        // the job's own estimate) -- a window that needs more sends the job over the bucketing path, which takes such a job with
        // 20 bytes per record (pp_polish_finish).  The room is this job's: what an earlier, deeper job of the context grew it
        // to is kept only while it stays under the cap.
        static const uint64_t xent_budget = getenv("PP_XENT_BUDGET") ? strtoull(getenv("PP_XENT_BUDGET"), nullptr, 10) : (8ull << 30);
        ctx->xcap_limit = (size_t)std::max<uint64_t>(want, xent_budget / ((uint64_t)nwin * 16));
        ctx->xcap = std::min<size_t>(std::max<size_t>(ctx->xcap, (size_t)want), ctx->xcap_limit);
        ctx->cap_ent = std::max<size_t>(ctx->cap_ent, (size_t)1 << 18);  // (work items in memory: only what k_xmat writes out)
    } else {
        ctx->cap_ent = std::max<size_t>(ctx->cap_ent, (size_t)(n + n / 4 + 4096));
    }
    const uint64_t coarse1_words = ((uint64_t)nwin / WIN_COARSE + 1 + 3) / 4 * 4 + 4, coarse2_words = ((uint64_t)nwin / WIN_COARSE2 + 1 + 3) / 4 * 4 + 4;
#define ENS(buf, bytes) if ((rc = dev_ensure(ctx, ctx->buf, (size_t)(bytes)))) return rc
    // metadata block (u64 words): 0 status | 1-2 counters | 3 work items | 4 scratch elements (10: the same, counted
    // as the positions are listed) |
    // 5 polished bytes | 6 ordered replay items | 8 key records | 9 longest fast read | 16.. contig output offsets
    // (nc+1) | then per-contig stats (3 words each)
    // ... | then the heavy-window list (HEAVY_WORDS u32: count, windows, arrival tickets)
    const size_t heavy_at = 16 + (size_t)nc + 1 + 3 * (size_t)nc;
    const size_t meta_words = heavy_at + (HEAVY_WORDS + 1) / 2;
    ENS(b_meta, meta_words * 8);
    ENS(b_vote_tab, (size_t)VOTE_TAB_N * 8);
    ENS(b_win_heavy, nwin);
    ENS(b_hslab, (size_t)HEAVY_SLOTS * HEAVY_PARTS * HSLAB_WORDS * 4);
    if (!direct) { ENS(b_gstart, n * 4); ENS(b_nkeep, n * 4); }
    if (direct) {
        ENS(b_first, (uint64_t)n_runs * (nwin + 1) * 4); ENS(b_xcnt, (uint64_t)nwin * 8);  /* extras per window | entries that are not bulk per window */ if (dev_ensure(ctx, ctx->b_xent, (size_t)((uint64_t)nwin * ctx->xcap * 16))) { (void)hipGetLastError(); ctx->no_direct = true; return run_pipeline(ctx, meta, n_entries_out); }  /* (no room for the windows' extras: the bucketing path) */
        ENS(b_need_win, (uint64_t)nwin * 4); ENS(b_win_lo, (uint64_t)nwin * 4); ENS(b_win_hi, (uint64_t)nwin * 4);
        ENS(b_later, PP_PREPD_TAIL ? 64 : cap_later * 32);  /* (the list of noted records in memory: the two-kernel build only) */
        std::vector<uint32_t> ends(ctx->wo_runs.begin(), ctx->wo_runs.end());
        if (!(ends == ctx->runs_on_dev && ctx->b_runs.p)) {  // (the same table as the job before: already there)
            const void *dummy;
            ctx->runs_on_dev.clear();
            if ((rc = upload(ctx, ctx->b_runs, ends.data(), ends.size() * 4, &dummy))) return rc;
            PP_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));  // (`ends` is a local)
            ctx->runs_on_dev = ends;
        }
    }
    // One level (items straight into their windows) while all windows fit one LDS pass of k_fill; two levels
    // (coarse buckets of COARSE_WINDOWS windows, then k_regroup) beyond that: there the single-level k_fill
    // would re-read its records once per range of 16384 windows.  Measured on MI355X: 5 Mbp: one level 0.19 ms
    // vs two 0.21 ms (+ k_tile 4 % slower on the regrouped order); 250 Mbp: one level 6.8 ms vs two 4.1 ms.
    static const int forced_levels = getenv("PP_BUCKET_LEVELS") ? atoi(getenv("PP_BUCKET_LEVELS")) : 0;  // tuning / tests
    const bool two_level = forced_levels ? forced_levels == 2 : nranges > 1;
    const uint32_t cw = !two_level ? 1 : (nwin >= COARSE_BIG_FROM ? COARSE_WINDOWS_BIG : COARSE_WINDOWS);
    const uint32_t ncoarse = (nwin + cw - 1) / cw;
    // columns (windows, or coarse buckets) per pass of k_fill: PP_FILL_RANGE (tuning), default one LDS range
    static const long forced_frange = getenv("PP_FILL_RANGE") ? atol(getenv("PP_FILL_RANGE")) : 0;
    const uint32_t frange = forced_frange > 0 ? (uint32_t)std::min<long>(forced_frange, COUNT_RANGE) : (uint32_t)COUNT_RANGE;
    const uint32_t ncranges = (ncoarse + frange - 1) / frange;
    if (!direct) {
        ENS(b_hist, (uint64_t)NB * ncoarse * 4); ENS(b_wincnt, (uint64_t)nwin * 4); ENS(b_winoff, ((uint64_t)nwin + 1) * 4);
        ENS(b_ccnt, (uint64_t)ncoarse * 4); ENS(b_coff, ((uint64_t)ncoarse + 1) * 4);
        if (two_level) ENS(b_entB, ctx->cap_ent * 16);
    }
    ENS(b_code, G); ENS(b_winlen, ((uint64_t)nwin + 3) / 4 * 16 + 16);  /* (room for a multiple of four windows: k_emit's fused prefix reads 16 bytes at a time) */ ENS(b_winout, ((uint64_t)nwin + 1) * 8); ENS(b_wincoarse, (coarse1_words + coarse2_words) * 4);  /* sums of win_len per 64 and per 4,096 windows, each padded for 16-byte loads */
    ENS(b_entA, ctx->cap_ent * 16);
    ENS(b_flag_pos, ctx->cap_flag * 4); ENS(b_flag_cov, ctx->cap_flag * 4); ENS(b_flag_scr, (ctx->cap_flag + 1) * 8);
    ENS(b_flag_bits, (uint64_t)nwin * (TILE / 8)); ENS(b_win_nflag, (uint64_t)nwin * 4);
    ENS(b_win_slab, (uint64_t)nwin * 4); ENS(b_slab_win, (uint64_t)ctx->cap_slabs * 4); ENS(b_slabs, (uint64_t)ctx->cap_slabs * 6 * TILE * 4); ENS(b_ents, (uint64_t)ctx->cap_ents * 16);
    if (ctx->debug) ENS(b_keys, (uint64_t)ctx->cap_keys * sizeof(KeyRec));
    ENS(b_scratch, ctx->cap_scr * 16); ENS(b_multi, ctx->cap_multi * sizeof(MultiEnt)); ENS(b_out, ctx->cap_out);
    if (ctx->debug) { ENS(b_dbg_depth, G * 8); ENS(b_dbg_counts, G * 28); ENS(b_dbg_status, G); }
#undef ENS
    u64 *d_meta = (u64 *)ctx->b_meta.p;
    u64 *d_status = d_meta;
    u32 *d_counters = (u32 *)(d_meta + 1);
    u64 *d_ctg_out = d_meta + 16;
    ContigStatsDev *d_stats = (ContigStatsDev *)(d_meta + 17 + nc);
    // zeros, status word = "no error"; a sharded job also gets its per-window output lengths and flag counts zeroed (the
    // windows nobody works on emit nothing and have nothing flagged) -- one launch instead of a kernel and two memsets
    // (the direct path: the same blocks zero its per-window counts of extras)
    // A job that went through leaves this done for the NEXT one (see the end of this function): a context that polishes
    // job after job of one shape -- a rank's share, a service -- starts with its first real kernel, and the 6 us of this one
    // run while the host is busy with the results of the job before.
    const bool sharded_job = !ctx->emit.empty();
    const pp_ctx::MetaReady meta_key{d_meta, (u32)meta_words, sharded_job ? ctx->b_winlen.p : nullptr, sharded_job ? ctx->b_win_nflag.p : nullptr,
                                     direct ? ctx->b_xcnt.p : nullptr, nwin, ctx->b_vote_tab.p, ctx->params.fraction_valid, ctx->params.fraction_invalid, ctx->b_wincoarse.p};
    const u32 n_coarse = (u32)(coarse1_words + coarse2_words);  // the sums of the windows' output lengths (k_emit's offsets), both levels
    auto launch_meta_init = [&]() {
        hipLaunchKernelGGL(k_meta_init, dim3(sharded_job || direct ? 1u + (nwin + 4095u) / 4096u : 1u), dim3(256), 0, st, d_meta, (u32)meta_words,
                           sharded_job ? (u32 *)ctx->b_winlen.p : (u32 *)nullptr, sharded_job ? (u32 *)ctx->b_win_nflag.p : (u32 *)nullptr,
                           direct ? (u32 *)ctx->b_xcnt.p : (u32 *)nullptr, direct ? (u32 *)ctx->b_xcnt.p + nwin : (u32 *)nullptr, nwin,
                           (u32 *)ctx->b_wincoarse.p, n_coarse, (u32 *)ctx->b_vote_tab.p, ctx->params.fraction_valid, ctx->params.fraction_invalid);
    };
    static const bool env_no_ahead = getenv("PP_INIT_AHEAD") && atoi(getenv("PP_INIT_AHEAD")) == 0;  // tuning / tests
    if (!(ctx->meta_ready_valid && ctx->meta_ready == meta_key)) launch_meta_init();
    ctx->meta_ready_valid = false;
    u32 *d_heavy = (u32 *)(d_meta + heavy_at);
    u8 *d_win_heavy = (u8 *)ctx->b_win_heavy.p;
    // A window is heavy from 1.5x the average number of records per window on (a few per cent below the items).  In a job
    // of uniform coverage no window gets there: 2,900 +- 60 items at 200x, and ~3,450 where the assembly has an indel --
    // the ~200 reads over it are three items each.  (At 1.25x those windows took the 32 slots of the list in a job with
    // planted indels, and the collapsed repeats the list is for went the ordinary way: configs[2] 0.98 -> 1.55 ms.)
    static const long forced_heavy = getenv("PP_HEAVY_MIN") ? atol(getenv("PP_HEAVY_MIN")) : 0;  // tuning / tests

    u32 *d_gstart = (u32 *)ctx->b_gstart.p, *d_nkeep = (u32 *)ctx->b_nkeep.p;
    u32 *d_hist = (u32 *)ctx->b_hist.p, *d_wincnt = (u32 *)ctx->b_wincnt.p, *d_winoff = (u32 *)ctx->b_winoff.p;
    const u64 *d_ctg = (const u64 *)ctx->b_contig_off.p;  // the RUN's contig table (the compact one below, if any)
    uint4 *d_entA = (uint4 *)ctx->b_entA.p, *d_entB = (uint4 *)ctx->b_entB.p;
    u32 *d_ccnt = (u32 *)ctx->b_ccnt.p, *d_coff = (u32 *)ctx->b_coff.p;

    const u32 *d_own = nullptr;       // (lo, hi) per contig of the RUN that this context emits (pp_polish_set_emit), or everything
    const u32 *d_own_full = nullptr;  // the same per contig of the JOB (k_prep looks records up by the job's contig index)
    const u64 *d_gbase = d_ctg;       // per contig of the job: where it starts in the run's coordinates
    const u32 *d_slice = nullptr;     // compact run: [lo, hi) of every contig of the job that the run holds
    const u8 *d_bases = ctx->d_bases;
    // Sharded job: only the windows that touch a range this context emits are worked on.  Their ranges (sorted, merged)
    // follow the (lo, hi) pairs in the same upload: [n_ranges | first window of each | windows before each (n + 1)].
    const u32 *d_own_win = nullptr;
    uint32_t n_own_win = nwin;
    if (!ctx->emit.empty()) {
        // one upload: [g_base (u64 x job contigs) | run contig table (u64 x nc + 1)] (compact runs only), then the u32 words
        std::vector<uint64_t> up64;
        if (compact) {
            up64.insert(up64.end(), g_base.begin(), g_base.end());
            up64.insert(up64.end(), run_off.begin(), run_off.end());
            up64.insert(up64.end(), src_start.begin(), src_start.end());
        }
        std::vector<uint32_t> up(ctx->emit);
        const size_t at_run = up.size();
        if (compact) up.insert(up.end(), run_emit.begin(), run_emit.end());
        std::vector<std::pair<uint32_t, uint32_t>> rng;  // [first, last] window
        for (uint32_t c = 0; c < nc; c++) {
            const uint64_t lo = run_emit[2 * c], hi = run_emit[2 * c + 1];
            if (hi <= lo) continue;
            const uint32_t w0 = (uint32_t)((run_off[c] + lo) / TILE), w1 = (uint32_t)((run_off[c] + hi - 1) / TILE);
            if (!rng.empty() && w0 <= rng.back().second + 1) rng.back().second = std::max(rng.back().second, w1);
            else rng.emplace_back(w0, w1);
        }
        const size_t at = up.size();
        up.push_back((uint32_t)rng.size());
        for (auto &r : rng) up.push_back(r.first);
        uint32_t before = 0;
        for (auto &r : rng) { up.push_back(before); before += r.second - r.first + 1; }
        up.push_back(before);
        n_own_win = before;
        const size_t at_slice = up.size();
        if (compact) up.insert(up.end(), slice.begin(), slice.end());  // compact run: the stretch of every contig of the job that the run holds
        std::vector<uint8_t> blob(up64.size() * 8 + up.size() * 4);
        if (!up64.empty()) memcpy(blob.data(), up64.data(), up64.size() * 8);
        memcpy(blob.data() + up64.size() * 8, up.data(), up.size() * 4);
        const void *p_own = ctx->b_own.p;
        if (!(ctx->own_blob == blob && ctx->b_own.p && ctx->b_own.cap >= blob.size())) {  // (the same ranges as the job before: already there)
            ctx->own_blob.clear();
            if (int rc = upload(ctx, ctx->b_own, blob.data(), blob.size(), &p_own)) return rc;
            ctx->own_blob = blob;
        }
        const u32 *words = (const u32 *)((const u8 *)p_own + up64.size() * 8);
        d_own_full = words;
        d_own = compact ? words + at_run : words;
        d_own_win = words + at;
        if (compact) {
            d_gbase = (const u64 *)p_own;
            d_ctg = d_gbase + nc_full;
            d_slice = words + at_slice;
            if (int rc = dev_ensure(ctx, ctx->b_sub_bases, G + 64)) return rc;
            hipLaunchKernelGGL(k_sub_bases, dim3((unsigned)((G + 8 * 256 - 1) / (8 * 256))), dim3(256), 0, st, ctx->d_bases,
                               d_ctg, nc, d_ctg + nc + 1, (u8 *)ctx->b_sub_bases.p, (u64)G);
            d_bases = (const u8 *)ctx->b_sub_bases.p;
        }
        // (the windows nobody works on emit nothing and have nothing flagged: k_meta_init zeroes win_len / win_nflag)
    }
    // (the average over the windows the context works on: a sharded job that is not compacted has its records there)
    const u32 heavy_min = forced_heavy > 0 ? (u32)forced_heavy
                                           : (u32)std::min<uint64_t>(MAX_BUCKET, std::max<uint64_t>(HEAVY_MIN_ITEMS, 3 * n / 2 / std::max<uint32_t>(1, n_own_win)));
    // The mirror checked against the arrays before anything reads the records through it: every mirror that is not one of the
    // library's own (pp_polish_add), and any with PP_CHECK_WO=1.  One that does not stand the check (DE_BAD_MIRROR) is left
    // aside: pp_polish_finish runs the job again without it.
    static const bool env_check_wo = getenv("PP_CHECK_WO") && atoi(getenv("PP_CHECK_WO")) != 0;
    const bool check_wo = env_check_wo || (ctx->wo_untrusted && !ctx->trust_all);
    if (check_wo && B.wo && !env_no_wo && !ctx->no_wo && n) {
        if ((rc = dev_ensure(ctx, ctx->b_aflag, (size_t)((n + 31) / 32) * 4))) return rc;
        PP_HIPCHK(ctx, hipMemsetAsync(ctx->b_aflag.p, 0, (size_t)((n + 31) / 32) * 4, st));
        hipLaunchKernelGGL(k_check_wo, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (u64)n, B.wo, B.contig, B.ref_start, B.k,
                           (const u64 *)B.seq_off, B.seq_len, (const u64 *)B.cig_off, B.n_cig, B.cigar, (u32 *)ctx->b_aflag.p, d_status);
    }
    // records -> (global start, kept entries, class); with all windows in one LDS range the same pass counts the
    // records of every block per window (two-level path: k_count, per range of windows)
    const bool fused_count = ncoarse <= (uint32_t)COUNT_RANGE;  // the columns (windows, or coarse buckets) fit one LDS range
    // the records through the batch's window-order mirror when it brings one (pp_aln_batch.wo; PP_WO=0: tuning / tests)
    static const bool no_wo = getenv("PP_WO") && atoi(getenv("PP_WO")) == 0;
    const pp_wo_rec *d_wo = no_wo || ctx->no_wo ? nullptr : B.wo;
#ifdef PP_PREP_STAMPS
    static DevBuf b_pstamps;
    const size_t pstamp_bytes = (size_t)2 * 16384 * 64;  // (8 ticks per workgroup, 16384 workgroups per kernel, two kernels)
    if (int rc2 = dev_ensure(ctx, b_pstamps, pstamp_bytes)) return rc2;
    PP_HIPCHK(ctx, hipMemsetAsync(b_pstamps.p, 0, pstamp_bytes, st));
    {
        u64 *sp = (u64 *)b_pstamps.p;
        PP_HIPCHK(ctx, hipMemcpyToSymbolAsync(HIP_SYMBOL(pp::g_prep_stamps), &sp, sizeof sp, 0, hipMemcpyHostToDevice, st));
    }
#endif
    if (direct) {
        // one pass over the mirror: validation, where the windows begin in every run, extras; then what each window holds.
        // (Geometry measured on configs[1]: one resident wave of 1024-thread workgroups, as k_prep / k_fill; more, smaller
        // ones only add their fixed round trips.)
        static const long forced_nbd = getenv("PP_PREPD_BLOCKS") ? atol(getenv("PP_PREPD_BLOCKS")) : 0;  // tuning
        static const long nbd_threads = getenv("PP_PREPD_THREADS") ? atol(getenv("PP_PREPD_THREADS")) : 1024;
        // (one resident wave of workgroups for the 5 Mbp job -- 13,000 entries each: a workgroup's list of noted entries holds
        // 768 of them, 6 % -- and as many more of that size as a larger job needs)
        // (PP_PREPD_TAIL, the noted records inside k_prepd: a round and a half of workgroups at least -- the chains of round trips of
        // one round under the streaming of the next --, 8,700 entries each: `profiles/r6zz_prepd_tail_ab.txt`)
        // ... and no more records that are not bulk than half a workgroup's list holds (768): the runs beyond one per record say how many
        // there are at most -- a read with one indel has three --, so a job with indels in every tenth read gets smaller workgroups
        // instead of lists that overflow (a workgroup then walks its entries once more: prep 0.35 ms at 10 %, 1.27 ms at 30 %)
        const uint64_t noted_est = B.n_cig_total > n ? (B.n_cig_total - n) / 2 : 0;
        const uint64_t nbd_default = PP_PREPD_TAIL ? std::max<uint64_t>(std::max<uint64_t>(768, n / 8700), noted_est / 400) : std::max<uint64_t>(512, n / 13000);
        const uint32_t NBD = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(forced_nbd > 0 ? (uint64_t)forced_nbd : nbd_default, (n + 2047) / 2048));
        const uint64_t chunk_d = (n + NBD - 1) / NBD;
        timer_begin(ctx, "prep");
PrepdArgs PA;
        PA.n = n; PA.wo = d_wo; PA.cig_off = (const u64 *)B.cig_off; PA.n_cig = B.n_cig; PA.cigar = B.cigar; PA.seq = B.seq;
        PA.contig_off = (const u64 *)ctx->b_contig_off.p; PA.n_contigs = nc_full; PA.nwin = nwin;
        PA.run_end = (const u32 *)ctx->b_runs.p; PA.n_runs = n_runs; PA.first = (u32 *)ctx->b_first.p;
        PA.x_cnt = (u32 *)ctx->b_xcnt.p; PA.x_nb = (u32 *)ctx->b_xcnt.p + nwin; PA.xent = (uint4 *)ctx->b_xent.p; PA.xcap = (u32)ctx->xcap;
        PA.maxlen = (u32 *)(d_meta + 9); PA.x_need = d_meta + 12;
        PA.g_later = (uint4 *)ctx->b_later.p; PA.g_nlater = d_meta + 15; PA.cap_later = cap_later; PA.seq_bytes = B.seq_bytes;
        PA.status = d_status;
        if (nbd_threads == 1024) hipLaunchKernelGGL(k_prepd<1024>, dim3(NBD), dim3(1024), 0, st, (u64)chunk_d, PA);
        else hipLaunchKernelGGL(k_prepd<512>, dim3(NBD), dim3(512), 0, st, (u64)chunk_d, PA);
        // ... and the records it only noted (indels, long reads: a few per cent), a lane each
        static const long prepg_div = getenv("PP_PREPG_DIV") && atol(getenv("PP_PREPG_DIV")) > 0 ? atol(getenv("PP_PREPG_DIV")) : 4096;  // records of the job per workgroup of k_prepg (tuning; 8192 until round 6: -1.5 us, tools/exp_prepg_sweep.sh)
        if (!PP_PREPD_TAIL) hipLaunchKernelGGL(k_prepg<256>, dim3((unsigned)std::max<uint64_t>(64, std::min<uint64_t>(16384, n / (uint64_t)prepg_div + 1))), dim3(256), 0, st, PA);
        timer_end(ctx);
        timer_begin(ctx, "bucket");
        hipLaunchKernelGGL(k_winplan, dim3((nwin + 255) / 256), dim3(256), 0, st, nwin, n_runs, (const u32 *)ctx->b_first.p,
                           (const u32 *)ctx->b_xcnt.p, (const u32 *)ctx->b_xcnt.p + nwin, (u32)ctx->xcap, heavy_min, d_heavy, d_win_heavy,
                           d_meta + 3, d_status);
        timer_end(ctx);
    } else {
    timer_begin(ctx, "prep");
#define PP_PREP_ARGS dim3(NB), dim3(1024), 0, st, (u64)n, (u64)chunk, d_wo, B.contig, B.ref_start, (const u64 *)B.seq_off, B.seq_len, \
                     (const u64 *)B.cig_off, B.n_cig, B.cigar, B.seq, (const u64 *)ctx->b_contig_off.p, nc_full, d_gbase, d_slice, d_own_full, \
                     d_gstart, d_nkeep, (u32 *)(d_meta + 9), nwin, cw, ncoarse
    if (!fused_count) {
        if (d_wo) hipLaunchKernelGGL((k_prep<false, true>), PP_PREP_ARGS, (u32 *)nullptr, d_status);
        else hipLaunchKernelGGL((k_prep<false, false>), PP_PREP_ARGS, (u32 *)nullptr, d_status);
    } else {
        if (d_wo) hipLaunchKernelGGL((k_prep<true, true>), PP_PREP_ARGS, d_hist, d_status);
        else hipLaunchKernelGGL((k_prep<true, false>), PP_PREP_ARGS, d_hist, d_status);
    }
#undef PP_PREP_ARGS
    timer_end(ctx);
    timer_begin(ctx, "bucket");
#define PP_FILL(CWV, ENT, OFF)                                                                                                  \
    do {                                                                                                                        \
        if (d_wo)                                                                                                               \
            hipLaunchKernelGGL((k_fill<CWV, true>), dim3(NB, ncranges), dim3(1024), 0, st, (u64)n, (u64)chunk, d_wo, d_gstart,   \
                               d_nkeep, B.k, (const u64 *)B.seq_off, B.seq_len, nwin, ncoarse, (const u32 *)d_hist,            \
                               (const u32 *)(OFF), ENT, frange, (u64)B.seq_bytes, d_status);                                   \
        else                                                                                                                    \
            hipLaunchKernelGGL((k_fill<CWV, false>), dim3(NB, ncranges), dim3(1024), 0, st, (u64)n, (u64)chunk, d_wo, d_gstart,  \
                               d_nkeep, B.k, (const u64 *)B.seq_off, B.seq_len, nwin, ncoarse, (const u32 *)d_hist,            \
                               (const u32 *)(OFF), ENT, frange, (u64)B.seq_bytes, d_status);                                   \
    } while (0)
    if (two_level) {
        if (!fused_count) {  // more than 16384 coarse buckets (a 2 Gbp assembly): counted range by range
            if (cw == (uint32_t)COARSE_WINDOWS_BIG)
                hipLaunchKernelGGL(k_count<COARSE_WINDOWS_BIG>, dim3(NB, nranges), dim3(1024), 0, st, (u64)n, (u64)chunk,
                                   d_gstart, d_nkeep, nwin, ncoarse, d_hist);
            else
                hipLaunchKernelGGL(k_count<COARSE_WINDOWS>, dim3(NB, nranges), dim3(1024), 0, st, (u64)n, (u64)chunk, d_gstart,
                                   d_nkeep, nwin, ncoarse, d_hist);
        }
        hipLaunchKernelGGL(k_scan_cols, dim3((ncoarse + 3) / 4), dim3(256), 0, st, ncoarse, NB, d_hist, d_ccnt, 0u,
                           (u32 *)nullptr, (u8 *)nullptr);
        hipLaunchKernelGGL(k_scan<u32>, dim3(1), dim3(1024), 0, st, (const u32 *)d_ccnt, (u64)ncoarse, (const u32 *)nullptr,
                           d_coff, d_meta + 3, (u64)ctx->cap_ent, d_status);
        if (cw == (uint32_t)COARSE_WINDOWS_BIG) {
            if (n) PP_FILL(COARSE_WINDOWS_BIG, d_entB, d_coff);
            hipLaunchKernelGGL(k_regroup<COARSE_WINDOWS_BIG>, dim3(ncoarse), dim3(1024), 0, st, nwin, ncoarse,
                               (const u32 *)d_coff, d_winoff, (const uint4 *)d_entB, d_entA, d_status);
        } else {
            if (n) PP_FILL(COARSE_WINDOWS, d_entB, d_coff);
            hipLaunchKernelGGL(k_regroup<COARSE_WINDOWS>, dim3(ncoarse), dim3(1024), 0, st, nwin, ncoarse,
                               (const u32 *)d_coff, d_winoff, (const uint4 *)d_entB, d_entA, d_status);
        }
        hipLaunchKernelGGL(k_heavy, dim3((nwin + 255) / 256), dim3(256), 0, st, nwin, (const u32 *)d_winoff, heavy_min,
                           d_heavy, d_win_heavy);
    } else {
        if (!fused_count)  // one level forced beyond one LDS range of windows (tuning): counted range by range
            hipLaunchKernelGGL(k_count<1>, dim3(NB, nranges), dim3(1024), 0, st, (u64)n, (u64)chunk, d_gstart, d_nkeep, nwin,
                               nwin, d_hist);
        hipLaunchKernelGGL(k_scan_cols, dim3((nwin + 3) / 4), dim3(256), 0, st, nwin, NB, d_hist, d_wincnt, heavy_min,
                           d_heavy, d_win_heavy);
        hipLaunchKernelGGL(k_scan<u32>, dim3(1), dim3(1024), 0, st, (const u32 *)d_wincnt, (u64)nwin, (const u32 *)nullptr,
                           d_winoff, d_meta + 3, (u64)ctx->cap_ent, d_status);
        if (n) PP_FILL(1, d_entA, d_winoff);
    }
#undef PP_FILL
    timer_end(ctx);
    }
#ifdef PP_PREP_STAMPS
    if (const char *path = getenv("PP_PREP_STAMPS_FILE")) {
        std::vector<uint64_t> hs(pstamp_bytes / 8);
        PP_HIPCHK(ctx, hipMemcpyAsync(hs.data(), b_pstamps.p, pstamp_bytes, hipMemcpyDeviceToHost, st));
        PP_HIPCHK(ctx, hipStreamSynchronize(st));
        if (FILE *f = fopen(path, "wb")) { fwrite(hs.data(), 1, pstamp_bytes, f); fclose(f); }
    }
#endif

    TileArgs T;
    T.entA = d_entA; T.win_off = d_winoff; T.nwin = nwin;
    T.seq = B.seq; T.seq_off = (const u64 *)B.seq_off; T.cig_off = (const u64 *)B.cig_off;
    static const bool no_seq4 = getenv("PP_SEQ4") && atoi(getenv("PP_SEQ4")) == 0;  // tuning / tests: ignore a batch's 4-bit mirror
    T.seq4 = no_seq4 ? nullptr : B.seq4;
    T.n_cig = B.n_cig; T.cigar = B.cigar; T.kk = B.k;
    T.bases = d_bases; T.G = G; T.contig_off = d_ctg; T.n_contigs = nc;
    T.min_depth = ctx->params.min_depth; T.fv = ctx->params.fraction_valid; T.fi = ctx->params.fraction_invalid;
    T.code = (u8 *)ctx->b_code.p; T.win_len = (u32 *)ctx->b_winlen.p; T.win_coarse = (u32 *)ctx->b_wincoarse.p; T.win_coarse2 = T.win_coarse + coarse1_words;
    T.counters = d_counters; T.cap_flag = (u32)ctx->cap_flag;
    T.vote_tab = (const u32 *)ctx->b_vote_tab.p;
    T.multi = (MultiEnt *)ctx->b_multi.p; T.cap_multi = (u32)ctx->cap_multi;
    T.flag_pos = (u32 *)ctx->b_flag_pos.p; T.flag_cov = (u32 *)ctx->b_flag_cov.p;
    T.flag_bits = (u32 *)ctx->b_flag_bits.p; T.win_nflag = (u32 *)ctx->b_win_nflag.p;
    T.win_slab = (u32 *)ctx->b_win_slab.p; T.slab_win = (u32 *)ctx->b_slab_win.p; T.slabs = (u32 *)ctx->b_slabs.p; T.cap_slabs = (u32)ctx->cap_slabs;
    T.stats = d_stats;
    T.maxlen = (const u32 *)(d_meta + 9);
    T.scr_need = d_meta + 10;
    T.flag_scr = (u64 *)ctx->b_flag_scr.p;
    T.seq_bytes = B.seq_bytes;
    T.own = d_own;
    T.own_win = d_own_win;
    T.heavy = d_heavy; T.win_heavy = d_win_heavy; T.hslab = (u32 *)ctx->b_hslab.p;
    T.dbg_depth = (double *)ctx->b_dbg_depth.p; T.dbg_counts = (u32 *)ctx->b_dbg_counts.p;
    T.dbg_status = (u8 *)ctx->b_dbg_status.p; T.status = d_status;
    // PP_DEBUG_REPLAY2=1 (tests): per-position records while order-dependent positions still go through k_exact2,
    // so that its f64 depths can be compared bit for bit (the key records of the TSV are then incomplete)
    static const bool dbg_replay2 = getenv("PP_DEBUG_REPLAY2") && atoi(getenv("PP_DEBUG_REPLAY2")) != 0;
    // debug_level 3 (tests: pp_polish_set_debug(ctx, 3)): per-position records of what k_tile itself decides -- positions with
    // inexact depth shares that its interval test settles are NOT sent to the replay, their record holds the thresholds and
    // the status as voted and the fixed-point depth (within the interval of the exact one)
    T.dbg = ctx->debug ? (ctx->debug_level == 3 ? 3 : (dbg_replay2 ? 2 : 1)) : 0;
    T.wo = (const uint4 *)d_wo; T.first = (const u32 *)ctx->b_first.p; T.n_runs = n_runs; T.xcap = (u32)ctx->xcap;
    T.x_cnt = (const u32 *)ctx->b_xcnt.p; T.xent = (const uint4 *)ctx->b_xent.p; T.need_win = (u32 *)ctx->b_need_win.p; T.n_need = d_meta + 13;
    const uint32_t per = (n_own_win + 7) / 8;  // windows to work on, dealt to the eight XCDs in stretches
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    const bool ev_launch = timer_for_launch(ctx, "tile", &ev_start, &ev_stop);
    if (!ev_launch) timer_begin(ctx, "tile");
#ifdef PP_TILE_STAMPS
    static DevBuf b_stamps;
    const size_t stamp_bytes = (size_t)(HEAVY_BLOCKS + per * 8) * 64;
    if (int rc2 = dev_ensure(ctx, b_stamps, stamp_bytes)) return rc2;
    PP_HIPCHK(ctx, hipMemsetAsync(b_stamps.p, 0, stamp_bytes, st));
    T.stamps = (u64 *)b_stamps.p;
#endif
    // one kernel per (lane-group width of the plain class, read fetch): the instance for the longest fast-class read the
    // context's job before had (a first job: short reads); an instance that does not take this job's raises DE_GW_HINT
    // (pp_polish_finish reruns with the job's own figure)
    {
        const uint32_t hint = ctx->maxlen_hint;
        const int gw = hint <= PlainCfg<5>::MAXL ? 5 : (hint <= PlainCfg<6>::MAXL ? 6 : 8);
        const dim3 grid(HEAVY_BLOCKS + per * 8), block(TILE_THREADS);
#define PP_TILE_LAUNCH(KERNEL)                                                                             \
        do {                                                                                               \
            if (ev_launch) hipExtLaunchKernelGGL(KERNEL, grid, block, 0, st, ev_start, ev_stop, 0, T);     \
            else hipLaunchKernelGGL(KERNEL, grid, block, 0, st, T);                                        \
        } while (0)
#define PP_TILE_PICK(NAME)                                                                                 \
        do {                                                                                               \
            if (T.seq4) {                                                                                  \
                if (gw == 5) PP_TILE_LAUNCH((NAME<5, true>));                                              \
                else if (gw == 6) PP_TILE_LAUNCH((NAME<6, true>));                                         \
                else PP_TILE_LAUNCH((NAME<8, true>));                                                      \
            } else if (gw == 5) PP_TILE_LAUNCH((NAME<5, false>));                                          \
            else if (gw == 6) PP_TILE_LAUNCH((NAME<6, false>));                                            \
            else PP_TILE_LAUNCH((NAME<8, false>));                                                         \
        } while (0)
        if (direct) PP_TILE_PICK(k_tile_direct);
        else PP_TILE_PICK(k_tile);
#undef PP_TILE_PICK
#undef PP_TILE_LAUNCH
    }
#ifdef PP_TILE_STAMPS
    if (const char *path = getenv("PP_TILE_STAMPS_FILE")) {
        std::vector<uint64_t> hs(stamp_bytes / 8);
        PP_HIPCHK(ctx, hipMemcpyAsync(hs.data(), b_stamps.p, stamp_bytes, hipMemcpyDeviceToHost, st));
        PP_HIPCHK(ctx, hipStreamSynchronize(st));
        if (FILE *f = fopen(path, "wb")) { fwrite(hs.data(), 1, stamp_bytes, f); fclose(f); }
    }
#endif
    timer_end(ctx);

    u64 *d_scr = (u64 *)ctx->b_flag_scr.p;
    ExactArgs E;
    E.cap_multi = (u32)ctx->cap_multi; E.cap_flag = (u32)ctx->cap_flag;
    E.flag_pos_w = T.flag_pos; E.flag_cov_w = T.flag_cov; E.flag_bits = T.flag_bits; E.win_nflag = T.win_nflag;
    E.win_slab = T.win_slab; E.slab_win = T.slab_win; E.cap_slabs = T.cap_slabs; E.slabs = T.slabs; E.ents = (ulonglong2 *)ctx->b_ents.p; E.cap_ents = ctx->cap_ents;
    E.ents_cursor = d_meta + 6;
    E.scr_need = d_meta + 10; E.cap_scr = (u64)ctx->cap_scr; E.flag_scr_w = d_scr;
    E.keys = (KeyRec *)ctx->b_keys.p; E.cap_keys = ctx->debug ? ctx->cap_keys : 0; E.n_keys = d_meta + 8;
    E.flag_pos = T.flag_pos; E.flag_cov = T.flag_cov; E.flag_scr = d_scr;
    E.entA = d_entA; E.seq = B.seq;
    E.win_lo = direct ? (const u32 *)ctx->b_win_lo.p : d_winoff;
    E.win_hi = direct ? (const u32 *)ctx->b_win_hi.p : d_winoff + 1; E.seq_off = (const u64 *)B.seq_off;
    E.cig_off = (const u64 *)B.cig_off; E.n_cig = B.n_cig; E.cigar = B.cigar; E.kk = B.k;
    E.bases = d_bases; E.G = G; E.contig_off = d_ctg; E.n_contigs = nc; E.seq_bytes = B.seq_bytes;
    E.min_depth = T.min_depth; E.fv = T.fv; E.fi = T.fi;
    E.scratch = (ulonglong2 *)ctx->b_scratch.p; E.code = T.code; E.win_len = T.win_len; E.win_coarse = T.win_coarse; E.win_coarse2 = T.win_coarse2;
    E.counters = d_counters; E.multi = (MultiEnt *)ctx->b_multi.p; E.stats = d_stats;
    E.dbg_depth = T.dbg_depth; E.dbg_counts = T.dbg_counts; E.dbg_status = T.dbg_status;
    E.status = d_status; E.dbg = T.dbg;
    E.heavy = d_heavy; E.win_heavy = d_win_heavy;
    // windows of up to SORT_MAX items: wave-per-position replay; the rest (and key-table overflows)
    // go through the global list to the thread-serial k_exact
    auto launch_exact = [&]() {
    timer_begin(ctx, "exact");
    if (direct)  // the windows k_tile listed for a replay get their work items written out: the replays read items
        hipLaunchKernelGGL(k_xmat, dim3((unsigned)std::min<uint32_t>(nwin, 512)), dim3(1024), 0, st, (const u32 *)ctx->b_need_win.p,
                           (const u64 *)(d_meta + 13), nwin, d_wo, n_runs, (const u32 *)ctx->b_first.p, (const u32 *)ctx->b_xcnt.p,
                           (const uint4 *)ctx->b_xent.p, (u32)ctx->xcap, d_ctg, nc, d_entA, (u64)ctx->cap_ent, d_meta + 14,
                           (u32 *)ctx->b_win_lo.p, (u32 *)ctx->b_win_hi.p, d_status);
    const unsigned n_replay = (unsigned)std::min<uint64_t>(nwin, ctx->cap_slabs);  // one block per window with a tally slab
    hipLaunchKernelGGL((k_exact2<SORT_SMALL, 0u, 1u>), dim3(n_replay), dim3(1024), 0, st, E, nwin);
    hipLaunchKernelGGL((k_exact2<SORT_MAX, SORT_SMALL, 1u>), dim3(n_replay), dim3(1024), 0, st, E, nwin);
    hipLaunchKernelGGL((k_exact2<SORT_SMALL, 0u, HEAVY_SUB>), dim3(HEAVY_SLOTS * HEAVY_SUB), dim3(1024), 0, st, E, nwin);
    // one workgroup per listed position; how many there will be is only known on the device -- the grid follows what the
    // context's job before listed (twice that, at least 256 and at most EXW_BLOCKS workgroups; the kernel strides)
    {
        uint32_t blocks = 256;
        while (blocks < EXW_BLOCKS && blocks < 2 * ctx->last_listed) blocks <<= 1;
        hipLaunchKernelGGL(k_exact, dim3(ctx->last_listed == ~0u ? EXW_BLOCKS / 4 : blocks), dim3(EXW_THREADS), 0, st, E);
    }
    timer_end(ctx);
    };

    u64 *d_winout = (u64 *)ctx->b_winout.p;
    static const bool env_no_spec = getenv("PP_SPECULATE") && atoi(getenv("PP_SPECULATE")) == 0;  // tuning / tests (see below)
    const bool speculate = ctx->nothing_flagged_last && !ctx->debug && !env_no_spec;
    const bool speculate_now = speculate;
    // The job's one read-back lands in pinned host memory -- written by k_emit's last workgroup (EmitTail, pp_k_emit.h), which also
    // sets the metadata block up for the next job when this one is through; PP_RESULT_COPY=1 (tuning / tests): the copy on the
    // stream and k_meta_init behind it, as until round 6.
    if (ctx->h_meta_words < meta_words + 2) {
        if (ctx->h_meta) (void)hipHostFree(ctx->h_meta);
        ctx->h_meta = nullptr;
        ctx->h_meta_words = 0;
        PP_HIPCHK(ctx, hipHostMalloc((void **)&ctx->h_meta, (meta_words + 64) * 8, hipHostMallocCoherent));  // (fine-grained: what the kernel writes is seen while it runs -- the polling below -- whatever HIP_HOST_COHERENT says)
        ctx->h_meta_words = meta_words + 64;
        PP_HIPCHK(ctx, hipHostGetDevicePointer((void **)&ctx->d_hmeta, ctx->h_meta, 0));
    }
    {   // k_emit's "who is last" counters: zero between launches (they reset themselves); zeroed here when they are new, or a launch may have been cut short
        const size_t want = (size_t)emit_done_words((uint64_t)nwin + 2048) * 8;
        const bool fresh = ctx->b_emit_done.cap < want;
        if ((rc = dev_ensure(ctx, ctx->b_emit_done, want))) return rc;
        if (fresh || !ctx->emit_done_clean) PP_HIPCHK(ctx, hipMemsetAsync(ctx->b_emit_done.p, 0, ctx->b_emit_done.cap, st));
        ctx->emit_done_clean = false;  // (until this pass's read-back says the emission ended as it should)
    }
    static const bool env_result_copy = getenv("PP_RESULT_COPY") && atoi(getenv("PP_RESULT_COPY")) != 0;
    u64 *const d_hmeta = env_result_copy ? nullptr : (u64 *)ctx->d_hmeta;
    // (the last workgroup alone zeroes the per-window counts: a job of up to EMIT_FUSE_MAX windows -- a larger one keeps k_meta_init's blocks)
    const bool tail_reinit = d_hmeta && !env_no_ahead && !ctx->debug && nwin <= EMIT_FUSE_MAX;
    // The host watches the serial in the pinned block instead of waiting for the stream: it has the results when the last workgroup
    // has written them, not when the kernel's end has been signalled and hipStreamSynchronize has noticed (-6 us a step;
    // `profiles/r6zz_results_to_host_ab.txt`).  The stream is asked every few thousand looks: a launch that failed never writes the
    // serial.  PP_SYNC=wait (tuning / tests): hipStreamSynchronize; PP_SYNC=query: hipStreamQuery in a loop.
    static const bool env_sync_wait = getenv("PP_SYNC") && strcmp(getenv("PP_SYNC"), "poll") != 0;
    const bool sync_by_poll = !env_sync_wait && d_hmeta;
    uint32_t emit_round = 0;
    auto launch_emit = [&]() {
    timer_begin(ctx, "emit");
    EmitTail Z{};
    Z.meta = d_meta; Z.words = (u32)meta_words; Z.host = d_hmeta; Z.serial = ++ctx->emit_serial; Z.done = (u64 *)ctx->b_emit_done.p;
    Z.reinit = !tail_reinit ? 0u : (speculate_now && emit_round == 0 ? 1u : 2u);
    emit_round++;
    Z.zero_a = sharded_job ? (u32 *)ctx->b_winlen.p : nullptr; Z.zero_b = sharded_job ? (u32 *)ctx->b_win_nflag.p : nullptr;
    Z.zero_c = direct ? (u32 *)ctx->b_xcnt.p : nullptr; Z.zero_d = direct ? (u32 *)ctx->b_xcnt.p + nwin : nullptr; Z.n_zero = nwin; Z.zero_e = T.win_coarse; Z.n_zero_e = n_coarse;
    Z.ordered = sync_by_poll ? 1u : 0u;
    // (no scan kernel -- k_emit's workgroups add the lengths in front of their window up themselves, from the coarse sums and their
    // group's windows; PP_EMIT_FUSE=0: tuning / tests)
    static const bool env_no_fuse = getenv("PP_EMIT_FUSE") && atoi(getenv("PP_EMIT_FUSE")) == 0;
    const bool fuse = !env_no_fuse;
    if (!fuse)
        hipLaunchKernelGGL(k_scan<u64>, dim3(1), dim3(1024), 0, st, (const u32 *)T.win_len, (u64)nwin, (const u32 *)nullptr,
                           d_winout, d_meta + 5, (u64)ctx->cap_out, d_status);
    // multi-byte winners + contig starts: one wave each, grid-stride.  How many winners there are is known on the device only: as many
    // waves as twice the job before had (a context's first job: as the room for them) -- 2,048 workgroups that found nothing to do
    // were half of a 5 Mbp job's launch.
    const uint64_t nfin = (ctx->last_multi == ~0u ? (uint64_t)ctx->cap_multi : std::min<uint64_t>(ctx->cap_multi, 2ull * ctx->last_multi + 64)) + nc + 1;
    const unsigned fin_blocks = (unsigned)std::min<uint64_t>(2048, (nfin + 3) / 4);
    #define PP_EMIT_ARGS dim3(n_own_win + fin_blocks), dim3(COMPACT_THREADS), 0, st, (const u8 *)T.code, (u64)G, (const u64 *)d_winout, (const u32 *)T.win_len, (const u32 *)T.win_coarse, (const u32 *)T.win_coarse2, (u64)ctx->cap_out, d_meta + 5, nwin, n_own_win, d_own_win, (const MultiEnt *)ctx->b_multi.p, (const u32 *)d_counters, B.seq, d_ctg, nc, (u8 *)ctx->b_out.p, d_ctg_out, d_status, Z
    if (fuse) hipLaunchKernelGGL(k_emit<true>, PP_EMIT_ARGS); else hipLaunchKernelGGL(k_emit<false>, PP_EMIT_ARGS);
#undef PP_EMIT_ARGS
    timer_end(ctx);
    };
    // The replays' five launches cost 23 us even when k_tile flagged nothing for them (4.5 us apiece: rocprofv3, round 5) --
    // a tenth of a 5 Mbp job's kernels.  A context whose job before had nothing flagged leaves them out, looks at what THIS
    // job flagged when its metadata are back, and only then -- anything flagged -- runs them and the emission once more
    // (a second synchronisation: ~35 us; the polished bytes are the same either way -- a flagged position emits nothing
    // until its replay has decided it).  Not with per-position records (they are the replays' to write).
    if (!speculate) launch_exact();
    launch_emit();
    PP_HIPCHK(ctx, hipGetLastError());

    // the job's one read-back: k_emit's last workgroup wrote it (or, PP_RESULT_COPY=1, a copy on the stream into the pinned block:
    // a copy into pageable memory goes through the runtime's staging buffer and keeps the host waiting for longer than the GPU needs)
    auto read_back = [&]() -> int {
        if (!d_hmeta) PP_HIPCHK(ctx, hipMemcpyAsync(ctx->h_meta, d_meta, meta_words * 8, hipMemcpyDeviceToHost, st));
        // PP_SYNC=query (tuning): poll the stream instead of waiting in hipStreamSynchronize
        static const bool sync_by_query = getenv("PP_SYNC") && !strcmp(getenv("PP_SYNC"), "query");
        if (sync_by_poll) {
            ctx->stream_may_be_busy = true;
            for (uint32_t spins = 0;; spins++) {
                if (__atomic_load_n(&ctx->h_meta[meta_words + 1], __ATOMIC_ACQUIRE) == ctx->emit_serial) break;
                if ((spins & 0xFFFu) == 0xFFFu) {
                    const hipError_t q = hipStreamQuery(st);
                    if (q != hipErrorNotReady) { PP_HIPCHK(ctx, q); break; }  // (done: the check below decides)
                }
            }
        } else if ((ctx->stream_may_be_busy = false), sync_by_query) {
            hipError_t q;
            while ((q = hipStreamQuery(st)) == hipErrorNotReady) {}
            PP_HIPCHK(ctx, q);
        } else PP_HIPCHK(ctx, hipStreamSynchronize(st));
        if (d_hmeta && __atomic_load_n(&ctx->h_meta[meta_words + 1], __ATOMIC_ACQUIRE) != ctx->emit_serial)
            return ctx->fail(PP_ERR_HIP, "the emission ended without its results in host memory");
        return PP_OK;
    };
    if ((rc = read_back())) return rc;
    ctx->emit_done_clean = true;
    if (speculate && ctx->h_meta[0] == ~0ull && (((const uint32_t *)&ctx->h_meta[1])[0] || ((const uint32_t *)&ctx->h_meta[1])[2])) {
        launch_exact();  // something was flagged after all
        launch_emit();
        PP_HIPCHK(ctx, hipGetLastError());
        if ((rc = read_back())) return rc;
    }
    meta.assign(ctx->h_meta, ctx->h_meta + meta_words);
    *n_entries_out = (uint32_t)meta[3];
    if (meta[0] == ~0ull && !env_no_ahead) {  // the job is through: the metadata block (and the per-window zeros) of the next one
        if (!(d_hmeta && ctx->h_meta[meta_words] == 1ull)) {  // (not already done by k_emit's last workgroup)
            launch_meta_init();
            PP_HIPCHK(ctx, hipGetLastError());
        }
        ctx->meta_ready = meta_key;
        ctx->meta_ready_valid = true;
    }
    if (getenv("PP_TRACE_FLAGGED")) {  // tuning: the positions this pass listed for k_exact
        const uint32_t *c = (const uint32_t *)&meta[1];
        const uint32_t nl = std::min<uint32_t>(c[0], 64);
        std::vector<uint32_t> pos(nl ? nl : 1), cov(nl ? nl : 1);
        if (nl) {
            PP_HIPCHK(ctx, hipMemcpy(pos.data(), ctx->b_flag_pos.p, nl * 4, hipMemcpyDeviceToHost));
            PP_HIPCHK(ctx, hipMemcpy(cov.data(), ctx->b_flag_cov.p, nl * 4, hipMemcpyDeviceToHost));
        }
        fprintf(stderr, "[flagged] %s path: listed %u, flagged in all %u, windows written out %llu:", direct ? "direct" : "bucketing", c[0], c[2],
                (unsigned long long)meta[13]);
        for (uint32_t i = 0; i < nl; i++) fprintf(stderr, " %u(cov %u)", pos[i], cov[i]);
        fprintf(stderr, "\n");
    }
    return PP_OK;
}

// the record-level device errors of the CIGAR walk, for callers that renumber records (a sharded job's ranks)
extern "C" int pp_polish_error_record(const pp_ctx *ctx, uint64_t *record, uint32_t *kind) {
    if (!ctx || ctx->last_dev_error == ~0ull) return 0;
    const uint32_t code = (uint32_t)(ctx->last_dev_error & 0xFF);
    if (code < DE_UNEXPECTED_OP || code > DE_BAD_ENDS) return 0;  // not about one alignment record
    if (record) *record = ctx->last_dev_error >> 8;
    if (kind) *kind = code;
    return 1;
}
extern "C" int pp_polish_error_text(pp_ctx *ctx, uint32_t kind, uint64_t record) {
    if (!ctx || kind < DE_UNEXPECTED_OP || kind > DE_BAD_ENDS) return PP_ERR_ARG;
    return map_device_error(ctx, (record << 8) | kind);
}

extern "C" int pp_polish_finish(pp_ctx *ctx) {
    if (!ctx) return PP_ERR_ARG;
    if (int rdy = pp_ctx_wait(ctx)) return rdy;
    if (!ctx->job_open) return ctx->fail(PP_ERR_ARG, "pp_polish_finish without pp_polish_begin");
    ctx->last_dev_error = ~0ull;
    PP_HIPCHK(ctx, hipSetDevice(ctx->device));
    const uint64_t G = ctx->G;
    const uint32_t nc = ctx->n_contigs;
    // optimistic capacities (grow-only across jobs; the work items' is set by run_pipeline: it depends on the path)
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
    ctx->no_compact = false;
    ctx->no_direct = false;
    ctx->no_wo = false;
    for (;; attempt++) {
        timers_release(ctx);
        int rc = run_pipeline(ctx, meta, &n_entries);
        if (rc) return rc;
        const uint64_t key = meta[0];
        if (key == ~0ull) break;
        if ((key & 0xFF) == DE_HALO && !ctx->no_compact) {  // a read longer than the halo of a compact run: run over the whole assembly
            ctx->no_compact = true;
            continue;
        }
        if ((key & 0xFF) == DE_GW_HINT && ctx->maxlen_hint != (uint32_t)meta[9]) {  // k_tile's instance does not take this job's reads: the one that does
            ctx->maxlen_hint = (uint32_t)meta[9];
            continue;
        }
        if ((key & 0xFF) == DE_BAD_MIRROR && !ctx->no_wo) {  // the mirror is not what its name says (a hint: every result the same without it)
            static const bool trace_m = getenv("PP_TIMING") != nullptr;
            if (trace_m) fprintf(stderr, "[timing] pass %d: entry %llu of the window-order mirror does not mirror its record -> the job runs without the mirror\n",
                                 attempt + 1, (unsigned long long)(key >> 8));
            ctx->no_wo = true;
            continue;
        }
        if ((key & 0xFF) == DE_MIRROR_ORDER && !ctx->no_direct) {  // the mirror is not in the order its run table promises: the bucketing path
            static const bool trace_d = getenv("PP_TIMING") != nullptr;
            if (trace_d) fprintf(stderr, "[timing] pass %d: the window-order mirror is not in run order -> bucketing path\n", attempt + 1);
            ctx->no_direct = true;
            continue;
        }
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
        grow(ctx->cap_ent, ctx->last_direct ? meta[14] : meta[3], "work items");
        if (ctx->last_direct && meta[12] > ctx->xcap) {
            if (meta[12] + meta[12] / 8 + 1024 > ctx->xcap_limit) {  // one window needs more room than every window can be given: the bucketing path
                if (trace) fprintf(stderr, "[timing] pass %d: a window needs %llu extras (room for %zu at most) -> bucketing path\n", attempt + 1,
                                   (unsigned long long)meta[12], ctx->xcap_limit);
                ctx->no_direct = true;
                grew = true;
            } else grow(ctx->xcap, meta[12], "extras per window");
        }
        grow(ctx->cap_flag, std::min<uint64_t>(cnt[0], G), "listed positions");
        grow(ctx->cap_scr, meta[10], "replay scratch");
        grow(ctx->cap_multi, cnt[1], "multi-byte winners");
        grow(ctx->cap_slabs, cnt[3], "tally slabs");
        grow(ctx->cap_ents, meta[6], "ordered replay items");
        if (ctx->debug) grow(ctx->cap_keys, meta[8], "key records");
        grow(ctx->cap_out, meta[5], "polished bytes");
        if (!grew) return ctx->fail(PP_ERR_HIP, "device reported a capacity overflow that the host cannot locate");
    }
    const uint32_t *cnt = (const uint32_t *)&meta[1];
    ctx->total_out = meta[5];
    ctx->last_listed = cnt[0];
    ctx->nothing_flagged_last = cnt[0] == 0 && cnt[2] == 0;
    ctx->maxlen_hint = (uint32_t)meta[9];
    ctx->n_multi = cnt[1];
    ctx->last_multi = cnt[1];
    ctx->n_keys = meta[8];
    // per-contig results: the run's contigs are the job's, or (compact run) the ones this context owns -- the others
    // have no bytes and no statistics here
    const uint32_t rnc = ctx->run_nc;
    const uint64_t *r_out = &meta[16];
    const ContigStatsDev *hs = (const ContigStatsDev *)&meta[17 + rnc];
    ctx->contig_out_off.assign((size_t)nc + 1, 0);
    ctx->stats.assign(nc, pp_contig_stats{0, 0, 0, 0.0});
    uint32_t j = 0;  // contigs of the run seen so far
    for (uint32_t c = 0; c < nc; c++) {
        ctx->contig_out_off[c] = r_out[j];
        const bool mine = ctx->run_full_of.empty() || (j < rnc && ctx->run_full_of[j] == c);
        if (!mine) continue;
        ctx->stats[c].polished_len = r_out[j + 1] - r_out[j];
        ctx->stats[c].changed = hs[j].changed;
        ctx->stats[c].zero_depth = hs[j].zero_depth;
        ctx->stats[c].depth_sum = (double)hs[j].depth_fx / (double)(1u << DEPTH_FX_BITS);
        j++;
    }
    ctx->contig_out_off[nc] = r_out[rnc];
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
    if (!(ctx && ctx->job_done)) return nullptr;
    // (pp_polish_finish returns when k_emit's last workgroup has handed the results over -- the kernel's end, with its write-back of
    // the bytes for readers outside this stream, may be a few microseconds away: the caller reads through a stream of its own)
    if (ctx->stream_may_be_busy) {
        (void)hipStreamSynchronize(ctx->stream);
        ctx->stream_may_be_busy = false;
    }
    return (const uint8_t *)ctx->b_out.p;
}

extern "C" int pp_polish_set_debug(pp_ctx *ctx, int enable) {
    if (!ctx) return PP_ERR_ARG;
    ctx->debug = enable != 0;
    ctx->debug_level = enable;
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

extern "C" int pp_polish_took_direct_path(const pp_ctx *ctx) { return ctx && ctx->last_direct ? 1 : 0; }

extern "C" int pp_polish_kernel_times(pp_ctx *ctx, pp_kernel_times *out) {
    if (!ctx || !out) return PP_ERR_ARG;
    *out = ctx->last_times;
    return PP_OK;
}

// ---- context -----------------------------------------------------------------------------------
extern "C" void pp_tokenize_warm_(hipStream_t st);
extern "C" void pp_filter_warm_(hipStream_t st);
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

extern "C" int pp_device_count(void) {
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess && n > 0 ? n : 0;
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
        if (ctx->init_rc == PP_OK) {  // load the code objects (one per translation unit) and spin up the queue while the host parses
            hipLaunchKernelGGL(k_warm, dim3(1), dim3(64), 0, ctx->stream, (uint32_t *)nullptr);
            pp_tokenize_warm_(ctx->stream);
            pp_filter_warm_(ctx->stream);
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
    pp_comm_destroy(ctx);
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    DevBuf *all[] = {&ctx->b_comm, &ctx->b_gather, &ctx->b_bases, &ctx->b_contig_off, &ctx->b_status, &ctx->b_gstart, &ctx->b_nkeep,
                     &ctx->b_aflag, &ctx->b_hist, &ctx->b_wincnt, &ctx->b_winoff, &ctx->b_entA, &ctx->b_entB, &ctx->b_ccnt, &ctx->b_coff,
                     &ctx->b_code, &ctx->b_winlen, &ctx->b_winout, &ctx->b_flag_pos, &ctx->b_flag_cov,
                     &ctx->b_flag_scr, &ctx->b_scratch, &ctx->b_multi, &ctx->b_meta, &ctx->b_vote_tab, &ctx->b_flag_bits, &ctx->b_win_nflag, &ctx->b_win_slab, &ctx->b_slab_win, &ctx->b_slabs, &ctx->b_ents, &ctx->b_keys, &ctx->b_own,
                     &ctx->b_win_heavy, &ctx->b_hslab, &ctx->b_sub_bases,
                     &ctx->b_runs, &ctx->b_first, &ctx->b_xcnt, &ctx->b_xent, &ctx->b_need_win, &ctx->b_win_lo, &ctx->b_win_hi, &ctx->b_later,
                     &ctx->b_out, &ctx->b_emit_done, &ctx->b_wincoarse, &ctx->b_dbg_depth, &ctx->b_dbg_counts, &ctx->b_dbg_status,
                     &ctx->f_refend[0], &ctx->f_refend[1], &ctx->f_pass[0], &ctx->f_pass[1], &ctx->f_orient, &ctx->f_poisoned,
                     &ctx->f_insert, &ctx->f_list, &ctx->f_blkcnt};
    for (DevBuf *b : all) dev_free(*b);
    for (auto &b : ctx->b_in) dev_free(b);
    for (auto &b : ctx->b_split) dev_free(b);
    for (auto &f : ctx->f_in) for (auto &b : f) dev_free(b);
    timers_release(ctx);
    if (ctx->h_meta) (void)hipHostFree(ctx->h_meta);
    for (hipEvent_t e : ctx->event_pool) (void)hipEventDestroy(e);
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

// Several contexts of one process on different GPUs (pp_polish_files_multi): let every GPU reach the others' memory
// directly, so that the records travel GPU to GPU over xGMI (PP_MEM_PEER copies); best effort -- without peer access the
// runtime stages such a copy through the host.
extern "C" void pp_ctx_enable_peers_(pp_ctx *const *ctxs, int n) {
    for (int i = 0; i < n; i++) {
        if (!ctxs[i] || pp_ctx_wait(ctxs[i]) != PP_OK) continue;
        for (int j = 0; j < n; j++) {
            if (!ctxs[j] || ctxs[j]->device == ctxs[i]->device || pp_ctx_wait(ctxs[j]) != PP_OK) continue;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, ctxs[i]->device, ctxs[j]->device) != hipSuccess || !can) continue;
            if (hipSetDevice(ctxs[i]->device) != hipSuccess) continue;
            (void)hipDeviceEnablePeerAccess(ctxs[j]->device, 0);  // "already enabled" is fine
        }
    }
    (void)hipGetLastError();
}

extern "C" int pp_ctx_device_(const pp_ctx *ctx) { return ctx ? ctx->device : -1; }
// (internal: bench / tests that lay their batches out as the library's ingests do) every window-order mirror this context is
// given counts as one of the library's own: not compared with the arrays before it is used
extern "C" int pp_ctx_trust_mirrors_(pp_ctx *ctx, int on) {
    if (!ctx) return PP_ERR_ARG;
    ctx->trust_all = on != 0;
    return PP_OK;
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
    PP_HIPCHK(ctx, hipSetDevice(ctx->device));
    PP_HIPCHK(ctx, hipMemcpyAsync(host_dst, dev_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    PP_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return PP_OK;
}

extern "C" void *pp_ctx_stream(pp_ctx *ctx) { return ctx && pp_ctx_wait(ctx) == PP_OK ? (void *)ctx->stream : nullptr; }
extern "C" const char *pp_version(void) { return "polypolish-mi355x 0.1.0 (parity target v0.6.1)"; }
