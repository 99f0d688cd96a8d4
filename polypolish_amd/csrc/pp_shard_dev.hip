// pp_shard_dev.hip -- pp_shard_split / pp_shard_count: the RECORDS of a sharded polish job go where they are needed.
//
// A position's pileup only ever sees the alignments of its own contig that cover it (src/alignment.rs:297-303 picks the
// pileup by RNAME, src/pileup.rs:189-200 indexes it by ref_start + j), so a rank that polishes some contigs -- or one
// window of a large contig -- needs only the records that reach them ("each rank receives only its
// contigs' bases and alignment records"; "a record is sent to every window it overlaps").  pp_shard_split picks those
// records out of a batch in file order, for one destination rank of a pp_shard_plan:
//   * span of a record = the sum of its M / = / X / D / N run lengths (what get_ref_end adds up, src/alignment.rs:
//     138-149), at least 1: a generous bound of the positions it can touch (the homopolymer trim only shortens it);
//   * the record goes to the rank of EVERY unit its [ref_start, ref_start + span) touches -- a read across a window
//     boundary goes to both neighbours, each of which emits only its own positions (pp_polish_set_emit), so an owned
//     position still sees all of its alignments, in file order, and the order-dependent f64 depth stays exact;
//   * a record that touches no unit (contig index out of range, start beyond the contig's end) goes to ONE fixed rank,
//     which reports it: whatever is wrong with a record is found by somebody.
// The part keeps orig[i] = index of its record i in the source batch, so that a rank-local record number in a device
// error can be turned back into the job's (the first bad record of the job is the minimum over the ranks).
// Device batches are split by kernels on the context's stream (flag, three scans, gather of the SoA, the SEQ bytes
// and the CIGAR runs); host batches by a plain loop.  No reference counterpart: the reference is one thread.
#include "pp_devtext.h"

#include <algorithm>
#include <cstring>
#include <vector>

struct pp_shard_part {
    pp_ctx *ctx = nullptr;
    int mem = PP_MEM_HOST;
    pp::DevBuf d_all;   // one allocation, carved into: contig ref_start k seq_off seq_len cig_off n_cig seq cigar orig seq4 wo
    void *d[12] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    bool has_wo = false;
    std::vector<pp_wo_rec> h_wo;
    std::vector<uint64_t> wo_runs;  // the ends of the part's mirror's runs (the source's runs, restricted), or empty: not known
    std::vector<uint32_t> h_contig, h_ref_start, h_k, h_seq_len, h_n_cig, h_cigar, h_orig;
    std::vector<uint64_t> h_seq_off, h_cig_off;
    std::vector<uint8_t> h_seq;
    pp_aln_batch view{};
    const uint32_t *orig = nullptr;
};

namespace {
// bytes of the part's seq array a record takes: its SEQ bytes up to the next PP_SEQ_ALIGN boundary (include/polypolish_hip.h)
__host__ __device__ inline u32 seq_room(u32 n) { return (n + (u32)PP_SEQ_ALIGN - 1u) & ~((u32)PP_SEQ_ALIGN - 1u); }

struct UnitTable {       // the plan, per contig: units [first[c], first[c + 1]), in position order
    const u32 *first;    // n_contigs + 1
    const u32 *lo, *hi;  // [lo, hi) of the contig
    const u32 *rank;
    u32 n_contigs, fallback;
};

__host__ __device__ inline bool ref_consuming(u32 op) {
    return op == PP_OP_M || op == PP_OP_EQ || op == PP_OP_X || op == PP_OP_D || op == PP_OP_N;
}

// does record (c, rs, span) go to `dest`?
__host__ __device__ inline bool goes_to(const UnitTable &T, u32 c, u64 rs, u64 span, u32 dest) {
    if (c >= T.n_contigs) return dest == T.fallback;
    bool hit = false, mine = false;
    for (u32 u = T.first[c]; u < T.first[c + 1]; u++)
        if (rs < (u64)T.hi[u] && rs + span > (u64)T.lo[u]) {
            hit = true;
            mine |= T.rank[u] == dest;
        }
    if (!hit) return T.first[c + 1] > T.first[c] ? T.rank[T.first[c + 1] - 1] == dest : dest == T.fallback;
    return mine;
}

__global__ __launch_bounds__(256) void k_split_flag(u64 n, const u32 *__restrict__ contig, const u32 *__restrict__ ref_start,
                                                    const u32 *__restrict__ seq_len, const u64 *__restrict__ cig_off,
                                                    const u32 *__restrict__ n_cig, const u32 *__restrict__ cigar, UnitTable T,
                                                    u32 dest, u32 *__restrict__ flag, u32 *__restrict__ sel_seq,
                                                    u32 *__restrict__ sel_cig, const u64 *__restrict__ seq_off,
                                                    u32 *__restrict__ slots, u64 n_slots, u32 *__restrict__ bad) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u32 nr = n_cig[i];
    const u32 *cg = cigar + cig_off[i];
    u64 span = 0;
    for (u32 r = 0; r < nr; r++) {
        const u32 op = cg[r];
        if (ref_consuming(op & 15u)) span += op >> 4;
    }
    if (span == 0) span = 1;
    const bool sel = goes_to(T, contig[i], ref_start[i], span, dest);
    flag[i] = sel ? 1u : 0u;
    sel_seq[i] = sel ? seq_room(seq_len[i]) : 0u;
    sel_cig[i] = sel ? nr : 0u;
    // the order of the SEQ bytes in the source's seq array is kept (a window-grouped batch gives window-grouped parts):
    // the array as slots of PP_SEQ_ALIGN bytes, a selected record's room noted at its first slot, scanned by the caller.
    // A batch whose records do not start on slots of their own (not from this library's producers) says so in *bad and
    // the part is laid out in the order of the records.
    if (sel && slots) {
        const u64 so = seq_off[i];
        if ((so & ((u64)PP_SEQ_ALIGN - 1u)) || (so >> 5) >= n_slots || atomicExch(&slots[so >> 5], seq_room(seq_len[i]) >> 5) != 0u) atomicOr(bad, 1u);
    }
}

// a selected record's place in the part's seq array: the scan of the slots in front of its own
__global__ __launch_bounds__(256) void k_split_place(u64 n, const u32 *__restrict__ flag, const u64 *__restrict__ seq_off,
                                                     const u32 *__restrict__ slot_scan, u64 *__restrict__ seq_scan) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && flag[i]) seq_scan[i] = (u64)slot_scan[seq_off[i] >> 5] << 5;
}

struct SplitOut {
    u32 *contig, *ref_start, *k, *seq_len, *n_cig, *cigar, *orig;
    u64 *seq_off, *cig_off;
    u8 *seq;
};

__global__ __launch_bounds__(256) void k_split_meta(u64 n, const u32 *__restrict__ contig, const u32 *__restrict__ ref_start,
                                                    const u32 *__restrict__ kk, const u32 *__restrict__ seq_len,
                                                    const u32 *__restrict__ n_cig, const u32 *__restrict__ flag,
                                                    const u32 *__restrict__ out_idx, const u64 *__restrict__ seq_scan,
                                                    const u64 *__restrict__ cig_scan, SplitOut O) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !flag[i]) return;
    const u32 o = out_idx[i];
    O.contig[o] = contig[i];
    O.ref_start[o] = ref_start[i];
    O.k[o] = kk[i];
    O.seq_len[o] = seq_len[i];
    O.n_cig[o] = n_cig[i];
    O.seq_off[o] = seq_scan[i];
    O.cig_off[o] = cig_scan[i];
    O.orig[o] = (u32)i;
}

// SEQ bytes and their 4-bit mirror (pp_aln_batch.seq4: the bytes are in registers anyway -- a part always brings one):
// eight lanes per selected record, one 16-byte chunk of its room per lane and trip (gfx950 global accesses need no alignment)
__device__ __forceinline__ u32 split_code4(u32 c) {
    const u32 t = (c >> 1) & 3u;
    const u32 expect = (0x47544341u >> (t * 8u)) & 0xFFu;
    return c == expect ? t : (c == (u32)'N' ? (u32)PP_SEQ4_N : (c == (u32)'-' ? (u32)PP_SEQ4_DASH : (u32)PP_SEQ4_OTHER));
}
__global__ __launch_bounds__(256) void k_split_seq(u64 n, const u64 *__restrict__ seq_off, const u32 *__restrict__ seq_len,
                                                   const u8 *__restrict__ seq, const u32 *__restrict__ flag,
                                                   const u64 *__restrict__ seq_scan, u8 *__restrict__ out, u8 *__restrict__ out4) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x, i = t >> 3;
    const u32 s = (u32)(t & 7u);
    if (i >= n || !flag[i]) return;
    const u8 *in = seq + seq_off[i];
    const u64 at = seq_scan[i];  // a multiple of PP_SEQ_ALIGN
    u8 *o = out + at, *o4 = out4 + (at >> 1);
    const u32 len = seq_len[i], room = seq_room(len);
    for (u32 b = 16u * s; b < room; b += 128u) {
        u32 w[4] = {0, 0, 0, 0};
        if (b + 16u <= len) {
            uint4 v;
            __builtin_memcpy(&v, in + b, 16);
            w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
        } else {
            for (u32 j = 0; b + j < len; j++) w[j >> 2] |= (u32)in[b + j] << (8u * (j & 3u));  // zeros up to the next boundary
        }
        const uint4 v = make_uint4(w[0], w[1], w[2], w[3]);
        __builtin_memcpy(o + b, &v, 16);
        u32 q4[2];
#pragma unroll
        for (int q = 0; q < 2; q++) {
            u32 x = 0;
#pragma unroll
            for (int j = 0; j < 8; j++) x |= split_code4((w[2 * q + (j >> 2)] >> (8 * (j & 3))) & 0xFFu) << (4 * j);
            q4[q] = x;
        }
        const uint2 p4 = make_uint2(q4[0], q4[1]);
        __builtin_memcpy(o4 + (b >> 1), &p4, 8);
    }
}

// The part's window-order mirror (pp_aln_batch.wo): the source's mirror restricted to the selected records, in the source's
// order -- entry j of the source goes along if its record does (mflag, scanned into mpos by the caller), with the record's
// new place in the part's seq array and its new index among the part's records.  idx_base: what the source mirror's
// file_idx count from (a view on records [lo, hi) of a batch keeps the batch's numbering).
__global__ __launch_bounds__(256) void k_split_wo_flag(u64 n, const pp_wo_rec *__restrict__ wo, u32 idx_base, const u32 *__restrict__ flag,
                                                       u32 *__restrict__ mflag, u32 *__restrict__ bad) {
    const u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const u32 fi = wo[j].file_idx - idx_base;
    if (fi >= n) { atomicOr(bad, 2u); mflag[j] = 0; return; }  // not this view's mirror: the part goes without one
    mflag[j] = flag[fi];
}
__global__ __launch_bounds__(256) void k_split_wo(u64 n, const pp_wo_rec *__restrict__ wo, u32 idx_base, const u32 *__restrict__ mflag,
                                                  const u32 *__restrict__ mpos, const u32 *__restrict__ out_idx,
                                                  const u64 *__restrict__ seq_scan, pp_wo_rec *__restrict__ out) {
    const u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n || !mflag[j]) return;
    pp_wo_rec w = wo[j];
    const u32 fi = w.file_idx - idx_base;
    w.seq_off = seq_scan[fi];
    w.file_idx = out_idx[fi];
    out[mpos[j]] = w;
}

__global__ __launch_bounds__(256) void k_split_cigar(u64 n, const u64 *__restrict__ cig_off, const u32 *__restrict__ n_cig,
                                                     const u32 *__restrict__ cigar, const u32 *__restrict__ flag,
                                                     const u64 *__restrict__ cig_scan, u32 *__restrict__ out) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !flag[i]) return;
    const u32 *in = cigar + cig_off[i];
    u32 *o = out + cig_scan[i];
    const u32 nr = n_cig[i];
    for (u32 r = 0; r < nr; r++) o[r] = in[r];
}

// alignment records per contig: per-block LDS histogram while the contigs fit, global atomics beyond
constexpr u32 HIST_LDS = 8192;
__global__ __launch_bounds__(1024) void k_contig_hist(u64 n, const u32 *__restrict__ contig, u32 n_contigs,
                                                      u64 *__restrict__ counts) {
    __shared__ u32 h[HIST_LDS];
    const bool lds = n_contigs <= HIST_LDS;
    if (lds)
        for (u32 c = threadIdx.x; c < n_contigs; c += blockDim.x) h[c] = 0;
    __syncthreads();
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
        const u32 c = contig[i];
        if (c >= n_contigs) continue;
        if (lds) atomicAdd(&h[c], 1u); else atomicAdd(&counts[c], 1ull);
    }
    __syncthreads();
    if (lds)
        for (u32 c = threadIdx.x; c < n_contigs; c += blockDim.x)
            if (h[c]) atomicAdd(&counts[c], (u64)h[c]);
}

// the plan as per-contig unit lists (units come contig by contig, windows in position order)
struct HostUnits {
    std::vector<u32> first, lo, hi, rank;
    u32 fallback = 0;
    bool build(const pp_shard_plan *p) {
        first.assign((size_t)p->n_contigs + 1, 0);
        lo.resize(p->n_units); hi.resize(p->n_units); rank.resize(p->n_units);
        u32 prev = 0;
        for (u32 u = 0; u < p->n_units; u++) {
            if (p->contig[u] >= p->n_contigs || p->contig[u] < prev || p->hi[u] > 0xFFFFFFFFull || p->rank[u] >= p->world) return false;
            prev = p->contig[u];
            first[p->contig[u] + 1]++;
            lo[u] = (u32)p->lo[u]; hi[u] = (u32)p->hi[u]; rank[u] = p->rank[u];
        }
        for (u32 c = 0; c < p->n_contigs; c++) first[c + 1] += first[c];
        fallback = p->n_units ? p->rank[p->n_units - 1] : 0;
        return true;
    }
    UnitTable table(u32 n_contigs) const { return UnitTable{first.data(), lo.data(), hi.data(), rank.data(), n_contigs, fallback}; }
};

void set_view(pp_shard_part *P, uint64_t n, uint64_t seq_bytes, uint64_t n_cig_total) {
    pp_aln_batch &v = P->view;
    v.n_aln = n; v.seq_bytes = seq_bytes; v.n_cig_total = n_cig_total;
    v.seq4 = P->mem == PP_MEM_DEVICE ? (const u8 *)P->d[10] : nullptr;  // a device part brings the 4-bit mirror of its seq array
    v.wo = !P->has_wo || n == 0 ? nullptr : (P->mem == PP_MEM_DEVICE ? (const pp_wo_rec *)P->d[11] : P->h_wo.data());  // ... and both kinds the window-order mirror, when the source has one
    pp_mirror_register_(P, v.wo, v.wo ? (size_t)n * sizeof(pp_wo_rec) : 0);  // (one of the library's own: pp_polish_add takes it unchecked)
    v.wo_n_runs = v.wo ? (uint32_t)P->wo_runs.size() : 0;  // ... with its runs (a restriction of an ascending run is one)
    v.wo_run_end = v.wo_n_runs ? P->wo_runs.data() : nullptr;
    if (P->mem == PP_MEM_DEVICE) {
        v.contig = (const u32 *)P->d[0]; v.ref_start = (const u32 *)P->d[1]; v.k = (const u32 *)P->d[2];
        v.seq_off = (const uint64_t *)P->d[3]; v.seq_len = (const u32 *)P->d[4]; v.cig_off = (const uint64_t *)P->d[5];
        v.n_cig = (const u32 *)P->d[6]; v.seq = (const u8 *)P->d[7]; v.cigar = (const u32 *)P->d[8];
        P->orig = (const u32 *)P->d[9];
    } else {
        v.contig = P->h_contig.data(); v.ref_start = P->h_ref_start.data(); v.k = P->h_k.data(); v.seq_off = P->h_seq_off.data();
        v.seq_len = P->h_seq_len.data(); v.cig_off = P->h_cig_off.data(); v.n_cig = P->h_n_cig.data(); v.seq = P->h_seq.data();
        v.cigar = P->h_cigar.data();
        P->orig = P->h_orig.data();
    }
}

// the source's run table is usable: ends ascending, the last one the record count (a view on records [lo, hi) of a larger batch
// brings its own table, counted from lo -- the multi-GPU driver cuts the tokenizer's for its slices)
bool src_runs_ok(const pp_aln_batch *b, u32 wo_base) {
    (void)wo_base;
    if (!b->wo || !b->wo_n_runs || b->wo_n_runs > PP_WO_MAX_RUNS || !b->wo_run_end) return false;
    uint64_t prev = 0;
    for (u32 r = 0; r < b->wo_n_runs; r++) {
        if (b->wo_run_end[r] < prev) return false;
        prev = b->wo_run_end[r];
    }
    return prev == b->n_aln;
}

bool batch_ok(const pp_aln_batch *b) {
    return b && (b->n_aln == 0 || (b->contig && b->ref_start && b->k && b->seq_off && b->seq_len && b->cig_off && b->n_cig &&
                                  b->seq && b->cigar));
}

int split_host(pp_shard_part *P, const HostUnits &U, u32 n_contigs, u32 dest, const pp_aln_batch *B, u32 wo_base) {
    const UnitTable T = U.table(n_contigs);
    const uint64_t n = B->n_aln;
    uint64_t seq_total = 0, cig_total = 0, cnt = 0;
    std::vector<uint8_t> sel((size_t)n);
    for (uint64_t i = 0; i < n; i++) {
        uint64_t span = 0;
        const u32 *cg = B->cigar + B->cig_off[i];
        for (u32 r = 0; r < B->n_cig[i]; r++)
            if (ref_consuming(cg[r] & 15u)) span += cg[r] >> 4;
        if (span == 0) span = 1;
        sel[i] = goes_to(T, B->contig[i], B->ref_start[i], span, dest);
        if (sel[i]) { cnt++; seq_total += seq_room(B->seq_len[i]); cig_total += B->n_cig[i]; }
    }
    P->h_contig.reserve(cnt); P->h_ref_start.reserve(cnt); P->h_k.reserve(cnt); P->h_seq_len.reserve(cnt); P->h_n_cig.reserve(cnt);
    P->h_orig.reserve(cnt); P->h_seq_off.reserve(cnt); P->h_cig_off.reserve(cnt);
    P->h_seq.resize(seq_total + 64); P->h_cigar.resize(cig_total + 1);
    // where the part's SEQ bytes go: in the order of the SOURCE's seq array (a window-grouped batch gives window-grouped
    // parts), which for a batch in file order is the order of the records
    std::vector<uint64_t> picked;
    picked.reserve(cnt);
    bool sorted = true;
    for (uint64_t i = 0; i < n; i++)
        if (sel[i]) {
            if (!picked.empty() && B->seq_off[i] < B->seq_off[picked.back()]) sorted = false;
            picked.push_back(i);
        }
    std::vector<uint64_t> place(cnt);  // per part record: its offset in the part's seq array
    {
        std::vector<uint32_t> ord(cnt);
        for (uint64_t j = 0; j < cnt; j++) ord[j] = (uint32_t)j;
        if (!sorted) std::stable_sort(ord.begin(), ord.end(), [&](uint32_t a, uint32_t b) { return B->seq_off[picked[a]] < B->seq_off[picked[b]]; });
        uint64_t so = 0;
        for (uint64_t j = 0; j < cnt; j++) { place[ord[j]] = so; so += seq_room(B->seq_len[picked[ord[j]]]); }
    }
    uint64_t co = 0;
    for (uint64_t j = 0; j < cnt; j++) {
        const uint64_t i = picked[j], so = place[j];
        P->h_contig.push_back(B->contig[i]); P->h_ref_start.push_back(B->ref_start[i]); P->h_k.push_back(B->k[i]);
        P->h_seq_len.push_back(B->seq_len[i]); P->h_n_cig.push_back(B->n_cig[i]); P->h_orig.push_back((u32)i);
        P->h_seq_off.push_back(so); P->h_cig_off.push_back(co);
        memcpy(P->h_seq.data() + so, B->seq + B->seq_off[i], B->seq_len[i]);
        memcpy(P->h_cigar.data() + co, B->cigar + B->cig_off[i], (size_t)B->n_cig[i] * 4);
        co += B->n_cig[i];  // (h_seq was zero-filled: so are the bytes up to the boundary)
    }
    if (B->wo && cnt) {  // the source's window-order mirror, restricted to the part (see k_split_wo)
        std::vector<uint32_t> new_idx((size_t)n, 0xFFFFFFFFu);
        for (uint64_t j = 0; j < cnt; j++) new_idx[picked[j]] = (uint32_t)j;
        P->h_wo.reserve(cnt);
        bool ok = true;
        const bool runs = src_runs_ok(B, wo_base);
        uint32_t run = 0;
        for (uint64_t j = 0; j < n && ok; j++) {
            while (runs && run < B->wo_n_runs && j == B->wo_run_end[run]) { P->wo_runs.push_back(P->h_wo.size()); run++; }
            pp_wo_rec w = B->wo[j];
            const uint32_t fi = w.file_idx - wo_base;
            if (fi >= n) { ok = false; break; }
            if (new_idx[fi] == 0xFFFFFFFFu) continue;
            w.seq_off = place[new_idx[fi]];
            w.file_idx = new_idx[fi];
            P->h_wo.push_back(w);
        }
        while (runs && run < B->wo_n_runs) { P->wo_runs.push_back(P->h_wo.size()); run++; }
        P->has_wo = ok && P->h_wo.size() == cnt;
        if (!P->has_wo) P->wo_runs.clear();
    }
    set_view(P, cnt, seq_total, cig_total);
    return PP_OK;
}

int split_device(pp_shard_part *P, const HostUnits &U, u32 n_contigs, u32 dest, const pp_aln_batch *B, u32 wo_base) {
    pp_ctx *ctx = P->ctx;
    hipStream_t st = ctx->stream;
    const uint64_t n = B->n_aln;
    if (n == 0) { set_view(P, 0, 0, 0); return PP_OK; }
    PP_HIPCHK(ctx, hipSetDevice(ctx->device));
    int rc;
    // the unit table on the device (a few hundred bytes), and the scratch of this call: the context's own buffers, grow-only
    // (a job is split into world x files parts: a hipMalloc per array and call was most of the split's time)
    pp::DevBuf &t_first = ctx->b_split[0], &t_units = ctx->b_split[1], &flag = ctx->b_split[2], &sel_seq = ctx->b_split[3],
               &sel_cig = ctx->b_split[4], &out_idx = ctx->b_split[5], &seq_scan = ctx->b_split[6], &cig_scan = ctx->b_split[7],
               &sums = ctx->b_split[8], &sums_off = ctx->b_split[9];
    auto done = [&](int r) { (void)hipStreamSynchronize(st); return r; };
    const size_t nu = U.lo.size();
    if ((rc = pp::dev_ensure(ctx, t_first, ((size_t)n_contigs + 1) * 4)) || (rc = pp::dev_ensure(ctx, t_units, (nu ? nu : 1) * 12)) ||
        (rc = pp::dev_ensure(ctx, flag, n * 4)) || (rc = pp::dev_ensure(ctx, sel_seq, n * 4)) || (rc = pp::dev_ensure(ctx, sel_cig, n * 4)) ||
        (rc = pp::dev_ensure(ctx, out_idx, (n + 1) * 4)) || (rc = pp::dev_ensure(ctx, seq_scan, (n + 1) * 8)) ||
        (rc = pp::dev_ensure(ctx, cig_scan, (n + 1) * 8)))
        return done(rc);
    u32 *d_units = (u32 *)t_units.p;
    if (hipMemcpyAsync(t_first.p, U.first.data(), ((size_t)n_contigs + 1) * 4, hipMemcpyHostToDevice, st) != hipSuccess ||
        (nu && (hipMemcpyAsync(d_units, U.lo.data(), nu * 4, hipMemcpyHostToDevice, st) != hipSuccess ||
                hipMemcpyAsync(d_units + nu, U.hi.data(), nu * 4, hipMemcpyHostToDevice, st) != hipSuccess ||
                hipMemcpyAsync(d_units + 2 * nu, U.rank.data(), nu * 4, hipMemcpyHostToDevice, st) != hipSuccess)))
        return done(ctx->fail(PP_ERR_HIP, "pp_shard_split: uploading the plan failed"));
    const UnitTable T{(const u32 *)t_first.p, d_units, d_units + nu, d_units + 2 * nu, n_contigs, U.fallback};
    const unsigned blocks = (unsigned)((n + 255) / 256);
    // slots of the source's seq array (see k_split_flag): [0, n_slots) notes, [n_slots + 1 ...) their scan, one word "bad"
    const u64 n_slots = (B->seq_bytes + (u64)PP_SEQ_ALIGN - 1) / (u64)PP_SEQ_ALIGN;
    const bool by_slots = n_slots > 0 && n_slots < 0xFFFFFFF0ull;  // (the part's bytes / 32 fit 32 bits)
    pp::DevBuf &slots = ctx->b_split[10], &slot_scan = ctx->b_split[11];
    if (by_slots) {
        if ((rc = pp::dev_ensure(ctx, slots, (n_slots + 2) * 4)) || (rc = pp::dev_ensure(ctx, slot_scan, (n_slots + 2) * 4))) return done(rc);
        if (hipMemsetAsync(slots.p, 0, (n_slots + 2) * 4, st) != hipSuccess) return done(ctx->fail(PP_ERR_HIP, "pp_shard_split: memset failed"));
    }
    u32 *d_bad = by_slots ? (u32 *)slots.p + n_slots + 1 : nullptr;
    hipLaunchKernelGGL(k_split_flag, dim3(blocks), dim3(256), 0, st, (u64)n, B->contig, B->ref_start, B->seq_len,
                       (const u64 *)B->cig_off, B->n_cig, B->cigar, T, dest, (u32 *)flag.p, (u32 *)sel_seq.p, (u32 *)sel_cig.p,
                       (const u64 *)B->seq_off, by_slots ? (u32 *)slots.p : (u32 *)nullptr, n_slots, d_bad);
    if ((rc = scan_u32<u32>(ctx, sums, sums_off, (const u32 *)flag.p, (u64)n, (u32 *)out_idx.p)) ||
        (rc = scan_u32<u64>(ctx, sums, sums_off, (const u32 *)sel_seq.p, (u64)n, (u64 *)seq_scan.p)) ||
        (rc = scan_u32<u64>(ctx, sums, sums_off, (const u32 *)sel_cig.p, (u64)n, (u64 *)cig_scan.p)))
        return done(rc);
    u32 cnt = 0, bad = 1;
    u64 seq_total = 0, cig_total = 0;
    if ((rc = fetch(ctx, (const u32 *)out_idx.p + n, &cnt)) || (rc = fetch(ctx, (const u64 *)seq_scan.p + n, &seq_total)) ||
        (rc = fetch(ctx, (const u64 *)cig_scan.p + n, &cig_total)) || (by_slots && (rc = fetch(ctx, (const u32 *)d_bad, &bad))))
        return done(rc);
    if (by_slots && !bad && cnt) {  // the part's SEQ bytes in the order of the source's seq array
        if ((rc = scan_u32<u32>(ctx, sums, sums_off, (const u32 *)slots.p, n_slots, (u32 *)slot_scan.p))) return done(rc);
        hipLaunchKernelGGL(k_split_place, dim3(blocks), dim3(256), 0, st, (u64)n, (const u32 *)flag.p, (const u64 *)B->seq_off,
                           (const u32 *)slot_scan.p, (u64 *)seq_scan.p);
    }
    const size_t esz[12] = {4, 4, 4, 8, 4, 8, 4, 1, 4, 4, 1, sizeof(pp_wo_rec)};
    const uint64_t ecnt[12] = {cnt, cnt, cnt, cnt, cnt, cnt, cnt, seq_total + 64, cig_total, cnt, (seq_total + 1) / 2 + 96, B->wo ? cnt : 0};
    size_t at[13] = {0};
    for (int a = 0; a < 12; a++) at[a + 1] = (at[a] + (size_t)ecnt[a] * esz[a] + 255) / 256 * 256;
    if ((rc = pp::dev_ensure(ctx, P->d_all, at[12]))) return done(rc);
    for (int a = 0; a < 12; a++) P->d[a] = (char *)P->d_all.p + at[a];
    SplitOut O{(u32 *)P->d[0], (u32 *)P->d[1], (u32 *)P->d[2], (u32 *)P->d[4], (u32 *)P->d[6], (u32 *)P->d[8],
               (u32 *)P->d[9], (u64 *)P->d[3], (u64 *)P->d[5], (u8 *)P->d[7]};
    if (cnt) {
        hipLaunchKernelGGL(k_split_meta, dim3(blocks), dim3(256), 0, st, (u64)n, B->contig, B->ref_start, B->k, B->seq_len,
                           B->n_cig, (const u32 *)flag.p, (const u32 *)out_idx.p, (const u64 *)seq_scan.p,
                           (const u64 *)cig_scan.p, O);
        hipLaunchKernelGGL(k_split_seq, dim3((unsigned)((n * 8 + 255) / 256)), dim3(256), 0, st, (u64)n, (const u64 *)B->seq_off,
                           B->seq_len, B->seq, (const u32 *)flag.p, (const u64 *)seq_scan.p, O.seq, (u8 *)P->d[10]);
        hipLaunchKernelGGL(k_split_cigar, dim3(blocks), dim3(256), 0, st, (u64)n, (const u64 *)B->cig_off, B->n_cig, B->cigar,
                           (const u32 *)flag.p, (const u64 *)cig_scan.p, O.cigar);
        if (hipMemsetAsync(O.seq + seq_total, 0, 64, st) != hipSuccess) return done(ctx->fail(PP_ERR_HIP, "pp_shard_split: memset failed"));
        if (B->wo) {  // the part's window-order mirror: the source's, restricted (sel_seq / sel_cig are free again: mflag / mpos)
            u32 *mflag = (u32 *)sel_seq.p, *wbad = (u32 *)sel_cig.p;
            pp::DevBuf &mpos = ctx->b_split[12];
            if ((rc = pp::dev_ensure(ctx, mpos, (n + 1) * 4))) return done(rc);
            if (hipMemsetAsync(wbad, 0, 4, st) != hipSuccess) return done(ctx->fail(PP_ERR_HIP, "pp_shard_split: memset failed"));
            hipLaunchKernelGGL(k_split_wo_flag, dim3(blocks), dim3(256), 0, st, (u64)n, B->wo, wo_base, (const u32 *)flag.p, mflag, wbad);
            if ((rc = scan_u32<u32>(ctx, sums, sums_off, (const u32 *)mflag, (u64)n, (u32 *)mpos.p))) return done(rc);
            u32 wb = 0, mcnt = 0;
            if ((rc = fetch(ctx, (const u32 *)wbad, &wb)) || (rc = fetch(ctx, (const u32 *)mpos.p + n, &mcnt))) return done(rc);
            if (!wb && mcnt == cnt) {
                hipLaunchKernelGGL(k_split_wo, dim3(blocks), dim3(256), 0, st, (u64)n, B->wo, wo_base, (const u32 *)mflag, (const u32 *)mpos.p,
                                   (const u32 *)out_idx.p, (const u64 *)seq_scan.p, (pp_wo_rec *)P->d[11]);
                P->has_wo = true;
                if (src_runs_ok(B, wo_base)) {  // where the source's runs end in the part's mirror: the scan at their ends
                    u32 ends[PP_WO_MAX_RUNS];
                    for (u32 r = 0; r < B->wo_n_runs; r++)
                        if (hipMemcpyAsync(&ends[r], (const u32 *)mpos.p + B->wo_run_end[r], 4, hipMemcpyDeviceToHost, st) != hipSuccess)
                            return done(ctx->fail(PP_ERR_HIP, "pp_shard_split: copy failed"));
                    if (hipStreamSynchronize(st) != hipSuccess) return done(ctx->fail(PP_ERR_HIP, "pp_shard_split: synchronize failed"));
                    for (u32 r = 0; r < B->wo_n_runs; r++) P->wo_runs.push_back(ends[r]);
                }
            }
        }
    }
    if (hipGetLastError() != hipSuccess) return done(ctx->fail(PP_ERR_HIP, "pp_shard_split: a kernel launch failed"));
    set_view(P, cnt, seq_total, cig_total);
    return done(PP_OK);  // synchronises: the part is complete when the call returns
}

}  // namespace

// (internal) wo_idx_base: what the file_idx of batch->wo count from -- a view on records [lo, hi) of a larger batch (the
// multi-GPU driver's (file, slice) pieces: batch->wo = the larger batch's mirror + lo) keeps that batch's numbering
extern "C" int pp_shard_split_view_(pp_ctx *ctx, const pp_shard_plan *plan, uint32_t dest, const pp_aln_batch *batch, int mem,
                                    uint32_t wo_idx_base, pp_shard_part **out);
extern "C" int pp_shard_split(pp_ctx *ctx, const pp_shard_plan *plan, uint32_t dest, const pp_aln_batch *batch, int mem,
                              pp_shard_part **out) {
    return pp_shard_split_view_(ctx, plan, dest, batch, mem, 0, out);
}
extern "C" int pp_shard_split_view_(pp_ctx *ctx, const pp_shard_plan *plan, uint32_t dest, const pp_aln_batch *batch, int mem,
                                    uint32_t wo_idx_base, pp_shard_part **out) {
    if (!out) return PP_ERR_ARG;
    *out = nullptr;
    if (!plan || dest >= plan->world || !batch_ok(batch) || (mem != PP_MEM_HOST && mem != PP_MEM_DEVICE)) return PP_ERR_ARG;
    if (mem == PP_MEM_DEVICE) {
        if (!ctx) return PP_ERR_ARG;
        if (int rdy = pp_ctx_wait(ctx)) return rdy;
    }
    if (batch->n_aln >= 0xFFFFFFFFull) return ctx ? ctx->fail(PP_ERR_LIMIT, "more than 2^32-1 alignments in one batch") : PP_ERR_LIMIT;
    HostUnits U;
    if (!U.build(plan)) return ctx ? ctx->fail(PP_ERR_ARG, "pp_shard_split: the plan's units are not listed contig by contig") : PP_ERR_ARG;
    pp_shard_part *P = new pp_shard_part();
    P->ctx = ctx;
    P->mem = mem;
    const int rc = mem == PP_MEM_DEVICE ? split_device(P, U, plan->n_contigs, dest, batch, wo_idx_base)
                                        : split_host(P, U, plan->n_contigs, dest, batch, wo_idx_base);
    if (rc) { pp_shard_part_free(P); return rc; }
    *out = P;
    return PP_OK;
}

extern "C" int pp_shard_part_mem(const pp_shard_part *part) { return part ? part->mem : PP_MEM_HOST; }

extern "C" void pp_shard_part_batch(const pp_shard_part *part, pp_aln_batch *out, const uint32_t **orig) {
    if (!part) return;
    if (out) *out = part->view;
    if (orig) *orig = part->orig;
}

extern "C" void pp_shard_part_free(pp_shard_part *P) {
    if (!P) return;
    pp_mirror_forget_(P);
    if (P->mem == PP_MEM_DEVICE && P->ctx) {
        (void)hipSetDevice(P->ctx->device);
        (void)hipStreamSynchronize(P->ctx->stream);
        pp::dev_free(P->d_all);
    }
    delete P;
}

// alignment records per contig (the planner's weights), ADDED to aln_per_contig (HOST, n_contigs)
extern "C" int pp_shard_count(pp_ctx *ctx, const pp_aln_batch *batch, int mem, uint32_t n_contigs, uint64_t *aln_per_contig) {
    if (!batch_ok(batch) || !aln_per_contig || n_contigs == 0) return PP_ERR_ARG;
    if (mem == PP_MEM_HOST) {
        for (uint64_t i = 0; i < batch->n_aln; i++)
            if (batch->contig[i] < n_contigs) aln_per_contig[batch->contig[i]]++;
        return PP_OK;
    }
    if (!ctx || mem != PP_MEM_DEVICE) return PP_ERR_ARG;
    if (int rdy = pp_ctx_wait(ctx)) return rdy;
    if (batch->n_aln == 0) return PP_OK;
    PP_HIPCHK(ctx, hipSetDevice(ctx->device));
    pp::DevBuf d_cnt;
    if (int rc = pp::dev_ensure(ctx, d_cnt, (size_t)n_contigs * 8)) return rc;
    std::vector<uint64_t> h(n_contigs);
    int rc = PP_OK;
    if (hipMemsetAsync(d_cnt.p, 0, (size_t)n_contigs * 8, ctx->stream) != hipSuccess) rc = ctx->fail(PP_ERR_HIP, "pp_shard_count: memset failed");
    if (!rc) {
        const unsigned blocks = (unsigned)std::min<uint64_t>(1024, (batch->n_aln + 1023) / 1024);
        hipLaunchKernelGGL(k_contig_hist, dim3(blocks), dim3(1024), 0, ctx->stream, (u64)batch->n_aln, batch->contig, n_contigs,
                           (u64 *)d_cnt.p);
        rc = fetch(ctx, (const uint64_t *)d_cnt.p, h.data(), n_contigs);
    }
    pp::dev_free(d_cnt);
    if (rc) return rc;
    for (uint32_t c = 0; c < n_contigs; c++) aln_per_contig[c] += h[c];
    return PP_OK;
}
