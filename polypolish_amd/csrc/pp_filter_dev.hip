// pp_filter_dev.hip -- the filter's load_alignments (src/filter.rs:91-145) on the device: both SAM texts are
// uploaded, every line gets Alignment::new_quick (src/alignment.rs:102-128) and get_ref_end (:138-149) from
// one lane, QNAMEs and RNAMEs are interned through device hash tables (the reference's
// HashMap<String, Vec<Alignment>> becomes a read number shared by both files plus a per-file group index),
// and the arrays of pp_filter_input stay in HBM for pp_filter_samples / pp_filter_pairs.
//
// The result equals pp_filter_load's (pp_filter_host.cpp) array by array -- read numbers are the ranks of
// the first record of each name, groups are in file order -- except that RNAME ids are arbitrary (only their
// equality matters).  As in pp_tokenize.hip the device only finds the first failing line; the message comes
// from the host parser run on that line.
#include "pp_devtext.h"
#include "pp_host.h"

#include <chrono>
#include <cstring>
#include <string>
#include <vector>

extern "C" int pp_filter_line_error_(const char *line, size_t n, const char *path, uint64_t line_no, char *err, size_t errlen);

namespace {

// ---- per-line quick parse --------------------------------------------------------------------------
struct FqLines {  // one entry per line of the file
    u32 *is_aln, *flag, *ref_start, *name_len, *ref_off, *ref_len;
    u64 *ref_end;
};

// new_quick for every line of the file: one lane per line, the wave's lines staged through LDS (pp_devtext.h)
template <u32 TOK_STAGE>
__global__ __launch_bounds__(64) void k_fq_parse(const u8 *__restrict__ text, u64 size, const u64 *__restrict__ nl_pos,
                                                 u64 n_nl, u64 n_lines, FqLines O, u64 *__restrict__ status) {
    __shared__ __attribute__((aligned(16))) u8 stage[TOK_STAGE + 32];
    const u8 *L;
    u32 n;
    u64 li;
    if (!stage_wave_lines<TOK_STAGE>(text, size, nl_pos, n_nl, n_lines, stage, &L, &n, &li)) return;
    O.is_aln[li] = 0;
    if (n > 0 && L[0] == (u8)'@') return;  // header lines are skipped; an EMPTY line is fatal (filter.rs:126-130)
    u32 cs[11], cl[11], nc = 0, q = 0;
    while (nc < 11) {
        const u32 t = find_tab(L, q, n);
        cs[nc] = q;
        cl[nc] = t - q;
        nc++;
        if (t >= n) break;
        q = t + 1;
    }
    if (nc < 11) { report(status, li); return; }
    u64 flags, pos;
    if (!parse_u(L + cs[1], cl[1], 0xFFFFFFFFull, flags) || !parse_u(L + cs[3], cl[3], ~0ull, pos)) { report(status, li); return; }
    if (flags & 4) return;
    if (pos > 0) pos -= 1;
    if (pos > 0xFFFFFFFFull) { report(status, li); return; }
    // get_ref_end: regex \d+[MIDNSHP=X] over the CIGAR column; text that does not match is skipped
    const u8 *c = L + cs[5];
    const u32 cgl = cl[5];
    u64 end = pos;
    u32 i = 0;
    while (i < cgl) {
        if (c[i] >= (u8)'0' && c[i] <= (u8)'9') {
            u32 j = i;
            while (j < cgl && c[j] >= (u8)'0' && c[j] <= (u8)'9') j++;
            const int op = j < cgl ? op_code(c[j]) : -1;
            if (op >= 0) {
                u64 num;
                if (!parse_u(c + i, j - i, ~0ull, num)) { end = PP_REF_END_UNPARSEABLE; break; }  // only fatal if needed (lazy)
                if (op == PP_OP_M || op == PP_OP_D || op == PP_OP_N || op == PP_OP_EQ || op == PP_OP_X) end += num;
                i = j + 1;
            } else {
                i = j;
            }
        } else {
            i++;
        }
    }
    O.flag[li] = (u32)flags;
    O.ref_start[li] = (u32)pos;
    O.ref_end[li] = end;
    O.name_len[li] = cl[0];
    O.ref_off[li] = cs[2];
    O.ref_len[li] = cl[2];
    O.is_aln[li] = 1;
}

struct NameRef {  // a name somewhere in one of the two device texts
    const u8 *p;
    u32 len, pad;
};

__global__ __launch_bounds__(256) void k_fq_compact(const u8 *__restrict__ text, const u64 *__restrict__ nl_pos, u64 n_lines,
                                                    FqLines O, const u32 *__restrict__ rec_of_line, u64 base,
                                                    u32 *__restrict__ flags, u32 *__restrict__ ref_start,
                                                    u64 *__restrict__ ref_end, NameRef *__restrict__ qname,
                                                    NameRef *__restrict__ rname) {
    const u64 li = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (li >= n_lines || !O.is_aln[li]) return;
    const u32 r = rec_of_line[li];
    const u8 *L = text + (li ? nl_pos[li - 1] + 1 : 0);
    flags[r] = O.flag[li];
    ref_start[r] = O.ref_start[li];
    ref_end[r] = O.ref_end[li];
    qname[base + r] = NameRef{L, O.name_len[li], 0};
    rname[base + r] = NameRef{L + O.ref_off[li], O.ref_len[li], 0};
}

// ---- name interning: open addressing over record indices; the representative of a name is its first record
__device__ __forceinline__ u32 hash_name(const NameRef &a) {
    u32 h = 2166136261u;
    for (u32 i = 0; i < a.len; i++) h = (h ^ a.p[i]) * 16777619u;
    h ^= h >> 15;
    return h * 2246822519u;
}
__device__ __forceinline__ bool same_name(const NameRef &a, const NameRef &b) {
    if (a.len != b.len) return false;
    for (u32 i = 0; i < a.len; i++)
        if (a.p[i] != b.p[i]) return false;
    return true;
}

__global__ __launch_bounds__(256) void k_ht_insert(u64 lo, u64 hi, const NameRef *__restrict__ names, u32 *__restrict__ slots,
                                                   u32 mask) {
    const u64 me = lo + (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (me >= hi) return;
    const NameRef a = names[me];
    u32 i = hash_name(a) & mask;
    for (;;) {
        // look before touching the slot with an atomic: when every record carries the same RNAME (or the mate's
        // QNAME is already there) millions of atomics on one address would serialise in L2
        u32 v = __atomic_load_n(&slots[i], __ATOMIC_RELAXED);
        if (v == 0) v = atomicCAS(&slots[i], 0u, (u32)me + 1u);
        if (v == 0) return;
        if (same_name(names[v - 1], a)) {  // a slot only ever moves to a smaller index of the SAME name
            if ((u32)me + 1u < v) atomicMin(&slots[i], (u32)me + 1u);
            return;
        }
        i = (i + 1) & mask;
    }
}

__global__ __launch_bounds__(256) void k_ht_find(u64 n, const NameRef *__restrict__ names, const u32 *__restrict__ slots, u32 mask,
                                                 u32 *__restrict__ rep, u32 *__restrict__ is_rep) {
    const u64 me = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (me >= n) return;
    const NameRef a = names[me];
    u32 i = hash_name(a) & mask;
    for (;;) {
        const u32 v = slots[i];
        if (same_name(names[v - 1], a)) {
            rep[me] = v - 1;
            if (is_rep) is_rep[me] = (v - 1 == (u32)me);
            return;
        }
        i = (i + 1) & mask;
    }
}

// names of file 2 that file 1 holds as well (for the "alignments from N reads" line of file 2)
__global__ __launch_bounds__(256) void k_mark_shared(u64 n0, u64 n, const u32 *__restrict__ rep, u32 *__restrict__ hit) {
    const u64 i = n0 + (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && rep[i] < n0) hit[rep[i]] = 1;
}

__global__ __launch_bounds__(256) void k_assign(u32 n_aln, u64 base, const u32 *__restrict__ rep, const u32 *__restrict__ id_scan,
                                                const u32 *__restrict__ rep_ref, u32 *__restrict__ read,
                                                u32 *__restrict__ ref_id, u32 *__restrict__ grp_cnt) {
    const u32 r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_aln) return;
    const u32 id = id_scan[rep[base + r]];
    read[r] = id;
    ref_id[r] = rep_ref[base + r];
    atomicAdd(&grp_cnt[id], 1u);
}

__global__ __launch_bounds__(256) void k_grp_scatter(u32 n_aln, const u32 *__restrict__ read, const u32 *__restrict__ grp_off,
                                                     u32 *__restrict__ cursor, u32 *__restrict__ grp_idx) {
    const u32 r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_aln) return;
    const u32 id = read[r];
    grp_idx[grp_off[id] + atomicAdd(&cursor[id], 1u)] = r;
}

// file order inside every group (the scatter's order is whatever the atomics made it)
__global__ __launch_bounds__(256) void k_grp_sort(u32 n_reads, const u32 *__restrict__ grp_off, u32 *__restrict__ grp_idx) {
    const u32 r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    const u32 lo = grp_off[r], hi = grp_off[r + 1];
    for (u32 i = lo + 1; i < hi; i++) {
        const u32 v = grp_idx[i];
        u32 j = i;
        while (j > lo && grp_idx[j - 1] > v) { grp_idx[j] = grp_idx[j - 1]; j--; }
        grp_idx[j] = v;
    }
}

struct DevFile {
    pph::FileText text;  // the host mapping stays: messages and the tagged output are built from it
    pp::DevBuf d_text, d_nl, d_isaln, d_flag, d_start, d_end, d_namelen, d_refoff, d_reflen, d_recofline;
    pp::DevBuf flags, ref_start, ref_end, read, ref_id, grp_off, grp_idx;
    u64 n_nl = 0, n_lines = 0;
    u32 n_aln = 0;
};

}  // namespace

struct pp_filter_dev {
    pp_ctx *ctx;
    DevFile F[2];
    pp::DevBuf d_blk, d_blkoff, d_status, d_sums, d_sumsoff, qname, rname, q_slots, r_slots, rep, rep_ref, is_rep, id_scan, hit,
        hit_scan, cursor;
    u32 n_reads = 0;
    u64 names[2] = {0, 0};
};

extern "C" void pp_filter_dev_free(pp_filter_dev *D) {
    if (!D) return;
    (void)hipStreamSynchronize(D->ctx->stream);
    for (int f = 0; f < 2; f++) {
        DevFile &X = D->F[f];
        pp::DevBuf *all[] = {&X.d_text, &X.d_nl, &X.d_isaln, &X.d_flag, &X.d_start, &X.d_end, &X.d_namelen, &X.d_refoff, &X.d_reflen,
                             &X.d_recofline, &X.flags, &X.ref_start, &X.ref_end, &X.read, &X.ref_id, &X.grp_off, &X.grp_idx};
        for (pp::DevBuf *b : all) pp::dev_free(*b);
    }
    pp::DevBuf *all[] = {&D->d_blk, &D->d_blkoff, &D->d_status, &D->d_sums, &D->d_sumsoff, &D->qname, &D->rname, &D->q_slots,
                         &D->r_slots, &D->rep, &D->rep_ref, &D->is_rep, &D->id_scan, &D->hit, &D->hit_scan, &D->cursor};
    for (pp::DevBuf *b : all) pp::dev_free(*b);
    delete D;
}

// the host text of file f (for the tagged output) and the device view of pp_filter_input
extern "C" const char *pp_filter_dev_text(const pp_filter_dev *D, int f, uint64_t *size) {
    if (size) *size = D->F[f].text.size;
    return D->F[f].text.text;
}

extern "C" void pp_filter_dev_input(const pp_filter_dev *D, pp_filter_input *in) {
    memset(in, 0, sizeof *in);
    in->n_reads = D->n_reads;
    for (int f = 0; f < 2; f++) {
        const DevFile &X = D->F[f];
        pp_filter_file &d = in->file[f];
        d.n_aln = X.n_aln;
        d.ref_id = (const u32 *)X.ref_id.p; d.ref_start = (const u32 *)X.ref_start.p; d.flags = (const u32 *)X.flags.p;
        d.read = (const u32 *)X.read.p; d.grp_off = (const u32 *)X.grp_off.p; d.grp_idx = (const u32 *)X.grp_idx.p;
        d.ref_end = (const uint64_t *)X.ref_end.p;
    }
}

extern "C" int pp_filter_load_device(pp_ctx *ctx, const char *in1, const char *in2, pp_filter_dev **out,
                                     pp_filter_file_counts counts[2]) {
    if (!ctx || !in1 || !in2 || !out) return PP_ERR_ARG;
    *out = nullptr;
    if (int rdy = pp_ctx_wait(ctx)) return rdy;
    pp_filter_file_counts local[2];
    if (!counts) counts = local;
    memset(counts, 0, 2 * sizeof(pp_filter_file_counts));
    hipStream_t st = ctx->stream;
    pp_filter_dev *D = new pp_filter_dev();
    D->ctx = ctx;
    struct Guard {
        pp_filter_dev *D;
        ~Guard() { pp_filter_dev_free(D); }
    } guard{D};
    const char *ins[2] = {in1, in2};
    int rc;
#define ENS(buf, bytes) if ((rc = pp::dev_ensure(ctx, buf, (size_t)(bytes)))) return rc
    ENS(D->d_status, 8);
    u64 *d_status = (u64 *)D->d_status.p;
    const bool timing = getenv("PP_TIMING") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!timing) return;
        (void)hipStreamSynchronize(st);
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[timing]   device load: %-18s %.3f s\n", what, std::chrono::duration<double>(now - t_last).count());
        t_last = now;
    };

    // ---- per file: text, newline index, quick parse; errors of file 1 before anything about file 2 ----
    // (file 2 is parsed before file 1's names are interned, but its errors are held back until then)
    int deferred_rc = PP_OK;
    std::string deferred_msg;
    for (int f = 0; f < 2; f++) {
        DevFile &X = D->F[f];
        auto fail2 = [&](int code, const std::string &msg) {  // file 2: report after file 1 has been accepted
            if (f == 0) return ctx->fail(code, "%s", msg.c_str());
            deferred_rc = code;
            deferred_msg = msg;
            return (int)PP_OK;
        };
        if (!X.text.open_file(ins[f])) {
            if ((rc = fail2(PP_ERR_QUIT, std::string("unable to load alignments from \"") + ins[f] + "\""))) return rc;
            break;
        }
        const u64 size = X.text.size;
        if (size >= (1ull << 40)) return ctx->fail(PP_ERR_LIMIT, "\"%s\" is larger than the 1 TiB this loader indexes", ins[f]);
        const u64 n_blk = (size + NL_BLOCK - 1) / NL_BLOCK, padded = std::max<u64>(1, n_blk) * NL_BLOCK;
        ENS(X.d_text, padded + 64);
        if (size) PP_HIPCHK(ctx, hipMemcpyAsync(X.d_text.p, X.text.text, size, hipMemcpyHostToDevice, st));
        PP_HIPCHK(ctx, hipMemsetAsync((u8 *)X.d_text.p + size, 0, padded + 64 - size, st));
        PP_HIPCHK(ctx, hipMemsetAsync(d_status, 0xFF, 8, st));
        const u8 *d_text = (const u8 *)X.d_text.p;
        {
            u32 not_ascii = 0;
            if ((rc = newline_index(ctx, d_text, size, D->d_blk, D->d_blkoff, X.d_nl, &X.n_nl, &not_ascii))) return rc;
            if (not_ascii)  // the host loader knows which lines are valid UTF-8
                return ctx->fail(PP_ERR_NOT_ASCII, "\"%s\" holds bytes outside ASCII: left to the host loader", ins[f]);
        }
        X.n_lines = X.n_nl + ((size > 0 && X.text.text[size - 1] != '\n') ? 1 : 0);
        if (X.n_lines >= 0x7FFFFFFFull) return ctx->fail(PP_ERR_LIMIT, "\"%s\" has more than 2^31-1 lines", ins[f]);
        if (X.n_lines) {
            const u64 nl = X.n_lines;
            ENS(X.d_isaln, nl * 4); ENS(X.d_flag, nl * 4); ENS(X.d_start, nl * 4); ENS(X.d_end, nl * 8);
            ENS(X.d_namelen, nl * 4); ENS(X.d_refoff, nl * 4); ENS(X.d_reflen, nl * 4); ENS(X.d_recofline, (nl + 1) * 4);
            FqLines O{(u32 *)X.d_isaln.p, (u32 *)X.d_flag.p, (u32 *)X.d_start.p, (u32 *)X.d_namelen.p, (u32 *)X.d_refoff.p,
                      (u32 *)X.d_reflen.p, (u64 *)X.d_end.p};
            {
                const u32 stage_bytes = tok_stage_for(size, nl);
                const dim3 grid((unsigned)((nl + 63) / 64));
                if (stage_bytes == TOK_STAGE_S) hipLaunchKernelGGL(k_fq_parse<TOK_STAGE_S>, grid, dim3(64), 0, st, d_text, size, (const u64 *)X.d_nl.p,
                               X.n_nl, nl, O, d_status);
                else if (stage_bytes == TOK_STAGE_M) hipLaunchKernelGGL(k_fq_parse<TOK_STAGE_M>, grid, dim3(64), 0, st, d_text, size, (const u64 *)X.d_nl.p,
                               X.n_nl, nl, O, d_status);
                else hipLaunchKernelGGL(k_fq_parse<TOK_STAGE_L>, grid, dim3(64), 0, st, d_text, size, (const u64 *)X.d_nl.p,
                               X.n_nl, nl, O, d_status);
            }
            if ((rc = scan_u32<u32>(ctx, D->d_sums, D->d_sumsoff, (const u32 *)X.d_isaln.p, nl, (u32 *)X.d_recofline.p))) return rc;
            if ((rc = fetch(ctx, (const u32 *)X.d_recofline.p + nl, &X.n_aln))) return rc;
        }
        u64 status = ~0ull;
        if ((rc = fetch(ctx, d_status, &status))) return rc;
        if (status != ~0ull) {  // the first failing line: the host parser words the message
            u64 off = 0;
            if (status > 0 && (rc = fetch(ctx, (const u64 *)X.d_nl.p + (status - 1), &off))) return rc;
            if (status > 0) off += 1;
            const char *line = X.text.text + off;
            const char *nl = (const char *)memchr(line, '\n', (size_t)(size - off));
            char err[1400] = "";
            const int code = pp_filter_line_error_(line, nl ? (size_t)(nl - line) : (size_t)(size - off), ins[f], status + 1, err, sizeof err);
            if (code == PP_OK) return ctx->fail(PP_ERR_HIP, "device loader flagged line %llu of \"%s\" but the host parser accepts it",
                                                (unsigned long long)(status + 1), ins[f]);
            X.n_aln = 0;
            if ((rc = fail2(code, err))) return rc;
            break;
        }
        lap(f == 0 ? "file 1 parsed" : "file 2 parsed");
    }
    const u64 n0 = D->F[0].n_aln, n1 = deferred_rc ? 0 : D->F[1].n_aln, N = n0 + n1;
    if (N >= 0xFFFFFFFFull) return ctx->fail(PP_ERR_LIMIT, "more than 2^32-1 alignments in the two files");

    // ---- per-record arrays and the names of both files ----
    ENS(D->qname, std::max<u64>(1, N) * sizeof(NameRef));
    ENS(D->rname, std::max<u64>(1, N) * sizeof(NameRef));
    for (int f = 0; f < 2; f++) {
        DevFile &X = D->F[f];
        const u64 n = f == 0 ? n0 : n1;
        ENS(X.flags, std::max<u64>(1, n) * 4); ENS(X.ref_start, std::max<u64>(1, n) * 4); ENS(X.ref_end, std::max<u64>(1, n) * 8);
        ENS(X.read, std::max<u64>(1, n) * 4); ENS(X.ref_id, std::max<u64>(1, n) * 4); ENS(X.grp_idx, std::max<u64>(1, n) * 4);
        if (!n) continue;
        FqLines O{(u32 *)X.d_isaln.p, (u32 *)X.d_flag.p, (u32 *)X.d_start.p, (u32 *)X.d_namelen.p, (u32 *)X.d_refoff.p,
                  (u32 *)X.d_reflen.p, (u64 *)X.d_end.p};
        hipLaunchKernelGGL(k_fq_compact, dim3((unsigned)((X.n_lines + 255) / 256)), dim3(256), 0, st, (const u8 *)X.d_text.p,
                           (const u64 *)X.d_nl.p, X.n_lines, O, (const u32 *)X.d_recofline.p, f == 0 ? 0ull : n0, (u32 *)X.flags.p,
                           (u32 *)X.ref_start.p, (u64 *)X.ref_end.p, (NameRef *)D->qname.p, (NameRef *)D->rname.p);
    }
    // ---- intern QNAMEs (file 1, then file 2) and RNAMEs ----
    u32 cap = 1024;
    while (cap < 2 * N + 2) cap <<= 1;
    ENS(D->q_slots, (u64)cap * 4); ENS(D->r_slots, (u64)cap * 4);
    ENS(D->rep, std::max<u64>(1, N) * 4); ENS(D->rep_ref, std::max<u64>(1, N) * 4); ENS(D->is_rep, std::max<u64>(1, N) * 4);
    ENS(D->id_scan, (N + 1) * 4);
    PP_HIPCHK(ctx, hipMemsetAsync(D->q_slots.p, 0, (u64)cap * 4, st));
    PP_HIPCHK(ctx, hipMemsetAsync(D->r_slots.p, 0, (u64)cap * 4, st));
    if (N) {
        const unsigned gb = (unsigned)((N + 255) / 256);
        hipLaunchKernelGGL(k_ht_insert, dim3(gb), dim3(256), 0, st, 0ull, N, (const NameRef *)D->qname.p, (u32 *)D->q_slots.p, cap - 1);
        hipLaunchKernelGGL(k_ht_insert, dim3(gb), dim3(256), 0, st, 0ull, N, (const NameRef *)D->rname.p, (u32 *)D->r_slots.p, cap - 1);
        hipLaunchKernelGGL(k_ht_find, dim3(gb), dim3(256), 0, st, N, (const NameRef *)D->qname.p, (const u32 *)D->q_slots.p, cap - 1,
                           (u32 *)D->rep.p, (u32 *)D->is_rep.p);
        hipLaunchKernelGGL(k_ht_find, dim3(gb), dim3(256), 0, st, N, (const NameRef *)D->rname.p, (const u32 *)D->r_slots.p, cap - 1,
                           (u32 *)D->rep_ref.p, (u32 *)nullptr);
        if ((rc = scan_u32<u32>(ctx, D->d_sums, D->d_sumsoff, (const u32 *)D->is_rep.p, N, (u32 *)D->id_scan.p))) return rc;
    } else {
        PP_HIPCHK(ctx, hipMemsetAsync(D->id_scan.p, 0, 4, st));
    }
    u32 names_f1 = 0, n_reads = 0, shared = 0;
    if ((rc = fetch(ctx, (const u32 *)D->id_scan.p + n0, &names_f1))) return rc;
    if ((rc = fetch(ctx, (const u32 *)D->id_scan.p + N, &n_reads))) return rc;
    if (n1 && n0) {
        ENS(D->hit, n0 * 4); ENS(D->hit_scan, (n0 + 1) * 4);
        PP_HIPCHK(ctx, hipMemsetAsync(D->hit.p, 0, n0 * 4, st));
        hipLaunchKernelGGL(k_mark_shared, dim3((unsigned)((n1 + 255) / 256)), dim3(256), 0, st, n0, N, (const u32 *)D->rep.p, (u32 *)D->hit.p);
        if ((rc = scan_u32<u32>(ctx, D->d_sums, D->d_sumsoff, (const u32 *)D->hit.p, n0, (u32 *)D->hit_scan.p))) return rc;
        if ((rc = fetch(ctx, (const u32 *)D->hit_scan.p + n0, &shared))) return rc;
    }
    D->n_reads = n_reads;
    D->names[0] = names_f1;
    D->names[1] = (u64)(n_reads - names_f1) + shared;
    lap("names interned");
    // what the reference reports while loading, in its order
    counts[0].alignments = n0; counts[0].reads = D->names[0]; counts[0].loaded = 1;
    if (n0 == 0) return ctx->fail(PP_ERR_QUIT, "no alignments found in \"%s\"", in1);
    if (deferred_rc) return ctx->fail(deferred_rc, "%s", deferred_msg.c_str());
    counts[1].alignments = n1; counts[1].reads = D->names[1]; counts[1].loaded = 1;

    // ---- read numbers, RNAME ids, per-file groups in file order ----
    ENS(D->cursor, std::max<u64>(1, (u64)n_reads) * 4);
    for (int f = 0; f < 2; f++) {
        DevFile &X = D->F[f];
        const u32 n = X.n_aln;
        ENS(X.grp_off, ((u64)n_reads + 1) * 4);
        PP_HIPCHK(ctx, hipMemsetAsync(D->cursor.p, 0, std::max<u64>(1, (u64)n_reads) * 4, st));
        if (n)
            hipLaunchKernelGGL(k_assign, dim3((n + 255) / 256), dim3(256), 0, st, n, f == 0 ? 0ull : n0, (const u32 *)D->rep.p,
                               (const u32 *)D->id_scan.p, (const u32 *)D->rep_ref.p, (u32 *)X.read.p, (u32 *)X.ref_id.p, (u32 *)D->cursor.p);
        if ((rc = scan_u32<u32>(ctx, D->d_sums, D->d_sumsoff, (const u32 *)D->cursor.p, (u64)n_reads, (u32 *)X.grp_off.p))) return rc;
        PP_HIPCHK(ctx, hipMemsetAsync(D->cursor.p, 0, std::max<u64>(1, (u64)n_reads) * 4, st));
        if (n) {
            hipLaunchKernelGGL(k_grp_scatter, dim3((n + 255) / 256), dim3(256), 0, st, n, (const u32 *)X.read.p, (const u32 *)X.grp_off.p,
                               (u32 *)D->cursor.p, (u32 *)X.grp_idx.p);
            hipLaunchKernelGGL(k_grp_sort, dim3((n_reads + 255) / 256), dim3(256), 0, st, n_reads, (const u32 *)X.grp_off.p, (u32 *)X.grp_idx.p);
        }
    }
    PP_HIPCHK(ctx, hipStreamSynchronize(st));
    PP_HIPCHK(ctx, hipGetLastError());
    lap("groups built");
#undef ENS
    guard.D = nullptr;
    *out = D;
    return PP_OK;
}
