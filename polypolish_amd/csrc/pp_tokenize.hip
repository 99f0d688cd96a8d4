// pp_tokenize.hip -- SAM text -> pp_aln_batch on the device: the polish ingest without the
// host parse.  The raw text is uploaded once; newline index, field split, number/CIGAR/tag validation,
// contig lookup, read grouping, the gates of process_one_read, the 1/k share, "*" fill (+ reverse
// complement), upper-casing and CIGAR packing all run as kernels, and the batch stays in HBM for
// pp_polish_add(PP_MEM_DEVICE).
//
// Semantics are those of pp_ingest.cpp (the host ingest), which mirrors
//   Alignment::new               src/alignment.rs:49-98   (+ get_expanded_cigar :325-346)
//   add_to_pileup (grouping)     src/alignment.rs:225-272
//   process_one_read             src/alignment.rs:275-322
// and the result is bit-identical to it (tests compare the two batches array by array).  The device only
// has to DECIDE whether a line or a read group is in error and which event comes first in the reference's
// streaming order; the message itself is produced by running the host ingest on the offending lines.
//   event key = 2 * line          a line that fails to parse
//               2 * line + 1      a read group that fails when it is flushed, i.e. when `line` -- the first
//                                 record of the next group -- has been parsed (EOF: line = number of lines)
#include "pp_devtext.h"
#include "pp_host.h"

#include <algorithm>
#include <chrono>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

extern "C" int pp_ingest_text_(pp_ingest *I, const char *path, const char *text, size_t size, uint64_t line_base,
                               pp_sam_counts *counts, char *err, size_t errlen);

namespace {

enum { K_SKIP = 0, K_UNALIGNED = 1, K_ALIGNED = 2 };

struct LineRec {          // one parsed line (offsets relative to the start of the line)
    u32 flag, contig, ref_start, nm;
    u32 name_len, cig_off, cig_len, n_runs;
    u32 seq_off, seq_len, bits, pad;  // bits: first op | last op << 4 | pass_qc << 8 | start beyond u32 << 9
};

// ---- per-line parse (Alignment::new) ---------------------------------------------------------------
__device__ __forceinline__ u32 fnv1a(const u8 *s, u32 n) {
    u32 h = 2166136261u;
    for (u32 i = 0; i < n; i++) h = (h ^ s[i]) * 16777619u;
    return h;
}

struct ContigTable {      // RNAME -> contig index: open addressing over the assembly's names
    const u32 *slots;     // contig index + 1, 0 = empty
    const u32 *name_off;  // n_contigs + 1
    const u8 *names;
    u32 mask;
};

__device__ __forceinline__ int lookup_contig(const ContigTable &T, const u8 *s, u32 n) {
    u32 i = fnv1a(s, n) & T.mask;
    for (;;) {
        const u32 v = T.slots[i];
        if (!v) return -1;
        const u32 o = T.name_off[v - 1], l = T.name_off[v] - o;
        if (l == n) {
            u32 j = 0;
            while (j < n && T.names[o + j] == s[j]) j++;
            if (j == n) return (int)(v - 1);
        }
        i = (i + 1) & T.mask;
    }
}

__device__ __forceinline__ void tok_parse_line(const u8 *L, u32 n, u64 li, const ContigTable &T, LineRec *__restrict__ rec,
                                               u32 *__restrict__ is_aln, u64 *__restrict__ status);

// Alignment::new for every line of the file: one lane per line, the wave's lines staged through LDS (pp_devtext.h)
template <u32 TOK_STAGE>
__global__ __launch_bounds__(64) void k_tok_parse(const u8 *__restrict__ text, u64 size, const u64 *__restrict__ nl_pos,
                                                  u64 n_nl, u64 n_lines, ContigTable T, LineRec *__restrict__ rec,
                                                  u32 *__restrict__ is_aln, u64 *__restrict__ status) {
    __shared__ __attribute__((aligned(16))) u8 stage[TOK_STAGE + 32];
    const u8 *L;
    u32 n;
    u64 li;
    if (stage_wave_lines<TOK_STAGE>(text, size, nl_pos, n_nl, n_lines, stage, &L, &n, &li)) tok_parse_line(L, n, li, T, rec, is_aln, status);
}

__device__ __forceinline__ void tok_parse_line(const u8 *L, u32 n, u64 li, const ContigTable &T, LineRec *__restrict__ rec,
                                               u32 *__restrict__ is_aln, u64 *__restrict__ status) {
    is_aln[li] = 0;
    if (n == 0 || L[0] == (u8)'@') return;  // alignment.rs:241
    u32 cs[11], cl[11], nc = 0, q = 0;
    while (nc < 11) {
        const u32 t = find_tab(L, q, n);
        cs[nc] = q;
        cl[nc] = t - q;
        nc++;
        if (t >= n) { q = n + 1; break; }
        q = t + 1;
    }
    if (nc < 11) { report(status, 2 * li); return; }  // too few columns
    u64 flags, pos;
    if (!parse_u(L + cs[1], cl[1], 0xFFFFFFFFull, flags) || !parse_u(L + cs[3], cl[3], ~0ull, pos)) { report(status, 2 * li); return; }
    if (pos > 0) pos -= 1;
    u32 nm = 0xFFFFFFFFu, pass_qc = 1;
    u32 tg = q;  // start of the tag fields; q == n means one empty field after a trailing tab; n + 1: none
    while (tg <= n) {
        const u32 t = find_tab(L, tg, n);
        const u32 tl = t - tg;
        const u8 *f = L + tg;
        if (tl >= 5 && f[0] == 'N' && f[1] == 'M' && f[2] == ':' && f[3] == 'i' && f[4] == ':') {
            u64 v;
            if (!parse_u(f + 5, tl - 5, 0xFFFFFFFFull, v)) { report(status, 2 * li); return; }
            nm = (u32)v;
        }
        if (tl == 9 && (f[0] | 32) == 'z' && (f[1] | 32) == 'p' && f[2] == ':' && (f[3] | 32) == 'z' && f[4] == ':' &&
            (f[5] | 32) == 'f' && (f[6] | 32) == 'a' && (f[7] | 32) == 'i' && (f[8] | 32) == 'l')
            pass_qc = 0;
        if (t >= n) break;
        tg = t + 1;
    }
    if (nm == 0xFFFFFFFFu && (flags & 4) == 0) { report(status, 2 * li); return; }  // missing NM tag
    // get_expanded_cigar (alignment.rs:325-346), kept as runs: zero-length runs vanish, long runs split
    const u8 *cg = L + cs[5];
    const u32 cgl = cl[5];
    u32 n_runs = 0, first_op = 15, last_op = 15;
    if (!(cgl == 1 && cg[0] == (u8)'*')) {
        u32 i = 0;
        while (i < cgl) {
            u32 j = i;
            while (j < cgl && cg[j] >= (u8)'0' && cg[j] <= (u8)'9') j++;
            const int op = j < cgl ? op_code(cg[j]) : -1;
            u64 num;
            if (j == i || op < 0 || !parse_u(cg + i, j - i, 0xFFFFFFFFull, num)) { report(status, 2 * li); return; }
            while (num > 0) {
                const u32 piece = num > 0x0FFFFFFFull ? 0x0FFFFFFFu : (u32)num;
                if (!n_runs) first_op = (u32)op;
                last_op = (u32)op;
                n_runs++;
                num -= piece;
            }
            i = j + 1;
        }
    }
    if (flags & 4) return;  // parsed, then skipped (alignment.rs:250)
    LineRec r;
    r.flag = (u32)flags;
    r.contig = (u32)lookup_contig(T, L + cs[2], cl[2]);
    r.ref_start = (u32)pos;
    r.nm = nm;
    r.name_len = cl[0];
    r.cig_off = cs[5];
    r.cig_len = cgl;
    r.n_runs = n_runs;
    r.seq_off = cs[9];
    r.seq_len = cl[9];
    r.bits = first_op | (last_op << 4) | (pass_qc << 8) | ((pos > 0xFFFFFFFEull ? 1u : 0u) << 9);
    r.pad = 0;
    rec[li] = r;
    is_aln[li] = 1;
}

__global__ __launch_bounds__(256) void k_tok_compact(u64 n_lines, const u32 *__restrict__ is_aln,
                                                     const u32 *__restrict__ rec_of_line, u32 *__restrict__ rec_line) {
    const u64 li = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (li < n_lines && is_aln[li]) rec_line[rec_of_line[li]] = (u32)li;
}

__device__ __forceinline__ u64 line_start(const u64 *nl_pos, u32 li) { return li ? nl_pos[li - 1] + 1 : 0; }

// does record r open a new read group?  (alignment.rs:255: it joins when the previous QNAME is empty or equal)
__global__ __launch_bounds__(256) void k_tok_group_start(const u8 *__restrict__ text, const u64 *__restrict__ nl_pos,
                                                         const LineRec *__restrict__ rec, const u32 *__restrict__ rec_line,
                                                         u32 n_aln, u32 *__restrict__ is_start) {
    const u32 r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_aln) return;
    u32 start = 1;
    if (r > 0) {
        const u32 lp = rec_line[r - 1], lc = rec_line[r];
        const u32 np = rec[lp].name_len, ncur = rec[lc].name_len;
        if (np == 0) start = 0;
        else if (np == ncur) {
            // eight bytes at a time (unaligned loads; the text buffer is padded, and a QNAME is followed by ten more
            // columns): names of neighbouring records tend to differ in their last characters only
            const u8 *a = text + line_start(nl_pos, lp), *b = text + line_start(nl_pos, lc);
            bool same = true;
            for (u32 j = 0; j < np && same; j += 8) {
                u64 x, y;
                __builtin_memcpy(&x, a + j, 8);
                __builtin_memcpy(&y, b + j, 8);
                const u32 rem = np - j;
                const u64 mask = rem < 8u ? (1ull << (8u * rem)) - 1ull : ~0ull;
                same = ((x ^ y) & mask) == 0;
            }
            if (same) start = 0;
        }
    }
    is_start[r] = start;
}

__global__ __launch_bounds__(256) void k_tok_group_first(u32 n_aln, const u32 *__restrict__ is_start,
                                                         const u32 *__restrict__ grp_of_rec, u32 n_groups,
                                                         u32 *__restrict__ group_first) {
    const u32 r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n_aln && is_start[r]) group_first[grp_of_rec[r]] = r;
    if (r == 0) group_first[n_groups] = n_aln;
}

constexpr u32 WIN_NONE = 0xFFFFFFFFu;  // win_of[] of a record that is not good
// process_one_read (alignment.rs:275-322), one lane per read group
__global__ __launch_bounds__(256) void k_tok_group(const u8 *__restrict__ text, const u64 *__restrict__ nl_pos,
                                                   const LineRec *__restrict__ rec, const u32 *__restrict__ rec_line,
                                                   const u32 *__restrict__ group_first, u32 n_groups, u64 n_lines,
                                                   u32 max_errors, int careful, const u8 *__restrict__ pass,
                                                   u32 *__restrict__ good,
                                                   u32 *__restrict__ kk, u32 *__restrict__ src_rec,
                                                   u32 *__restrict__ g_seq_len, u32 *__restrict__ g_ncig,
                                                   const u64 *__restrict__ ctg_off, u32 n_win, u32 *__restrict__ win_of,
                                                   u64 *__restrict__ status) {
    const u32 g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_groups) return;
    const u32 r0 = group_first[g], r1 = group_first[g + 1];
    const u64 key = 2ull * (g + 1 < n_groups ? (u64)rec_line[r1] : n_lines) + 1ull;
    for (u32 r = r0; r < r1; r++) { good[r] = 0; kk[r] = 0; src_rec[r] = r; g_seq_len[r] = 0; g_ncig[r] = 0; win_of[r] = WIN_NONE; }
    if (careful && r1 - r0 > 1) return;
    u32 src = 0xFFFFFFFFu;
    for (u32 r = r0; r < r1; r++) {
        const LineRec &a = rec[rec_line[r]];
        const bool star = a.seq_len == 1 && text[line_start(nl_pos, rec_line[r]) + a.seq_off] == (u8)'*';
        if (!star) { src = r; break; }
    }
    if (src == 0xFFFFFFFFu) { report(status, key); return; }  // no alignment of the read contains sequence
    u32 n_good = 0;
    for (u32 r = r0; r < r1; r++) {
        const LineRec &a = rec[rec_line[r]];
        if (a.n_runs == 0) { report(status, key); return; }  // empty expanded CIGAR: the reference panics
        const u32 f = a.bits & 15u, l = (a.bits >> 4) & 15u;
        const bool ends_ok = (f == PP_OP_M || f == PP_OP_EQ) && (l == PP_OP_M || l == PP_OP_EQ);
        // pass: the filter's verdict for this aligned record, as if "ZP:Z:fail" were on its line
        if (ends_ok && a.nm <= max_errors && ((a.bits >> 8) & 1u) && (!pass || pass[r])) { good[r] = 1; n_good++; }
    }
    const LineRec &s = rec[rec_line[src]];
    for (u32 r = r0; r < r1; r++) {
        if (!good[r]) continue;
        const LineRec &a = rec[rec_line[r]];
        if ((int)a.contig < 0 || ((a.bits >> 9) & 1u)) { report(status, key); return; }  // not in assembly / start beyond u32
        const bool star = a.seq_len == 1 && text[line_start(nl_pos, rec_line[r]) + a.seq_off] == (u8)'*';
        kk[r] = n_good;
        src_rec[r] = star ? src : r;
        g_seq_len[r] = star ? s.seq_len : a.seq_len;
        g_ncig[r] = a.n_runs;
        // the 2048-position window of the assembly the record starts in (the window-grouped SEQ layout places it there)
        const u64 w = (ctg_off[a.contig] + a.ref_start) / (u64)pp::TILE;
        win_of[r] = w < n_win ? (u32)w : n_win - 1u;
    }
}

// bytes of the seq array a good record takes: its SEQ bytes up to the next PP_SEQ_ALIGN boundary (include/polypolish_hip.h)
__device__ __host__ __forceinline__ u32 seq_room(u32 n) { return (n + (u32)PP_SEQ_ALIGN - 1u) & ~((u32)PP_SEQ_ALIGN - 1u); }
__global__ __launch_bounds__(256) void k_tok_room(u32 n_aln, const u32 *__restrict__ g_seq_len, u32 *__restrict__ g_room) {
    const u32 r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n_aln) g_room[r] = seq_room(g_seq_len[r]);
}

// ---- output: the structure of arrays of pp_aln_batch ----------------------------------------------
struct OutArrays {
    u32 *contig, *ref_start, *k, *seq_len, *n_cig, *cigar;
    u64 *seq_off, *cig_off;
    u8 *seq;
};

__global__ __launch_bounds__(256) void k_tok_meta(const u8 *__restrict__ text, const u64 *__restrict__ nl_pos,
                                                  const LineRec *__restrict__ rec, const u32 *__restrict__ rec_line,
                                                  u32 n_aln, const u32 *__restrict__ good, const u32 *__restrict__ kk,
                                                  const u32 *__restrict__ g_seq_len, const u32 *__restrict__ out_idx,
                                                  const u64 *__restrict__ seq_scan, const u64 *__restrict__ cig_scan,
                                                  const u32 *__restrict__ slot, pp_wo_rec *__restrict__ wo,
                                                  OutArrays O, u64 out_base, u64 seq_base, u64 cig_base) {
    const u32 r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_aln || !good[r]) return;
    const LineRec &a = rec[rec_line[r]];
    const u64 o = out_base + out_idx[r];
    O.contig[o] = a.contig;
    O.ref_start[o] = a.ref_start;
    O.k[o] = kk[r];
    O.seq_len[o] = g_seq_len[r];
    O.n_cig[o] = a.n_runs;
    O.seq_off[o] = seq_base + seq_scan[r];
    O.cig_off[o] = cig_base + cig_scan[r];
    // the CIGAR as packed runs (validated by k_tok_parse; zero-length runs vanish, long runs split: get_expanded_cigar) --
    // the one pass over the record's CIGAR text also yields the mirror's op0
    const u8 *cg = text + line_start(nl_pos, rec_line[r]) + a.cig_off;
    u32 *runs = O.cigar + cig_base + cig_scan[r];
    u32 i = 0, nw = 0, first_run = PP_WO_MULTI_RUN;
    while (i < a.cig_len) {
        u64 num = 0;
        while (cg[i] >= (u8)'0' && cg[i] <= (u8)'9') num = num * 10 + (u64)(cg[i++] - (u8)'0');
        const u32 op = (u32)op_code(cg[i++]);
        while (num > 0) {
            const u32 piece = num > 0x0FFFFFFFull ? 0x0FFFFFFFu : (u32)num;
            const u32 run = (piece << 4) | op;
            if (nw == 0) first_run = run;
            runs[nw++] = run;
            num -= piece;
        }
    }
    if (wo) {
        // the record once more, at its place in window order (pp_aln_batch.wo): one 32-byte store.  op0 = its only CIGAR
        // run, or the marker for a record of several runs
        pp_wo_rec w;
        w.contig = a.contig; w.ref_start = a.ref_start; w.k = kk[r]; w.seq_len = g_seq_len[r];
        w.seq_off = seq_base + seq_scan[r];
        w.op0 = a.n_runs == 1u ? first_run : PP_WO_MULTI_RUN; w.file_idx = (u32)o;
        wo[out_base + slot[r]] = w;
    }
}

__device__ __forceinline__ u8 comp_upper(u8 c) {  // misc.rs:170-182 on the upper-cased base
    switch (c) {
    case 'A': return 'T'; case 'T': return 'A'; case 'G': return 'C'; case 'C': return 'G';
    case 'R': return 'Y'; case 'Y': return 'R'; case 'S': return 'S'; case 'W': return 'W';
    case 'K': return 'M'; case 'M': return 'K'; case 'B': return 'V'; case 'V': return 'B';
    case 'D': return 'H'; case 'H': return 'D'; case 'N': return 'N'; case '.': return '.';
    case '-': return '-'; case '?': return '?'; default: return 'N';
    }
}

// ---- SEQ bytes and their 4-bit mirror (pp_aln_batch.seq4), one pass ------------------------------------------------
// Eight lanes per good record, one 16-byte chunk of its room (SEQ bytes up to the next PP_SEQ_ALIGN boundary) per lane and
// trip: upper-cased (alignment.rs:94), zeros past the read, and -- the bytes being in registers anyway -- their 4-bit codes
// packed into the mirror (base i of the ARRAY in bits 4*(i&1).. of seq4[i >> 1]: a room starts on a multiple of 32 bytes,
// so a chunk's eight mirror bytes are its own).  A "*" record takes the group's sequence, reverse-complemented when the
// strands differ (alignment.rs:161-167, 288-296).  Round 3 packed the mirror in a kernel of its own that read the seq
// array back (0.4 ms per 2.4 GB of text); here it costs the stores.
__device__ __forceinline__ u32 seq4_code(u32 c) {
    const u32 t = (c >> 1) & 3u;  // A->0 C->1 T->2 G->3: the counter rows
    const u32 expect = (0x47544341u >> (t * 8u)) & 0xFFu;
    return c == expect ? t : (c == (u32)'N' ? (u32)PP_SEQ4_N : (c == (u32)'-' ? (u32)PP_SEQ4_DASH : (u32)PP_SEQ4_OTHER));
}
__device__ __forceinline__ uint2 pack4_16(const u32 w[4]) {  // 16 bytes -> 16 nibbles
    u32 o[2];
#pragma unroll
    for (int q = 0; q < 2; q++) {
        u32 v = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) v |= seq4_code((w[2 * q + (j >> 2)] >> (8 * (j & 3))) & 0xFFu) << (4 * j);
        o[q] = v;
    }
    return make_uint2(o[0], o[1]);
}
__global__ __launch_bounds__(256) void k_tok_seq(const u8 *__restrict__ text, const u64 *__restrict__ nl_pos,
                                                 const LineRec *__restrict__ rec, const u32 *__restrict__ rec_line,
                                                 u32 n_aln, const u32 *__restrict__ good, const u32 *__restrict__ src_rec,
                                                 const u64 *__restrict__ seq_scan, u8 *__restrict__ seq, u8 *__restrict__ seq4,
                                                 u64 seq_base) {
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x, r = t >> 3, s = t & 7u;
    if (r >= n_aln || !good[r]) return;
    const u32 sr = src_rec[r];
    const LineRec &a = rec[rec_line[r]], &b = rec[rec_line[sr]];
    const u8 *in = text + line_start(nl_pos, rec_line[sr]) + b.seq_off;
    const u32 n = b.seq_len, room = seq_room(n);
    const u64 at = seq_base + seq_scan[r];  // a multiple of PP_SEQ_ALIGN
    u8 *out = seq + at;
    u8 *out4 = seq4 ? seq4 + (at >> 1) : nullptr;
    const bool rc = sr != r && ((a.flag & 16u) == 0) != ((b.flag & 16u) == 0);
    for (u32 i = 16u * s; i < room; i += 128u) {
        u32 w[4] = {0, 0, 0, 0};
        if (!rc) {
            if (i < n) {
                // 16 bytes per lane and trip (gfx950 global accesses need no alignment; the text buffer is padded, the bytes
                // past the read -- the next column -- are cut off), upper-cased four at a time: bit 7 of every byte in
                // 'a'..'z' (ASCII only), shifted down to the 0x20 that is taken off
                uint4 v;
                __builtin_memcpy(&v, in + i, 16);
                w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
                const u32 live = n - i;  // bytes of this chunk that belong to the read (>= 1)
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const u32 have = live > 4u * q ? min(4u, live - 4u * q) : 0u;
                    w[q] &= have == 4u ? 0xFFFFFFFFu : ((1u << (8u * have)) - 1u);
                    const u32 x = w[q] & 0x7F7F7F7Fu;
                    const u32 m = (x + 0x1F1F1F1Fu) & ~(x + 0x05050505u) & ~w[q] & 0x80808080u;
                    w[q] -= m >> 2;
                }
            }
        } else {  // a "*" record on the other strand (rare): byte-wise, reversed and complemented
            for (u32 j = 0; j < 16u && i + j < n; j++) {
                u8 c = in[n - 1 - (i + j)];
                if (c >= (u8)'a' && c <= (u8)'z') c = (u8)(c - 32);
                w[j >> 2] |= (u32)comp_upper(c) << (8u * (j & 3u));
            }
        }
        const uint4 v = make_uint4(w[0], w[1], w[2], w[3]);
        __builtin_memcpy(out + i, &v, 16);
        if (out4) {
            const uint2 p4 = pack4_16(w);
            __builtin_memcpy(out4 + (i >> 1), &p4, 8);
        }
    }
}

// ---- window-grouped SEQ layout (pp_dev_ingest_set_seq_layout; the default) -------------------------------------------
// Where a good record's SEQ bytes go when the reads of one 2048-position window are to be adjacent in the seq array.  The
// record's window comes from k_tok_group (win_of[]: it has the record in hand); what is left is a multisplit of the rooms
// into the windows, done like the polish pipeline's own (pp_k_bucket.h), without global atomics: a workgroup adds its
// records' rooms up per window in LDS and leaves its row of a blocks x windows matrix (k_tok_win_hist), one wave per
// window scans its column (k_tok_win_cols: the window's bytes, and every block's offset inside the window's region), the
// windows' bytes are scanned into the regions' starts, and the workgroups place their records with LDS cursors that
// start at their offsets (k_tok_win_place).  Round 3 went to one global counter per window from every workgroup -- a
// device-scope atomic on one address is served every ~0.6 us on this chip (the XCDs share no L2) -- and read the 48-byte
// parse record of every line twice: 0.3 ms per SAM file of 3.3 M records, now 0.05.
// Inside a window the order is that of the workgroups (file order) and, within one, whatever the LDS atomics make it:
// seq_off goes with the record, nothing depends on it.
// Two things are placed at once: a record's SEQ bytes in its window's region of the seq array, and the record itself in the
// batch's window-order mirror (pp_aln_batch.wo: the records of a window adjacent) -- one 64-bit counter per window, the
// records in bits 40.. and the bytes below (a file holds < 2^40 SEQ bytes and a window < 2^24 records: MAX_BUCKET).
constexpr u32 WIN_LDS_MAX = 8192;   // windows in LDS (16.7 Mbp); beyond: the global-atomic kernels below
constexpr u64 WIN_BYTES_MASK = (1ull << 40) - 1ull;
__global__ __launch_bounds__(1024) void k_tok_win_hist(const u32 *__restrict__ win_of, const u32 *__restrict__ g_seq_len, u32 n_aln,
                                                       u32 per_block, u32 n_win, u64 *__restrict__ mat) {
    __shared__ u64 hist[WIN_LDS_MAX];
    for (u32 w = threadIdx.x; w < n_win; w += 1024u) hist[w] = 0;
    __syncthreads();
    const u32 lo = blockIdx.x * per_block, hi = min(n_aln, lo + per_block);
    for (u32 r = lo + threadIdx.x; r < hi; r += 1024u) {
        const u32 w = win_of[r];
        if (w != WIN_NONE) atomicAdd(&hist[w], (1ull << 40) | (u64)seq_room(g_seq_len[r]));
    }
    __syncthreads();
    for (u32 w = threadIdx.x; w < n_win; w += 1024u) mat[(u64)blockIdx.x * n_win + w] = hist[w];
}
// one wave per window: exclusive scan of its column over the blocks (in place), the totals to wbytes / wcount
__global__ __launch_bounds__(256) void k_tok_win_cols(u32 n_win, u32 n_blocks, u64 *__restrict__ mat, u32 *__restrict__ wbytes,
                                                      u32 *__restrict__ wcount) {
    const u32 w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63u;
    if (w >= n_win) return;
    u64 carry = 0;
    for (u32 b0 = 0; b0 < n_blocks; b0 += 64u) {
        const u32 b = b0 + lane;
        const u64 v = b < n_blocks ? mat[(u64)b * n_win + w] : 0ull;
        u64 inc = v;
        for (int o = 1; o < 64; o <<= 1) {
            const u64 t = (u64)__shfl_up((long long)inc, o, 64);
            if ((int)lane >= o) inc += t;
        }
        if (b < n_blocks) mat[(u64)b * n_win + w] = carry + inc - v;
        carry += (u64)__shfl((long long)inc, 63, 64);
    }
    if (lane == 0) { wbytes[w] = (u32)(carry & WIN_BYTES_MASK); wcount[w] = (u32)(carry >> 40); }
}
__global__ __launch_bounds__(1024) void k_tok_win_place(const u32 *__restrict__ win_of, const u32 *__restrict__ g_seq_len, u32 n_aln,
                                                        u32 per_block, u32 n_win, const u64 *__restrict__ mat,
                                                        const u64 *__restrict__ wbase, const u32 *__restrict__ wcbase,
                                                        u64 *__restrict__ seq_pos, u32 *__restrict__ slot) {
    __shared__ u64 cur[WIN_LDS_MAX];
    for (u32 w = threadIdx.x; w < n_win; w += 1024u) cur[w] = mat[(u64)blockIdx.x * n_win + w];
    __syncthreads();
    const u32 lo = blockIdx.x * per_block, hi = min(n_aln, lo + per_block);
    for (u32 r = lo + threadIdx.x; r < hi; r += 1024u) {
        const u32 w = win_of[r];
        if (w == WIN_NONE) continue;
        const u64 old = atomicAdd(&cur[w], (1ull << 40) | (u64)seq_room(g_seq_len[r]));
        seq_pos[r] = wbase[w] + (old & WIN_BYTES_MASK);
        slot[r] = wcbase[w] + (u32)(old >> 40);
    }
}
// more windows than LDS holds (from 16.7 Mbp on): global counters per window -- there are enough of them then
__global__ __launch_bounds__(256) void k_tok_win_bytes_g(const u32 *__restrict__ win_of, const u32 *__restrict__ g_seq_len, u32 n_aln,
                                                         u32 *__restrict__ wbytes, u32 *__restrict__ wcount) {
    const u32 r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_aln) return;
    const u32 w = win_of[r];
    if (w != WIN_NONE) { atomicAdd(&wbytes[w], seq_room(g_seq_len[r])); atomicAdd(&wcount[w], 1u); }
}
// (one 64-bit counter per window, records in bits 40.. and bytes below, as in k_tok_win_place: a record's room and its entry of
// the mirror get their places from ONE atomic, so the rooms of a window follow each other in the order of its entries -- with
// reads of one length a whole file's rooms then lie at one pitch along the mirror, which k_tile's first loads count on)
__global__ __launch_bounds__(256) void k_tok_win_place_g(const u32 *__restrict__ win_of, const u32 *__restrict__ g_seq_len, u32 n_aln,
                                                         const u64 *__restrict__ wbase, const u32 *__restrict__ wcbase, u64 *__restrict__ wcur,
                                                         u64 *__restrict__ seq_pos, u32 *__restrict__ slot) {
    const u32 r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_aln) return;
    const u32 w = win_of[r];
    if (w == WIN_NONE) return;
    const u64 old = atomicAdd((unsigned long long *)&wcur[w], (1ull << 40) | (u64)seq_room(g_seq_len[r]));
    seq_pos[r] = wbase[w] + (old & WIN_BYTES_MASK);
    slot[r] = wcbase[w] + (u32)(old >> 40);
}

}  // namespace

// =================================================================================================
struct pp_dev_ingest {
    pp_ctx *ctx;
    const pp_assembly *asmb;
    u32 max_errors;
    int careful;
    // contig table
    pp::DevBuf t_slots, t_off, t_names, t_ctgoff;
    u32 t_mask = 0;
    int seq_layout = PP_SEQ_WINDOW_GROUPED;  // the default since round 4 (PP_SEQ_LAYOUT=file: in the order of the records)
    pp::DevBuf d_wbytes, d_wbase, d_wcur, d_seqpos, d_win, d_wcount, d_wcbase, d_slot;
    pp::DevBuf o_wo;         // the batch's window-order mirror (pp_aln_batch.wo)
    bool wo_mirror = true;   // PP_WO=0: none
    // per-file scratch
    pp::DevBuf d_text, d_blk, d_blkoff, d_nl, d_rec, d_isaln, d_recofline, d_recline, d_isstart, d_grpofrec, d_gfirst,
        d_good, d_k, d_src, d_gseq, d_groom, d_gcig, d_outidx, d_seqscan, d_cigscan, d_status, d_sums, d_sumsoff, d_pass;
    // output (grows over the files)
    pp::DevBuf o_contig, o_ref_start, o_k, o_seq_len, o_n_cig, o_cigar, o_seq_off, o_cig_off, o_seq;
    pp::DevBuf o_seq4;   // the 4-bit mirror of o_seq (pp_aln_batch.seq4)
    u64 expect_total = 0;  // pp_dev_ingest_expect: text bytes of all the files to come (sizes the arrays once)
    int seq4 = 1;        // the mirror goes with every batch (PP_SEQ4=0: none)
    bool mirror() const { return seq4 != 0; }
    u64 n_out = 0, seq_bytes = 0, n_cig_total = 0;
    std::vector<uint64_t> wo_run_end;  // the mirror's runs -- one per file (or slice of a file) -- as pp_aln_batch.wo_run_end
    // The text of the NEXT file goes up while the current one is tokenized (pp_dev_ingest_prefetch_): two text buffers, an
    // upload stream of its own, and a helper thread per upload (a copy out of pageable memory keeps its caller busy).
    pp::DevBuf d_text2;
    hipStream_t up_stream = nullptr;
    int text_sel = 0;        // the buffer the last tokenized file's text went to (0: d_text, 1: d_text2)
    hipEvent_t up_gate = nullptr;  // recorded behind the current file's own upload: the next file's copy waits for it
    bool gate_set = false;
    std::string next_path;   // the file the driver announced as the next one (pp_dev_ingest_prefetch_)
    uint64_t next_bytes = 0;
    struct Prefetch {
        std::string path;
        pph::FileText *F = nullptr;
        std::thread th;
        int sel = 0;
        bool ok = false;     // the text is in the device buffer (set by the thread)
        double wait_s = 0, copy_s = 0;
    } *pf = nullptr;
};

// the mapping of a file that has been tokenized goes away in the background: unmapping 15 GB of populated pages takes 0.14 s
static void reap_mapping(pph::FileText *F) {
    if (!F) return;
    if (pph::process_leaving_soon()) return;  // the CLI: left to the exit (see pp_host.h)
    std::thread([F] { delete F; }).detach();
}
static pp::DevBuf &text_buf(pp_dev_ingest *D, int sel) { return sel ? D->d_text2 : D->d_text; }

namespace {

}  // namespace

extern "C" int pp_dev_ingest_create(pp_ctx *ctx, const pp_assembly *a, uint32_t max_errors, int careful,
                                    pp_dev_ingest **out) {
    if (!ctx || !a || !out) return PP_ERR_ARG;
    *out = nullptr;
    if (int rdy = pp_ctx_wait(ctx)) return rdy;
    PP_HIPCHK(ctx, hipSetDevice(ctx->device));  // (the device is a per-thread setting: the multi-GPU driver calls from worker threads)
    pp_dev_ingest *D = new pp_dev_ingest();
    D->ctx = ctx;
    D->asmb = a;
    D->max_errors = max_errors;
    if (const char *e = getenv("PP_SEQ4")) D->seq4 = atoi(e) != 0;  // PP_SEQ4=0: no 4-bit mirror
    if (const char *e = getenv("PP_WO")) D->wo_mirror = atoi(e) != 0;  // PP_WO=0: no window-order mirror of the records
    D->careful = careful != 0;
    // RNAME table
    const u32 nc = pp_assembly_n_contigs(a);
    u32 cap = 16;
    while (cap < 2 * nc + 2) cap <<= 1;
    std::vector<u32> slots(cap, 0), off(nc + 1, 0);
    std::string names;
    for (u32 c = 0; c < nc; c++) {
        const char *nm = pp_assembly_name(a, c);
        const u32 n = (u32)strlen(nm);
        u32 h = 2166136261u;
        for (u32 i = 0; i < n; i++) h = (h ^ (u8)nm[i]) * 16777619u;
        u32 i = h & (cap - 1);
        while (slots[i]) i = (i + 1) & (cap - 1);
        slots[i] = c + 1;
        off[c] = (u32)names.size();
        names.append(nm, n);
    }
    off[nc] = (u32)names.size();
    D->t_mask = cap - 1;
    int rc = pp::dev_ensure(ctx, D->t_slots, cap * 4);
    if (!rc) rc = pp::dev_ensure(ctx, D->t_off, (nc + 1) * 4);
    if (!rc) rc = pp::dev_ensure(ctx, D->t_names, names.size() + 16);
    if (!rc) rc = pp::dev_ensure(ctx, D->t_ctgoff, ((size_t)nc + 1) * 8);
    if (!rc && hipMemcpy(D->t_ctgoff.p, pp_assembly_offsets(a), ((size_t)nc + 1) * 8, hipMemcpyHostToDevice) != hipSuccess)
        rc = ctx->fail(PP_ERR_HIP, "uploading the contig offsets failed");
    if (const char *e = getenv("PP_SEQ_LAYOUT")) D->seq_layout = !strcmp(e, "file") ? PP_SEQ_FILE_ORDER : PP_SEQ_WINDOW_GROUPED;
    if (!rc && (hipMemcpy(D->t_slots.p, slots.data(), cap * 4, hipMemcpyHostToDevice) != hipSuccess ||
                hipMemcpy(D->t_off.p, off.data(), (nc + 1) * 4, hipMemcpyHostToDevice) != hipSuccess ||
                (names.size() && hipMemcpy(D->t_names.p, names.data(), names.size(), hipMemcpyHostToDevice) != hipSuccess)))
        rc = ctx->fail(PP_ERR_HIP, "uploading the contig table failed");
    if (rc) { delete D; return rc; }
    *out = D;
    return PP_OK;
}

extern "C" void pp_dev_ingest_free(pp_dev_ingest *D) {
    if (!D) return;
    pp_mirror_forget_(D);
    if (D->pf) {
        if (D->pf->th.joinable()) D->pf->th.join();
        reap_mapping(D->pf->F);
        delete D->pf;
        D->pf = nullptr;
    }
    (void)hipSetDevice(D->ctx->device);
    if (D->up_stream) { (void)hipStreamSynchronize(D->up_stream); (void)hipStreamDestroy(D->up_stream); }
    if (D->up_gate) (void)hipEventDestroy(D->up_gate);
    pp::dev_free(D->d_text2);
    (void)hipStreamSynchronize(D->ctx->stream);
    pp::DevBuf *all[] = {&D->t_ctgoff, &D->d_wbytes, &D->d_wbase, &D->d_wcur, &D->d_seqpos, &D->d_win, &D->d_wcount, &D->d_wcbase, &D->d_slot, &D->o_wo, &D->t_slots, &D->t_off, &D->t_names, &D->d_text, &D->d_blk, &D->d_blkoff, &D->d_nl, &D->d_rec,
                         &D->d_isaln, &D->d_recofline, &D->d_recline, &D->d_isstart, &D->d_grpofrec, &D->d_gfirst, &D->d_good,
                         &D->d_k, &D->d_src, &D->d_gseq, &D->d_groom, &D->d_gcig, &D->d_outidx, &D->d_seqscan, &D->d_cigscan, &D->d_status, &D->d_sums, &D->d_sumsoff, &D->d_pass,
                         &D->o_contig, &D->o_ref_start, &D->o_k, &D->o_seq_len, &D->o_n_cig, &D->o_cigar, &D->o_seq_off,
                         &D->o_cig_off, &D->o_seq, &D->o_seq4};
    for (pp::DevBuf *b : all) pp::dev_free(*b);
    delete D;
}

extern "C" int pp_dev_ingest_set_seq_layout(pp_dev_ingest *D, int layout) {
    if (!D || (layout != PP_SEQ_FILE_ORDER && layout != PP_SEQ_WINDOW_GROUPED)) return PP_ERR_ARG;
    if (D->n_out && layout != D->seq_layout) return PP_ERR_ARG;  // (the 4-bit mirror goes with the layout: before the first file)
    D->seq_layout = layout;
    return PP_OK;
}

extern "C" int pp_dev_ingest_expect(pp_dev_ingest *D, uint64_t total_text_bytes) {
    if (!D) return PP_ERR_ARG;
    D->expect_total = total_text_bytes;
    return PP_OK;
}

extern "C" void pp_dev_ingest_batch(const pp_dev_ingest *D, pp_aln_batch *out) {
    out->n_aln = D->n_out;
    out->contig = (const u32 *)D->o_contig.p;
    out->ref_start = (const u32 *)D->o_ref_start.p;
    out->k = (const u32 *)D->o_k.p;
    out->seq_off = (const uint64_t *)D->o_seq_off.p;
    out->seq_len = (const u32 *)D->o_seq_len.p;
    out->cig_off = (const uint64_t *)D->o_cig_off.p;
    out->n_cig = (const u32 *)D->o_n_cig.p;
    out->seq = (const u8 *)D->o_seq.p;
    out->seq_bytes = D->seq_bytes;
    out->seq4 = D->mirror() && D->seq_bytes ? (const u8 *)D->o_seq4.p : nullptr;
    out->wo = D->wo_mirror && D->n_out ? (const pp_wo_rec *)D->o_wo.p : nullptr;
    pp_mirror_register_(D, out->wo, out->wo ? (size_t)D->n_out * sizeof(pp_wo_rec) : 0);  // (one of the library's own: pp_polish_add takes it unchecked)
    const bool runs = out->wo && !D->wo_run_end.empty() && D->wo_run_end.back() == D->n_out;
    out->wo_n_runs = runs ? (uint32_t)D->wo_run_end.size() : 0;  // (HOST memory, whatever the batch's)
    out->wo_run_end = runs ? D->wo_run_end.data() : nullptr;
    out->cigar = (const u32 *)D->o_cigar.p;
    out->n_cig_total = D->n_cig_total;
}

// The reference's message for the first event of the file: run the host ingest over the lines involved.
static int describe_error(pp_dev_ingest *D, const char *path, const char *text, size_t size, u64 key, u64 n_lines,
                          u64 n_nl, const std::vector<u32> &rec_line, const std::vector<u32> &group_first) {
    pp_ctx *ctx = D->ctx;
    auto line_off = [&](u64 li, u64 *off) -> int {  // byte offset of the start of line li (li <= n_lines)
        if (li == 0) { *off = 0; return PP_OK; }
        if (li > n_nl) { *off = size; return PP_OK; }
        u64 p;
        if (int rc = fetch(ctx, (const u64 *)D->d_nl.p + (li - 1), &p)) return rc;
        *off = p + 1;
        return PP_OK;
    };
    u64 first_line, end_line;  // the slice [first_line, end_line) reproduces the event
    if ((key & 1) == 0) {
        first_line = key / 2;
        end_line = first_line + 1;
    } else {
        const u64 flush_line = key / 2;  // the failing group ends just before this line
        // its first record: the last group start whose line is < flush_line
        size_t lo = 0, hi = group_first.size() - 1;  // group_first[n_groups] = n_aln
        while (hi - lo > 1) {
            const size_t mid = (lo + hi) / 2;
            if (group_first[mid] < rec_line.size() && rec_line[group_first[mid]] < flush_line) lo = mid; else hi = mid;
        }
        first_line = rec_line[group_first[lo]];
        end_line = flush_line;
    }
    u64 a = 0, b = size;
    if (int rc = line_off(first_line, &a)) return rc;
    if (int rc = line_off(std::min(end_line, n_lines), &b)) return rc;
    if (end_line >= n_lines) b = size;
    pp_ingest *H = nullptr;
    if (int rc = pp_ingest_create(D->asmb, D->max_errors, D->careful, &H)) return rc;
    char err[1024] = "";
    pp_sam_counts c;
    const int rc = pp_ingest_text_(H, path, text + a, (size_t)(b - a), first_line, &c, err, sizeof err);
    pp_ingest_free(H);
    if (rc == PP_OK) return ctx->fail(PP_ERR_HIP, "device tokenizer flagged \"%s\" (event %llu) but the host ingest accepts it", path,
                                      (unsigned long long)key);
    return ctx->fail(rc, "%s", err);
}

// room for the text of the largest file of the job, once (internal: the driver knows the sizes; a text buffer that grows
// from file to file is fresh device memory every time, touched for the first time by the upload)
extern "C" int pp_dev_ingest_reserve_text_(pp_dev_ingest *D, uint64_t bytes) {
    if (!D) return PP_ERR_ARG;
    const u64 n_blk = (bytes + NL_BLOCK - 1) / NL_BLOCK, padded = std::max<u64>(1, n_blk) * NL_BLOCK;
    return pp::dev_ensure(D->ctx, D->d_text, (size_t)(padded + 64));
}

// Start the upload of `path` -- the file pp_dev_ingest_sam* will be given NEXT -- into the text buffer the current file does
// not use, on the upload stream: it then overlaps the tokenizer kernels of the current file (and whatever the caller does
// in between).  Internal (the file drivers call it); a prefetch that cannot be made is simply not there, the ingest then
// uploads as usual.  second_buffer_bytes: room to make for the second text buffer (the largest file of the job).
extern "C" void pp_dev_ingest_prefetch_(pp_dev_ingest *D, const char *path, uint64_t second_buffer_bytes) {
    if (!D || !path) return;
    D->next_path = path;              // (started by the ingest of the current file, once ITS text has gone up: two uploads
    D->next_bytes = second_buffer_bytes;  //  at a time would only share the link)
}
static void start_prefetch(pp_dev_ingest *D) {
    if (D->next_path.empty() || D->pf) return;
    const std::string path_s = D->next_path;
    const char *path = path_s.c_str();
    const uint64_t second_buffer_bytes = D->next_bytes;
    D->next_path.clear();
    pp_ctx *ctx = D->ctx;
    if (hipSetDevice(ctx->device) != hipSuccess) return;
    if (!D->up_stream && hipStreamCreateWithFlags(&D->up_stream, hipStreamNonBlocking) != hipSuccess) { D->up_stream = nullptr; return; }
    if (D->gate_set && hipStreamWaitEvent(D->up_stream, D->up_gate, 0) != hipSuccess) return;  // behind the current file's upload
    D->gate_set = false;
    const int sel = 1 - D->text_sel;  // the file that is tokenized meanwhile uses (or used) buffer text_sel
    {
        const u64 n_blk = (second_buffer_bytes + NL_BLOCK - 1) / NL_BLOCK, padded = std::max<u64>(1, n_blk) * NL_BLOCK;
        pp::DevBuf &b = text_buf(D, sel);
        if (b.cap < padded + 64) {  // (made here, on the caller's thread: the helper only copies)
            if (b.p) return;        // in use or too small: no prefetch rather than a reallocation under a running kernel
            if (hipMalloc(&b.p, (size_t)(padded + 64)) != hipSuccess) { b.p = nullptr; (void)hipGetLastError(); return; }
            b.cap = (size_t)(padded + 64);
        }
    }
    auto *P = new pp_dev_ingest::Prefetch();
    P->path = path;
    P->sel = sel;
    P->F = new pph::FileText();
    D->pf = P;
    const int device = ctx->device;
    hipStream_t us = D->up_stream;
    pp::DevBuf *buf = &text_buf(D, sel);
    P->th = std::thread([P, device, us, buf] {
        const auto t0 = std::chrono::steady_clock::now();
        if (!P->F->open_file(P->path.c_str())) return;  // (waits for the pre-faulting of the mapping)
        const auto t1 = std::chrono::steady_clock::now();
        P->wait_s = std::chrono::duration<double>(t1 - t0).count();
        const u64 size = P->F->size, n_blk = (size + NL_BLOCK - 1) / NL_BLOCK, padded = std::max<u64>(1, n_blk) * NL_BLOCK;
        if (buf->cap < padded + 64 || hipSetDevice(device) != hipSuccess) return;
        if (size && hipMemcpyAsync(buf->p, P->F->text, size, hipMemcpyHostToDevice, us) != hipSuccess) return;
        if (hipMemsetAsync((u8 *)buf->p + size, 0, padded + 64 - size, us) != hipSuccess) return;
        if (hipStreamSynchronize(us) != hipSuccess) return;
        P->copy_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
        P->ok = true;
    });
}

extern "C" int pp_dev_ingest_sam(pp_dev_ingest *D, const char *path, pp_sam_counts *counts) {
    return pp_dev_ingest_sam_filtered(D, path, nullptr, 0, counts);
}

static int ingest_text(pp_dev_ingest *D, const char *path, const char *text, u64 size, bool slice, const uint8_t *pass,
                       uint64_t n_pass, pp_sam_counts *counts, const u8 *uploaded = nullptr);

extern "C" int pp_dev_ingest_sam_filtered(pp_dev_ingest *D, const char *path, const uint8_t *pass, uint64_t n_pass,
                                          pp_sam_counts *counts) {
    if (!D || !path) return PP_ERR_ARG;
    pp_ctx *ctx = D->ctx;
    pp_sam_counts c{0, 0, 0};
    if (counts) *counts = c;
    const bool timing = getenv("PP_TIMING") != nullptr;
    pph::FileText *F = nullptr;
    const u8 *uploaded = nullptr;  // the text is on the device already (prefetched while the file before was tokenized)
    const auto t_open = std::chrono::steady_clock::now();
    if (D->pf && D->pf->path == path) {  // (a prefetch for a LATER file stays pending: it is for the file after this one)
        pp_dev_ingest::Prefetch *P = D->pf;
        D->pf = nullptr;
        if (P->th.joinable()) P->th.join();
        if (P->ok) {
            F = P->F;
            uploaded = (const u8 *)text_buf(D, P->sel).p;
            D->text_sel = P->sel;
            if (timing)
                fprintf(stderr, "[timing]   tokenizer: %-20s %.4f s  (%.2f GB: mapping %.4f s + copy %.4f s on the upload stream, %.4f s of it still to wait for)\n",
                        "text prefetched", P->wait_s + P->copy_s, 1e-9 * (double)F->size, P->wait_s, P->copy_s,
                        std::chrono::duration<double>(std::chrono::steady_clock::now() - t_open).count());
        } else {
            if (timing) fprintf(stderr, "[timing]   tokenizer: the prefetch of this file did not come about: uploading it now\n");
            reap_mapping(P->F);
        }
        delete P;
    }
    if (!F) {
        F = new pph::FileText();
        if (!F->open_file(path)) { delete F; return ctx->fail(PP_ERR_QUIT, "unable to load alignments from \"%s\"", path); }
        if (timing)
            fprintf(stderr, "[timing]   tokenizer: %-20s %.4f s  (%.2f GB)\n", "mapping ready", std::chrono::duration<double>(std::chrono::steady_clock::now() - t_open).count(), 1e-9 * (double)F->size);
    }
    const int rc = ingest_text(D, path, F->text, F->size, false, pass, n_pass, counts, uploaded);
    reap_mapping(F);
    return rc;
}

// One SLICE of a SAM file -- a byte range that starts and ends on read-group boundaries (src/alignment.rs:255-263: a
// group is a run of adjacent aligned lines with one QNAME), cut by the multi-GPU driver so that every GPU uploads and
// tokenizes its own part of the text.  The records are appended to the batch like a file's.  A slice without aligned
// records is fine (the driver looks at the file's total); any defect in the text only says so (PP_ERR_QUIT): line
// numbers and the order of two defects are a whole-file matter, the driver then takes the file through the host parsers.
extern "C" int pp_dev_ingest_slice_(pp_dev_ingest *D, const char *path, const char *text, uint64_t size, pp_sam_counts *counts) {
    if (!D || !path || (size && !text)) return PP_ERR_ARG;
    pp_sam_counts c{0, 0, 0};
    if (counts) *counts = c;
    if (size == 0) return PP_OK;
    return ingest_text(D, path, text, size, true, nullptr, 0, counts);
}

static int ingest_text(pp_dev_ingest *D, const char *path, const char *text, u64 size, bool slice, const uint8_t *pass,
                       uint64_t n_pass, pp_sam_counts *counts, const u8 *uploaded) {
    pp_ctx *ctx = D->ctx;
    hipStream_t st = ctx->stream;
    pp_sam_counts c{0, 0, 0};
    struct { const char *text; } F{text};
    if (size >= (1ull << 40)) return ctx->fail(PP_ERR_LIMIT, "\"%s\" is larger than the 1 TiB this tokenizer indexes", path);
    PP_HIPCHK(ctx, hipSetDevice(ctx->device));
    int rc;
    const bool timing = getenv("PP_TIMING") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!timing) return;
        (void)hipStreamSynchronize(st);
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[timing]   tokenizer: %-20s %.4f s\n", what, std::chrono::duration<double>(now - t_last).count());
        t_last = now;
    };
#define ENS(buf, bytes) if ((rc = pp::dev_ensure(ctx, D->buf, (size_t)(bytes)))) return rc
    // ---- text + newline index ----
    const u64 n_blk = (size + NL_BLOCK - 1) / NL_BLOCK, padded = std::max<u64>(1, n_blk) * NL_BLOCK;
    ENS(d_status, 8);
    const u8 *d_text = uploaded;
    if (!uploaded) {  // (else: it went up on the upload stream while the file before was tokenized, pp_dev_ingest_prefetch_)
        pp::DevBuf &tb = text_buf(D, D->text_sel);
        if ((rc = pp::dev_ensure(ctx, tb, (size_t)(padded + 64)))) return rc;
        if (size) PP_HIPCHK(ctx, hipMemcpyAsync(tb.p, F.text, size, hipMemcpyHostToDevice, st));
        PP_HIPCHK(ctx, hipMemsetAsync((u8 *)tb.p + size, 0, padded + 64 - size, st));
        d_text = (const u8 *)tb.p;
        if (!D->up_gate && hipEventCreateWithFlags(&D->up_gate, hipEventDisableTiming) != hipSuccess) D->up_gate = nullptr;
        if (D->up_gate && hipEventRecord(D->up_gate, st) == hipSuccess) D->gate_set = true;
    }
    PP_HIPCHK(ctx, hipMemsetAsync(D->d_status.p, 0xFF, 8, st));
    u64 *d_status = (u64 *)D->d_status.p;
    if (!uploaded) lap("text uploaded");
    if (!slice) start_prefetch(D);  // this file's text is on its way (a copy out of pageable memory returns when it is staged): the next file's may follow
    u64 n_nl = 0;
    {
        u32 not_ascii = 0;
        if ((rc = newline_index(ctx, d_text, size, D->d_blk, D->d_blkoff, D->d_nl, &n_nl, &not_ascii))) return rc;
        if (not_ascii)  // the host ingest knows which lines are valid UTF-8 (the driver reruns with it, as for any defect)
            return ctx->fail(PP_ERR_NOT_ASCII, "\"%s\" holds bytes outside ASCII: left to the host ingest", path);
    }
    lap("newline index");
    const u64 n_lines = n_nl + ((size > 0 && F.text[size - 1] != '\n') ? 1 : 0);
    if (n_lines >= 0x7FFFFFFFull) return ctx->fail(PP_ERR_LIMIT, "\"%s\" has more than 2^31-1 lines", path);
    // ---- per-line parse ----
    u32 n_aln = 0;
    if (n_lines) {
        ENS(d_rec, n_lines * sizeof(LineRec));
        ENS(d_isaln, n_lines * 4);
        ENS(d_recofline, (n_lines + 1) * 4);
        ContigTable T{(const u32 *)D->t_slots.p, (const u32 *)D->t_off.p, (const u8 *)D->t_names.p, D->t_mask};
        {
            const u32 stage_bytes = tok_stage_for(size, n_lines);
            const dim3 grid((unsigned)((n_lines + 63) / 64));
            if (stage_bytes == TOK_STAGE_S) hipLaunchKernelGGL(k_tok_parse<TOK_STAGE_S>, grid, dim3(64), 0, st, d_text, size,
                           (const u64 *)D->d_nl.p, n_nl, n_lines, T, (LineRec *)D->d_rec.p, (u32 *)D->d_isaln.p, d_status);
            else if (stage_bytes == TOK_STAGE_M) hipLaunchKernelGGL(k_tok_parse<TOK_STAGE_M>, grid, dim3(64), 0, st, d_text, size,
                           (const u64 *)D->d_nl.p, n_nl, n_lines, T, (LineRec *)D->d_rec.p, (u32 *)D->d_isaln.p, d_status);
            else hipLaunchKernelGGL(k_tok_parse<TOK_STAGE_L>, grid, dim3(64), 0, st, d_text, size,
                           (const u64 *)D->d_nl.p, n_nl, n_lines, T, (LineRec *)D->d_rec.p, (u32 *)D->d_isaln.p, d_status);
        }
        if ((rc = scan_u32<u32>(ctx, D->d_sums, D->d_sumsoff, (const u32 *)D->d_isaln.p, n_lines, (u32 *)D->d_recofline.p))) return rc;
        if ((rc = fetch(ctx, (const u32 *)D->d_recofline.p + n_lines, &n_aln))) return rc;
    }
    lap("lines parsed");
    // ---- read groups and gates ----
    u32 n_groups = 0;
    const u8 *d_pass = nullptr;
    if (pass && n_aln) {
        if (n_pass != n_aln)  // cannot be a parse error's doing: those are reported first, below
            pass = nullptr;
        else {
            ENS(d_pass, n_aln);
            PP_HIPCHK(ctx, hipMemcpyAsync(D->d_pass.p, pass, n_aln, hipMemcpyHostToDevice, st));
            d_pass = (const u8 *)D->d_pass.p;
        }
    }
    const bool pass_mismatch = n_pass && !d_pass && n_aln;
    const bool window_layout = D->seq_layout == PP_SEQ_WINDOW_GROUPED;
    const u64 G_asm = pp_assembly_offsets(D->asmb)[pp_assembly_n_contigs(D->asmb)];
    const u32 n_win = (u32)std::max<u64>(1, (G_asm + pp::TILE - 1) / pp::TILE);
    const u64 *seq_place = nullptr, *seq_total_at = nullptr;
    if (n_aln) {
        ENS(d_recline, (u64)n_aln * 4);
        ENS(d_isstart, (u64)n_aln * 4);
        ENS(d_grpofrec, ((u64)n_aln + 1) * 4);
        hipLaunchKernelGGL(k_tok_compact, dim3((unsigned)((n_lines + 255) / 256)), dim3(256), 0, st, n_lines,
                           (const u32 *)D->d_isaln.p, (const u32 *)D->d_recofline.p, (u32 *)D->d_recline.p);
        hipLaunchKernelGGL(k_tok_group_start, dim3((n_aln + 255) / 256), dim3(256), 0, st, d_text, (const u64 *)D->d_nl.p,
                           (const LineRec *)D->d_rec.p, (const u32 *)D->d_recline.p, n_aln, (u32 *)D->d_isstart.p);
        if ((rc = scan_u32<u32>(ctx, D->d_sums, D->d_sumsoff, (const u32 *)D->d_isstart.p, (u64)n_aln, (u32 *)D->d_grpofrec.p))) return rc;
        if ((rc = fetch(ctx, (const u32 *)D->d_grpofrec.p + n_aln, &n_groups))) return rc;
        ENS(d_gfirst, ((u64)n_groups + 1) * 4);
        ENS(d_good, (u64)n_aln * 4); ENS(d_k, (u64)n_aln * 4); ENS(d_src, (u64)n_aln * 4);
        ENS(d_gseq, (u64)n_aln * 4); ENS(d_groom, (u64)n_aln * 4); ENS(d_gcig, (u64)n_aln * 4); ENS(d_win, (u64)n_aln * 4);
        ENS(d_outidx, ((u64)n_aln + 1) * 4); ENS(d_seqscan, ((u64)n_aln + 1) * 8); ENS(d_cigscan, ((u64)n_aln + 1) * 8);
        hipLaunchKernelGGL(k_tok_group_first, dim3((n_aln + 255) / 256), dim3(256), 0, st, n_aln, (const u32 *)D->d_isstart.p,
                           (const u32 *)D->d_grpofrec.p, n_groups, (u32 *)D->d_gfirst.p);
        hipLaunchKernelGGL(k_tok_group, dim3((n_groups + 255) / 256), dim3(256), 0, st, d_text, (const u64 *)D->d_nl.p,
                           (const LineRec *)D->d_rec.p, (const u32 *)D->d_recline.p, (const u32 *)D->d_gfirst.p, n_groups,
                           n_lines, D->max_errors, D->careful, d_pass, (u32 *)D->d_good.p, (u32 *)D->d_k.p, (u32 *)D->d_src.p,
                           (u32 *)D->d_gseq.p, (u32 *)D->d_gcig.p, (const u64 *)D->t_ctgoff.p, n_win, (u32 *)D->d_win.p, d_status);
        if ((rc = scan_u32<u32>(ctx, D->d_sums, D->d_sumsoff, (const u32 *)D->d_good.p, (u64)n_aln, (u32 *)D->d_outidx.p))) return rc;
        if ((rc = scan_u32<u64>(ctx, D->d_sums, D->d_sumsoff, (const u32 *)D->d_gcig.p, (u64)n_aln, (u64 *)D->d_cigscan.p))) return rc;
        if (timing) lap("groups + gates");
        // where every good record's SEQ bytes go inside this file's stretch of the seq array, and -- for the batch's
        // window-order mirror -- where the record itself goes among the file's records in window order
        if (!window_layout) {  // SEQ bytes in file order: a scan of the rooms
            hipLaunchKernelGGL(k_tok_room, dim3((n_aln + 255) / 256), dim3(256), 0, st, n_aln, (const u32 *)D->d_gseq.p, (u32 *)D->d_groom.p);
            if ((rc = scan_u32<u64>(ctx, D->d_sums, D->d_sumsoff, (const u32 *)D->d_groom.p, (u64)n_aln, (u64 *)D->d_seqscan.p))) return rc;
        }
        if (window_layout || D->wo_mirror) {  // the multisplit into the windows the records start in (see k_tok_win_hist)
            ENS(d_wbytes, ((u64)n_win + 1) * 4); ENS(d_wbase, ((u64)n_win + 1) * 8); ENS(d_seqpos, (u64)n_aln * 8);
            ENS(d_wcount, ((u64)n_win + 1) * 4); ENS(d_wcbase, ((u64)n_win + 1) * 4); ENS(d_slot, (u64)n_aln * 4);
            if (n_win <= WIN_LDS_MAX) {
                // 16384 records per workgroup, more when the matrix would pass 1024 rows
                const u32 per_block = std::max<u32>(16384u, (u32)((((u64)n_aln + 1023u) / 1024u + 1023u) & ~1023ull));
                const u32 nb = (n_aln + per_block - 1u) / per_block;
                ENS(d_wcur, (u64)nb * n_win * 8);
                hipLaunchKernelGGL(k_tok_win_hist, dim3(nb), dim3(1024), 0, st, (const u32 *)D->d_win.p, (const u32 *)D->d_gseq.p, n_aln,
                                   per_block, n_win, (u64 *)D->d_wcur.p);
                hipLaunchKernelGGL(k_tok_win_cols, dim3((n_win + 3u) / 4u), dim3(256), 0, st, n_win, nb, (u64 *)D->d_wcur.p, (u32 *)D->d_wbytes.p,
                                   (u32 *)D->d_wcount.p);
                hipLaunchKernelGGL(k_tscan<u64>, dim3(1), dim3(1024), 0, st, (const u32 *)D->d_wbytes.p, (u64)n_win, (u64 *)D->d_wbase.p);
                hipLaunchKernelGGL(k_tscan<u32>, dim3(1), dim3(1024), 0, st, (const u32 *)D->d_wcount.p, (u64)n_win, (u32 *)D->d_wcbase.p);
                hipLaunchKernelGGL(k_tok_win_place, dim3(nb), dim3(1024), 0, st, (const u32 *)D->d_win.p, (const u32 *)D->d_gseq.p, n_aln,
                                   per_block, n_win, (const u64 *)D->d_wcur.p, (const u64 *)D->d_wbase.p, (const u32 *)D->d_wcbase.p,
                                   (u64 *)D->d_seqpos.p, (u32 *)D->d_slot.p);
            } else {
                ENS(d_wcur, (u64)n_win * 8);  // one 64-bit cursor per window
                PP_HIPCHK(ctx, hipMemsetAsync(D->d_wbytes.p, 0, ((size_t)n_win + 1) * 4, st));
                PP_HIPCHK(ctx, hipMemsetAsync(D->d_wcount.p, 0, ((size_t)n_win + 1) * 4, st));
                PP_HIPCHK(ctx, hipMemsetAsync(D->d_wcur.p, 0, (size_t)n_win * 8, st));
                hipLaunchKernelGGL(k_tok_win_bytes_g, dim3((n_aln + 255) / 256), dim3(256), 0, st, (const u32 *)D->d_win.p,
                                   (const u32 *)D->d_gseq.p, n_aln, (u32 *)D->d_wbytes.p, (u32 *)D->d_wcount.p);
                if ((rc = scan_u32<u64>(ctx, D->d_sums, D->d_sumsoff, (const u32 *)D->d_wbytes.p, (u64)n_win, (u64 *)D->d_wbase.p))) return rc;
                if ((rc = scan_u32<u32>(ctx, D->d_sums, D->d_sumsoff, (const u32 *)D->d_wcount.p, (u64)n_win, (u32 *)D->d_wcbase.p))) return rc;
                hipLaunchKernelGGL(k_tok_win_place_g, dim3((n_aln + 255) / 256), dim3(256), 0, st, (const u32 *)D->d_win.p,
                                   (const u32 *)D->d_gseq.p, n_aln, (const u64 *)D->d_wbase.p, (const u32 *)D->d_wcbase.p, (u64 *)D->d_wcur.p,
                                   (u64 *)D->d_seqpos.p, (u32 *)D->d_slot.p);
            }
        }
        if (window_layout) {
            seq_place = (const u64 *)D->d_seqpos.p;
            seq_total_at = (const u64 *)D->d_wbase.p + n_win;
        } else {
            seq_place = (const u64 *)D->d_seqscan.p;
            seq_total_at = (const u64 *)D->d_seqscan.p + n_aln;
        }
        if (timing) lap(window_layout ? "window layout" : "file-order layout");
    }
    u64 status = ~0ull;
    if ((rc = fetch(ctx, d_status, &status))) return rc;
    c.alignments = n_aln;
    if (status != ~0ull) {
        std::vector<u32> rec_line(n_aln), group_first((size_t)n_groups + 1);
        if (n_aln && (rc = fetch(ctx, (const u32 *)D->d_recline.p, rec_line.data(), n_aln))) return rc;
        if (n_aln && (rc = fetch(ctx, (const u32 *)D->d_gfirst.p, group_first.data(), (size_t)n_groups + 1))) return rc;
        if (counts) *counts = c;
        if (slice) return ctx->fail(PP_ERR_QUIT, "a defect in the text of \"%s\" (event %llu of a slice): left to the host ingest", path,
                                    (unsigned long long)status);
        return describe_error(D, path, F.text, size, status, n_lines, n_nl, rec_line, group_first);
    }
    if (pass_mismatch)
        return ctx->fail(PP_ERR_ARG, "%llu filter verdicts for the %u aligned records of \"%s\"", (unsigned long long)n_pass, n_aln, path);
    if (n_aln == 0) {  // the EOF flush of an empty group (alignment.rs:268, :319) -- of a FILE: a slice may well be empty
        if (slice) { if (counts) *counts = c; return PP_OK; }
        return ctx->fail(PP_ERR_PANIC, "no aligned records to process (the reference panics on an empty read group)");
    }
    u32 n_good = 0;
    u64 seq_total = 0, cig_total = 0;
    if ((rc = fetch(ctx, (const u32 *)D->d_outidx.p + n_aln, &n_good))) return rc;
    if ((rc = fetch(ctx, seq_total_at, &seq_total))) return rc;
    if ((rc = fetch(ctx, (const u64 *)D->d_cigscan.p + n_aln, &cig_total))) return rc;
    // ---- append to the batch ----
    const u64 no = D->n_out;
    // the first file of several (pp_dev_ingest_expect): room for the others as well, going by this one's yield per byte of text
    const double room = (no == 0 && !slice && D->expect_total > size && size) ? 1.03 * (double)D->expect_total / (double)size : 1.0;
    const u64 r_good = (u64)((double)n_good * room) + 16, r_seq = (u64)((double)seq_total * room) + 64, r_cig = (u64)((double)cig_total * room) + 16;
#define GROW(buf, elem, count, used) if ((rc = dev_grow(ctx, D->buf, (size_t)(count) * (elem), (size_t)(used) * (elem)))) return rc
    GROW(o_contig, 4, no + r_good, no); GROW(o_ref_start, 4, no + r_good, no); GROW(o_k, 4, no + r_good, no);
    GROW(o_seq_len, 4, no + r_good, no); GROW(o_n_cig, 4, no + r_good, no);
    GROW(o_seq_off, 8, no + r_good, no); GROW(o_cig_off, 8, no + r_good, no);
    GROW(o_seq, 1, D->seq_bytes + r_seq + 64, D->seq_bytes); GROW(o_cigar, 4, D->n_cig_total + r_cig, D->n_cig_total);
    if (D->mirror()) GROW(o_seq4, 1, (D->seq_bytes + r_seq) / 2 + 96, (D->seq_bytes + 1) / 2);
    if (D->wo_mirror) GROW(o_wo, sizeof(pp_wo_rec), no + r_good, no);
#undef GROW
    OutArrays O{(u32 *)D->o_contig.p, (u32 *)D->o_ref_start.p, (u32 *)D->o_k.p, (u32 *)D->o_seq_len.p, (u32 *)D->o_n_cig.p,
                (u32 *)D->o_cigar.p, (u64 *)D->o_seq_off.p, (u64 *)D->o_cig_off.p, (u8 *)D->o_seq.p};
    hipLaunchKernelGGL(k_tok_meta, dim3((n_aln + 255) / 256), dim3(256), 0, st, d_text, (const u64 *)D->d_nl.p, (const LineRec *)D->d_rec.p,
                       (const u32 *)D->d_recline.p, n_aln, (const u32 *)D->d_good.p, (const u32 *)D->d_k.p,
                       (const u32 *)D->d_gseq.p, (const u32 *)D->d_outidx.p, seq_place,
                       (const u64 *)D->d_cigscan.p, D->wo_mirror ? (const u32 *)D->d_slot.p : (const u32 *)nullptr,
                       D->wo_mirror ? (pp_wo_rec *)D->o_wo.p : (pp_wo_rec *)nullptr, O, no, D->seq_bytes, D->n_cig_total);
    // the SEQ bytes and, in the same pass, their 4-bit mirror (this file's stretch of the seq array starts on a multiple of
    // PP_SEQ_ALIGN: every stretch so far was a sum of rooms)
    hipLaunchKernelGGL(k_tok_seq, dim3((unsigned)(((u64)n_aln * 8 + 255) / 256)), dim3(256), 0, st, d_text, (const u64 *)D->d_nl.p,
                       (const LineRec *)D->d_rec.p, (const u32 *)D->d_recline.p, n_aln, (const u32 *)D->d_good.p,
                       (const u32 *)D->d_src.p, seq_place, O.seq, D->mirror() ? (u8 *)D->o_seq4.p : (u8 *)nullptr, D->seq_bytes);
    if (timing) lap("seq bytes + mirror");
    PP_HIPCHK(ctx, hipStreamSynchronize(st));  // the text mapping goes away with F
    lap("batch filled");
    D->n_out += n_good;
    if (D->wo_mirror && n_good) D->wo_run_end.push_back(D->n_out);  // this file's entries of the mirror: one run, in ascending window order
    D->seq_bytes += seq_total;
    D->n_cig_total += cig_total;
    c.used = n_good;
    c.reads = n_groups;
    if (counts) *counts = c;
#undef ENS
    return PP_OK;
}

// The HIP runtime loads a translation unit's code object when one of its kernels is launched for the first time (~10 ms for
// this one: it showed up as the first file's "newline index" stage).  pp_ctx_create_async launches this on its helper
// thread, while the host still loads the assembly.
__global__ void k_tok_warm(u32 *p) {
    if (p) p[threadIdx.x] = 0;
}
extern "C" void pp_tokenize_warm_(hipStream_t st) { hipLaunchKernelGGL(k_tok_warm, dim3(1), dim3(64), 0, st, (u32 *)nullptr); }
