// pp_filter_host.cpp -- pp_filter_files = filter::filter (src/filter.rs:26-37): the host half of the
// paired-read insert-size filter behind the `polypolish filter` CLI contract.
//
//   load_alignments / load_alignments_one_file   src/filter.rs:91-145  (Alignment::new_quick, alignment.rs:102-128)
//   get_insert_size_thresholds (reduction)       src/filter.rs:148-186, 221-270
//   filter_sams / filter_sam (re-emit + tag)     src/filter.rs:273-349
// The per-alignment work in between (get_ref_end, get_orientation, get_insert_size, alignment_pass_qc)
// runs on the device through pp_filter_begin / pp_filter_samples / pp_filter_pairs.
//
// Everything per-line is multi-threaded: the text is parsed in line-aligned slices, QNAMEs are interned
// through a lock-free open-addressing table (the reference's HashMap<String, Vec<Alignment>>, keyed by
// name + "_1"/"_2", becomes a read number shared by both files plus a per-file group index), and the
// output is formatted per slice and written with pwrite.  Results do not depend on the thread count.
#include <sys/stat.h>
#include <sys/uio.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <functional>
#include <cmath>
#include <cstdarg>
#include <string>
#include <unordered_map>

#include "polypolish_hip.h"
#include "pp_host.h"

using pph::HugeBuf;
using pph::parallel_for;

extern "C" int pp_ctx_set_error_(pp_ctx *ctx, int code, const char *msg);

namespace {

struct Log {
    bool quiet;
    void operator()(const char *fmt, ...) const {
        if (quiet) return;
        va_list ap;
        va_start(ap, fmt);
        vfprintf(stderr, fmt, ap);
        va_end(ap);
    }
};

std::string commas(uint64_t v) {  // num-format's Locale::en grouping
    std::string s = std::to_string(v), out;
    const int n = (int)s.size();
    for (int i = 0; i < n; i++) {
        out.push_back(s[i]);
        const int left = n - 1 - i;
        if (left > 0 && left % 3 == 0) out.push_back(',');
    }
    return out;
}

using pph::format_duration;
using pph::percentile_name;

struct FilterErr {
    int code;
    std::string msg;
};

[[noreturn]] void fail(int code, const char *fmt, ...) {
    char buf[1200];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    throw FilterErr{code, buf};
}

bool parse_u(const char *s, size_t n, uint64_t max, uint64_t &out) {  // str::parse::<uN>()
    size_t i = 0;
    if (n == 0) return false;
    if (s[0] == '+') { i = 1; if (n == 1) return false; }
    uint64_t v = 0;
    for (; i < n; i++) {
        if (s[i] < '0' || s[i] > '9') return false;
        const uint64_t d = (uint64_t)(s[i] - '0');
        if (v > (max - d) / 10) return false;
        v = v * 10 + d;
    }
    out = v;
    return true;
}

int cigar_op(char c) {
    switch (c) {
    case 'M': return PP_OP_M; case 'I': return PP_OP_I; case 'D': return PP_OP_D; case 'N': return PP_OP_N;
    case 'S': return PP_OP_S; case 'H': return PP_OP_H; case 'P': return PP_OP_P; case '=': return PP_OP_EQ;
    case 'X': return PP_OP_X; default: return -1;
    }
}

struct FLine {        // every line of the file, re-emitted by filter_sam
    uint64_t off;     // into the text
    uint32_t len;     // without the newline / CR
    int32_t aln;      // slice-local alignment index, -1 for header and unaligned lines
};
struct FAln {         // Alignment::new_quick, alignment.rs:102-128
    const char *name;
    uint32_t name_n, ref_local, ref_start, flags, run_lo, run_n;
};

enum { E_NONE = 0, E_COLUMNS, E_NUMBER, E_POS_LIMIT, E_UTF8 };

struct Slice {
    const char *beg = nullptr, *end = nullptr;
    HugeBuf<FLine> lines;
    HugeBuf<FAln> alns;
    HugeBuf<uint32_t> runs;
    std::vector<std::string> refs;  // slice-local RNAME numbering
    int err = E_NONE;               // first failing line of the slice (lines.size() = its 1-based index in the slice)
};

void parse_slice(Slice &S, const char *text) {
    std::unordered_map<std::string, uint32_t> ref_ids;
    std::string key, last_ref;
    uint32_t last_id = 0;
    bool have_last = false;
    const char *p = S.beg;
    while (p < S.end) {
        const char *nl = (const char *)memchr(p, '\n', (size_t)(S.end - p));
        size_t l = nl ? (size_t)(nl - p) : (size_t)(S.end - p);
        const char *line = p;
        p += l + (nl ? 1 : 0);
        if (l > 0 && line[l - 1] == '\r') l--;
        S.lines.push_back(FLine{(uint64_t)(line - text), (uint32_t)l, -1});
        if (!pph::valid_utf8(line, l)) { S.err = E_UTF8; return; }  // `let sam_line = line?`, filter.rs:121
        if (l > 0 && line[0] == '@') continue;
        const char *col[11];
        size_t len[11];
        size_t nc = 0;
        const char *q = line, *le = line + l;
        while (nc < 11) {
            const char *t = (const char *)memchr(q, '\t', (size_t)(le - q));
            col[nc] = q;
            len[nc] = t ? (size_t)(t - q) : (size_t)(le - q);
            nc++;
            if (!t) break;
            q = t + 1;
        }
        if (nc < 11) { S.err = E_COLUMNS; return; }  // an empty line is fatal here (filter.rs:126-130)
        uint64_t flags, pos;
        if (!parse_u(col[1], len[1], 0xFFFFFFFFull, flags) || !parse_u(col[3], len[3], UINT64_MAX, pos)) { S.err = E_NUMBER; return; }
        if (flags & 4) continue;
        if (pos > 0) pos -= 1;
        if (pos > 0xFFFFFFFFull) { S.err = E_POS_LIMIT; return; }
        FAln a;
        a.name = col[0]; a.name_n = (uint32_t)len[0];
        a.ref_start = (uint32_t)pos;
        a.flags = (uint32_t)flags;
        if (have_last && last_ref.size() == len[2] && memcmp(last_ref.data(), col[2], len[2]) == 0) {
            a.ref_local = last_id;
        } else {
            key.assign(col[2], len[2]);
            auto it = ref_ids.emplace(key, (uint32_t)S.refs.size());
            if (it.second) S.refs.push_back(key);
            a.ref_local = it.first->second;
            last_ref = key; last_id = a.ref_local; have_last = true;
        }
        a.run_lo = (uint32_t)S.runs.size();
        // Regex::find_iter over \d+[MIDNSHP=X] (alignment.rs:140): text that does not match is skipped
        const char *c = col[5];
        const size_t cl = len[5];
        size_t i = 0;
        while (i < cl) {
            if (c[i] >= '0' && c[i] <= '9') {
                size_t j = i;
                while (j < cl && c[j] >= '0' && c[j] <= '9') j++;
                const int op = j < cl ? cigar_op(c[j]) : -1;
                if (op >= 0) {
                    uint64_t num;
                    if (!parse_u(c + i, j - i, UINT64_MAX, num)) {
                        // a length the reference could not parse either -- but it only tries when a pair comparison
                        // needs this alignment's end (get_ref_end is lazy): the alignment is marked, not refused
                        S.runs.resize(a.run_lo);
                        S.runs.push_back((uint32_t)PP_OP_UNPARSEABLE);
                        break;
                    }
                    while (num > 0) {  // a packed run holds 28 bits of length
                        const uint32_t piece = num > 0x0FFFFFFFull ? 0x0FFFFFFFu : (uint32_t)num;
                        S.runs.push_back((piece << 4) | (uint32_t)op);
                        num -= piece;
                    }
                    i = j + 1;
                } else {
                    i = j;
                }
            } else {
                i++;
            }
        }
        a.run_n = (uint32_t)S.runs.size() - a.run_lo;
        S.lines.back().aln = (int32_t)S.alns.size();
        S.alns.push_back(a);
    }
}

struct FilterFile {
    pph::FileText text;
    std::vector<Slice> slices;
    std::vector<uint64_t> aln_first, run_first;  // prefix sums over slices
    uint64_t n_aln = 0, n_runs = 0;
    HugeBuf<uint32_t> ref_id, ref_start, flags, n_cig, cigar, read, grp_off, grp_idx;
    HugeBuf<uint64_t> cig_off;
};

inline uint64_t hash_name(const char *s, uint32_t n) {
    uint64_t h = 0x9E3779B97F4A7C15ull ^ ((uint64_t)n * 0xff51afd7ed558ccdull);
    while (n >= 8) {
        uint64_t w;
        memcpy(&w, s, 8);
        h = (h ^ w) * 0xff51afd7ed558ccdull;
        h ^= h >> 29;
        s += 8; n -= 8;
    }
    uint64_t w = 0;
    memcpy(&w, s, n);
    h = (h ^ w) * 0xc4ceb9fe1a85ec53ull;
    h ^= h >> 32;
    return h;
}

// QNAME -> read number.  Open addressing over record indices; a slot, once taken by a name, only ever
// moves to a smaller record index with the same name, so the representative of a name is its first
// record and the numbering is deterministic.
// QNAME -> representative record (the first record of the name in file-1-then-file-2 order), lock-free.
// A slot is 16 bytes: word A = hash tag << 32 | record index + 1 (0 = empty; fresh anonymous pages are zero),
// word B = the name of whoever claimed the slot, as address | min(length, 0xFFFF) << 48.  A probe compares the tag,
// then the bytes at B: two cache misses for a hit (slot, name text) instead of four through record pointers, and
// none for most foreign slots.  Callers run in batches that prefetch the slots of the next records.
struct NameTable {
    struct Slot { std::atomic<uint64_t> a, b; };
    HugeBuf<Slot> slots;
    uint64_t mask = 0;
    static constexpr unsigned BATCH = 16;
    void init(uint64_t n_records) {
        uint64_t cap = 1024;
        while (cap < 2 * n_records + 2) cap <<= 1;
        slots.resize(cap);
        mask = cap - 1;
    }
    static uint64_t pack(const FAln *r) {
        static_assert(sizeof(void *) == 8, "name addresses are packed into 48 bits");
        return ((uint64_t)(uintptr_t)r->name & 0xFFFFFFFFFFFFull) | ((uint64_t)std::min<uint32_t>(r->name_n, 0xFFFFu) << 48);
    }
    // does the name published in word B equal r's?  (lengths of 65535 and more are told apart by the bytes: a
    // QNAME ends at a tab, so the longer one differs from the shorter one at the shorter one's tab)
    static bool same(uint64_t b, const FAln *r) {
        const uint32_t n16 = (uint32_t)(b >> 48);
        if (n16 != std::min<uint32_t>(r->name_n, 0xFFFFu)) return false;
        const char *q = (const char *)(uintptr_t)(b & 0xFFFFFFFFFFFFull);
        if (n16 < 0xFFFFu) return memcmp(q, r->name, n16) == 0;  // equal lengths: n16 bytes exist on both sides
        // saturated length: the stored name may be SHORTER than r's -- compare byte by byte and stop at its tab, so that
        // nothing past its line (or past the mapping, for the last line) is read
        for (uint32_t i = 0; i < r->name_n; i++)
            if (q[i] != r->name[i] || q[i] == '\t') return false;
        return q[r->name_n] == '\t';
    }
    static uint64_t wait_b(const Slot &S) {  // the claimer publishes B right after winning A
        uint64_t b;
        while ((b = S.b.load(std::memory_order_acquire)) == 0) {}
        return b;
    }
    void prefetch(uint64_t h) const { __builtin_prefetch(&slots.data()[h & mask], 1, 1); }
    // returns true when the name was new
    bool insert(const FAln *r, uint32_t me, uint64_t h) {
        Slot *S = slots.data();
        const uint64_t tag = h >> 32 << 32, mine = tag | (uint64_t)(me + 1u);
        for (uint64_t i = h & mask;; i = (i + 1) & mask) {
            uint64_t a = S[i].a.load(std::memory_order_acquire);
            if (a == 0 && S[i].a.compare_exchange_strong(a, mine, std::memory_order_acq_rel)) {
                S[i].b.store(pack(r), std::memory_order_release);
                return true;
            }
            if ((a >> 32 << 32) != tag || !same(wait_b(S[i]), r)) continue;
            while ((uint32_t)a - 1u > me && !S[i].a.compare_exchange_weak(a, mine, std::memory_order_acq_rel)) {}
            return false;
        }
    }
    uint32_t find(const FAln *r, uint64_t h) const {
        const Slot *S = slots.data();
        const uint64_t tag = h >> 32 << 32;
        for (uint64_t i = h & mask;; i = (i + 1) & mask) {
            const uint64_t a = S[i].a.load(std::memory_order_relaxed);
            if ((a >> 32 << 32) == tag && a != 0 && same(S[i].b.load(std::memory_order_relaxed), r)) return (uint32_t)a - 1u;
        }
    }
};

// get_percentile, filter.rs:249-259 (on unsorted data: only one order statistic is needed)
uint32_t percentile(std::vector<uint32_t> &v, double p) {
    if (v.empty()) return 0;
    const double fraction = p / 100.0;
    const double r = ceil(fraction * (double)v.size());
    size_t rank = r <= 0.0 ? 0 : (r >= 1.8e19 ? SIZE_MAX : (size_t)r);
    if (rank < 1) rank = 1;
    if (rank - 1 >= v.size()) return 0;
    std::nth_element(v.begin(), v.begin() + (rank - 1), v.end());
    return v[rank - 1];
}

}  // namespace

struct pp_filter_loaded {
    FilterFile F[2];
    uint32_t n_reads = 0;
    uint64_t before = 0;
    unsigned threads = 1;
};

// load_alignments (filter.rs:91-145) for both files -> the SoA + read groups of pp_filter_input
extern "C" int pp_filter_load(const char *in1, const char *in2, pp_filter_loaded **out, pp_filter_file_counts counts[2],
                              char *err, size_t errlen) {
    if (!in1 || !in2 || !out) return PP_ERR_ARG;
    *out = nullptr;
    pp_filter_file_counts local[2];
    if (!counts) counts = local;
    memset(counts, 0, 2 * sizeof(pp_filter_file_counts));
    pp_filter_loaded *L = new pp_filter_loaded();
    FilterFile *F = L->F;
    const char *ins[2] = {in1, in2};
    HugeBuf<const FAln *> recs;  // both files, file 1 first, file order
    HugeBuf<uint32_t> rep;       // first record with the same QNAME
    NameTable table;
    uint64_t &before = L->before;
    uint32_t &n_reads = L->n_reads;
    unsigned &threads = L->threads;
    const bool timing = getenv("PP_TIMING") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (timing) fprintf(stderr, "[timing]   load: %-22s %8.3f s\n", what,
                            std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    };
    try {
        // Both files are parsed before any name is interned (the table is sized from the exact record
        // count); what the reference would report while loading file 1 still comes first.
        auto parse_file = [&](int f) {
            FilterFile &X = F[f];
            if (!X.text.open_file(ins[f])) fail(PP_ERR_QUIT, "unable to load alignments from \"%s\"", ins[f]);
            std::vector<const char *> cut;
            pph::line_slices(X.text.text, X.text.size, threads, cut);
            X.slices.resize(threads);
            for (unsigned t = 0; t < threads; t++) { X.slices[t].beg = cut[t]; X.slices[t].end = cut[t + 1]; }
            parallel_for(threads, threads, [&](size_t lo, size_t hi, unsigned) {
                for (size_t t = lo; t < hi; t++) parse_slice(X.slices[t], X.text.text);
            });
            uint64_t line_no = 0;
            for (unsigned t = 0; t < threads; t++) {  // the first failing line in file order
                const Slice &S = X.slices[t];
                line_no += S.lines.size();
                switch (S.err) {
                case E_NONE: break;
                case E_COLUMNS: fail(PP_ERR_QUIT, "too few columns in \"%s\" (line %llu)", ins[f], (unsigned long long)line_no);
                case E_NUMBER: fail(PP_ERR_PANIC, "could not parse FLAG or POS in \"%s\" (line %llu)", ins[f], (unsigned long long)line_no);
                case E_POS_LIMIT: fail(PP_ERR_LIMIT, "POS beyond 2^32 in \"%s\" (line %llu)", ins[f], (unsigned long long)line_no);
                case E_UTF8: fail(PP_ERR_QUIT, "unable to load alignments from \"%s\"", ins[f]);
                default: fail(PP_ERR_HIP, "internal: unknown parse error in \"%s\" (line %llu)", ins[f], (unsigned long long)line_no);
                }
            }
            X.aln_first.assign(threads + 1, 0);
            X.run_first.assign(threads + 1, 0);
            for (unsigned t = 0; t < threads; t++) {
                X.aln_first[t + 1] = X.aln_first[t] + X.slices[t].alns.size();
                X.run_first[t + 1] = X.run_first[t] + X.slices[t].runs.size();
            }
            X.n_aln = X.aln_first[threads];
            X.n_runs = X.run_first[threads];
        };
        {
            struct stat st1, st2;
            const uint64_t b1 = stat(in1, &st1) == 0 ? (uint64_t)st1.st_size : 0, b2 = stat(in2, &st2) == 0 ? (uint64_t)st2.st_size : 0;
            threads = pph::host_threads((size_t)(std::max(b1, b2)));
        }
        parse_file(0);
        bool deferred = false;
        FilterErr second{0, ""};
        try {
            parse_file(1);
        } catch (const FilterErr &e) {
            deferred = true;
            second = e;
        }
        lap("parsed");
        if (F[0].n_aln + (deferred ? 0 : F[1].n_aln) >= 0xFFFFFFFFull) fail(PP_ERR_LIMIT, "more than 2^32-1 alignments in the two files");
        table.init(F[0].n_aln + (deferred ? 0 : F[1].n_aln));
        uint64_t names_new[2] = {0, 0};
        for (int f = 0; f < 2; f++) {
            FilterFile &X = F[f];
            if (f == 1 && deferred) throw second;
            // intern this file's QNAMEs
            const uint64_t base = before;
            recs.resize(base + X.n_aln);
            std::vector<uint64_t> fresh(threads, 0);
            parallel_for(threads, threads, [&](size_t lo, size_t hi, unsigned) {
                for (size_t t = lo; t < hi; t++) {
                    Slice &S = X.slices[t];
                    const uint64_t g0 = base + X.aln_first[t];
                    for (size_t i = 0; i < S.alns.size(); i++) recs[g0 + i] = &S.alns[i];
                }
            });
            parallel_for(X.n_aln, threads, [&](size_t lo, size_t hi, unsigned t) {
                uint64_t n = 0, h[NameTable::BATCH];
                for (size_t i0 = lo; i0 < hi; i0 += NameTable::BATCH) {  // hash + prefetch a batch, then probe it
                    const size_t nb = std::min<size_t>(NameTable::BATCH, hi - i0);
                    for (size_t j = 0; j < nb; j++) {
                        const FAln *r = recs[base + i0 + j];
                        h[j] = hash_name(r->name, r->name_n);
                        table.prefetch(h[j]);
                    }
                    for (size_t j = 0; j < nb; j++) n += table.insert(recs[base + i0 + j], (uint32_t)(base + i0 + j), h[j]);
                }
                fresh[t] = n;
            });
            for (uint64_t n : fresh) names_new[f] += n;
            before += X.n_aln;
            uint64_t names = names_new[f];
            if (f == 1) {  // + names first seen in file 1 that file 2 also holds
                rep.resize(before);
                HugeBuf<uint8_t> hit;  // fresh pages: zero
                hit.resize(base ? base : 1);
                parallel_for(before, threads, [&](size_t lo, size_t hi, unsigned) {
                    uint64_t h[NameTable::BATCH];
                    for (size_t i0 = lo; i0 < hi; i0 += NameTable::BATCH) {
                        const size_t nb = std::min<size_t>(NameTable::BATCH, hi - i0);
                        for (size_t j = 0; j < nb; j++) {
                            h[j] = hash_name(recs[i0 + j]->name, recs[i0 + j]->name_n);
                            table.prefetch(h[j]);
                        }
                        for (size_t j = 0; j < nb; j++) {
                            const size_t i = i0 + j;
                            const uint32_t r = table.find(recs[i], h[j]);
                            rep[i] = r;
                            if (i >= base && r < base) __atomic_store_n(&hit[r], (uint8_t)1, __ATOMIC_RELAXED);
                        }
                    }
                });
                std::vector<uint64_t> part(threads, 0);
                parallel_for(base, threads, [&](size_t lo, size_t hi, unsigned t) {
                    uint64_t n = 0;
                    for (size_t i = lo; i < hi; i++) n += hit[i];
                    part[t] = n;
                });
                for (uint64_t n : part) names += n;
            }
            counts[f].alignments = X.n_aln;
            counts[f].reads = names;
            counts[f].loaded = 1;
            if (before == 0) fail(PP_ERR_QUIT, "no alignments found in \"%s\"", ins[f]);
        }
        lap("parsed + names interned");

        // read numbers: rank of the representative among representatives
        HugeBuf<uint32_t> id;
        id.resize(before);
        {
            std::vector<uint64_t> part(threads + 1, 0);
            parallel_for(before, threads, [&](size_t lo, size_t hi, unsigned t) {
                uint64_t n = 0;
                for (size_t i = lo; i < hi; i++) n += rep[i] == i;
                part[t + 1] = n;
            });
            for (unsigned t = 0; t < threads; t++) part[t + 1] += part[t];
            n_reads = (uint32_t)part[threads];
            parallel_for(before, threads, [&](size_t lo, size_t hi, unsigned t) {
                uint32_t n = (uint32_t)part[t];
                for (size_t i = lo; i < hi; i++)
                    if (rep[i] == i) id[i] = n++;
            });
        }
        // RNAME numbering across slices and files
        std::unordered_map<std::string, uint32_t> refs;
        std::vector<std::vector<uint32_t>> ref_map[2];
        for (int f = 0; f < 2; f++) {
            ref_map[f].resize(threads);
            for (unsigned t = 0; t < threads; t++)
                for (const std::string &r : F[f].slices[t].refs)
                    ref_map[f][t].push_back(refs.emplace(r, (uint32_t)refs.size()).first->second);
        }
        // the SoA of pp_filter_file, filled per slice
        uint64_t base = 0;
        for (int f = 0; f < 2; f++) {
            FilterFile &X = F[f];
            const size_t n = X.n_aln ? X.n_aln : 1;
            X.ref_id.resize(n); X.ref_start.resize(n); X.flags.resize(n); X.n_cig.resize(n); X.read.resize(n);
            X.cig_off.resize(n); X.cigar.resize(X.n_runs ? X.n_runs : 1); X.grp_idx.resize(n);
            X.grp_off.resize((size_t)n_reads + 1);  // fresh pages: zero
            std::atomic<uint32_t> *cnt = (std::atomic<uint32_t> *)X.grp_off.data();
            parallel_for(threads, threads, [&](size_t lo, size_t hi, unsigned) {
                for (size_t t = lo; t < hi; t++) {
                    const Slice &S = X.slices[t];
                    if (S.runs.size()) memcpy(X.cigar.data() + X.run_first[t], S.runs.data(), S.runs.size() * 4);
                    for (size_t i = 0; i < S.alns.size(); i++) {
                        const FAln &a = S.alns[i];
                        const size_t d = X.aln_first[t] + i;
                        X.ref_id[d] = ref_map[f][t][a.ref_local];
                        X.ref_start[d] = a.ref_start;
                        X.flags[d] = a.flags;
                        X.cig_off[d] = X.run_first[t] + a.run_lo;
                        X.n_cig[d] = a.run_n;
                        const uint32_t r = id[rep[base + d]];
                        X.read[d] = r;
                        cnt[r + 1].fetch_add(1, std::memory_order_relaxed);
                    }
                }
            });
            // group index: alignments of each read in file order
            for (uint32_t r = 0; r < n_reads; r++) X.grp_off[r + 1] += X.grp_off[r];
            HugeBuf<uint32_t> cur;
            cur.resize((size_t)n_reads + 1);
            parallel_for(n_reads, threads, [&](size_t lo, size_t hi, unsigned) {
                memcpy(cur.data() + lo, X.grp_off.data() + lo, (hi - lo) * 4);
            });
            std::atomic<uint32_t> *cursor = (std::atomic<uint32_t> *)cur.data();
            parallel_for(X.n_aln, threads, [&](size_t lo, size_t hi, unsigned) {
                for (size_t i = lo; i < hi; i++)
                    X.grp_idx[cursor[X.read[i]].fetch_add(1, std::memory_order_relaxed)] = (uint32_t)i;
            });
            parallel_for(n_reads, threads, [&](size_t lo, size_t hi, unsigned) {
                for (size_t r = lo; r < hi; r++)
                    if (X.grp_off[r + 1] - X.grp_off[r] > 1)
                        std::sort(X.grp_idx.data() + X.grp_off[r], X.grp_idx.data() + X.grp_off[r + 1]);
            });
            base += X.n_aln;
        }
        lap("records + groups built");
    } catch (const FilterErr &e) {
        if (err && errlen) snprintf(err, errlen, "%s", e.msg.c_str());
        delete L;
        return e.code;
    } catch (const std::bad_alloc &) {
        if (err && errlen) snprintf(err, errlen, "out of host memory while loading the alignments");
        delete L;
        return PP_ERR_LIMIT;
    }
    *out = L;
    return PP_OK;
}

extern "C" void pp_filter_loaded_free(pp_filter_loaded *L) { delete L; }

extern "C" void pp_filter_loaded_input(const pp_filter_loaded *L, pp_filter_input *in) {
    in->n_reads = L->n_reads;
    for (int f = 0; f < 2; f++) {
        const FilterFile &X = L->F[f];
        pp_filter_file &d = in->file[f];
        d.n_aln = X.n_aln;
        d.ref_id = X.ref_id.data(); d.ref_start = X.ref_start.data(); d.flags = X.flags.data();
        d.cig_off = X.cig_off.data(); d.n_cig = X.n_cig.data(); d.cigar = X.cigar.data();
        d.n_cig_total = X.n_runs; d.read = X.read.data();
        d.grp_off = X.grp_off.data(); d.grp_idx = X.grp_idx.data();
        d.ref_end = nullptr;  // computed on the device from the CIGAR runs
    }
}

// filter_sam (filter.rs:309-349): every input line again, "\tZP:Z:fail" appended where pass == 0
extern "C" int pp_filter_write(const pp_filter_loaded *L, int f, const uint8_t *pass_f, const char *out_path,
                               uint64_t *pass_count, uint64_t *fail_count, char *err, size_t errlen) {
    if (!L || f < 0 || f > 1 || !out_path || (!pass_f && L->F[f].n_aln)) return PP_ERR_ARG;
    const FilterFile &X = L->F[f];
    const unsigned threads = L->threads;
    auto write_failed = [&]() {
        if (err && errlen) snprintf(err, errlen, "unable to write alignments to \"%s\"", out_path);
        return PP_ERR_QUIT;
    };
    const int fd = open(out_path, O_WRONLY | O_CREAT | O_TRUNC, 0666);
    if (fd < 0) return write_failed();
    // size of every slice's output, then format + pwrite in parallel
    std::vector<uint64_t> out_first(threads + 1, 0), n_pass(threads, 0), n_fail(threads, 0);
    parallel_for(threads, threads, [&](size_t lo_t, size_t hi_t, unsigned) {
        for (size_t t = lo_t; t < hi_t; t++) {
            const Slice &S = X.slices[t];
            uint64_t bytes = 0, p_ = 0, f_ = 0;
            for (size_t i = 0; i < S.lines.size(); i++) {
                bytes += S.lines[i].len + 1;
                if (S.lines[i].aln >= 0) {
                    if (pass_f[X.aln_first[t] + (size_t)S.lines[i].aln]) p_++;
                    else { f_++; bytes += 10; }
                }
            }
            out_first[t + 1] = bytes; n_pass[t] = p_; n_fail[t] = f_;
        }
    });
    for (unsigned t = 0; t < threads; t++) out_first[t + 1] += out_first[t];
    std::atomic<int> bad{0};
    // A pipe, /dev/stdout or a process substitution cannot be written at offsets (ESPIPE): there the slices are
    // formatted and written one after the other, as the reference's buffered writer streams them (src/filter.rs:305-306).
    struct stat st_out;
    const bool seekable = fstat(fd, &st_out) == 0 && S_ISREG(st_out.st_mode);
    auto write_slice = [&](size_t t) {
        const Slice &S = X.slices[t];
        const uint64_t bytes = out_first[t + 1] - out_first[t];
        if (!bytes) return;
        HugeBuf<char> buf;
        buf.resize(bytes);
        char *w = buf.data();
        for (size_t i = 0; i < S.lines.size(); i++) {
            memcpy(w, X.text.text + S.lines[i].off, S.lines[i].len);
            w += S.lines[i].len;
            if (S.lines[i].aln >= 0 && !pass_f[X.aln_first[t] + (size_t)S.lines[i].aln]) {
                memcpy(w, "\tZP:Z:fail", 10);
                w += 10;
            }
            *w++ = '\n';
        }
        uint64_t done = 0;
        while (done < bytes) {
            const ssize_t r = seekable ? pwrite(fd, buf.data() + done, bytes - done, (off_t)(out_first[t] + done))
                                       : write(fd, buf.data() + done, bytes - done);
            if (r <= 0) { bad = 1; break; }
            done += (uint64_t)r;
        }
    };
    if (!seekable) {
        for (size_t t = 0; t < threads && !bad; t++) write_slice(t);
    } else {
        parallel_for(threads, threads, [&](size_t lo_t, size_t hi_t, unsigned) {
            for (size_t t = lo_t; t < hi_t; t++) write_slice(t);
        });
    }
    if (close(fd) != 0 || bad) return write_failed();
    uint64_t p_ = 0, f_ = 0;
    for (unsigned t = 0; t < threads; t++) { p_ += n_pass[t]; f_ += n_fail[t]; }
    if (pass_count) *pass_count = p_;
    if (fail_count) *fail_count = f_;
    return PP_OK;
}

// Internal (the device loader's error path): what load_alignments says about ONE line.
extern "C" int pp_filter_line_error_(const char *line, size_t n, const char *path, uint64_t line_no, char *err, size_t errlen) {
    Slice S;
    S.beg = line;
    S.end = line + n;
    if (n == 0) S.err = E_COLUMNS;  // an empty line (parse_slice would not even see it without its newline)
    else parse_slice(S, line);
    const unsigned long long ln = (unsigned long long)line_no;
    switch (S.err) {
    case E_NONE: return PP_OK;
    case E_COLUMNS: snprintf(err, errlen, "too few columns in \"%s\" (line %llu)", path, ln); return PP_ERR_QUIT;
    case E_NUMBER: snprintf(err, errlen, "could not parse FLAG or POS in \"%s\" (line %llu)", path, ln); return PP_ERR_PANIC;
    case E_POS_LIMIT: snprintf(err, errlen, "POS beyond 2^32 in \"%s\" (line %llu)", path, ln); return PP_ERR_LIMIT;
    case E_UTF8: snprintf(err, errlen, "unable to load alignments from \"%s\"", path); return PP_ERR_QUIT;
    default: snprintf(err, errlen, "internal: unknown parse error in \"%s\" (line %llu)", path, ln); return PP_ERR_HIP;
    }
}

// filter_sam (filter.rs:309-349) from a text in memory: the lines are found again here (in parallel slices);
// a line is an aligned record when it is not a header and FLAG & 4 is clear -- the text has been validated
// by the loader that produced the verdicts.
// Tagged copy of one SAM text into an already open (and truncated) file; closes fd.
static int write_text_fd(const char *text, uint64_t size, const uint8_t *pass, uint64_t n_pass, int fd,
                         const char *out_path, uint64_t *pass_count, uint64_t *fail_count, char *err, size_t errlen) {
    auto write_failed = [&]() {
        if (err && errlen) snprintf(err, errlen, "unable to write alignments to \"%s\"", out_path);
        return PP_ERR_QUIT;
    };
    const unsigned threads = pph::host_threads((size_t)size);
    std::vector<const char *> cut;
    pph::line_slices(text, (size_t)size, threads, cut);
    auto is_aligned = [](const char *line, size_t l) {
        if (l == 0 || line[0] == '@') return false;
        const char *t1 = (const char *)memchr(line, '\t', l);
        if (!t1) return false;
        const char *f = t1 + 1, *le = line + l;
        const char *t2 = (const char *)memchr(f, '\t', (size_t)(le - f));
        uint64_t flags = 0;
        if (!parse_u(f, (size_t)((t2 ? t2 : le) - f), 0xFFFFFFFFull, flags)) return false;
        return (flags & 4) == 0;
    };
    // aligned records per slice -> the ordinal of each slice's first one
    std::vector<uint64_t> first(threads + 1, 0);
    parallel_for(threads, threads, [&](size_t lo, size_t hi, unsigned) {
        for (size_t t = lo; t < hi; t++) {
            uint64_t n = 0;
            for (const char *p = cut[t]; p < cut[t + 1];) {
                const char *nl = (const char *)memchr(p, '\n', (size_t)(cut[t + 1] - p));
                size_t l = nl ? (size_t)(nl - p) : (size_t)(cut[t + 1] - p);
                const char *line = p;
                p += l + (nl ? 1 : 0);
                if (l > 0 && line[l - 1] == '\r') l--;
                n += is_aligned(line, l);
            }
            first[t + 1] = n;
        }
    });
    for (unsigned t = 0; t < threads; t++) first[t + 1] += first[t];
    if (first[threads] != n_pass) {
        if (err && errlen) snprintf(err, errlen, "%llu verdicts for %llu aligned records", (unsigned long long)n_pass,
                                    (unsigned long long)first[threads]);
        close(fd);
        return PP_ERR_ARG;
    }
    // The output is the input with a tag spliced in here and there: every slice becomes a list of iovecs that
    // point INTO the input mapping (stretches of untouched lines) or at the constant tag / newline, written
    // with pwritev -- no intermediate copy of the gigabytes that do not change.
    static const char TAG_NL[] = "\tZP:Z:fail\n";
    std::vector<std::vector<struct iovec>> iov(threads);
    std::vector<uint64_t> off(threads + 1, 0), n_ok(threads, 0), n_bad(threads, 0);
    parallel_for(threads, threads, [&](size_t lo, size_t hi, unsigned) {
        for (size_t t = lo; t < hi; t++) {
            std::vector<struct iovec> &V = iov[t];
            uint64_t a = first[t], bytes = 0;
            const char *s0 = cut[t];  // start of the current untouched stretch
            auto emit = [&](const char *p, size_t n) {
                if (!n) return;
                V.push_back(iovec{(void *)p, n});
                bytes += n;
            };
            for (const char *p = cut[t]; p < cut[t + 1];) {
                const char *nl = (const char *)memchr(p, '\n', (size_t)(cut[t + 1] - p));
                size_t l = nl ? (size_t)(nl - p) : (size_t)(cut[t + 1] - p);
                const char *line = p;
                p += l + (nl ? 1 : 0);
                const bool cr = l > 0 && line[l - 1] == '\r';
                if (cr) l--;
                bool failed = false;
                if (is_aligned(line, l)) {
                    if (pass[a++]) n_ok[t]++;
                    else { failed = true; n_bad[t]++; }
                }
                if (failed || cr || !nl) {  // the stretch ends with this line's content; its ending is rewritten
                    emit(s0, (size_t)(line + l - s0));
                    if (failed) emit(TAG_NL, 11); else emit(TAG_NL + 10, 1);
                    s0 = p;
                }
            }
            emit(s0, (size_t)(cut[t + 1] - s0));
            off[t + 1] = bytes;
        }
    });
    for (unsigned t = 0; t < threads; t++) off[t + 1] += off[t];
    std::atomic<int> bad{0};
    // A pipe, /dev/stdout or a process substitution cannot be written at offsets (ESPIPE): there the slices go out
    // one after the other with writev, as the reference's buffered writer would stream them (src/filter.rs:305-306).
    struct stat st_out;
    const bool seekable = fstat(fd, &st_out) == 0 && S_ISREG(st_out.st_mode);
    if (!seekable) {
        for (unsigned t = 0; t < threads && !bad; t++) {
            std::vector<struct iovec> &V = iov[t];
            size_t i = 0;
            while (i < V.size()) {
                const int cnt = (int)std::min<size_t>(1024, V.size() - i);
                ssize_t r = writev(fd, &V[i], cnt);
                if (r <= 0) { bad = 1; break; }
                while (r > 0 && i < V.size()) {
                    if ((size_t)r >= V[i].iov_len) { r -= (ssize_t)V[i].iov_len; i++; }
                    else { V[i].iov_base = (char *)V[i].iov_base + r; V[i].iov_len -= (size_t)r; r = 0; }
                }
            }
        }
    } else
    parallel_for(threads, threads, [&](size_t lo, size_t hi, unsigned) {
        for (size_t t = lo; t < hi; t++) {
            std::vector<struct iovec> &V = iov[t];
            uint64_t pos = off[t];
            size_t i = 0;
            while (i < V.size()) {
                const int cnt = (int)std::min<size_t>(1024, V.size() - i);
                ssize_t r = pwritev(fd, &V[i], cnt, (off_t)pos);
                if (r <= 0) { bad = 1; return; }
                pos += (uint64_t)r;
                while (r > 0 && i < V.size()) {  // advance over what was written (a short write cuts an iovec)
                    if ((size_t)r >= V[i].iov_len) { r -= (ssize_t)V[i].iov_len; i++; }
                    else { V[i].iov_base = (char *)V[i].iov_base + r; V[i].iov_len -= (size_t)r; r = 0; }
                }
            }
        }
    });
    if (close(fd) != 0 || bad) return write_failed();
    uint64_t p_ = 0, f_ = 0;
    for (unsigned t = 0; t < threads; t++) { p_ += n_ok[t]; f_ += n_bad[t]; }
    if (pass_count) *pass_count = p_;
    if (fail_count) *fail_count = f_;
    return PP_OK;
}

static int open_output(const char *out_path, char *err, size_t errlen) {
    const int fd = open(out_path, O_WRONLY | O_CREAT | O_TRUNC, 0666);
    if (fd < 0 && err && errlen) snprintf(err, errlen, "unable to write alignments to \"%s\"", out_path);
    return fd;
}

extern "C" int pp_filter_write_text(const char *text, uint64_t size, const uint8_t *pass, uint64_t n_pass,
                                    const char *out_path, uint64_t *pass_count, uint64_t *fail_count, char *err,
                                    size_t errlen) {
    if ((!text && size) || !out_path || (!pass && n_pass)) return PP_ERR_ARG;
    const int fd = open_output(out_path, err, errlen);
    if (fd < 0) return PP_ERR_QUIT;
    return write_text_fd(text, size, pass, n_pass, fd, out_path, pass_count, fail_count, err, errlen);
}

namespace {

struct FilterRun {
    pp_filter_loaded *L = nullptr;  // host load ...
    pp_filter_dev *DL = nullptr;    // ... or device load (PP_DEVICE_FILTER=1)
    uint64_t n_aln[2] = {0, 0};
    std::vector<uint8_t> pass[2];
    uint64_t before = 0;
    uint32_t lo = 0, hi = 0;
    int correct = -1;
    uint64_t counts[4] = {0, 0, 0, 0};
    ~FilterRun() {
        pp_filter_loaded_free(L);
        pp_filter_dev_free(DL);
    }
};

// load_alignments + get_insert_size_thresholds + the pass/fail verdicts (filter.rs:26-34 without filter_sams)
int filter_core(pp_ctx *ctx, const Log &log, const std::function<void(const char *)> &lap, const char *in1, const char *in2,
                const char *orientation, double low, double high, FilterRun &R) {
    auto set_err = [&](int code, const char *msg) { return pp_ctx_set_error_(ctx, code, msg); };
    log("Loading alignments\n");
    const char *ins[2] = {in1, in2};
    // PP_DEVICE_FILTER=1: load_alignments as kernels (pp_filter_dev.hip).  Default is the host loader: its parse
    // overlaps the HIP runtime's start-up, which the device loader has to wait for (measured: 0.86 s vs 1.07 s).
    bool dev_load = getenv("PP_DEVICE_FILTER") && atoi(getenv("PP_DEVICE_FILTER")) != 0;
    pp_filter_file_counts fc[2];
    char err[1400] = "";
    int rc = PP_OK;
    if (dev_load) {
        rc = pp_filter_load_device(ctx, in1, in2, &R.DL, fc);
        if (rc == PP_ERR_NOT_ASCII) {  // whether every line is valid UTF-8 is the host loader's call
            pp_filter_dev_free(R.DL);
            R.DL = nullptr;
            dev_load = false;
        }
    }
    if (!dev_load) rc = pp_filter_load(in1, in2, &R.L, fc, err, sizeof err);
    for (int f = 0; f < 2; f++)
        if (fc[f].loaded)
            log("%s: %s alignments from %s reads\n", ins[f], commas(fc[f].alignments).c_str(), commas(fc[f].reads).c_str());
    if (rc) return dev_load ? rc : set_err(rc, err);
    lap("alignments loaded");
    log("\n");
    pp_filter_input in{};
    if (dev_load) pp_filter_dev_input(R.DL, &in);
    else pp_filter_loaded_input(R.L, &in);
    const uint32_t n_reads = in.n_reads;
    R.n_aln[0] = in.file[0].n_aln;
    R.n_aln[1] = in.file[1].n_aln;
    R.before = R.n_aln[0] + R.n_aln[1];
    rc = pp_filter_begin(ctx, &in, dev_load ? PP_MEM_DEVICE : PP_MEM_HOST);
    if (rc) return rc;

    // get_insert_size_thresholds, filter.rs:148-186 (samples from the device, reduction on the host)
    log("Finding insert size thresholds\n");
    std::vector<uint8_t> orient(n_reads ? n_reads : 1);
    std::vector<uint32_t> insert(n_reads ? n_reads : 1);
    rc = pp_filter_samples(ctx, orient.data(), insert.data());
    if (rc) return rc;
    lap("samples from the device");
    uint64_t counts[4] = {0, 0, 0, 0};
    const unsigned red_threads = std::max(1u, std::min({std::thread::hardware_concurrency(), 16u, n_reads / 65536u + 1u}));
    {   // orientation counts, a slice of the reads per thread
        std::vector<std::array<uint64_t, 4>> part(red_threads, std::array<uint64_t, 4>{0, 0, 0, 0});
        std::vector<std::thread> th;
        for (unsigned t = 0; t < red_threads; t++)
            th.emplace_back([&, t] {
                const uint32_t a = (uint32_t)((uint64_t)n_reads * t / red_threads), b = (uint32_t)((uint64_t)n_reads * (t + 1) / red_threads);
                for (uint32_t r = a; r < b; r++)
                    if (orient[r] < 4) part[t][orient[r]]++;
            });
        for (auto &x : th) x.join();
        for (auto &p4 : part)
            for (int o = 0; o < 4; o++) counts[o] += p4[o];
    }
    if (counts[0] + counts[1] + counts[2] + counts[3] == 0)
        return set_err(PP_ERR_QUIT, "no one-alignment-per-read pairs available to determine orientation and "
                                    "insert size thresholds");
    static const char *ONAMES[4] = {"fr", "rf", "ff", "rr"};
    for (int o = 0; o < 4; o++) log("%s: %s pairs\n", ONAMES[o], commas(counts[o]).c_str());
    int correct = -1;
    if (strcmp(orientation, "auto") == 0) {  // auto_determine_orientation, filter.rs:238-246
        const uint64_t mx = *std::max_element(counts, counts + 4);
        int n_max = 0;
        for (int o = 0; o < 4; o++)
            if (counts[o] == mx) { n_max++; correct = o; }
        if (n_max != 1) return set_err(PP_ERR_QUIT, "could not automatically determine read pair orientation");
        log("\nAutomatically determined correct orientation: %s\n\n", ONAMES[correct]);
    } else {
        for (int o = 0; o < 4; o++)
            if (strcmp(orientation, ONAMES[o]) == 0) correct = o;
        log("\nUser-specified correct orientation: %s\n\n", orientation);
    }
    // get_percentile (filter.rs:249-259) is an order statistic of the correctly oriented pairs' insert sizes: taken
    // from a histogram (one per thread over its slice of the reads; sizes of 2^16 and more, if any, are kept as a list)
    // instead of gathering and partially sorting 3.3 M values on one core
    const uint64_t n_sizes = correct >= 0 ? counts[correct] : 0;
    if (n_sizes == 0) return set_err(PP_ERR_QUIT, "no read pairs available to determine insert size thresholds");
    constexpr uint32_t HBINS = 1u << 16;
    std::vector<uint64_t> hist(HBINS, 0);
    std::vector<uint32_t> big;
    {
        std::vector<std::vector<uint32_t>> hpart(red_threads), bpart(red_threads);
        std::vector<std::thread> th;
        for (unsigned t = 0; t < red_threads; t++)
            th.emplace_back([&, t] {
                hpart[t].assign(HBINS, 0);
                const uint32_t a = (uint32_t)((uint64_t)n_reads * t / red_threads), b = (uint32_t)((uint64_t)n_reads * (t + 1) / red_threads);
                for (uint32_t r = a; r < b; r++) {
                    if (orient[r] != correct) continue;
                    if (insert[r] < HBINS) hpart[t][insert[r]]++;
                    else bpart[t].push_back(insert[r]);
                }
            });
        for (auto &x : th) x.join();
        for (unsigned t = 0; t < red_threads; t++) {
            for (uint32_t i = 0; i < HBINS; i++) hist[i] += hpart[t][i];
            big.insert(big.end(), bpart[t].begin(), bpart[t].end());
        }
    }
    auto order_stat = [&](double p) -> uint32_t {  // = percentile() above on the gathered values
        const double r = ceil(p / 100.0 * (double)n_sizes);
        size_t rank = r <= 0.0 ? 0 : (r >= 1.8e19 ? SIZE_MAX : (size_t)r);
        if (rank < 1) rank = 1;
        if (rank - 1 >= n_sizes) return 0;
        uint64_t seen = 0;
        for (uint32_t i = 0; i < HBINS; i++) {
            seen += hist[i];
            if (seen >= rank) return i;
        }
        const size_t k = rank - 1 - (size_t)seen;  // among the sizes of 2^16 and more
        std::nth_element(big.begin(), big.begin() + (long)k, big.end());
        return big[k];
    };
    const uint32_t lo = order_stat(low), hi = order_stat(high);
    log("Low threshold:  %u (%s)\nHigh threshold: %u (%s)\n\n", lo, percentile_name(low).c_str(), hi,
        percentile_name(high).c_str());
    lap("thresholds");

    // alignment_pass_qc for every alignment of both files (filter.rs:352-377)
    for (int f = 0; f < 2; f++) R.pass[f].resize(R.n_aln[f] ? R.n_aln[f] : 1);
    rc = pp_filter_pairs(ctx, lo, hi, (uint8_t)correct, R.pass[0].data(), R.pass[1].data());
    if (rc) return rc;
    lap("pass flags from the device");
    R.lo = lo; R.hi = hi; R.correct = correct;
    for (int o = 0; o < 4; o++) R.counts[o] = counts[o];
    return PP_OK;
}

// filter_sams, filter.rs:273-349.  The outputs are opened in the reference's order (a second file is not created
// when the first cannot be), then both are written at the same time: buffered writes serialise on the inode, so
// two files in flight take about as long as one.  (Copying into a shared mapping of the output instead of
// pwritev was 3-4x slower on the 256-core box: 1.0-1.5 s of page faults for the 1.36 GB.)
int filter_write(pp_ctx *ctx, const Log &log, const FilterRun &R, const char *const ins[2], const char *const outs[2],
                 uint64_t *after) {
    char err[2][1400] = {"", ""};
    *after = 0;
    log("Filtering SAM files\n");
    uint64_t p_[2] = {0, 0}, f_[2] = {0, 0};
    int rc[2] = {PP_OK, PP_OK};
    auto write_one = [&](int f, int fd) {
        // both loaders keep the input mapped: the tags are spliced in with pwritev straight from it
        uint64_t size = 0;
        const char *text = R.DL ? pp_filter_dev_text(R.DL, f, &size) : R.L->F[f].text.text;
        if (!R.DL) size = R.L->F[f].text.size;
        rc[f] = write_text_fd(text, size, R.pass[f].data(), R.n_aln[f], fd, outs[f], &p_[f], &f_[f], err[f], sizeof err[f]);
    };
    const int fd0 = open_output(outs[0], err[0], sizeof err[0]);
    if (fd0 < 0) return pp_ctx_set_error_(ctx, PP_ERR_QUIT, err[0]);
    std::thread first(write_one, 0, fd0);
    const int fd1 = open_output(outs[1], err[1], sizeof err[1]);
    if (fd1 >= 0) write_one(1, fd1);
    else rc[1] = PP_ERR_QUIT;
    first.join();
    for (int f = 0; f < 2; f++) {
        if (rc[f]) return pp_ctx_set_error_(ctx, rc[f], err[f]);
        log("Filtering %s:\n  %s pass\n  %s fail\n\n", ins[f], commas(p_[f]).c_str(), commas(f_[f]).c_str());
        *after += p_[f];
    }
    return PP_OK;
}

int check_filter_options(pp_ctx *ctx, const char *const *names, int n_names, double low, double high) {
    // check_inputs, filter.rs:40-53
    for (int i = 0; i < n_names; i++)
        for (int j = i + 1; j < n_names; j++)
            if (strcmp(names[i], names[j]) == 0)
                return pp_ctx_set_error_(ctx, PP_ERR_QUIT, "--in1, --in2, --out1 and --out2 must all have unique values");
    if (low <= 0.0 || low >= 50.0) return pp_ctx_set_error_(ctx, PP_ERR_QUIT, "--low must be greater than 0 and less than 50");
    if (high <= 50.0 || high >= 100.0) return pp_ctx_set_error_(ctx, PP_ERR_QUIT, "--high must be greater than 50 and less than 100");
    return PP_OK;
}

}  // namespace

extern "C" int pp_filter_files(pp_ctx *ctx, const char *in1, const char *in2, const char *out1,
                               const char *out2, const char *orientation, double low, double high, int quiet,
                               pp_filter_report *report) {
    if (!ctx || !in1 || !in2 || !out1 || !out2 || !orientation) return PP_ERR_ARG;
    Log log{quiet != 0};
    const auto t0 = std::chrono::steady_clock::now();
    const bool timing = getenv("PP_TIMING") != nullptr;
    auto lap = [&](const char *what) {
        if (timing) fprintf(stderr, "[timing] %-28s %8.3f s\n", what,
                            std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    };
    const char *f4[4] = {in1, in2, out1, out2};
    if (int rc = check_filter_options(ctx, f4, 4, low, high)) return rc;
    log("\nStarting Polypolish filter\n%s\n\nInput alignments:\n  %s\n  %s\n\nOutput alignments:\n  %s\n  %s\n\n"
        "Settings:\n  --orientation %s\n  --low %g\n  --high %g\n\n", pp_version(), in1, in2, out1, out2, orientation, low, high);
    FilterRun R;
    if (int rc = filter_core(ctx, log, lap, in1, in2, orientation, low, high, R)) return rc;
    const char *ins[2] = {in1, in2}, *outs[2] = {out1, out2};
    uint64_t after = 0;
    if (int rc = filter_write(ctx, log, R, ins, outs, &after)) return rc;
    lap("filtered SAMs written");
    if (report) {
        report->before_count = R.before;
        report->after_count = after;
        report->low_threshold = R.lo;
        report->high_threshold = R.hi;
        report->orientation = R.correct;
        for (int o = 0; o < 4; o++) report->orientation_counts[o] = R.counts[o];
    }
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    log("Finished!\nAlignments before filtering: %s\nAlignments after filtering:  %s\n\nTime to run: %s\n\n",
        commas(R.before).c_str(), commas(after).c_str(), format_duration(secs).c_str());
    return PP_OK;
}

extern "C" int pp_polish_files_filtered_(pp_ctx *ctx, const char *assembly, const char *const *sams, int n_sams,
                                         const pp_polish_options *opt, pp_bytes *fasta,
                                         const uint8_t *const *pass, const uint64_t *n_pass);

extern "C" int pp_filter_polish_files(pp_ctx *ctx, const char *assembly, const char *in1, const char *in2,
                                      const char *out1, const char *out2, const char *orientation, double low,
                                      double high, const pp_polish_options *opt, pp_filter_report *report,
                                      pp_bytes *fasta) {
    if (!ctx || !assembly || !in1 || !in2 || !orientation || !opt || !fasta) return PP_ERR_ARG;
    if ((out1 == nullptr) != (out2 == nullptr))
        return pp_ctx_set_error_(ctx, PP_ERR_ARG, "pp_filter_polish_files: give both --out1 and --out2 or neither");
    Log log{opt->quiet != 0};
    const auto t0 = std::chrono::steady_clock::now();
    const bool timing = getenv("PP_TIMING") != nullptr;
    auto lap = [&](const char *what) {
        if (timing) fprintf(stderr, "[timing] %-28s %8.3f s\n", what,
                            std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    };
    const char *f4[4] = {in1, in2, out1, out2};
    if (int rc = check_filter_options(ctx, f4, out1 ? 4 : 2, low, high)) return rc;
    log("\nStarting Polypolish filter + polish (fused)\n%s\n\nInput alignments:\n  %s\n  %s\n\n"
        "Filter settings:\n  --orientation %s\n  --low %g\n  --high %g\n\n", pp_version(), in1, in2, orientation, low, high);
    FilterRun R;
    if (int rc = filter_core(ctx, log, lap, in1, in2, orientation, low, high, R)) return rc;
    const char *ins[2] = {in1, in2}, *outs[2] = {out1, out2};
    uint64_t after = 0;
    if (out1) {
        if (int rc = filter_write(ctx, log, R, ins, outs, &after)) return rc;
        lap("filtered SAMs written");
    } else {
        for (int f = 0; f < 2; f++)
            for (uint64_t i = 0; i < R.n_aln[f]; i++) after += R.pass[f][i];
    }
    if (report) {
        report->before_count = R.before;
        report->after_count = after;
        report->low_threshold = R.lo;
        report->high_threshold = R.hi;
        report->orientation = R.correct;
        for (int o = 0; o < 4; o++) report->orientation_counts[o] = R.counts[o];
    }
    log("Alignments before filtering: %s\nAlignments after filtering:  %s\n", commas(R.before).c_str(), commas(after).c_str());
    // the verdicts go straight into the polish ingest; the filter's parse-time memory is released first
    const uint8_t *pass[2] = {R.pass[0].data(), R.pass[1].data()};
    const uint64_t n_pass[2] = {R.n_aln[0], R.n_aln[1]};
    pp_filter_loaded_free(R.L);
    R.L = nullptr;
    pp_filter_dev_free(R.DL);
    R.DL = nullptr;
    return pp_polish_files_filtered_(ctx, assembly, ins, 2, opt, fasta, pass, n_pass);
}
