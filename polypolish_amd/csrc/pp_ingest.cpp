// pp_ingest.cpp -- host ingest: FASTA / SAM text -> the structure-of-arrays of polypolish_hip.h.
//
// Mirrors, line for line in behaviour (not in code), the host half of the reference's polish path:
//   load_fasta / check_load_fasta        src/misc.rs:38-167
//   Alignment::new                       src/alignment.rs:49-98  (+ get_expanded_cigar :325-346)
//   add_to_pileup (grouping)             src/alignment.rs:225-272
//   process_one_read (gates, 1/k, "*")   src/alignment.rs:275-322
//   reverse_complement                   src/misc.rs:170-191
// The CIGAR is kept as run-length ops (never expanded); everything downstream of the gates
// (CIGAR walk, trim, pileup, vote) happens on the device.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "polypolish_hip.h"
#include "pp_host.h"

using pph::HugeBuf;
using pph::parallel_for;

namespace {
// bytes of the seq array a record of n SEQ bytes takes (include/polypolish_hip.h: PP_SEQ_ALIGN)
inline size_t seq_room(size_t n) { return (n + (size_t)PP_SEQ_ALIGN - 1) & ~((size_t)PP_SEQ_ALIGN - 1); }

struct IngestError {
    int code;
    std::string msg;
};

[[noreturn]] void fail(int code, const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    throw IngestError{code, buf};
}

bool read_file(const char *path, std::vector<char> &out) {
    FILE *f = fopen(path, "rb");
    if (!f) return false;
    if (fseek(f, 0, SEEK_END) == 0) {
        long sz = ftell(f);
        if (sz > 0) out.reserve((size_t)sz + 1);
        fseek(f, 0, SEEK_SET);
    }
    char tmp[1 << 16];
    size_t r;
    while ((r = fread(tmp, 1, sizeof tmp, f)) > 0) out.insert(out.end(), tmp, tmp + r);
    bool bad = ferror(f);
    fclose(f);
    return !bad;
}

bool read_gz(const char *path, std::vector<char> &out) {
    gzFile g = gzopen(path, "rb");
    if (!g) return false;
    char tmp[1 << 16];
    int r;
    while ((r = gzread(g, tmp, sizeof tmp)) > 0) out.insert(out.end(), tmp, tmp + r);
    gzclose(g);
    return r == 0;
}

// BufRead::lines(): split on '\n', drop one trailing '\r', no empty line after a final newline
struct LineReader {
    const char *p, *end;
    bool next(const char *&line, size_t &n) {
        if (p >= end) return false;
        const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
        size_t l = nl ? (size_t)(nl - p) : (size_t)(end - p);
        line = p;
        p += l + (nl ? 1 : 0);
        if (l > 0 && line[l - 1] == '\r') l--;
        n = l;
        return true;
    }
};

// char::is_whitespace (Unicode White_Space: what `splitn(2, char::is_whitespace)` splits a FASTA header at, misc.rs:118-120)
// at the start of the valid UTF-8 text p[0 .. left): the bytes of that character, 0 if it is not whitespace.
// U+0009-000D, 0020, 0085, 00A0, 1680, 2000-200A, 2028, 2029, 202F, 205F, 3000.
inline size_t rust_ws_len(const char *p, size_t left) {
    const unsigned char c = (unsigned char)p[0];
    if (c == ' ' || (c >= 0x09 && c <= 0x0D)) return 1;
    if (c == 0xC2 && left >= 2) {
        const unsigned char d = (unsigned char)p[1];
        return d == 0x85 || d == 0xA0 ? 2 : 0;
    }
    if (left >= 3) {
        const unsigned char d = (unsigned char)p[1], e = (unsigned char)p[2];
        if (c == 0xE1 && d == 0x9A && e == 0x80) return 3;
        if (c == 0xE2 && d == 0x80 && ((e >= 0x80 && e <= 0x8A) || e == 0xA8 || e == 0xA9 || e == 0xAF)) return 3;
        if (c == 0xE2 && d == 0x81 && e == 0x9F) return 3;
        if (c == 0xE3 && d == 0x80 && e == 0x80) return 3;
    }
    return 0;
}

// str::parse::<uN>(): optional '+', ASCII digits, overflow is an error
bool parse_unsigned(const char *s, size_t n, uint64_t max, uint64_t &out) {
    size_t i = 0;
    if (n == 0) return false;
    if (s[0] == '+') {
        i = 1;
        if (n == 1) return false;
    }
    uint64_t v = 0;
    for (; i < n; i++) {
        if (s[i] < '0' || s[i] > '9') return false;
        uint64_t d = (uint64_t)(s[i] - '0');
        if (v > (max - d) / 10) return false;
        v = v * 10 + d;
    }
    out = v;
    return true;
}

// misc.rs:170-182
struct CompTable {
    unsigned char t[256];
    CompTable() {
        memset(t, 'N', sizeof t);
        const char *a = "ATGCatgcNnRYSWKMBVDHryswkmbvdh.-?";
        const char *b = "TACGtacgNnYRSWMKVBHDyrswmkvbhd.-?";
        for (size_t i = 0; a[i]; i++) t[(unsigned char)a[i]] = (unsigned char)b[i];
    }
};
const CompTable COMP;

inline int op_code(char c) {
    switch (c) {
    case 'M': return PP_OP_M; case 'I': return PP_OP_I; case 'D': return PP_OP_D;
    case 'N': return PP_OP_N; case 'S': return PP_OP_S; case 'H': return PP_OP_H;
    case 'P': return PP_OP_P; case '=': return PP_OP_EQ; case 'X': return PP_OP_X;
    default: return -1;
    }
}

}  // namespace

// =================================================================================================
struct pp_assembly {
    std::vector<std::string> names, descs;
    std::vector<uint64_t> off;
    HugeBuf<uint8_t> bases;  // (anonymous huge pages: a 250 Mbp assembly is not zero-filled before it is written)
    std::unordered_map<std::string, uint32_t> index;
};

namespace {

// One sequence line appended to the assembly, ASCII-uppercased (make_ascii_uppercase, misc.rs:114,129).  Lines of a few
// MB and more -- a chromosome written on one line -- are done by several threads: 250 Mbp took 0.7 s with one thread,
// all of it in front of the first byte of SAM text going anywhere.  Returns false if the line holds a byte >= 0x80.
bool append_upper(HugeBuf<uint8_t> &dst, const char *line, size_t n) {
    const size_t base = dst.size();
    dst.resize(base + n);
    uint8_t *out = dst.data() + base;
    auto piece = [&](size_t lo, size_t hi) {
        unsigned char any = 0;
        for (size_t i = lo; i < hi; i++) {  // branch-free: the compiler vectorises it
            const unsigned char c = (unsigned char)line[i];
            any |= c;
            out[i] = (unsigned char)(c - (((unsigned char)(c - 'a') < 26u) ? 32u : 0u));
        }
        return (any & 0x80u) == 0;
    };
    if (n < (size_t(4) << 20)) return piece(0, n);
    const unsigned threads = std::max(1u, std::min({std::thread::hardware_concurrency(), 32u, (unsigned)(n >> 21)}));
    std::vector<char> ok(threads, 1);
    parallel_for(n, threads, [&](size_t lo, size_t hi, unsigned t) { ok[t] = piece(lo, hi); });
    return std::all_of(ok.begin(), ok.end(), [](char c) { return c != 0; });
}

}  // namespace

static void load_fasta(const char *path, pp_assembly &a) {
    // is_file_gzipped, misc.rs:81-99
    FILE *f = fopen(path, "rb");
    if (!f) fail(PP_ERR_QUIT, "unable to open \"%s\"", path);
    unsigned char magic[2];
    size_t got = fread(magic, 1, 2, f);
    fclose(f);
    if (got != 2) fail(PP_ERR_QUIT, "\"%s\" is too small", path);
    std::vector<char> text;
    pph::FileText mapped;  // a plain file is parsed out of its mapping, a gzipped one out of the inflated copy
    const char *tbeg, *tend;
    if (magic[0] == 31 && magic[1] == 139) {
        if (!read_gz(path, text)) fail(PP_ERR_QUIT, "unable to load \"%s\"", path);
        tbeg = text.data(); tend = text.data() + text.size();
    } else {
        if (!mapped.open_file(path)) fail(PP_ERR_QUIT, "unable to load \"%s\"", path);
        tbeg = mapped.text; tend = mapped.text + mapped.size;
    }
    a.bases.reserve((size_t)(tend - tbeg) + 64);

    LineReader lr{tbeg, tend};
    const char *line;
    size_t n;
    std::string name, desc;
    bool have = false;  // name.len() > 0
    std::vector<uint64_t> lens;
    a.off.push_back(0);
    auto push = [&]() {
        a.names.push_back(name);
        a.descs.push_back(desc);
        a.off.push_back(a.bases.size());
    };
    while (lr.next(line, n)) {
        if (n == 0) continue;
        const bool header = line[0] == '>';
        // lines() fails on a line that is not valid UTF-8 (misc.rs:109-111).  A sequence line is copied first and only
        // looked at closely if the copy saw a byte outside ASCII.
        if (header && !pph::valid_utf8(line, n)) fail(PP_ERR_QUIT, "unable to load \"%s\"", path);
        if (header) {
            if (have) push();
            size_t i = 1, wl = 0;
            while (i < n && !(wl = rust_ws_len(line + i, n - i))) i++;
            name.assign(line + 1, i - 1);
            desc = i < n ? std::string(line + i + wl, n - i - wl) : std::string();
            have = !name.empty();
        } else {
            if (!have) {
                if (!pph::valid_utf8(line, n)) fail(PP_ERR_QUIT, "unable to load \"%s\"", path);
                fail(PP_ERR_QUIT, "\"%s\" is not correctly formatted", path);
            }
            if (!append_upper(a.bases, line, n)) {
                if (!pph::valid_utf8(line, n)) fail(PP_ERR_QUIT, "unable to load \"%s\"", path);
                fail(PP_ERR_LIMIT, "\"%s\" contains a non-ASCII byte in a sequence line "
                                   "(not supported by this implementation)", path);
            }
        }
    }
    if (have) push();
    // check_load_fasta, misc.rs:56-75
    if (a.names.empty()) fail(PP_ERR_QUIT, "\"%s\" contains no sequences", path);
    for (size_t i = 0; i < a.names.size(); i++) {
        if (a.names[i].empty()) fail(PP_ERR_QUIT, "\"%s\" has an unnamed sequence", path);
        if (a.off[i + 1] == a.off[i]) fail(PP_ERR_QUIT, "\"%s\" has an empty sequence", path);
    }
    for (size_t i = 0; i < a.names.size(); i++)
        if (!a.index.emplace(a.names[i], (uint32_t)i).second)
            fail(PP_ERR_QUIT, "\"%s\" has a duplicated name", path);
}

extern "C" int pp_assembly_load(const char *path, pp_assembly **out, char *err, size_t errlen) {
    if (!path || !out) return PP_ERR_ARG;
    *out = nullptr;
    pp_assembly *a = new pp_assembly();
    try {
        load_fasta(path, *a);
    } catch (const IngestError &e) {
        if (err && errlen) snprintf(err, errlen, "%s", e.msg.c_str());
        delete a;
        return e.code;
    }
    *out = a;
    return PP_OK;
}
extern "C" void pp_assembly_free(pp_assembly *a) { delete a; }
extern "C" uint32_t pp_assembly_n_contigs(const pp_assembly *a) { return a ? (uint32_t)a->names.size() : 0; }
extern "C" const char *pp_assembly_name(const pp_assembly *a, uint32_t i) { return a->names[i].c_str(); }
extern "C" const char *pp_assembly_description(const pp_assembly *a, uint32_t i) { return a->descs[i].c_str(); }
extern "C" const uint64_t *pp_assembly_offsets(const pp_assembly *a) { return a->off.data(); }
extern "C" const uint8_t *pp_assembly_bases(const pp_assembly *a) { return a->bases.data(); }

// =================================================================================================
struct pp_ingest {
    const pp_assembly *asmb;
    uint32_t max_errors;
    bool careful;
    // how the SEQ bytes of a file are laid out in the seq array (include/polypolish_hip.h, pp_dev_ingest_set_seq_layout):
    // window-grouped by default, as the device tokenizer lays them out; PP_SEQ_LAYOUT=file / pp_ingest_set_seq_layout
    int seq_layout = PP_SEQ_WINDOW_GROUPED;
    HugeBuf<uint32_t> contig, ref_start, k, seq_len, n_cig, cigar;
    HugeBuf<uint64_t> seq_off, cig_off, name_off;
    HugeBuf<uint8_t> seq;
    HugeBuf<pp_wo_rec> wo;  // the records once more, in window order (pp_aln_batch.wo); empty with PP_WO=0
    std::vector<uint64_t> wo_run_end;  // ... one run per SAM file, in ascending window order: where each ends (pp_aln_batch.wo_run_end)
    bool wo_mirror = true;
    HugeBuf<char> names;  // NUL-separated QNAMEs, one per record
    std::vector<std::thread> reapers;  // parse-time memory of finished files being released in the background
    // After a failed pp_ingest_sam: the byte offset in that file of the first line of the read group that was pending
    // when the reference's streaming loop would have stopped -- everything before it HAD been handed to the pileup
    // (alignment.rs:238-303), so a defect only the CIGAR walk finds there comes first (pp_driver.cpp).
    uint64_t fail_cut = 0;
    bool fail_has_cut = false;
    ~pp_ingest() {
        for (auto &t : reapers) {
            if (pph::process_leaving_soon()) t.detach(); else t.join();  // (the CLI exits without waiting for the unmapping)
        }
    }
};

namespace {

// One aligned SAM record (Alignment::new, alignment.rs:49-98), parsed but not yet gated.
struct Rec {
    const char *name, *ref, *seq;
    uint32_t name_n, ref_n, seq_n;
    uint64_t ref_start;
    uint32_t flags, nm;
    int32_t contig;            // index into the assembly, -1 if RNAME is not in it
    uint32_t run_lo, run_n;    // packed CIGAR runs in the chunk's pool (zero-length runs dropped)
    const uint32_t *runs;      // = pool + run_lo, set when the chunk is complete
    uint8_t pass_qc;
};

static const char NOT_UTF8[] = "\x01not-utf8";  // err_what of a line that is not valid UTF-8 (the message names the file)

// A slice of the file, parsed by one thread.
struct Chunk {
    const char *beg = nullptr, *end = nullptr;
    HugeBuf<Rec> recs;
    HugeBuf<uint32_t> runs;
    uint64_t n_lines = 0;        // lines seen (up to and including a failing one)
    int err_code = 0;            // first parse error of the chunk, if any
    std::string err_what;        // message without the "in <file> (line N)" part where that applies
    bool err_has_line = false;
    size_t err_recs = 0;         // records parsed before the failing line
};

void parse_chunk(Chunk &c, const pp_assembly *asmb) {
    const char *p = c.beg;
    std::string key;
    const std::string *last_ref = nullptr;
    int32_t last_contig = -1;
    while (p < c.end) {
        const char *nl = (const char *)memchr(p, '\n', (size_t)(c.end - p));
        size_t n = nl ? (size_t)(nl - p) : (size_t)(c.end - p);
        const char *line = p;
        p += n + (nl ? 1 : 0);
        if (n > 0 && line[n - 1] == '\r') n--;
        c.n_lines++;
        auto fail = [&](int code, const char *what, bool with_line) {
            c.err_code = code; c.err_what = what; c.err_has_line = with_line; c.err_recs = c.recs.size();
        };
        // `let sam_line = line?` (alignment.rs:240): a line that is not UTF-8 ends the load before anything looks at it
        if (!pph::valid_utf8(line, n)) { fail(PP_ERR_QUIT, NOT_UTF8, false); return; }
        if (n == 0 || line[0] == '@') continue;
        const char *col[11];
        size_t len[11];
        size_t nc = 0;
        const char *q = line, *end = line + n, *tags = nullptr;
        while (nc < 11) {
            const char *t = (const char *)memchr(q, '\t', (size_t)(end - q));
            col[nc] = q;
            len[nc] = t ? (size_t)(t - q) : (size_t)(end - q);
            nc++;
            if (!t) { q = end + 1; break; }
            q = t + 1;
        }
        if (nc < 11) { fail(PP_ERR_QUIT, "too few columns", true); return; }
        tags = (q <= end) ? q : nullptr;
        uint64_t flags, pos;
        if (!parse_unsigned(col[1], len[1], 0xFFFFFFFFull, flags)) { fail(PP_ERR_PANIC, "could not parse the FLAG column as u32", true); return; }
        if (!parse_unsigned(col[3], len[3], UINT64_MAX, pos)) { fail(PP_ERR_PANIC, "could not parse the POS column", true); return; }
        if (pos > 0) pos -= 1;
        uint32_t nm = 0xFFFFFFFFu;
        bool pass_qc = true, bad_nm = false;
        while (tags && tags <= end) {
            const char *t = (const char *)memchr(tags, '\t', (size_t)(end - tags));
            size_t tl = t ? (size_t)(t - tags) : (size_t)(end - tags);
            if (tl >= 5 && memcmp(tags, "NM:i:", 5) == 0) {
                uint64_t v;
                if (!parse_unsigned(tags + 5, tl - 5, 0xFFFFFFFFull, v)) { bad_nm = true; break; }
                nm = (uint32_t)v;
            }
            if (tl == 9 && strncasecmp(tags, "ZP:Z:fail", 9) == 0) pass_qc = false;
            if (!t) break;
            tags = t + 1;
        }
        if (bad_nm) { fail(PP_ERR_PANIC, "could not parse the NM tag", true); return; }
        if (nm == 0xFFFFFFFFu && (flags & 4) == 0) { fail(PP_ERR_QUIT, "missing NM tag", true); return; }
        // get_expanded_cigar, alignment.rs:325-346: the whole string must be \d+[MIDNSHP=X] tokens
        const uint32_t run_lo = (uint32_t)c.runs.size();
        const char *cg = col[5];
        const size_t cl = len[5];
        if (!(cl == 1 && cg[0] == '*')) {
            size_t i = 0;
            bool ok = true, overflow = false;
            while (i < cl) {
                size_t j = i;
                while (j < cl && cg[j] >= '0' && cg[j] <= '9') j++;
                int op = (j < cl) ? op_code(cg[j]) : -1;
                if (j == i || op < 0) { ok = false; break; }
                uint64_t num;
                if (!parse_unsigned(cg + i, j - i, 0xFFFFFFFFull, num)) { overflow = true; break; }
                while (num > 0) {  // a packed run holds 28 bits of length
                    uint32_t piece = num > 0x0FFFFFFFull ? 0x0FFFFFFFu : (uint32_t)num;
                    c.runs.push_back((piece << 4) | (uint32_t)op);
                    num -= piece;
                }
                i = j + 1;
            }
            if (overflow) { c.runs.resize(run_lo); fail(PP_ERR_PANIC, "CIGAR run length does not fit u32", true); return; }
            if (!ok) {
                c.runs.resize(run_lo);
                char m[600];
                snprintf(m, sizeof m, "encountered an invalid CIGAR string for read %.*s: \"%.*s\"", (int)len[0], col[0], (int)cl, cg);
                fail(PP_ERR_QUIT, m, false);
                return;
            }
        }
        if (flags & 4) { c.runs.resize(run_lo); continue; }  // unaligned: parsed, then skipped (alignment.rs:250)
        Rec r;
        r.name = col[0]; r.name_n = (uint32_t)len[0];
        r.ref = col[2]; r.ref_n = (uint32_t)len[2];
        r.seq = col[9]; r.seq_n = (uint32_t)len[9];
        r.ref_start = pos;
        r.flags = (uint32_t)flags;
        r.nm = nm;
        r.pass_qc = pass_qc;
        r.run_lo = run_lo;
        r.run_n = (uint32_t)c.runs.size() - run_lo;
        if (last_ref && last_ref->size() == len[2] && memcmp(last_ref->data(), col[2], len[2]) == 0) {
            r.contig = last_contig;
        } else {
            key.assign(col[2], len[2]);
            auto it = asmb->index.find(key);
            r.contig = it == asmb->index.end() ? -1 : (int32_t)it->second;
            if (it != asmb->index.end()) { last_ref = &it->first; last_contig = r.contig; }
        }
        r.runs = nullptr;
        c.recs.push_back(r);
    }
}

struct OutRec {            // a good alignment, ready to be copied into the batch
    const Rec *rec, *src;  // src: the group's record that carries the SEQ (for "*" fills)
    const uint32_t *runs;
    uint32_t k;
    bool star, revcomp;
};

}  // namespace

extern "C" int pp_ingest_create(const pp_assembly *a, uint32_t max_errors, int careful, pp_ingest **out) {
    if (!a || !out) return PP_ERR_ARG;
    pp_ingest *g = new pp_ingest();
    g->asmb = a;
    g->max_errors = max_errors;
    g->careful = careful != 0;
    if (const char *e = getenv("PP_SEQ_LAYOUT")) g->seq_layout = !strcmp(e, "file") ? PP_SEQ_FILE_ORDER : PP_SEQ_WINDOW_GROUPED;
    if (const char *e = getenv("PP_WO")) g->wo_mirror = atoi(e) != 0;
    *out = g;
    return PP_OK;
}

extern "C" int pp_ingest_set_seq_layout(pp_ingest *g, int layout) {
    if (!g || (layout != PP_SEQ_FILE_ORDER && layout != PP_SEQ_WINDOW_GROUPED)) return PP_ERR_ARG;
    g->seq_layout = layout;
    return PP_OK;
}

// add_to_pileup (alignment.rs:225-272) + process_one_read (alignment.rs:275-322), multi-threaded:
// the text is parsed in parallel slices, the grouping/gates run once over the parsed records in file
// order (so errors surface in the order the reference's streaming loop would hit them), and the SoA
// is filled in parallel.  The result does not depend on the thread count.
extern "C" int pp_ingest_sam(pp_ingest *I, const char *path, pp_sam_counts *counts, char *err, size_t errlen) {
    return pp_ingest_sam_filtered(I, path, nullptr, 0, counts, err, errlen);
}

static int ingest_impl(pp_ingest *I, const char *path, const char *ext_text, size_t ext_size, uint64_t line_base,
                       const uint8_t *pass, uint64_t n_pass, pp_sam_counts *counts, char *err, size_t errlen,
                       uint64_t prefix_bytes = ~0ull);

extern "C" int pp_ingest_sam_filtered(pp_ingest *I, const char *path, const uint8_t *pass, uint64_t n_pass,
                                      pp_sam_counts *counts, char *err, size_t errlen) {
    if (!I || !path) return PP_ERR_ARG;
    return ingest_impl(I, path, nullptr, 0, 0, pass, n_pass, counts, err, errlen);
}

// Internal (the device tokenizer's error path): the same ingest over a slice of text already in memory;
// line numbers in messages start at line_base + 1.
extern "C" int pp_ingest_text_(pp_ingest *I, const char *path, const char *text, size_t size, uint64_t line_base,
                               pp_sam_counts *counts, char *err, size_t errlen) {
    if (!I || !path || !text) return PP_ERR_ARG;
    return ingest_impl(I, path, text, size, line_base, nullptr, 0, counts, err, errlen);
}

// Internal (pp_driver.cpp, error path): was the failure of the last pp_ingest_sam somewhere the records before it are
// known?  *cut = bytes of the file that hold exactly the read groups the reference had processed by then.
extern "C" int pp_ingest_fail_cut_(const pp_ingest *I, uint64_t *cut) {
    if (!I || !cut || !I->fail_has_cut) return 0;
    *cut = I->fail_cut;
    return 1;
}

// Internal: ingest only the first `cut` bytes of the file (whole read groups by construction; none at all is fine).
extern "C" int pp_ingest_sam_prefix_(pp_ingest *I, const char *path, uint64_t cut, const uint8_t *pass, uint64_t n_pass,
                                     pp_sam_counts *counts, char *err, size_t errlen) {
    if (!I || !path) return PP_ERR_ARG;
    return ingest_impl(I, path, nullptr, 0, 0, pass, n_pass, counts, err, errlen, cut);
}

static int ingest_impl(pp_ingest *I, const char *path, const char *ext_text, size_t ext_size, uint64_t line_base,
                       const uint8_t *pass, uint64_t n_pass, pp_sam_counts *counts, char *err, size_t errlen,
                       uint64_t prefix_bytes) {
    pp_sam_counts c{0, 0, 0};
    const bool prefix_mode = prefix_bytes != ~0ull;
    I->fail_has_cut = false;
    int fd = -1;
    void *map = nullptr;
    size_t map_len = 0;
    std::vector<char> fallback;
    auto cleanup = [&] {
        if (map) munmap(map, map_len);
        if (fd >= 0) close(fd);
    };
    try {
        const char *text = ext_text;
        size_t size = ext_size;
        struct stat st;
        if (ext_text) {
            // the caller's memory
        } else if ((fd = open(path, O_RDONLY)) < 0 || fstat(fd, &st) != 0) {
            fail(PP_ERR_QUIT, "unable to load alignments from \"%s\"", path);
        } else if (S_ISREG(st.st_mode) && st.st_size > 0) {
            map_len = (size_t)st.st_size;
            map = mmap(nullptr, map_len, PROT_READ, MAP_PRIVATE, fd, 0);
            if (map == MAP_FAILED) { map = nullptr; fail(PP_ERR_QUIT, "unable to load alignments from \"%s\"", path); }
            madvise(map, map_len, MADV_SEQUENTIAL);
            text = (const char *)map;
            size = map_len;
        } else {  // pipe or empty file: read the descriptor that is already open (a FIFO cannot be opened twice)
            char tmp[1 << 16];
            ssize_t r;
            while ((r = read(fd, tmp, sizeof tmp)) > 0) fallback.insert(fallback.end(), tmp, tmp + r);
            if (r < 0) fail(PP_ERR_QUIT, "unable to load alignments from \"%s\"", path);
            text = fallback.data();
            size = fallback.size();
        }
        if (prefix_mode) size = std::min<size_t>(size, (size_t)prefix_bytes);
        const bool regular_input = ext_text == nullptr && map != nullptr;  // a file that can be read again (not a pipe)
        const bool timing = getenv("PP_TIMING") != nullptr;
        auto t_last = std::chrono::steady_clock::now();
        auto lap = [&](const char *what) {
            if (!timing) return;
            auto now = std::chrono::steady_clock::now();
            fprintf(stderr, "[timing]   ingest: %-22s %.3f s\n", what, std::chrono::duration<double>(now - t_last).count());
            t_last = now;
        };
        unsigned threads;
        if (const char *e = getenv("PP_INGEST_THREADS")) threads = std::max(1, std::min(64, atoi(e)));
        else threads = std::max(1u, std::min({std::thread::hardware_concurrency(), 64u, (unsigned)(size / (4u << 20)) + 1u}));

        // ---- parallel parse of line-aligned slices ----
        std::vector<Chunk> chunks(threads);
        {
            const char *p = text, *end = text + size;
            for (unsigned t = 0; t < threads; t++) {
                chunks[t].beg = p;
                const char *want = (t + 1 == threads) ? end : text + (size / threads) * (t + 1);
                if (want < p) want = p;
                if (want < end) {
                    const char *nl = (const char *)memchr(want, '\n', (size_t)(end - want));
                    want = nl ? nl + 1 : end;
                }
                chunks[t].end = want;
                p = want;
            }
        }
        parallel_for(threads, threads, [&](size_t lo, size_t hi, unsigned) {
            for (size_t t = lo; t < hi; t++) parse_chunk(chunks[t], I->asmb);
        });
        lap("parse");
        // the first slice with a parse error bounds what the streaming loop would have processed
        size_t n_chunks_ok = threads;
        for (unsigned t = 0; t < threads; t++)
            if (chunks[t].err_code) { n_chunks_ok = t; break; }

        // ---- grouping and gates ----
        // Whether record i opens a new read group depends only on records i-1 and i (alignment.rs:255:
        // it joins when the previous QNAME is empty or equal), so the groups can be cut in parallel:
        // each part processes the groups that START inside its range, in file order.
        size_t total_recs = 0;
        std::vector<size_t> first(threads + 1, 0);
        for (size_t t = 0; t < threads; t++) {
            const size_t nrec = t > n_chunks_ok ? 0 : (chunks[t].err_code ? chunks[t].err_recs : chunks[t].recs.size());
            first[t + 1] = first[t] + nrec;
        }
        total_recs = first[threads];
        HugeBuf<const Rec *> all;
        all.resize(total_recs);
        parallel_for(threads, threads, [&](size_t lo, size_t hi, unsigned) {
            for (size_t t = lo; t < hi; t++) {
                Chunk &ch = chunks[t];
                const size_t nrec = first[t + 1] - first[t];
                for (size_t i = 0; i < nrec; i++) {
                    ch.recs[i].runs = ch.runs.data() + ch.recs[i].run_lo;
                    all[first[t] + i] = &ch.recs[i];
                    // the filter's verdict for this aligned record, as if "ZP:Z:fail" were on its line
                    if (pass && first[t] + i < n_pass && !pass[first[t] + i]) ch.recs[i].pass_qc = 0;
                }
            }
        });
        lap("flatten");
        if (pass && !prefix_mode && n_chunks_ok == threads && n_pass != total_recs)
            fail(PP_ERR_ARG, "%llu filter verdicts for the %llu aligned records of \"%s\"", (unsigned long long)n_pass,
                 (unsigned long long)total_recs, path);
        const bool parse_failed = n_chunks_ok < threads;  // the group pending at the failing line is never processed
        auto is_start = [&](size_t i) {
            if (i == 0) return true;
            const Rec &p = *all[i - 1], &r = *all[i];
            return !(p.name_n == 0 || (p.name_n == r.name_n && memcmp(p.name, r.name, r.name_n) == 0));
        };
        struct Part {
            HugeBuf<OutRec> outs;
            uint64_t reads = 0, used = 0, seq_sum = 0, cig_sum = 0, nam_sum = 0;
            size_t cur_g0 = 0;  // first record of the group being processed (where a failure leaves off)
            int err_code = 0;
            std::string err_msg;
        };
        std::vector<Part> parts(threads);
        auto process_one_read = [&](Part &P, size_t g0, size_t g1) {  // alignment.rs:275-322 on records [g0, g1)
            const size_t n = g1 - g0;
            if (I->careful && n > 1) return;
            const Rec *src = nullptr;
            for (size_t i = g0; i < g1; i++)
                if (!(all[i]->seq_n == 1 && all[i]->seq[0] == '*')) { src = all[i]; break; }
            if (!src) fail(PP_ERR_QUIT, "no alignments for read %.*s contain sequence", (int)all[g0]->name_n, all[g0]->name);
            const bool src_fwd = (src->flags & 16) == 0;
            uint32_t n_good = 0;
            const size_t first_out = P.outs.size();
            for (size_t i = g0; i < g1; i++) {
                const Rec &a = *all[i];
                if (a.run_n == 0)  // chars().next().unwrap() on an empty expanded CIGAR
                    fail(PP_ERR_PANIC, "aligned record of read %.*s has an empty CIGAR", (int)a.name_n, a.name);
                const uint32_t f = a.runs[0] & 15u, l = a.runs[a.run_n - 1] & 15u;
                const bool ends_ok = (f == PP_OP_M || f == PP_OP_EQ) && (l == PP_OP_M || l == PP_OP_EQ);
                if (!(ends_ok && a.nm <= I->max_errors && a.pass_qc)) continue;
                OutRec o;
                o.rec = &a; o.src = src; o.runs = a.runs; o.k = 0;
                o.star = a.seq_n == 1 && a.seq[0] == '*';
                o.revcomp = o.star && ((a.flags & 16) == 0) != src_fwd;
                P.outs.push_back(o);
                n_good++;
            }
            for (size_t i = first_out; i < P.outs.size(); i++) {
                OutRec &o = P.outs[i];
                const Rec &a = *o.rec;
                if (a.contig < 0) fail(PP_ERR_QUIT, "query name %.*s in SAM but not in assembly", (int)a.ref_n, a.ref);
                if (a.ref_start > 0xFFFFFFFEull)
                    fail(PP_ERR_PANIC, "alignment of read %.*s starts past the end of %.*s", (int)a.name_n, a.name, (int)a.ref_n, a.ref);
                o.k = n_good;
                P.seq_sum += seq_room(o.star ? o.src->seq_n : a.seq_n);
                P.cig_sum += a.run_n;
                P.nam_sum += a.name_n + 1;
            }
            P.used += n_good;
        };
        parallel_for(total_recs, threads, [&](size_t lo, size_t hi, unsigned t) {
            Part &P = parts[t];
            size_t i = lo;
            while (i < hi && !is_start(i)) i++;  // the tail of a group that started in an earlier part
            try {
                while (i < hi) {
                    size_t j = i + 1;
                    while (j < total_recs && !is_start(j)) j++;
                    if (j == total_recs && parse_failed) break;
                    P.cur_g0 = i;
                    process_one_read(P, i, j);
                    P.reads++;
                    i = j;
                }
            } catch (const IngestError &e) {
                P.err_code = e.code;
                P.err_msg = e.msg;
            }
        });
        c.alignments = total_recs;
        for (const Part &P : parts) {  // parts are in file order: the first failing group is the one the reference hits
            c.reads += P.reads;
            c.used += P.used;
            if (P.err_code) {
                if (regular_input) {  // every group before the failing one had been processed
                    I->fail_cut = (uint64_t)(all[P.cur_g0]->name - text);
                    I->fail_has_cut = true;
                }
                throw IngestError{P.err_code, P.err_msg};
            }
        }
        if (parse_failed && regular_input) {  // ... and so had every group before the one pending at the failing line
            size_t i = total_recs;
            if (i) {
                i--;
                while (i > 0 && !is_start(i)) i--;
                I->fail_cut = (uint64_t)(all[i]->name - text);
            } else {
                I->fail_cut = 0;
            }
            I->fail_has_cut = true;
        }
        if (parse_failed) {  // the streaming loop would have stopped at the failing line
            const Chunk &ch = chunks[n_chunks_ok];
            uint64_t line_no = line_base + ch.n_lines;
            for (size_t u = 0; u < n_chunks_ok; u++) line_no += chunks[u].n_lines;
            if (ch.err_has_line) fail(ch.err_code, "%s in \"%s\" (line %llu)", ch.err_what.c_str(), path, (unsigned long long)line_no);
            if (ch.err_what == NOT_UTF8) fail(ch.err_code, "unable to load alignments from \"%s\"", path);
            fail(ch.err_code, "%s", ch.err_what.c_str());
        }
        if (total_recs == 0 && prefix_mode) {  // nothing had been processed before the failure
            cleanup();
            if (counts) *counts = c;
            return PP_OK;
        }
        if (total_recs == 0)  // process_one_read on an empty group after the loop (alignment.rs:268)
            fail(PP_ERR_PANIC, "no aligned records to process (the reference panics on an empty read group)");
        lap("group + gates");

        // ---- parallel fill of the structure of arrays: every part copies its own good alignments ----
        std::vector<uint64_t> p_out(threads + 1, 0), p_seq(threads + 1, 0), p_cig(threads + 1, 0), p_nam(threads + 1, 0);
        for (size_t t = 0; t < threads; t++) {
            p_out[t + 1] = p_out[t] + parts[t].outs.size();
            p_seq[t + 1] = p_seq[t] + parts[t].seq_sum;
            p_cig[t + 1] = p_cig[t] + parts[t].cig_sum;
            p_nam[t + 1] = p_nam[t] + parts[t].nam_sum;
        }
        const size_t n_out = p_out[threads], base = I->contig.size();
        const uint64_t seq0 = I->seq.size(), cig0 = I->cigar.size(), nam0 = I->names.size();
        I->contig.resize(base + n_out); I->ref_start.resize(base + n_out); I->k.resize(base + n_out);
        I->seq_off.resize(base + n_out); I->seq_len.resize(base + n_out); I->cig_off.resize(base + n_out);
        I->n_cig.resize(base + n_out); I->name_off.resize(base + n_out);
        I->seq.resize(seq0 + p_seq[threads]); I->cigar.resize(cig0 + p_cig[threads]); I->names.resize(nam0 + p_nam[threads]);
        lap("offsets + resize");
        // Window-grouped SEQ layout (the default, as the device tokenizer's): the reads that start in one 2048-position
        // window of the assembly are adjacent in this file's stretch of the seq array -- the pileup kernel then fetches a
        // window's reads from one place.  A multisplit of the rooms: every part adds its records' rooms up per window, the
        // windows' totals are scanned, and a part's records of a window follow those of the parts before it, in file order
        // (so the layout does not depend on the thread count).  seq_off goes with the record; nothing else changes.
        constexpr uint64_t WINDOW = 2048;  // = pp::TILE (pp_internal.h), the pileup kernel's window
        const uint64_t *ctg_off = pp_assembly_offsets(I->asmb);
        const uint64_t G_asm = ctg_off[pp_assembly_n_contigs(I->asmb)];
        const size_t n_win = (size_t)std::max<uint64_t>(1, (G_asm + WINDOW - 1) / WINDOW);
        const bool grouped = I->seq_layout == PP_SEQ_WINDOW_GROUPED && n_out > 0;
        const bool mirror = I->wo_mirror && n_out > 0;  // the window-order mirror of the records: the same multisplit, counting records
        auto window_of = [&](const Rec &a) {
            const uint64_t w = (ctg_off[a.contig] + a.ref_start) / WINDOW;
            return (size_t)std::min<uint64_t>(w, n_win - 1);
        };
        // One row of the two tables for several parts in a row (file order) when a row per part would be large: the tables
        // are [rows][windows] of 8 bytes -- 64 threads over a 3 Gbp assembly would be 1.5 GB of them per SAM file, cleared and
        // scanned whatever the file holds -- so beyond 32 M entries a table the parts share rows, and the parts of one row are
        // filled by one thread, one after the other.  (PP_INGEST_ROWS: tuning / tests.)
        const long forced_rows = getenv("PP_INGEST_ROWS") ? atol(getenv("PP_INGEST_ROWS")) : 0;
        const size_t rows = forced_rows > 0 ? std::min<size_t>(threads, (size_t)forced_rows)
                                            : std::min<size_t>(threads, std::max<size_t>(1, ((size_t)32 << 20) / n_win));
        auto row_first = [&](size_t r) { return (r * threads + rows - 1) / rows; };  // the parts of row r: [row_first(r), row_first(r + 1))
        HugeBuf<uint64_t> wcur;  // [row][window]: first the bytes, then where the row's next record of the window goes
        HugeBuf<uint64_t> wcnt;  // [row][window]: the same for the records themselves (slots of the window-order mirror)
        if (grouped || mirror) {
            wcur.resize(rows * n_win);
            wcnt.resize(rows * n_win);
            parallel_for(rows, threads, [&](size_t lo, size_t hi, unsigned) {
                for (size_t r = lo; r < hi; r++) {
                    uint64_t *row = wcur.data() + r * n_win, *crow = wcnt.data() + r * n_win;
                    memset(row, 0, n_win * sizeof(uint64_t));
                    memset(crow, 0, n_win * sizeof(uint64_t));
                    for (size_t t = row_first(r); t < row_first(r + 1); t++)
                        for (size_t i = 0; i < parts[t].outs.size(); i++) {
                            const OutRec &o = parts[t].outs[i];
                            const size_t w = window_of(*o.rec);
                            row[w] += seq_room(o.star ? o.src->seq_n : o.rec->seq_n);
                            crow[w] += 1;
                        }
                }
            });
            std::vector<uint64_t> wtot(n_win + 1, 0), ctot(n_win + 1, 0);
            parallel_for(n_win, threads, [&](size_t lo, size_t hi, unsigned) {
                for (size_t w = lo; w < hi; w++) {
                    uint64_t sum = 0, cs = 0;
                    for (size_t t = 0; t < rows; t++) { sum += wcur[t * n_win + w]; cs += wcnt[t * n_win + w]; }
                    wtot[w + 1] = sum;
                    ctot[w + 1] = cs;
                }
            });
            for (size_t w = 0; w < n_win; w++) { wtot[w + 1] += wtot[w]; ctot[w + 1] += ctot[w]; }
            parallel_for(n_win, threads, [&](size_t lo, size_t hi, unsigned) {
                for (size_t w = lo; w < hi; w++) {
                    uint64_t run = seq0 + wtot[w], crun = base + ctot[w];
                    for (size_t t = 0; t < rows; t++) {
                        const uint64_t v = wcur[t * n_win + w], c = wcnt[t * n_win + w];
                        wcur[t * n_win + w] = run;
                        wcnt[t * n_win + w] = crun;
                        run += v;
                        crun += c;
                    }
                }
            });
            if (mirror) {
                I->wo.resize(base + n_out);
                I->wo_run_end.push_back(base + n_out);  // this file's entries: one run
            }
            lap("window layout");
        }
        parallel_for(rows, threads, [&](size_t lo, size_t hi, unsigned) {
            for (size_t row = lo; row < hi; row++)
            for (size_t t = row_first(row); t < row_first(row + 1); t++) {
                const Part &P = parts[t];
                uint64_t so = seq0 + p_seq[t], co = cig0 + p_cig[t], no = nam0 + p_nam[t];
                for (size_t i = 0; i < P.outs.size(); i++) {
                    const OutRec &o = P.outs[i];
                    const Rec &a = *o.rec;
                    const size_t d = base + p_out[t] + i;
                    I->contig[d] = (uint32_t)a.contig;
                    I->ref_start[d] = (uint32_t)a.ref_start;
                    I->k[d] = o.k;
                    const char *s = o.star ? o.src->seq : a.seq;
                    const size_t sn = o.star ? o.src->seq_n : a.seq_n;
                    if (grouped) {  // its place in its window's region
                        uint64_t &cur = wcur[row * n_win + window_of(a)];
                        so = cur;
                        cur += seq_room(sn);
                    }
                    I->seq_off[d] = so;
                    I->seq_len[d] = (uint32_t)sn;
                    uint8_t *dst = I->seq.data() + so;
                    if (o.revcomp) {  // add_read_seq (alignment.rs:161-167): reverse complement of the upper-cased group SEQ
                        for (size_t j = 0; j < sn; j++) {
                            unsigned char ch = (unsigned char)s[sn - 1 - j];
                            if (ch >= 'a' && ch <= 'z') ch = (unsigned char)(ch - 32);
                            dst[j] = COMP.t[ch];
                        }
                    } else {
                        for (size_t j = 0; j < sn; j++) {
                            unsigned char ch = (unsigned char)s[j];
                            dst[j] = (ch >= 'a' && ch <= 'z') ? (unsigned char)(ch - 32) : ch;  // to_ascii_uppercase
                        }
                    }
                    memset(dst + sn, 0, seq_room(sn) - sn);  // every record's SEQ starts on a PP_SEQ_ALIGN boundary
                    so += seq_room(sn);
                    I->cig_off[d] = co;
                    I->n_cig[d] = a.run_n;
                    memcpy(I->cigar.data() + co, o.runs, (size_t)a.run_n * 4);
                    if (mirror) {  // the record once more, at its place in window order (file order inside a window)
                        pp_wo_rec w;
                        w.contig = (uint32_t)a.contig; w.ref_start = (uint32_t)a.ref_start; w.k = o.k; w.seq_len = (uint32_t)sn;
                        w.seq_off = I->seq_off[d];
                        w.op0 = a.run_n == 1 ? o.runs[0] : PP_WO_MULTI_RUN;
                        w.file_idx = (uint32_t)d;
                        I->wo[wcnt[row * n_win + window_of(a)]++] = w;
                    }
                    co += a.run_n;
                    I->name_off[d] = no;
                    memcpy(I->names.data() + no, a.name, a.name_n);
                    I->names[no + a.name_n] = '\0';
                    no += a.name_n + 1;
                }
            }
        });
        lap("fill");
        // the parsed records, the group lists and the file mapping are released in the background
        struct Garbage {
            std::vector<Chunk> chunks;
            std::vector<Part> parts;
            HugeBuf<const Rec *> all;
            void *map;
            size_t map_len;
            int fd;
        };
        Garbage *gb = new Garbage{std::move(chunks), std::move(parts), std::move(all), map, map_len, fd};
        map = nullptr;
        fd = -1;
        I->reapers.emplace_back([gb] {
            if (gb->map) munmap(gb->map, gb->map_len);
            if (gb->fd >= 0) close(gb->fd);
            delete gb;
        });
    } catch (const IngestError &e) {
        if (err && errlen) snprintf(err, errlen, "%s", e.msg.c_str());
        if (counts) *counts = c;
        cleanup();
        return e.code;
    }
    cleanup();
    if (counts) *counts = c;
    return PP_OK;
}

extern "C" void pp_ingest_batch(const pp_ingest *I, pp_aln_batch *out) {
    out->n_aln = I->contig.size();
    out->contig = I->contig.data();
    out->ref_start = I->ref_start.data();
    out->k = I->k.data();
    out->seq_off = I->seq_off.data();
    out->seq_len = I->seq_len.data();
    out->cig_off = I->cig_off.data();
    out->n_cig = I->n_cig.data();
    out->seq = I->seq.data();
    out->seq_bytes = I->seq.size();
    out->seq4 = nullptr;
    out->wo = I->wo_mirror && I->wo.size() == I->contig.size() && I->contig.size() ? I->wo.data() : nullptr;
    pp_mirror_register_(I, out->wo, out->wo ? I->wo.size() * sizeof(pp_wo_rec) : 0);  // (one of the library's own: pp_polish_add takes it unchecked)
    const bool runs = out->wo && !I->wo_run_end.empty() && I->wo_run_end.back() == I->contig.size();
    out->wo_n_runs = runs ? (uint32_t)I->wo_run_end.size() : 0;
    out->wo_run_end = runs ? I->wo_run_end.data() : nullptr;
    out->cigar = I->cigar.data();
    out->n_cig_total = I->cigar.size();
}

extern "C" const char *pp_ingest_read_name(const pp_ingest *I, uint64_t i) {
    if (!I || i >= I->name_off.size()) return "?";
    return I->names.data() + I->name_off[i];
}

extern "C" void pp_ingest_free(pp_ingest *I) {
    pp_mirror_forget_(I);
    delete I;
}
