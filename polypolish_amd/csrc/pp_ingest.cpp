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
#include <zlib.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "polypolish_hip.h"

namespace {

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

inline bool rust_ws(char c) { return c == ' ' || (c >= 0x09 && c <= 0x0D); }

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
    std::vector<uint8_t> bases;
    std::unordered_map<std::string, uint32_t> index;
};

static void load_fasta(const char *path, pp_assembly &a) {
    // is_file_gzipped, misc.rs:81-99
    FILE *f = fopen(path, "rb");
    if (!f) fail(PP_ERR_QUIT, "unable to open \"%s\"", path);
    unsigned char magic[2];
    size_t got = fread(magic, 1, 2, f);
    fclose(f);
    if (got != 2) fail(PP_ERR_QUIT, "\"%s\" is too small", path);
    std::vector<char> text;
    bool ok = (magic[0] == 31 && magic[1] == 139) ? read_gz(path, text) : read_file(path, text);
    if (!ok) fail(PP_ERR_QUIT, "unable to load \"%s\"", path);

    LineReader lr{text.data(), text.data() + text.size()};
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
        if (line[0] == '>') {
            if (have) push();
            size_t i = 1;
            while (i < n && !rust_ws(line[i])) i++;
            name.assign(line + 1, i - 1);
            desc = i < n ? std::string(line + i + 1, n - i - 1) : std::string();
            have = !name.empty();
        } else {
            if (!have) fail(PP_ERR_QUIT, "\"%s\" is not correctly formatted", path);
            size_t base = a.bases.size();
            a.bases.resize(base + n);
            for (size_t i = 0; i < n; i++) {
                unsigned char c = (unsigned char)line[i];
                if (c >= 0x80)
                    fail(PP_ERR_LIMIT, "\"%s\" contains a non-ASCII byte in a sequence line "
                                       "(not supported by this implementation)", path);
                if (c >= 'a' && c <= 'z') c = (unsigned char)(c - 32);  // make_ascii_uppercase
                a.bases[base + i] = c;
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
    std::vector<uint32_t> contig, ref_start, k, seq_len, n_cig, cigar;
    std::vector<uint64_t> seq_off, cig_off, name_off;
    std::vector<uint8_t> seq;
    std::vector<char> names;  // NUL-separated QNAMEs, one per record
};

namespace {

struct Parsed {  // one aligned SAM record of the current read group
    const char *name;
    size_t name_n;
    const char *ref;
    size_t ref_n;
    uint32_t flags;
    uint64_t ref_start;
    const char *seq;
    size_t seq_n;
    uint32_t nm;
    bool pass_qc;
    uint32_t run_lo, run_hi;  // into the group's run pool (zero-length runs dropped)
};

struct Group {
    std::vector<Parsed> al;
    std::vector<uint32_t> runs;
    void clear() {
        al.clear();
        runs.clear();
    }
};

// Alignment::new, alignment.rs:49-98.  Returns false for an unaligned record (skipped by the
// caller after a successful parse, alignment.rs:250).
bool parse_line(const char *line, size_t n, const char *path, uint64_t line_no, Group &g) {
    const char *col[12];
    size_t len[12];
    size_t nc = 0;
    const char *p = line, *end = line + n;
    const char *tags = nullptr;
    while (nc < 11) {
        const char *t = (const char *)memchr(p, '\t', (size_t)(end - p));
        col[nc] = p;
        len[nc] = t ? (size_t)(t - p) : (size_t)(end - p);
        nc++;
        if (!t) { p = end + 1; break; }
        p = t + 1;
    }
    if (nc < 11) fail(PP_ERR_QUIT, "too few columns in \"%s\" (line %llu)", path, (unsigned long long)line_no);
    tags = (p <= end) ? p : nullptr;  // start of column 12, if any

    uint64_t flags, pos;
    if (!parse_unsigned(col[1], len[1], 0xFFFFFFFFull, flags))
        fail(PP_ERR_PANIC, "could not parse the FLAG column as u32 in \"%s\" (line %llu)", path,
             (unsigned long long)line_no);
    if (!parse_unsigned(col[3], len[3], UINT64_MAX, pos))
        fail(PP_ERR_PANIC, "could not parse the POS column in \"%s\" (line %llu)", path,
             (unsigned long long)line_no);
    if (pos > 0) pos -= 1;

    uint32_t nm = 0xFFFFFFFFu;
    bool pass_qc = true;
    while (tags && tags <= end) {
        const char *t = (const char *)memchr(tags, '\t', (size_t)(end - tags));
        size_t tl = t ? (size_t)(t - tags) : (size_t)(end - tags);
        if (tl >= 5 && memcmp(tags, "NM:i:", 5) == 0) {
            uint64_t v;
            if (!parse_unsigned(tags + 5, tl - 5, 0xFFFFFFFFull, v))
                fail(PP_ERR_PANIC, "could not parse the NM tag in \"%s\" (line %llu)", path,
                     (unsigned long long)line_no);
            nm = (uint32_t)v;
        }
        if (tl == 9 && strncasecmp(tags, "ZP:Z:fail", 9) == 0) pass_qc = false;
        if (!t) break;
        tags = t + 1;
    }
    if (nm == 0xFFFFFFFFu && (flags & 4) == 0)
        fail(PP_ERR_QUIT, "missing NM tag in \"%s\" (line %llu)", path, (unsigned long long)line_no);

    // get_expanded_cigar, alignment.rs:325-346: the whole string must be \d+[MIDNSHP=X] tokens
    uint32_t run_lo = (uint32_t)g.runs.size();
    const char *c = col[5];
    size_t cl = len[5];
    if (!(cl == 1 && c[0] == '*')) {
        size_t i = 0;
        bool ok = true;
        while (i < cl) {
            size_t j = i;
            while (j < cl && c[j] >= '0' && c[j] <= '9') j++;
            int op = (j < cl) ? op_code(c[j]) : -1;
            if (j == i || op < 0) { ok = false; break; }
            uint64_t num;
            if (!parse_unsigned(c + i, j - i, 0xFFFFFFFFull, num))
                fail(PP_ERR_PANIC, "CIGAR run length does not fit u32 in \"%s\" (line %llu)", path,
                     (unsigned long long)line_no);
            while (num > 0) {  // a packed run holds 28 bits of length
                uint32_t piece = num > 0x0FFFFFFFull ? 0x0FFFFFFFu : (uint32_t)num;
                g.runs.push_back((piece << 4) | (uint32_t)op);
                num -= piece;
            }
            i = j + 1;
        }
        if (!ok) {
            g.runs.resize(run_lo);
            fail(PP_ERR_QUIT, "encountered an invalid CIGAR string for read %.*s: \"%.*s\"", (int)len[0],
                 col[0], (int)cl, c);
        }
    }
    if (flags & 4) {
        g.runs.resize(run_lo);
        return false;
    }
    Parsed a;
    a.name = col[0]; a.name_n = len[0];
    a.ref = col[2]; a.ref_n = len[2];
    a.flags = (uint32_t)flags;
    a.ref_start = pos;
    a.seq = col[9]; a.seq_n = len[9];
    a.nm = nm;
    a.pass_qc = pass_qc;
    a.run_lo = run_lo;
    a.run_hi = (uint32_t)g.runs.size();
    g.al.push_back(a);
    return true;
}

// process_one_read, alignment.rs:275-305
uint64_t process_one_read(pp_ingest &I, Group &g) {
    size_t n = g.al.size();
    if (I.careful && n > 1) return 0;
    // get_read_seq_from_alignments, alignment.rs:311-322
    const Parsed *src = nullptr;
    for (size_t i = 0; i < n; i++)
        if (!(g.al[i].seq_n == 1 && g.al[i].seq[0] == '*')) { src = &g.al[i]; break; }
    if (!src) {
        if (n == 0) fail(PP_ERR_PANIC, "no aligned records to process (the reference panics on an empty read group)");
        fail(PP_ERR_QUIT, "no alignments for read %.*s contain sequence", (int)g.al[0].name_n, g.al[0].name);
    }
    const bool src_fwd = (src->flags & 16) == 0;

    size_t n_good = 0;
    std::vector<uint8_t> good(n, 0);
    for (size_t i = 0; i < n; i++) {
        const Parsed &a = g.al[i];
        if (a.run_lo == a.run_hi)  // chars().next().unwrap() on an empty expanded CIGAR
            fail(PP_ERR_PANIC, "aligned record of read %.*s has an empty CIGAR", (int)a.name_n, a.name);
        uint32_t f = g.runs[a.run_lo] & 15u, l = g.runs[a.run_hi - 1] & 15u;
        bool ends_ok = (f == PP_OP_M || f == PP_OP_EQ) && (l == PP_OP_M || l == PP_OP_EQ);
        if (ends_ok && a.nm <= I.max_errors && a.pass_qc) {
            good[i] = 1;
            n_good++;
        }
    }
    for (size_t i = 0; i < n; i++) {
        if (!good[i]) continue;
        const Parsed &a = g.al[i];
        auto it = I.asmb->index.find(std::string(a.ref, a.ref_n));
        if (it == I.asmb->index.end())
            fail(PP_ERR_QUIT, "query name %.*s in SAM but not in assembly", (int)a.ref_n, a.ref);
        if (a.ref_start > 0xFFFFFFFEull)
            fail(PP_ERR_PANIC, "alignment of read %.*s starts past the end of %.*s", (int)a.name_n, a.name,
                 (int)a.ref_n, a.ref);
        I.contig.push_back(it->second);
        I.ref_start.push_back((uint32_t)a.ref_start);
        I.k.push_back((uint32_t)n_good);
        I.seq_off.push_back(I.seq.size());
        const bool star = a.seq_n == 1 && a.seq[0] == '*';
        const char *s = star ? src->seq : a.seq;
        const size_t sn = star ? src->seq_n : a.seq_n;
        const size_t base = I.seq.size();
        I.seq.resize(base + sn);
        uint8_t *dst = I.seq.data() + base;
        if (star && ((a.flags & 16) == 0) != src_fwd) {
            // add_read_seq (alignment.rs:161-167): reverse complement of the (uppercased) group SEQ
            for (size_t j = 0; j < sn; j++) {
                unsigned char ch = (unsigned char)s[sn - 1 - j];
                if (ch >= 'a' && ch <= 'z') ch = (unsigned char)(ch - 32);
                dst[j] = COMP.t[ch];
            }
        } else {
            for (size_t j = 0; j < sn; j++) {
                unsigned char ch = (unsigned char)s[j];
                if (ch >= 'a' && ch <= 'z') ch = (unsigned char)(ch - 32);  // to_ascii_uppercase
                dst[j] = ch;
            }
        }
        I.seq_len.push_back((uint32_t)sn);
        I.cig_off.push_back(I.cigar.size());
        I.n_cig.push_back(a.run_hi - a.run_lo);
        I.cigar.insert(I.cigar.end(), g.runs.begin() + a.run_lo, g.runs.begin() + a.run_hi);
        I.name_off.push_back(I.names.size());
        I.names.insert(I.names.end(), a.name, a.name + a.name_n);
        I.names.push_back('\0');
    }
    return n_good;
}

}  // namespace

extern "C" int pp_ingest_create(const pp_assembly *a, uint32_t max_errors, int careful, pp_ingest **out) {
    if (!a || !out) return PP_ERR_ARG;
    pp_ingest *g = new pp_ingest();
    g->asmb = a;
    g->max_errors = max_errors;
    g->careful = careful != 0;
    *out = g;
    return PP_OK;
}

// add_to_pileup, alignment.rs:225-272
extern "C" int pp_ingest_sam(pp_ingest *I, const char *path, pp_sam_counts *counts, char *err, size_t errlen) {
    if (!I || !path) return PP_ERR_ARG;
    pp_sam_counts c{0, 0, 0};
    try {
        std::vector<char> text;
        if (!read_file(path, text)) fail(PP_ERR_QUIT, "unable to load alignments from \"%s\"", path);
        LineReader lr{text.data(), text.data() + text.size()};
        const char *line;
        size_t n;
        uint64_t line_no = 0;
        Group g;
        std::string current;  // current_read_name
        while (lr.next(line, n)) {
            line_no++;
            if (n == 0) continue;
            if (line[0] == '@') continue;
            // a record that does not continue the current group closes it first; the decision
            // needs the QNAME only, so peek at it before parsing into the (possibly flushed) group
            const char *tab = (const char *)memchr(line, '\t', n);
            size_t qn = tab ? (size_t)(tab - line) : n;
            Group tmp;
            bool same = current.empty() || (current.size() == qn && memcmp(current.data(), line, qn) == 0);
            Group &dst = same ? g : tmp;
            if (!parse_line(line, n, path, line_no, dst)) continue;  // unaligned: skipped, name not recorded
            c.alignments++;
            if (!same) {
                c.used += process_one_read(*I, g);
                c.reads++;
                g.clear();
                // move the freshly parsed record into the (now empty) group
                Parsed a = tmp.al[0];
                a.run_lo = 0;
                a.run_hi = (uint32_t)tmp.runs.size();
                g.runs = tmp.runs;
                g.al.push_back(a);
            }
            current.assign(line, qn);
        }
        c.used += process_one_read(*I, g);
        c.reads++;
        if (c.alignments == 0) fail(PP_ERR_QUIT, "no alignments in \"%s\"", path);
    } catch (const IngestError &e) {
        if (err && errlen) snprintf(err, errlen, "%s", e.msg.c_str());
        if (counts) *counts = c;
        return e.code;
    }
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
    out->cigar = I->cigar.data();
    out->n_cig_total = I->cigar.size();
}

extern "C" const char *pp_ingest_read_name(const pp_ingest *I, uint64_t i) {
    if (!I || i >= I->name_off.size()) return "?";
    return I->names.data() + I->name_off[i];
}

extern "C" void pp_ingest_free(pp_ingest *I) { delete I; }
