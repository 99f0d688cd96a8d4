/*
 * pp_oracle.c -- CPU oracle for the Polypolish filter + pileup/vote hot path.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT (see pp_oracle.h).  Single-threaded plain
 * C restatement of rrwick/Polypolish v0.6.1; every function cites the
 * reference file:line (relative to /root/reference/) it restates.  The data
 * model deliberately mirrors the reference (one struct per assembly base
 * holding four integer counters, an f64 depth and a string-keyed count table;
 * an expanded CIGAR string per alignment) so that it can be audited against
 * the Rust text line by line.  It is compiled with -ffp-contract=off so that
 * no multiply-add is fused (bankers_rounding depends on the unfused product).
 *
 * Parity status: pinned against the reference's own unit-test vectors
 * (T1..T12 of SURVEY.md section 4).  For everything the reference does not
 * test (read grouping and gates, 1/k shares, CIGAR walk + trim, the pair rule,
 * whole-program output bytes) the status is "parity unpinned": restated from
 * source and cross-checked against oracle/pyref.py only -- the reference is a
 * Rust crate and cannot be built or run in this image (no cargo / rustc).
 */
#define _GNU_SOURCE
#include "pp_oracle.h"

#include <math.h>
#include <setjmp.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

/* ------------------------------------------------------------------------ */
/* error plumbing: quit_with_error (misc.rs:29-33) and Rust panics           */
/* ------------------------------------------------------------------------ */
static __thread jmp_buf *g_jmp = NULL;
static __thread char g_err[1024];

static void bail(int code, const char *fmt, va_list ap) {
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    if (!g_jmp) {
        fprintf(stderr, "pp_oracle: %s outside an API call: %s\n",
                code == ORC_QUIT ? "quit" : "panic", g_err);
        abort();
    }
    longjmp(*g_jmp, code);
}
static void quit_with_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    bail(ORC_QUIT, fmt, ap);
    va_end(ap);
}
static void rust_panic(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    bail(ORC_PANIC, fmt, ap);
    va_end(ap);
}

#define API_ENTER(err, errlen)                         \
    jmp_buf jb_;                                       \
    jmp_buf *saved_ = g_jmp;                           \
    g_jmp = &jb_;                                      \
    int code_ = setjmp(jb_);                           \
    if (code_ != 0) {                                  \
        g_jmp = saved_;                                \
        if ((err) && (errlen)) {                       \
            snprintf((err), (errlen), "%s", g_err);    \
        }                                              \
        return code_;                                  \
    }
#define API_LEAVE() \
    do {            \
        g_jmp = saved_; \
    } while (0)

static void *xmalloc(size_t n) {
    void *p = malloc(n ? n : 1);
    if (!p) {
        fprintf(stderr, "pp_oracle: out of memory\n");
        abort();
    }
    return p;
}
static void *xrealloc(void *q, size_t n) {
    void *p = realloc(q, n ? n : 1);
    if (!p) {
        fprintf(stderr, "pp_oracle: out of memory\n");
        abort();
    }
    return p;
}
static char *xstrndup(const char *s, size_t n) {
    char *p = (char *)xmalloc(n + 1);
    memcpy(p, s, n);
    p[n] = 0;
    return p;
}

/* ------------------------------------------------------------------------ */
/* orc_buf                                                                   */
/* ------------------------------------------------------------------------ */
static void buf_reserve(orc_buf *b, size_t extra) {
    if (b->len + extra + 1 > b->cap) {
        size_t nc = b->cap ? b->cap * 2 : 256;
        while (nc < b->len + extra + 1) nc *= 2;
        b->data = (char *)xrealloc(b->data, nc);
        b->cap = nc;
    }
}
static void buf_append(orc_buf *b, const char *s, size_t n) {
    buf_reserve(b, n);
    memcpy(b->data + b->len, s, n);
    b->len += n;
    b->data[b->len] = 0;
}
static void buf_puts(orc_buf *b, const char *s) { buf_append(b, s, strlen(s)); }
static void buf_printf(orc_buf *b, const char *fmt, ...) {
    char tmp[512];
    va_list ap;
    va_start(ap, fmt);
    int n = vsnprintf(tmp, sizeof tmp, fmt, ap);
    va_end(ap);
    if (n < 0) return;
    if ((size_t)n < sizeof tmp) {
        buf_append(b, tmp, (size_t)n);
    } else {
        char *big = (char *)xmalloc((size_t)n + 1);
        va_start(ap, fmt);
        vsnprintf(big, (size_t)n + 1, fmt, ap);
        va_end(ap);
        buf_append(b, big, (size_t)n);
        free(big);
    }
}
void orc_buf_free(orc_buf *b) {
    if (!b) return;
    free(b->data);
    b->data = NULL;
    b->len = b->cap = 0;
}

/* ------------------------------------------------------------------------ */
/* misc.rs                                                                   */
/* ------------------------------------------------------------------------ */

/* misc.rs:208-215.  `float as u32` saturates (NaN -> 0); fract() is
 * x - trunc(x). */
uint32_t orc_bankers_rounding(double x) {
    uint32_t rounded_down;
    if (!(x == x) || x <= 0.0)
        rounded_down = 0;
    else if (x >= 4294967295.0)
        rounded_down = 4294967295u;
    else
        rounded_down = (uint32_t)x;
    double f = x - trunc(x);
    if (f < 0.5) return rounded_down;
    if (f > 0.5) return rounded_down + 1u;
    return rounded_down + (rounded_down & 1u);
}

/* misc.rs:170-182 */
static char complement_base(char base) {
    switch (base) {
    case 'A': return 'T'; case 'T': return 'A'; case 'G': return 'C'; case 'C': return 'G';
    case 'a': return 't'; case 't': return 'a'; case 'g': return 'c'; case 'c': return 'g';
    case 'N': return 'N'; case 'n': return 'n';
    case 'R': return 'Y'; case 'Y': return 'R'; case 'S': return 'S'; case 'W': return 'W';
    case 'K': return 'M'; case 'M': return 'K';
    case 'B': return 'V'; case 'V': return 'B'; case 'D': return 'H'; case 'H': return 'D';
    case 'r': return 'y'; case 'y': return 'r'; case 's': return 's'; case 'w': return 'w';
    case 'k': return 'm'; case 'm': return 'k';
    case 'b': return 'v'; case 'v': return 'b'; case 'd': return 'h'; case 'h': return 'd';
    case '.': return '.'; case '-': return '-'; case '?': return '?';
    default: return 'N';
    }
}

/* misc.rs:185-191 (byte-wise; the reference iterates chars, identical for ASCII) */
void orc_reverse_complement(const char *in, size_t n, char *out) {
    for (size_t i = 0; i < n; i++) out[i] = complement_base(in[n - 1 - i]);
}

/* ------------------------------------------------------------------------ */
/* number parsing the way Rust's str::parse::<uN>() does: optional '+',      */
/* ASCII digits only, overflow is an error (callers unwrap -> panic).        */
/* ------------------------------------------------------------------------ */
static int parse_unsigned(const char *s, size_t n, uint64_t max, uint64_t *out) {
    size_t i = 0;
    if (n == 0) return -1;
    if (s[0] == '+') {
        i = 1;
        if (n == 1) return -1;
    }
    uint64_t v = 0;
    for (; i < n; i++) {
        if (s[i] < '0' || s[i] > '9') return -1;
        uint64_t d = (uint64_t)(s[i] - '0');
        if (v > (max - d) / 10) return -1;
        v = v * 10 + d;
    }
    *out = v;
    return 0;
}

/* ------------------------------------------------------------------------ */
/* alignment.rs                                                              */
/* ------------------------------------------------------------------------ */
static int is_cigar_op(char c) { return c != 0 && strchr("MIDNSHP=X", c) != NULL; }

/* Iterate the non-overlapping leftmost matches of \d+[MIDNSHP=X]
 * (alignment.rs:27-29) the way Regex::find_iter does.  *pos is the scan
 * cursor; on a match returns 1 and the digit span [ds,de) with the operator at
 * de. */
static int next_cigar_token(const char *c, size_t n, size_t *pos, size_t *ds, size_t *de) {
    size_t i = *pos;
    while (i < n) {
        if (c[i] >= '0' && c[i] <= '9') {
            size_t j = i;
            while (j < n && c[j] >= '0' && c[j] <= '9') j++;
            if (j < n && is_cigar_op(c[j])) {
                *ds = i;
                *de = j;
                *pos = j + 1;
                return 1;
            }
            i = j; /* digits not followed by an operator: no match starts inside them */
        } else {
            i++;
        }
    }
    *pos = n;
    return 0;
}

/* alignment.rs:325-346 */
int orc_get_expanded_cigar(const char *cigar, char **expanded, size_t *exp_len) {
    size_t n = strlen(cigar);
    if (n == 1 && cigar[0] == '*') {
        *expanded = xstrndup("", 0);
        *exp_len = 0;
        return 0;
    }
    size_t cap = 256, len = 0, total_len = 0, pos = 0, ds, de;
    char *out = (char *)xmalloc(cap);
    while (next_cigar_token(cigar, n, &pos, &ds, &de)) {
        uint64_t num;
        if (parse_unsigned(cigar + ds, de - ds, 0xFFFFFFFFull, &num) != 0) {
            free(out);
            rust_panic("CIGAR length does not fit u32: %s", cigar);
        }
        if (len + num + 1 > cap) {
            while (len + num + 1 > cap) cap *= 2;
            out = (char *)xrealloc(out, cap);
        }
        memset(out + len, cigar[de], (size_t)num);
        len += (size_t)num;
        total_len += de + 1 - ds;
    }
    if (n != total_len) {
        free(out);
        return -1;
    }
    out[len] = 0;
    *expanded = out;
    *exp_len = len;
    return 0;
}

/* alignment.rs:138-149 */
uint64_t orc_get_ref_end(uint64_t ref_start, const char *cigar) {
    size_t n = strlen(cigar), pos = 0, ds, de;
    uint64_t ref_end = ref_start;
    while (next_cigar_token(cigar, n, &pos, &ds, &de)) {
        uint64_t num;
        if (parse_unsigned(cigar + ds, de - ds, UINT64_MAX, &num) != 0)
            rust_panic("CIGAR length does not fit usize: %s", cigar);
        switch (cigar[de]) {
        case 'M': case 'D': case 'N': case '=': case 'X': ref_end += num; break;
        default: break;
        }
    }
    return ref_end;
}

typedef struct {
    char *read_name;
    char *ref_name;
    uint32_t sam_flags;
    uint64_t ref_start;
    char *cigar;
    char *expanded_cigar;
    size_t expanded_len;
    char *read_seq;
    size_t read_seq_len;
    uint32_t mismatches;
    int pass_qc;
} Alignment;

static void alignment_free(Alignment *a) {
    free(a->read_name);
    free(a->ref_name);
    free(a->cigar);
    free(a->expanded_cigar);
    free(a->read_seq);
    memset(a, 0, sizeof *a);
}

typedef struct {
    const char *p;
    size_t n;
} span;

/* sam_line.split('\t') */
static size_t split_tabs(const char *line, size_t n, span **parts_out) {
    size_t cap = 16, cnt = 0;
    span *parts = (span *)xmalloc(cap * sizeof *parts);
    size_t start = 0;
    for (size_t i = 0; i <= n; i++) {
        if (i == n || line[i] == '\t') {
            if (cnt == cap) {
                cap *= 2;
                parts = (span *)xrealloc(parts, cap * sizeof *parts);
            }
            parts[cnt].p = line + start;
            parts[cnt].n = i - start;
            cnt++;
            start = i + 1;
        }
    }
    *parts_out = parts;
    return cnt;
}

static int span_eq_ignore_ascii_case(span s, const char *lit) {
    size_t n = strlen(lit);
    if (s.n != n) return 0;
    for (size_t i = 0; i < n; i++) {
        char a = s.p[i], b = lit[i];
        if (a >= 'A' && a <= 'Z') a = (char)(a + 32);
        if (b >= 'A' && b <= 'Z') b = (char)(b + 32);
        if (a != b) return 0;
    }
    return 1;
}

/* Alignment::new, alignment.rs:49-98.  Returns NULL on success or the
 * reference's Err string. */
static const char *alignment_new(const char *line, size_t n, Alignment *a) {
    memset(a, 0, sizeof *a);
    span *parts;
    size_t np = split_tabs(line, n, &parts);
    if (np < 11) {
        free(parts);
        return "too few columns";
    }
    uint64_t flags, ref_start;
    if (parse_unsigned(parts[1].p, parts[1].n, 0xFFFFFFFFull, &flags) != 0) {
        free(parts);
        rust_panic("could not parse SAM FLAG as u32");
    }
    if (parse_unsigned(parts[3].p, parts[3].n, UINT64_MAX, &ref_start) != 0) {
        free(parts);
        rust_panic("could not parse SAM POS as usize");
    }
    if (ref_start > 0) ref_start -= 1;

    uint32_t mismatches = 0xFFFFFFFFu;
    int pass_qc = 1;
    for (size_t i = 11; i < np; i++) {
        if (parts[i].n >= 5 && memcmp(parts[i].p, "NM:i:", 5) == 0) {
            uint64_t nm;
            if (parse_unsigned(parts[i].p + 5, parts[i].n - 5, 0xFFFFFFFFull, &nm) != 0) {
                free(parts);
                rust_panic("could not parse NM tag as u32");
            }
            mismatches = (uint32_t)nm;
        }
        if (span_eq_ignore_ascii_case(parts[i], "ZP:Z:fail")) pass_qc = 0;
    }
    if (mismatches == 0xFFFFFFFFu && (flags & 4) == 0) {
        free(parts);
        return "missing NM tag";
    }
    a->read_name = xstrndup(parts[0].p, parts[0].n);
    a->cigar = xstrndup(parts[5].p, parts[5].n);
    if (orc_get_expanded_cigar(a->cigar, &a->expanded_cigar, &a->expanded_len) != 0) {
        /* alignment.rs:82-83; {:?} quotes the string */
        char msg[512];
        snprintf(msg, sizeof msg, "encountered an invalid CIGAR string for read %s: \"%s\"",
                 a->read_name, a->cigar);
        free(parts);
        quit_with_error("%s", msg);
    }
    a->ref_name = xstrndup(parts[2].p, parts[2].n);
    a->sam_flags = (uint32_t)flags;
    a->ref_start = ref_start;
    a->read_seq = xstrndup(parts[9].p, parts[9].n);
    a->read_seq_len = parts[9].n;
    for (size_t i = 0; i < a->read_seq_len; i++) /* to_ascii_uppercase */
        if (a->read_seq[i] >= 'a' && a->read_seq[i] <= 'z') a->read_seq[i] = (char)(a->read_seq[i] - 32);
    a->mismatches = mismatches;
    a->pass_qc = pass_qc;
    free(parts);
    return NULL;
}

/* Alignment::new_quick, alignment.rs:102-128 */
static const char *alignment_new_quick(const char *line, size_t n, Alignment *a) {
    memset(a, 0, sizeof *a);
    span *parts;
    size_t np = split_tabs(line, n, &parts);
    if (np < 11) {
        free(parts);
        return "too few columns";
    }
    uint64_t flags, ref_start;
    if (parse_unsigned(parts[1].p, parts[1].n, 0xFFFFFFFFull, &flags) != 0) {
        free(parts);
        rust_panic("could not parse SAM FLAG as u32");
    }
    if (parse_unsigned(parts[3].p, parts[3].n, UINT64_MAX, &ref_start) != 0) {
        free(parts);
        rust_panic("could not parse SAM POS as usize");
    }
    if (ref_start > 0) ref_start -= 1;
    a->read_name = xstrndup(parts[0].p, parts[0].n);
    a->ref_name = xstrndup(parts[2].p, parts[2].n);
    a->sam_flags = (uint32_t)flags;
    a->ref_start = ref_start;
    a->cigar = xstrndup(parts[5].p, parts[5].n);
    a->expanded_cigar = xstrndup("", 0);
    a->read_seq = xstrndup("", 0);
    a->mismatches = 0;
    a->pass_qc = 1;
    free(parts);
    return NULL;
}

static int is_aligned(const Alignment *a) { return (a->sam_flags & 4) == 0; }            /* :130-132 */
static int is_on_forward_strand(const Alignment *a) { return (a->sam_flags & 16) == 0; } /* :151-153 */
static int get_strand(const Alignment *a) { return is_on_forward_strand(a) ? 1 : -1; }   /* :134-136 */

/* alignment.rs:155-159; chars().next().unwrap() panics on an empty string */
static int starts_and_ends_with_match(const Alignment *a) {
    if (a->expanded_len == 0) rust_panic("expanded CIGAR is empty for read %s", a->read_name);
    char f = a->expanded_cigar[0], l = a->expanded_cigar[a->expanded_len - 1];
    return (f == 'M' || f == '=') && (l == 'M' || l == '=');
}

/* alignment.rs:161-167 */
static void add_read_seq(Alignment *a, const char *read_seq, size_t n, int strand) {
    free(a->read_seq);
    a->read_seq = (char *)xmalloc(n + 1);
    if (get_strand(a) == strand)
        memcpy(a->read_seq, read_seq, n);
    else
        orc_reverse_complement(read_seq, n, a->read_seq);
    a->read_seq[n] = 0;
    a->read_seq_len = n;
}

typedef struct {
    size_t start, end;
} slice;

/* alignment.rs:364-378 */
static void trim_bases_for_homopolymers(slice *rb, size_t *n, const char *read_seq) {
    if (*n == 0) rust_panic("trim_bases_for_homopolymers on an empty list");
    slice last = rb[*n - 1];
    size_t last_len = last.end - last.start;
    const char *last_base = read_seq + last.start;
    while (*n > 0) {
        slice cur = rb[*n - 1];
        size_t cur_len = cur.end - cur.start;
        if (cur_len != last_len || memcmp(read_seq + cur.start, last_base, cur_len) != 0) break;
        (*n)--;
    }
    if (*n > 0) (*n)--;
}

/* alignment.rs:175-201.  Returns a malloc'd slice list. */
static slice *get_read_bases_for_each_target_base(const char *read_name, const char *cigar,
                                                  const char *expanded, size_t exp_len,
                                                  const char *read_seq, size_t seq_len,
                                                  size_t *n_out) {
    size_t i = 0, n = 0;
    slice *rb = (slice *)xmalloc((exp_len ? exp_len : 1) * sizeof *rb);
    for (size_t x = 0; x < exp_len; x++) {
        char c = expanded[x];
        if (c == 'M' || c == '=' || c == 'X') {
            rb[n].start = i;
            rb[n].end = i + 1;
            n++;
            i += 1;
        } else if (c == 'I') {
            if (n == 0) {
                free(rb);
                rust_panic("insertion before any reference base in CIGAR for read %s", read_name);
            }
            rb[n - 1].end = i + 1;
            i += 1;
        } else if (c == 'D') {
            rb[n].start = i;
            rb[n].end = i;
            n++;
        } else {
            free(rb);
            quit_with_error("unexpected character (other than M, =, X, I or D) in CIGAR string for "
                            "read %s: \"%s\" - did you use BWA MEM to generate your alignments?",
                            read_name, cigar);
        }
    }
    if (i != seq_len) {
        free(rb);
        quit_with_error("CIGAR string for read %s does not match read sequence", read_name);
    }
    /* Slices are byte ranges into read_seq; an I extending a slice never
     * leaves the string because i <= seq_len was just checked ... but the
     * check happens after the walk in the reference too, so a walk that
     * overruns is only caught here. */
    trim_bases_for_homopolymers(rb, &n, read_seq);
    *n_out = n;
    return rb;
}

/* ------------------------------------------------------------------------ */
/* pileup.rs                                                                 */
/* ------------------------------------------------------------------------ */
struct orc_pileup_base {
    char original;
    double depth;
    uint32_t count_a, count_c, count_g, count_t;
    /* counts: HashMap<String,u32> -- a small open list; iteration order is
     * irrelevant to every output. */
    char **keys;
    uint32_t *key_len;
    uint32_t *key_cnt;
    uint32_t n_keys, cap_keys;
};
typedef struct orc_pileup_base PileupBase;

static void pb_init(PileupBase *b, char original) { /* pileup.rs:44-54 */
    memset(b, 0, sizeof *b);
    b->original = original;
    b->depth = 0.0;
}
static void pb_clear(PileupBase *b) {
    for (uint32_t i = 0; i < b->n_keys; i++) free(b->keys[i]);
    free(b->keys);
    free(b->key_len);
    free(b->key_cnt);
}

/* pileup.rs:56-65 */
static void pb_add_seq(PileupBase *b, const char *seq, size_t n, double depth_contribution) {
    if (n == 1 && seq[0] == 'A') {
        b->count_a += 1;
    } else if (n == 1 && seq[0] == 'C') {
        b->count_c += 1;
    } else if (n == 1 && seq[0] == 'G') {
        b->count_g += 1;
    } else if (n == 1 && seq[0] == 'T') {
        b->count_t += 1;
    } else {
        uint32_t i;
        for (i = 0; i < b->n_keys; i++)
            if (b->key_len[i] == n && memcmp(b->keys[i], seq, n) == 0) break;
        if (i == b->n_keys) {
            if (b->n_keys == b->cap_keys) {
                b->cap_keys = b->cap_keys ? b->cap_keys * 2 : 2;
                b->keys = (char **)xrealloc(b->keys, b->cap_keys * sizeof *b->keys);
                b->key_len = (uint32_t *)xrealloc(b->key_len, b->cap_keys * sizeof *b->key_len);
                b->key_cnt = (uint32_t *)xrealloc(b->key_cnt, b->cap_keys * sizeof *b->key_cnt);
            }
            b->keys[i] = xstrndup(seq, n);
            b->key_len[i] = (uint32_t)n;
            b->key_cnt[i] = 0;
            b->n_keys++;
        }
        b->key_cnt[i] += 1;
    }
    b->depth += depth_contribution;
}

typedef struct {
    const char *new_base; /* points at b->original, a literal, or a key */
    size_t new_len;
    int status;
    uint32_t valid_threshold, invalid_threshold;
} vote;

static int cmp_str(const void *a, const void *b) {
    return strcmp(*(const char *const *)a, *(const char *const *)b);
}

/* pileup.rs:137-148 */
static void pb_get_count_str(const PileupBase *b, orc_buf *out) {
    size_t n = 0, cap = 4 + b->n_keys;
    char **items = (char **)xmalloc(cap * sizeof *items);
    char tmp[64];
    if (b->count_a > 0) { snprintf(tmp, sizeof tmp, "Ax%u", b->count_a); items[n++] = xstrndup(tmp, strlen(tmp)); }
    if (b->count_c > 0) { snprintf(tmp, sizeof tmp, "Cx%u", b->count_c); items[n++] = xstrndup(tmp, strlen(tmp)); }
    if (b->count_g > 0) { snprintf(tmp, sizeof tmp, "Gx%u", b->count_g); items[n++] = xstrndup(tmp, strlen(tmp)); }
    if (b->count_t > 0) { snprintf(tmp, sizeof tmp, "Tx%u", b->count_t); items[n++] = xstrndup(tmp, strlen(tmp)); }
    for (uint32_t i = 0; i < b->n_keys; i++) {
        size_t need = b->key_len[i] + 16;
        char *s = (char *)xmalloc(need);
        memcpy(s, b->keys[i], b->key_len[i]);
        snprintf(s + b->key_len[i], need - b->key_len[i], "x%u", b->key_cnt[i]);
        items[n++] = s;
    }
    qsort(items, n, sizeof *items, cmp_str); /* counts.sort(): byte-wise string order */
    for (size_t i = 0; i < n; i++) {
        if (i) buf_puts(out, ",");
        buf_puts(out, items[i]);
        free(items[i]);
    }
    free(items);
}

static const char *status_str(int s) { /* pileup.rs:156-163 */
    switch (s) {
    case ORC_ST_KEPT: return "kept";
    case ORC_ST_CHANGED: return "changed";
    case ORC_ST_LOW_DEPTH: return "low_depth";
    case ORC_ST_NONE: return "none";
    case ORC_ST_MULTIPLE: return "multiple";
    default: return "too_close";
    }
}

/* pileup.rs:67-134 */
static vote pb_get_polished_seq(const PileupBase *b, uint32_t min_depth, double fraction_valid,
                                double fraction_invalid) {
    vote v;
    uint32_t vt = orc_bankers_rounding(b->depth * fraction_valid);
    uint32_t valid_threshold = min_depth > vt ? min_depth : vt;
    uint32_t invalid_threshold = orc_bankers_rounding(b->depth * fraction_invalid);

    size_t n_valid = 0, n_intermediate = 0;
    const char *first_valid = NULL;
    size_t first_valid_len = 0;
    static const char *ACGT[4] = {"A", "C", "G", "T"};
    uint32_t c4[4] = {b->count_a, b->count_c, b->count_g, b->count_t};
    for (int i = 0; i < 4; i++) {
        if (c4[i] >= valid_threshold) {
            if (n_valid == 0) { first_valid = ACGT[i]; first_valid_len = 1; }
            n_valid++;
        } else if (c4[i] >= invalid_threshold) {
            n_intermediate++;
        }
    }
    for (uint32_t i = 0; i < b->n_keys; i++) {
        if (b->key_cnt[i] >= valid_threshold) {
            if (n_valid == 0) { first_valid = b->keys[i]; first_valid_len = b->key_len[i]; }
            n_valid++;
        } else if (b->key_cnt[i] >= invalid_threshold) {
            n_intermediate++;
        }
    }

    v.new_base = &b->original;
    v.new_len = 1;
    v.status = ORC_ST_KEPT;
    if (b->depth < (double)min_depth) {
        v.status = ORC_ST_LOW_DEPTH;
    } else if (n_valid == 1) {
        if (n_intermediate > 0) {
            v.status = ORC_ST_TOO_CLOSE;
        } else {
            v.new_base = first_valid;
            v.new_len = first_valid_len;
            if (!(v.new_len == 1 && v.new_base[0] == b->original)) v.status = ORC_ST_CHANGED;
        }
    } else if (n_valid == 0) {
        v.status = ORC_ST_NONE;
    } else {
        v.status = ORC_ST_MULTIPLE;
    }
    v.valid_threshold = valid_threshold;
    v.invalid_threshold = invalid_threshold;
    return v;
}

/* pileup.rs:150-166 (the part of the debug line after name and pos) */
static void pb_debug_line(const PileupBase *b, const vote *v, orc_buf *out) {
    buf_printf(out, "%c\t%.1f\t%u\t%u\t", b->original, b->depth, v->invalid_threshold,
               v->valid_threshold);
    pb_get_count_str(b, out);
    buf_printf(out, "\t%s\t", status_str(v->status));
    buf_append(out, v->new_base, v->new_len);
}

typedef struct {
    PileupBase *bases;
    size_t n;
} Pileup;

static void pileup_new(Pileup *p, const char *seq, size_t n) { /* pileup.rs:178-187 */
    p->bases = (PileupBase *)xmalloc((n ? n : 1) * sizeof *p->bases);
    p->n = n;
    for (size_t i = 0; i < n; i++) pb_init(&p->bases[i], seq[i]);
}
static void pileup_free(Pileup *p) {
    for (size_t i = 0; i < p->n; i++) pb_clear(&p->bases[i]);
    free(p->bases);
    p->bases = NULL;
    p->n = 0;
}

/* pileup.rs:189-200 */
static void pileup_add_alignment(Pileup *p, const Alignment *a, double depth_contribution) {
    size_t n;
    slice *rb = get_read_bases_for_each_target_base(a->read_name, a->cigar, a->expanded_cigar,
                                                    a->expanded_len, a->read_seq,
                                                    a->read_seq_len, &n);
    uint64_t i = a->ref_start;
    for (size_t j = 0; j < n; j++) {
        if (i >= p->n) {
            free(rb);
            rust_panic("index out of bounds: alignment of read %s runs past the end of %s",
                       a->read_name, a->ref_name);
        }
        if (rb[j].start == rb[j].end)
            pb_add_seq(&p->bases[i], "-", 1, depth_contribution);
        else
            pb_add_seq(&p->bases[i], a->read_seq + rb[j].start, rb[j].end - rb[j].start,
                       depth_contribution);
        i += 1;
    }
    free(rb);
}

/* ---- public single-base handles (tests T8) ------------------------------ */
orc_pileup_base *orc_pb_new(char original) {
    PileupBase *b = (PileupBase *)xmalloc(sizeof *b);
    pb_init(b, original);
    return b;
}
void orc_pb_free(orc_pileup_base *b) {
    if (!b) return;
    pb_clear(b);
    free(b);
}
void orc_pb_add_seq(orc_pileup_base *b, const char *seq, size_t n, double dc) {
    pb_add_seq(b, seq, n, dc);
}
int orc_pb_get_polished_seq(const orc_pileup_base *b, uint32_t min_depth, double fraction_valid,
                            double fraction_invalid, char *out, size_t cap) {
    vote v = pb_get_polished_seq(b, min_depth, fraction_valid, fraction_invalid);
    if (cap) {
        size_t n = v.new_len < cap - 1 ? v.new_len : cap - 1;
        memcpy(out, v.new_base, n);
        out[n] = 0;
    }
    return v.status;
}
void orc_pb_get_count_str(const orc_pileup_base *b, orc_buf *out) { pb_get_count_str(b, out); }

int orc_read_bases_for_each_target_base(const char *cigar, const char *read_seq, size_t seq_len,
                                        uint32_t **starts, uint32_t **ends, size_t *n,
                                        char *err, size_t errlen) {
    API_ENTER(err, errlen);
    char *expanded;
    size_t exp_len;
    if (orc_get_expanded_cigar(cigar, &expanded, &exp_len) != 0)
        quit_with_error("encountered an invalid CIGAR string for read r: \"%s\"", cigar);
    size_t cnt;
    slice *rb = get_read_bases_for_each_target_base("r", cigar, expanded, exp_len, read_seq,
                                                    seq_len, &cnt);
    *starts = (uint32_t *)xmalloc((cnt ? cnt : 1) * sizeof **starts);
    *ends = (uint32_t *)xmalloc((cnt ? cnt : 1) * sizeof **ends);
    for (size_t i = 0; i < cnt; i++) {
        (*starts)[i] = (uint32_t)rb[i].start;
        (*ends)[i] = (uint32_t)rb[i].end;
    }
    *n = cnt;
    free(rb);
    free(expanded);
    API_LEAVE();
    return ORC_OK;
}

int orc_parse_positions(const char *sam_line, uint64_t *ref_start, uint64_t *ref_end) {
    char err[1024];
    API_ENTER(err, sizeof err);
    Alignment a;
    const char *e = alignment_new(sam_line, strlen(sam_line), &a);
    if (e) quit_with_error("%s", e);
    *ref_start = a.ref_start;
    *ref_end = orc_get_ref_end(a.ref_start, a.cigar);
    alignment_free(&a);
    API_LEAVE();
    return ORC_OK;
}

/* ------------------------------------------------------------------------ */
/* line reader with BufRead::lines() semantics: split on '\n', strip one     */
/* trailing '\r', no final empty line for a trailing newline.                */
/* ------------------------------------------------------------------------ */
typedef struct {
    char *data;
    size_t len;
    size_t pos;
} text;

static int read_whole_file(const char *path, text *t) {
    FILE *f = fopen(path, "rb");
    if (!f) return -1;
    size_t cap = 1 << 16, len = 0;
    char *d = (char *)xmalloc(cap);
    for (;;) {
        if (len == cap) {
            cap *= 2;
            d = (char *)xrealloc(d, cap);
        }
        size_t r = fread(d + len, 1, cap - len, f);
        len += r;
        if (r == 0) break;
    }
    int bad = ferror(f);
    fclose(f);
    if (bad) {
        free(d);
        return -1;
    }
    t->data = d;
    t->len = len;
    t->pos = 0;
    return 0;
}

static int next_line(text *t, const char **line, size_t *n) {
    if (t->pos >= t->len) return 0;
    const char *s = t->data + t->pos;
    const char *nl = (const char *)memchr(s, '\n', t->len - t->pos);
    size_t l = nl ? (size_t)(nl - s) : t->len - t->pos;
    t->pos += l + (nl ? 1 : 0);
    if (l > 0 && s[l - 1] == '\r') l--;
    *line = s;
    *n = l;
    return 1;
}

/* lines() yields Err on invalid UTF-8, which every caller turns into
 * "unable to load ..." */
static int valid_utf8(const char *s, size_t n) {
    size_t i = 0;
    while (i < n) {
        unsigned char c = (unsigned char)s[i];
        if (c < 0x80) { i++; continue; }
        size_t need;
        unsigned int cp;
        if ((c & 0xE0) == 0xC0) { need = 1; cp = c & 0x1F; if (cp < 2) return 0; }
        else if ((c & 0xF0) == 0xE0) { need = 2; cp = c & 0x0F; }
        else if ((c & 0xF8) == 0xF0) { need = 3; cp = c & 0x07; if (cp > 4) return 0; }
        else return 0;
        for (size_t k = 1; k <= need; k++) {
            if (i + k >= n) return 0; /* truncated sequence */
            unsigned char cc = (unsigned char)s[i + k];
            if ((cc & 0xC0) != 0x80) return 0;
            cp = (cp << 6) | (cc & 0x3F);
        }
        if (need == 2 && (cp < 0x800 || (cp >= 0xD800 && cp <= 0xDFFF))) return 0;
        if (need == 3 && (cp < 0x10000 || cp > 0x10FFFF)) return 0;
        i += need + 1;
    }
    return 1;
}

/* ------------------------------------------------------------------------ */
/* misc.rs:38-167 FASTA loader                                               */
/* ------------------------------------------------------------------------ */
/* char::is_whitespace (Unicode White_Space, misc.rs:118-120) at the start of the valid UTF-8 text p[0 .. left): the
   number of bytes of that character, 0 if the character there is not whitespace.  U+0009-000D, 0020, 0085, 00A0, 1680,
   2000-200A, 2028, 2029, 202F, 205F, 3000. */
static size_t rust_whitespace_len(const char *p, size_t left) {
    const unsigned char c = (unsigned char)p[0];
    if (c == ' ' || (c >= 0x09 && c <= 0x0D)) return 1;
    if (c == 0xC2 && left >= 2) {
        const unsigned char d = (unsigned char)p[1];
        return d == 0x85 || d == 0xA0 ? 2 : 0;
    }
    if (left >= 3) {
        const unsigned char d = (unsigned char)p[1], e = (unsigned char)p[2];
        if (c == 0xE1 && d == 0x9A && e == 0x80) return 3;                               /* U+1680 */
        if (c == 0xE2 && d == 0x80 && ((e >= 0x80 && e <= 0x8A) || e == 0xA8 || e == 0xA9 || e == 0xAF)) return 3; /* U+2000-200A, 2028, 2029, 202F */
        if (c == 0xE2 && d == 0x81 && e == 0x9F) return 3;                               /* U+205F */
        if (c == 0xE3 && d == 0x80 && e == 0x80) return 3;                               /* U+3000 */
    }
    return 0;
}

void orc_fasta_free(orc_fasta *f) {
    if (!f) return;
    for (size_t i = 0; i < f->n; i++) {
        free(f->name[i]);
        free(f->desc[i]);
        free(f->seq[i]);
    }
    free(f->name);
    free(f->desc);
    free(f->seq);
    free(f->len);
    memset(f, 0, sizeof *f);
}

static void fasta_push(orc_fasta *f, size_t *cap, char *name, char *desc, char *seq, size_t len) {
    if (f->n == *cap) {
        *cap = *cap ? *cap * 2 : 8;
        f->name = (char **)xrealloc(f->name, *cap * sizeof *f->name);
        f->desc = (char **)xrealloc(f->desc, *cap * sizeof *f->desc);
        f->seq = (char **)xrealloc(f->seq, *cap * sizeof *f->seq);
        f->len = (size_t *)xrealloc(f->len, *cap * sizeof *f->len);
    }
    for (size_t i = 0; i < len; i++) /* make_ascii_uppercase, misc.rs:114,129 */
        if (seq[i] >= 'a' && seq[i] <= 'z') seq[i] = (char)(seq[i] - 32);
    seq[len] = 0; /* callers keep one spare byte */
    f->name[f->n] = name;
    f->desc[f->n] = desc;
    f->seq[f->n] = seq;
    f->len[f->n] = len;
    f->n++;
}

static void load_fasta_inner(const char *path, orc_fasta *out) {
    /* is_file_gzipped, misc.rs:81-99 */
    FILE *f = fopen(path, "rb");
    if (!f) quit_with_error("unable to open \"%s\"", path);
    unsigned char magic[2];
    size_t got = fread(magic, 1, 2, f);
    fclose(f);
    if (got != 2) quit_with_error("\"%s\" is too small", path);
    int gz = magic[0] == 31 && magic[1] == 139;

    text t = {0};
    if (gz) { /* misc.rs:136-167 */
        gzFile g = gzopen(path, "rb");
        if (!g) quit_with_error("unable to load \"%s\"", path);
        size_t cap = 1 << 16, len = 0;
        char *d = (char *)xmalloc(cap);
        for (;;) {
            if (len == cap) {
                cap *= 2;
                d = (char *)xrealloc(d, cap);
            }
            int r = gzread(g, d + len, (unsigned)(cap - len > (1u << 30) ? (1u << 30) : cap - len));
            if (r < 0) {
                gzclose(g);
                free(d);
                quit_with_error("unable to load \"%s\"", path);
            }
            if (r == 0) break;
            len += (size_t)r;
        }
        gzclose(g);
        t.data = d;
        t.len = len;
    } else { /* misc.rs:102-133 */
        if (read_whole_file(path, &t) != 0) quit_with_error("unable to load \"%s\"", path);
    }

    size_t cap = 0;
    char *name = xstrndup("", 0), *desc = xstrndup("", 0);
    size_t name_len = 0;
    char *seq = (char *)xmalloc(1);
    size_t seq_len = 0, seq_cap = 1;
    const char *line;
    size_t n;
    while (next_line(&t, &line, &n)) {
        if (!valid_utf8(line, n)) quit_with_error("unable to load \"%s\"", path);
        if (n == 0) continue;
        if (line[0] == '>') {
            if (name_len > 0) {
                fasta_push(out, &cap, name, desc, seq, seq_len);
                seq = (char *)xmalloc(1);
                seq_len = 0;
                seq_cap = 1;
            } else {
                free(name);
                free(desc);
            }
            /* text[1..].splitn(2, char::is_whitespace), misc.rs:118-120 */
            size_t i = 1, wl = 0;
            while (i < n && !(wl = rust_whitespace_len(line + i, n - i))) i++;
            name = xstrndup(line + 1, i - 1);
            name_len = i - 1;
            desc = i < n ? xstrndup(line + i + wl, n - i - wl) : xstrndup("", 0);
        } else {
            if (name_len == 0) quit_with_error("\"%s\" is not correctly formatted", path);
            if (seq_len + n + 1 > seq_cap) {
                while (seq_len + n + 1 > seq_cap) seq_cap *= 2;
                seq = (char *)xrealloc(seq, seq_cap);
            }
            memcpy(seq + seq_len, line, n);
            seq_len += n;
        }
    }
    if (name_len > 0) {
        fasta_push(out, &cap, name, desc, seq, seq_len);
    } else {
        free(name);
        free(desc);
        free(seq);
    }
    free(t.data);

    /* check_load_fasta, misc.rs:56-75 */
    if (out->n == 0) quit_with_error("\"%s\" contains no sequences", path);
    for (size_t i = 0; i < out->n; i++) {
        if (out->name[i][0] == 0) quit_with_error("\"%s\" has an unnamed sequence", path);
        if (out->len[i] == 0) quit_with_error("\"%s\" has an empty sequence", path);
    }
    for (size_t i = 0; i < out->n; i++)
        for (size_t j = i + 1; j < out->n; j++)
            if (strcmp(out->name[i], out->name[j]) == 0)
                quit_with_error("\"%s\" has a duplicated name", path);
}

int orc_load_fasta(const char *path, orc_fasta *out, char *err, size_t errlen) {
    memset(out, 0, sizeof *out);
    API_ENTER(err, errlen);
    load_fasta_inner(path, out);
    API_LEAVE();
    return ORC_OK;
}

/* ------------------------------------------------------------------------ */
/* polish.rs                                                                 */
/* ------------------------------------------------------------------------ */
void orc_positions_free(orc_positions *p) {
    if (!p) return;
    free(p->depth);
    free(p->count_a);
    free(p->count_c);
    free(p->count_g);
    free(p->count_t);
    free(p->count_other);
    free(p->valid_thr);
    free(p->invalid_thr);
    free(p->status);
    free(p->emit_len);
    memset(p, 0, sizeof *p);
}
static void positions_alloc(orc_positions *p, size_t n) {
    memset(p, 0, sizeof *p);
    size_t m = n ? n : 1;
    p->n_positions = n;
    p->depth = (double *)xmalloc(m * sizeof(double));
    p->count_a = (uint32_t *)xmalloc(m * 4);
    p->count_c = (uint32_t *)xmalloc(m * 4);
    p->count_g = (uint32_t *)xmalloc(m * 4);
    p->count_t = (uint32_t *)xmalloc(m * 4);
    p->count_other = (uint32_t *)xmalloc(m * 4);
    p->valid_thr = (uint32_t *)xmalloc(m * 4);
    p->invalid_thr = (uint32_t *)xmalloc(m * 4);
    p->status = (uint8_t *)xmalloc(m);
    p->emit_len = (uint32_t *)xmalloc(m * 4);
}

/* get_read_seq_from_alignments, alignment.rs:311-322 */
static void get_read_seq_from_alignments(Alignment *al, size_t n, const char **seq, size_t *len,
                                         int *strand) {
    for (size_t i = 0; i < n; i++) {
        if (al[i].read_seq_len == 1 && al[i].read_seq[0] == '*') continue;
        *seq = al[i].read_seq;
        *len = al[i].read_seq_len;
        *strand = get_strand(&al[i]);
        return;
    }
    if (n == 0) rust_panic("called `Option::unwrap()` on a `None` value (empty read group)");
    quit_with_error("no alignments for read %s contain sequence", al[0].read_name);
}

typedef struct {
    size_t n;
    char **names;
    Pileup *pileups;
} PileupSet;

static Pileup *find_pileup(PileupSet *ps, const char *name) {
    for (size_t i = 0; i < ps->n; i++)
        if (strcmp(ps->names[i], name) == 0) return &ps->pileups[i];
    return NULL;
}

/* process_one_read, alignment.rs:275-305.  Consumes (frees) the group. */
static size_t process_one_read(Alignment *al, size_t n, PileupSet *ps, uint32_t max_errors,
                               int careful) {
    size_t used = 0;
    if (careful && n > 1) goto done;
    const char *read_seq;
    size_t read_seq_len;
    int strand;
    get_read_seq_from_alignments(al, n, &read_seq, &read_seq_len, &strand);
    {
        size_t *good = (size_t *)xmalloc((n ? n : 1) * sizeof *good);
        size_t n_good = 0;
        for (size_t i = 0; i < n; i++)
            if (starts_and_ends_with_match(&al[i]) && al[i].mismatches <= max_errors &&
                al[i].pass_qc)
                good[n_good++] = i;
        double depth_contribution = 1.0 / (double)n_good;
        /* The group sequence may belong to a good alignment that is about to
         * be left untouched (its SEQ is not "*"), so copy it first. */
        char *group_seq = xstrndup(read_seq, read_seq_len);
        for (size_t g = 0; g < n_good; g++) {
            Alignment *a = &al[good[g]];
            if (a->read_seq_len == 1 && a->read_seq[0] == '*')
                add_read_seq(a, group_seq, read_seq_len, strand);
        }
        free(group_seq);
        for (size_t g = 0; g < n_good; g++) {
            Alignment *a = &al[good[g]];
            Pileup *p = find_pileup(ps, a->ref_name);
            if (!p) {
                free(good);
                quit_with_error("query name %s in SAM but not in assembly", a->ref_name);
            }
            pileup_add_alignment(p, a, depth_contribution);
        }
        used = n_good;
        free(good);
    }
done:
    for (size_t i = 0; i < n; i++) alignment_free(&al[i]);
    return used;
}

/* add_to_pileup, alignment.rs:225-272 */
static void add_to_pileup(const char *filename, PileupSet *ps, uint32_t max_errors, int careful,
                          uint64_t *alignment_count, uint64_t *used_count, uint64_t *read_count) {
    text t = {0};
    if (read_whole_file(filename, &t) != 0)
        quit_with_error("unable to load alignments from \"%s\"", filename);
    char *current_read_name = xstrndup("", 0);
    size_t cap = 8, n = 0;
    Alignment *cur = (Alignment *)xmalloc(cap * sizeof *cur);
    uint64_t line_count = 0;
    *alignment_count = *used_count = *read_count = 0;
    const char *line;
    size_t len;
    while (next_line(&t, &line, &len)) {
        line_count++;
        if (!valid_utf8(line, len))
            quit_with_error("unable to load alignments from \"%s\"", filename);
        if (len == 0) continue;
        if (line[0] == '@') continue;
        Alignment a;
        const char *e = alignment_new(line, len, &a);
        if (e) quit_with_error("%s in \"%s\" (line %llu)", e, filename, (unsigned long long)line_count);
        if (!is_aligned(&a)) {
            alignment_free(&a);
            continue;
        }
        *alignment_count += 1;
        char *read_name = xstrndup(a.read_name, strlen(a.read_name));
        if (current_read_name[0] == 0 || strcmp(current_read_name, a.read_name) == 0) {
            if (n == cap) {
                cap *= 2;
                cur = (Alignment *)xrealloc(cur, cap * sizeof *cur);
            }
            cur[n++] = a;
        } else {
            *used_count += process_one_read(cur, n, ps, max_errors, careful);
            *read_count += 1;
            n = 0;
            cur[n++] = a;
        }
        free(current_read_name);
        current_read_name = read_name;
    }
    *used_count += process_one_read(cur, n, ps, max_errors, careful);
    *read_count += 1;
    if (*alignment_count == 0) quit_with_error("no alignments in \"%s\"", filename);
    free(cur);
    free(current_read_name);
    free(t.data);
}

/* polish_one_sequence + print_seq_to_stdout, polish.rs:157-203, and the
 * debug TSV of polish.rs:230-266 */
static size_t polish_one_sequence(const char *name, const char *desc, const Pileup *pileup,
                                  uint32_t min_depth, double fv, double fi, orc_buf *fasta,
                                  int with_header, orc_buf *debug, orc_positions *pos,
                                  size_t pos_base) {
    orc_buf polished = {0};
    buf_reserve(&polished, pileup->n);
    for (size_t i = 0; i < pileup->n; i++) {
        const PileupBase *b = &pileup->bases[i];
        vote v = pb_get_polished_seq(b, min_depth, fv, fi);
        if (debug) {
            buf_printf(debug, "%s\t%zu\t", name, i);
            pb_debug_line(b, &v, debug);
            buf_puts(debug, "\n");
        }
        if (pos && pos->depth) {
            size_t k = pos_base + i;
            uint32_t other = 0;
            for (uint32_t q = 0; q < b->n_keys; q++) other += b->key_cnt[q];
            pos->depth[k] = b->depth;
            pos->count_a[k] = b->count_a;
            pos->count_c[k] = b->count_c;
            pos->count_g[k] = b->count_g;
            pos->count_t[k] = b->count_t;
            pos->count_other[k] = other;
            pos->valid_thr[k] = v.valid_threshold;
            pos->invalid_thr[k] = v.invalid_threshold;
            pos->status[k] = (uint8_t)v.status;
        }
        /* polished_seq.push_str(&seq) then .replace("-", ""), polish.rs:185-188 */
        const size_t before = polished.len;
        for (size_t q = 0; q < v.new_len; q++)
            if (v.new_base[q] != '-') buf_append(&polished, v.new_base + q, 1);
        if (pos && pos->depth) pos->emit_len[pos_base + i] = (uint32_t)(polished.len - before);
    }
    size_t out_len = polished.len;
    if (with_header) { /* polish.rs:196-203 */
        buf_puts(fasta, ">");
        buf_puts(fasta, name);
        if (desc[0]) {
            buf_puts(fasta, " ");
            buf_puts(fasta, desc);
        }
        buf_puts(fasta, " polypolish\n");
    }
    buf_append(fasta, polished.data ? polished.data : "", polished.len);
    if (with_header) buf_puts(fasta, "\n");
    orc_buf_free(&polished);
    return out_len;
}

static int file_exists(const char *p) {
    FILE *f = fopen(p, "rb");
    if (!f) return 0;
    fclose(f);
    return 1;
}

int orc_polish_files(const char *assembly, const char *const *sams, int n_sams,
                     const orc_polish_params *p, orc_buf *fasta, orc_buf *debug,
                     orc_positions *positions, orc_polish_counts *counts, char *err,
                     size_t errlen) {
    if (positions) memset(positions, 0, sizeof *positions);
    API_ENTER(err, errlen);
    /* check_option_values, polish.rs:277-287 */
    if (p->fraction_valid <= 0.0 || p->fraction_valid >= 1.0)
        quit_with_error("--fraction_valid must be between 0 and 1 (exclusive)");
    if (p->fraction_invalid <= 0.0 || p->fraction_invalid >= 1.0)
        quit_with_error("--fraction_invalid must be between 0 and 1 (exclusive)");
    if (p->fraction_invalid >= p->fraction_valid)
        quit_with_error("--fraction_invalid must be less than --fraction_valid");
    /* check_inputs_exist, polish.rs:269-274 */
    if (!file_exists(assembly)) quit_with_error("\"%s\" file does not exist", assembly);
    for (int i = 0; i < n_sams; i++)
        if (!file_exists(sams[i])) quit_with_error("\"%s\" file does not exist", sams[i]);

    /* load_assembly, polish.rs:93-106 */
    orc_fasta fa = {0};
    load_fasta_inner(assembly, &fa);
    PileupSet ps;
    ps.n = fa.n;
    ps.names = fa.name;
    ps.pileups = (Pileup *)xmalloc(fa.n * sizeof *ps.pileups);
    size_t total = 0;
    for (size_t i = 0; i < fa.n; i++) {
        pileup_new(&ps.pileups[i], fa.seq[i], fa.len[i]);
        total += fa.len[i];
    }
    /* load_alignments, polish.rs:109-134 */
    orc_polish_counts c = {0, 0, 0};
    for (int i = 0; i < n_sams; i++) {
        uint64_t ac, uc, rc;
        add_to_pileup(sams[i], &ps, p->max_errors, p->careful, &ac, &uc, &rc);
        c.alignment_total += ac;
        c.used_total += uc;
        c.read_total += rc;
    }
    if (counts) *counts = c;
    /* polish_sequences, polish.rs:137-154 */
    if (debug)
        buf_puts(debug, "name\tpos\tbase\tdepth\tinvalid\tvalid\tpileup\tstatus\tnew_base\n");
    if (positions) positions_alloc(positions, total);
    size_t base = 0;
    for (size_t i = 0; i < fa.n; i++) {
        polish_one_sequence(fa.name[i], fa.desc[i], &ps.pileups[i], p->min_depth,
                            p->fraction_valid, p->fraction_invalid, fasta, 1, debug, positions,
                            base);
        base += fa.len[i];
    }
    for (size_t i = 0; i < fa.n; i++) pileup_free(&ps.pileups[i]);
    free(ps.pileups);
    orc_fasta_free(&fa);
    API_LEAVE();
    return ORC_OK;
}

/* ------------------------------------------------------------------------ */
/* record-level entry: the same SoA the product's C ABI takes                */
/* ------------------------------------------------------------------------ */
static const char OPCH[] = "MIDNSHP=X";

int orc_polish_records(uint32_t n_contigs, const uint64_t *contig_off, const uint8_t *bases,
                       const orc_records *r, uint32_t min_depth, double fraction_valid,
                       double fraction_invalid, orc_buf *polished, uint64_t *polished_off,
                       orc_positions *positions, char *err, size_t errlen) {
    if (positions) memset(positions, 0, sizeof *positions);
    API_ENTER(err, errlen);
    Pileup *pl = (Pileup *)xmalloc((n_contigs ? n_contigs : 1) * sizeof *pl);
    for (uint32_t c = 0; c < n_contigs; c++)
        pileup_new(&pl[c], (const char *)bases + contig_off[c],
                   (size_t)(contig_off[c + 1] - contig_off[c]));
    for (uint64_t i = 0; i < r->n_aln; i++) {
        Alignment a;
        memset(&a, 0, sizeof a);
        char nm[32], cn[32];
        snprintf(nm, sizeof nm, "aln%llu", (unsigned long long)i);
        snprintf(cn, sizeof cn, "contig%u", r->contig[i]);
        a.read_name = nm;
        a.ref_name = cn;
        a.ref_start = r->ref_start[i];
        a.read_seq = (char *)(uintptr_t)(r->seq + r->seq_off[i]);
        a.read_seq_len = r->seq_len[i];
        /* rebuild the CIGAR text and its expansion from the packed runs */
        orc_buf cig = {0};
        size_t exp_len = 0;
        for (uint32_t q = 0; q < r->n_cig[i]; q++) {
            uint32_t op = r->cigar[r->cig_off[i] + q];
            if ((op & 15u) > 8u) quit_with_error("bad packed CIGAR op for record %s", nm);
            buf_printf(&cig, "%u%c", op >> 4, OPCH[op & 15u]);
            exp_len += op >> 4;
        }
        char *expanded = (char *)xmalloc(exp_len + 1);
        size_t w = 0;
        for (uint32_t q = 0; q < r->n_cig[i]; q++) {
            uint32_t op = r->cigar[r->cig_off[i] + q];
            memset(expanded + w, OPCH[op & 15u], op >> 4);
            w += op >> 4;
        }
        expanded[w] = 0;
        a.cigar = cig.data ? cig.data : (char *)"";
        a.expanded_cigar = expanded;
        a.expanded_len = exp_len;
        if (r->contig[i] >= n_contigs)
            quit_with_error("query name %s in SAM but not in assembly", cn);
        if (r->k[i] == 0) quit_with_error("record %s has k = 0", nm);
        pileup_add_alignment(&pl[r->contig[i]], &a, 1.0 / (double)r->k[i]);
        free(expanded);
        orc_buf_free(&cig);
    }
    size_t total = (size_t)contig_off[n_contigs];
    if (positions) positions_alloc(positions, total);
    for (uint32_t c = 0; c < n_contigs; c++) {
        polished_off[c] = polished->len;
        polish_one_sequence("", "", &pl[c], min_depth, fraction_valid, fraction_invalid, polished,
                            0, NULL, positions, (size_t)contig_off[c]);
    }
    polished_off[n_contigs] = polished->len;
    for (uint32_t c = 0; c < n_contigs; c++) pileup_free(&pl[c]);
    free(pl);
    API_LEAVE();
    return ORC_OK;
}

/* ------------------------------------------------------------------------ */
/* filter.rs                                                                 */
/* ------------------------------------------------------------------------ */

/* filter.rs:189-209: 0 fr, 1 rf, 2 ff, 3 rr */
static int get_orientation(const Alignment *a1, const Alignment *a2) {
    int f1 = is_on_forward_strand(a1), f2 = is_on_forward_strand(a2);
    uint64_t p1 = f1 ? a1->ref_start : orc_get_ref_end(a1->ref_start, a1->cigar);
    uint64_t p2 = f2 ? a2->ref_start : orc_get_ref_end(a2->ref_start, a2->cigar);
    if (f1 != f2) {
        /* "{s1}{s2}" if p1 < p2 else "{s2}{s1}" */
        int first_is_f = (p1 < p2) ? f1 : f2;
        return first_is_f ? 0 : 1;
    }
    if (f1) return p1 < p2 ? 2 : 3;
    return p2 < p1 ? 2 : 3;
}

/* filter.rs:212-218 */
static uint32_t get_insert_size(const Alignment *a1, const Alignment *a2) {
    uint64_t pos[4] = {a1->ref_start, orc_get_ref_end(a1->ref_start, a1->cigar), a2->ref_start,
                       orc_get_ref_end(a2->ref_start, a2->cigar)};
    uint64_t lo = pos[0], hi = pos[0];
    for (int i = 1; i < 4; i++) {
        if (pos[i] < lo) lo = pos[i];
        if (pos[i] > hi) hi = pos[i];
    }
    return (uint32_t)(hi - lo);
}

int orc_get_orientation(uint32_t flags1, uint64_t start1, const char *cigar1, uint32_t flags2,
                        uint64_t start2, const char *cigar2) {
    Alignment a, b;
    memset(&a, 0, sizeof a);
    memset(&b, 0, sizeof b);
    a.sam_flags = flags1; a.ref_start = start1; a.cigar = (char *)(uintptr_t)cigar1;
    b.sam_flags = flags2; b.ref_start = start2; b.cigar = (char *)(uintptr_t)cigar2;
    return get_orientation(&a, &b);
}
uint32_t orc_get_insert_size(uint64_t start1, const char *cigar1, uint64_t start2,
                             const char *cigar2) {
    Alignment a, b;
    memset(&a, 0, sizeof a);
    memset(&b, 0, sizeof b);
    a.ref_start = start1; a.cigar = (char *)(uintptr_t)cigar1;
    b.ref_start = start2; b.cigar = (char *)(uintptr_t)cigar2;
    return get_insert_size(&a, &b);
}

/* filter.rs:249-259 */
uint32_t orc_get_percentile(const uint32_t *sorted, size_t n, double percentile) {
    if (n == 0) return 0;
    double fraction = percentile / 100.0;
    double r = ceil(fraction * (double)n);
    size_t rank;
    if (!(r == r) || r <= 0.0) rank = 0; else if (r >= 1.8446744073709552e19) rank = SIZE_MAX; else rank = (size_t)r;
    if (rank < 1) rank = 1;
    if (rank - 1 < n) return sorted[rank - 1];
    return 0;
}

/* filter.rs:238-246 */
int orc_auto_determine_orientation(const uint64_t counts[4]) {
    uint64_t max_count = 0;
    for (int i = 0; i < 4; i++)
        if (counts[i] > max_count) max_count = counts[i];
    int n = 0, which = -1;
    for (int i = 0; i < 4; i++)
        if (counts[i] == max_count) {
            n++;
            which = i;
        }
    return n == 1 ? which : -1;
}

/* HashMap<String, Vec<Alignment>> keyed by QNAME + "_1"/"_2" (filter.rs:91-145),
 * held here as one entry per QNAME with one list per file -- the same
 * partition, because a "_1" key can only come from file 1 and a "_2" key only
 * from file 2. */
typedef struct {
    char *name;
    Alignment *al[2];
    uint32_t n[2], cap[2];
} ReadEntry;

typedef struct {
    ReadEntry *e;
    size_t n, cap;
    uint32_t *table; /* open addressing, value = index+1 */
    size_t tcap;
} ReadMap;

static uint64_t fnv1a(const char *s, size_t n) {
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; i++) {
        h ^= (unsigned char)s[i];
        h *= 1099511628211ull;
    }
    return h;
}
static void readmap_rehash(ReadMap *m, size_t tcap) {
    free(m->table);
    m->table = (uint32_t *)xmalloc(tcap * sizeof *m->table);
    memset(m->table, 0, tcap * sizeof *m->table);
    m->tcap = tcap;
    for (size_t i = 0; i < m->n; i++) {
        size_t h = fnv1a(m->e[i].name, strlen(m->e[i].name)) & (tcap - 1);
        while (m->table[h]) h = (h + 1) & (tcap - 1);
        m->table[h] = (uint32_t)(i + 1);
    }
}
static ReadEntry *readmap_get(ReadMap *m, const char *name, int create) {
    if (m->tcap == 0) readmap_rehash(m, 1024);
    size_t len = strlen(name);
    size_t h = fnv1a(name, len) & (m->tcap - 1);
    while (m->table[h]) {
        ReadEntry *e = &m->e[m->table[h] - 1];
        if (strcmp(e->name, name) == 0) return e;
        h = (h + 1) & (m->tcap - 1);
    }
    if (!create) return NULL;
    if (m->n == m->cap) {
        m->cap = m->cap ? m->cap * 2 : 1024;
        m->e = (ReadEntry *)xrealloc(m->e, m->cap * sizeof *m->e);
    }
    ReadEntry *e = &m->e[m->n];
    memset(e, 0, sizeof *e);
    e->name = xstrndup(name, len);
    m->n++;
    if (m->n * 2 > m->tcap) {
        readmap_rehash(m, m->tcap * 2);
    } else {
        m->table[h] = (uint32_t)m->n;
    }
    return &m->e[m->n - 1];
}
static void readmap_free(ReadMap *m) {
    for (size_t i = 0; i < m->n; i++) {
        for (int f = 0; f < 2; f++) {
            for (uint32_t j = 0; j < m->e[i].n[f]; j++) alignment_free(&m->e[i].al[f][j]);
            free(m->e[i].al[f]);
        }
        free(m->e[i].name);
    }
    free(m->e);
    free(m->table);
    memset(m, 0, sizeof *m);
}

/* load_alignments_one_file, filter.rs:110-145 */
static void load_alignments_one_file(const char *filename, ReadMap *m, int file_idx,
                                     uint64_t *total_alignments) {
    text t = {0};
    if (read_whole_file(filename, &t) != 0)
        quit_with_error("unable to load alignments from \"%s\"", filename);
    uint64_t line_count = 0;
    const char *line;
    size_t len;
    while (next_line(&t, &line, &len)) {
        line_count++;
        if (!valid_utf8(line, len))
            quit_with_error("unable to load alignments from \"%s\"", filename);
        if (len > 0 && line[0] == '@') continue;
        Alignment a;
        const char *e = alignment_new_quick(line, len, &a);
        if (e) quit_with_error("%s in \"%s\" (line %llu)", e, filename, (unsigned long long)line_count);
        if (!is_aligned(&a)) {
            alignment_free(&a);
            continue;
        }
        ReadEntry *re = readmap_get(m, a.read_name, 1);
        if (re->n[file_idx] == re->cap[file_idx]) {
            re->cap[file_idx] = re->cap[file_idx] ? re->cap[file_idx] * 2 : 1;
            re->al[file_idx] = (Alignment *)xrealloc(re->al[file_idx],
                                                     re->cap[file_idx] * sizeof(Alignment));
        }
        re->al[file_idx][re->n[file_idx]++] = a;
        *total_alignments += 1;
    }
    free(t.data);
    if (*total_alignments == 0) quit_with_error("no alignments found in \"%s\"", filename);
}

static int cmp_u32(const void *a, const void *b) {
    uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b;
    return x < y ? -1 : x > y;
}

/* alignment_pass_qc, filter.rs:352-377 */
static int alignment_pass_qc(const Alignment *a, const Alignment *this_al, uint32_t n_this,
                             const Alignment *pair_al, uint32_t n_pair, uint32_t low,
                             uint32_t high, int correct_orientation) {
    (void)this_al;
    if (n_pair == 0) return 1;
    if (n_this == 1) return 1;
    for (uint32_t i = 0; i < n_pair; i++) {
        const Alignment *p = &pair_al[i];
        int same_ref = strcmp(a->ref_name, p->ref_name) == 0;
        uint32_t insert = get_insert_size(a, p);
        int orientation = get_orientation(a, p);
        if (same_ref && low <= insert && insert <= high && orientation == correct_orientation)
            return 1;
    }
    return 0;
}

/* filter_sam, filter.rs:296-349 */
static uint64_t filter_sam(const char *in_filename, const char *out_filename, ReadMap *m,
                           uint32_t low, uint32_t high, int correct_orientation, int read_num) {
    text t = {0};
    if (read_whole_file(in_filename, &t) != 0)
        quit_with_error("unable to write alignments to \"%s\"", out_filename);
    FILE *out = fopen(out_filename, "wb");
    if (!out) quit_with_error("unable to write alignments to \"%s\"", out_filename);
    uint64_t pass_count = 0;
    const char *line;
    size_t len;
    int this_idx = read_num == 1 ? 0 : 1;
    while (next_line(&t, &line, &len)) {
        if (len > 0 && line[0] == '@') {
            fwrite(line, 1, len, out);
            fputc('\n', out);
            continue;
        }
        Alignment a;
        const char *e = alignment_new_quick(line, len, &a);
        if (e) {
            fclose(out);
            rust_panic("called `Result::unwrap()` on an `Err` value: %s", e);
        }
        if (!is_aligned(&a)) {
            fwrite(line, 1, len, out);
            fputc('\n', out);
            alignment_free(&a);
            continue;
        }
        ReadEntry *re = readmap_get(m, a.read_name, 0);
        if (!re || re->n[this_idx] == 0) {
            fclose(out);
            rust_panic("read %s not found in the loaded alignments", a.read_name);
        }
        if (alignment_pass_qc(&a, re->al[this_idx], re->n[this_idx], re->al[1 - this_idx],
                              re->n[1 - this_idx], low, high, correct_orientation)) {
            fwrite(line, 1, len, out);
            fputc('\n', out);
            pass_count++;
        } else {
            fwrite(line, 1, len, out);
            fputs("\tZP:Z:fail\n", out);
        }
        alignment_free(&a);
    }
    fclose(out);
    free(t.data);
    return pass_count;
}

int orc_filter_files(const char *in1, const char *in2, const char *out1, const char *out2,
                     const char *orientation, double low, double high, orc_filter_report *rep,
                     char *err, size_t errlen) {
    if (rep) memset(rep, 0, sizeof *rep);
    API_ENTER(err, errlen);
    /* check_inputs, filter.rs:40-53 */
    const char *f[4] = {in1, in2, out1, out2};
    for (int i = 0; i < 4; i++)
        for (int j = i + 1; j < 4; j++)
            if (strcmp(f[i], f[j]) == 0)
                quit_with_error("--in1, --in2, --out1 and --out2 must all have unique values");
    if (low <= 0.0 || low >= 50.0) quit_with_error("--low must be greater than 0 and less than 50");
    if (high <= 50.0 || high >= 100.0)
        quit_with_error("--high must be greater than 50 and less than 100");

    /* load_alignments, filter.rs:91-107 */
    ReadMap m;
    memset(&m, 0, sizeof m);
    uint64_t before = 0;
    load_alignments_one_file(in1, &m, 0, &before);
    load_alignments_one_file(in2, &m, 1, &before);

    /* get_insert_size_thresholds, filter.rs:148-186 */
    uint32_t *sizes[4] = {NULL, NULL, NULL, NULL};
    size_t ns[4] = {0, 0, 0, 0}, cs[4] = {0, 0, 0, 0};
    for (size_t i = 0; i < m.n; i++) {
        ReadEntry *re = &m.e[i];
        if (re->n[0] != 1) continue;
        if (re->n[1] == 1 && strcmp(re->al[0][0].ref_name, re->al[1][0].ref_name) == 0) {
            int o = get_orientation(&re->al[0][0], &re->al[1][0]);
            uint32_t ins = get_insert_size(&re->al[0][0], &re->al[1][0]);
            if (ns[o] == cs[o]) {
                cs[o] = cs[o] ? cs[o] * 2 : 1024;
                sizes[o] = (uint32_t *)xrealloc(sizes[o], cs[o] * sizeof(uint32_t));
            }
            sizes[o][ns[o]++] = ins;
        }
    }
    if (ns[0] + ns[1] + ns[2] + ns[3] == 0)
        quit_with_error("no one-alignment-per-read pairs available to determine orientation and "
                        "insert size thresholds");
    /* determine_correct_orientation, filter.rs:221-235 */
    int correct;
    uint64_t oc[4] = {ns[0], ns[1], ns[2], ns[3]};
    if (strcmp(orientation, "auto") == 0) {
        correct = orc_auto_determine_orientation(oc);
        if (correct < 0) quit_with_error("could not automatically determine read pair orientation");
    } else {
        static const char *names[4] = {"fr", "rf", "ff", "rr"};
        correct = -1;
        for (int i = 0; i < 4; i++)
            if (strcmp(orientation, names[i]) == 0) correct = i;
    }
    if (correct < 0 || ns[correct] == 0)
        quit_with_error("no read pairs available to determine insert size thresholds");
    qsort(sizes[correct], ns[correct], sizeof(uint32_t), cmp_u32);
    uint32_t low_threshold = orc_get_percentile(sizes[correct], ns[correct], low);
    uint32_t high_threshold = orc_get_percentile(sizes[correct], ns[correct], high);

    /* filter_sams, filter.rs:273-293 */
    uint64_t after = 0;
    after += filter_sam(in1, out1, &m, low_threshold, high_threshold, correct, 1);
    after += filter_sam(in2, out2, &m, low_threshold, high_threshold, correct, 2);
    if (rep) {
        rep->before_count = before;
        rep->after_count = after;
        rep->low_threshold = low_threshold;
        rep->high_threshold = high_threshold;
        rep->orientation = correct;
        for (int i = 0; i < 4; i++) rep->orientation_counts[i] = oc[i];
    }
    for (int i = 0; i < 4; i++) free(sizes[i]);
    readmap_free(&m);
    API_LEAVE();
    return ORC_OK;
}
