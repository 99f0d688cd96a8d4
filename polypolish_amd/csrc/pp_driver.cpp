// pp_driver.cpp -- whole-command drivers behind the `polypolish` CLI contract.
//   pp_polish_files  = polish::polish   (src/polish.rs:26-38)
// Output bytes (FASTA on stdout, filtered SAMs) are the contract; the stderr log reproduces the
// reference's numbers (counts, depth, changed positions, Q-score) in plain text -- the
// reference's colours, wrapping and timestamps (src/log.rs) are cosmetic and not reproduced.
#include <sys/stat.h>

#include <chrono>
#include <functional>
#include <future>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include <algorithm>

#include "polypolish_hip.h"
#include "pp_host.h"

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

// num-format's Locale::en grouping (polish.rs:99 etc.)
std::string commas(uint64_t v) {
    std::string s = std::to_string(v), out;
    int n = (int)s.size();
    for (int i = 0; i < n; i++) {
        out.push_back(s[i]);
        int left = n - 1 - i;
        if (left > 0 && left % 3 == 0) out.push_back(',');
    }
    return out;
}

bool exists(const char *p) {
    struct stat st;
    return stat(p, &st) == 0;
}

using pph::format_duration;
using pph::qscore;

int set_err(pp_ctx *ctx, int code, const char *msg);

}  // namespace

// pp_ctx is opaque here; error text is stored through a tiny hook exported by pp_kernels.hip
extern "C" int pp_ctx_set_error_(pp_ctx *ctx, int code, const char *msg);
namespace {
int set_err(pp_ctx *ctx, int code, const char *msg) { return pp_ctx_set_error_(ctx, code, msg); }
}  // namespace

// (internal, bin/polypolish only) the process exits right after the command: see pph::process_leaving_soon
extern "C" void pp_process_leaving_soon_(int yes) { pph::process_leaving_soon() = yes != 0; }

extern "C" void pp_bytes_free(pp_bytes *b) {
    if (!b) return;
    free(b->data);
    b->data = nullptr;
    b->len = 0;
}

// write_debug_line (polish.rs:257-266) + get_debug_line / get_count_str (pileup.rs:137-166) for every
// position, from the device's per-position records
static int write_debug_tsv(pp_ctx *ctx, FILE *f, const pp_assembly *a, const pp_aln_batch *batch) {
    const uint32_t nc = pp_assembly_n_contigs(a);
    const uint64_t *off = pp_assembly_offsets(a);
    const uint8_t *bases = pp_assembly_bases(a);
    const uint64_t G = off[nc];
    std::vector<double> depth(G ? G : 1);
    std::vector<uint32_t> ca(G ? G : 1), cc(G ? G : 1), cg(G ? G : 1), ct(G ? G : 1), vt(G ? G : 1), it(G ? G : 1);
    std::vector<uint8_t> st(G ? G : 1);
    pp_positions pos{depth.data(), ca.data(), cc.data(), cg.data(), ct.data(), nullptr, vt.data(), it.data(), st.data()};
    int rc = pp_polish_positions(ctx, &pos);
    if (rc) return rc;
    pp_debug_extra x;
    rc = pp_polish_debug_extra(ctx, &x);
    if (rc) return rc;
    // key records and multi-byte winners, grouped by position
    std::vector<uint64_t> korder(x.n_keys), morder(x.n_multi);
    for (uint64_t i = 0; i < x.n_keys; i++) korder[i] = i;
    for (uint64_t i = 0; i < x.n_multi; i++) morder[i] = i;
    std::sort(korder.begin(), korder.end(), [&](uint64_t l, uint64_t r) { return x.key_pos[l] < x.key_pos[r]; });
    std::sort(morder.begin(), morder.end(), [&](uint64_t l, uint64_t r) { return x.multi_pos[l] < x.multi_pos[r]; });
    static const char *STATUS[6] = {"kept", "changed", "low_depth", "none", "multiple", "too_close"};
    uint64_t ki = 0, mi = 0;
    std::string line;
    std::vector<std::string> items;
    char num[64];
    for (uint32_t c = 0; c < nc; c++) {
        const char *name = pp_assembly_name(a, c);
        for (uint64_t gp = off[c]; gp < off[c + 1]; gp++) {
            items.clear();
            if (ca[gp]) { snprintf(num, sizeof num, "Ax%u", ca[gp]); items.push_back(num); }
            if (cc[gp]) { snprintf(num, sizeof num, "Cx%u", cc[gp]); items.push_back(num); }
            if (cg[gp]) { snprintf(num, sizeof num, "Gx%u", cg[gp]); items.push_back(num); }
            if (ct[gp]) { snprintf(num, sizeof num, "Tx%u", ct[gp]); items.push_back(num); }
            for (; ki < x.n_keys && x.key_pos[korder[ki]] == gp; ki++) {
                const uint64_t r = korder[ki];
                std::string key = x.key_len[r] ? std::string((const char *)batch->seq + x.key_off[r], x.key_len[r]) : std::string("-");
                snprintf(num, sizeof num, "x%u", x.key_count[r]);
                items.push_back(key + num);
            }
            std::sort(items.begin(), items.end());
            const uint8_t e = x.emit[gp];
            std::string new_base;
            if (e == 0) new_base = st[gp] == PP_ST_CHANGED ? std::string("-") : std::string(1, (char)bases[gp]);
            else if (e < 0x80) new_base = std::string(1, (char)e);
            else {
                while (mi < x.n_multi && x.multi_pos[morder[mi]] < gp) mi++;
                if (mi < x.n_multi && x.multi_pos[morder[mi]] == gp)
                    new_base = std::string((const char *)batch->seq + x.multi_off[morder[mi]], x.multi_len[morder[mi]]);
            }
            line.assign(name);
            snprintf(num, sizeof num, "\t%llu\t%c\t%.1f\t%u\t%u\t", (unsigned long long)(gp - off[c]), (char)bases[gp],
                     depth[gp], it[gp], vt[gp]);
            line += num;
            for (size_t i = 0; i < items.size(); i++) { if (i) line += ','; line += items[i]; }
            line += '\t';
            line += STATUS[st[gp] < 6 ? st[gp] : 0];
            line += '\t';
            line += new_base;
            line += '\n';
            if (fwrite(line.data(), 1, line.size(), f) != line.size()) {
                pp_debug_extra_free(&x);
                return pp_ctx_set_error_(ctx, PP_ERR_QUIT, "unable to write to the --debug file");
            }
        }
    }
    pp_debug_extra_free(&x);
    return PP_OK;
}

static int polish_files_impl(pp_ctx *const *ctxs, int n_ctx, const char *assembly, const char *const *sams, int n_sams,
                             const pp_polish_options *opt, pp_bytes *fasta, const uint8_t *const *pass,
                             const uint64_t *n_pass, int resume_log_at = -1);
extern "C" int pp_dev_ingest_reserve_text_(pp_dev_ingest *D, uint64_t bytes);
extern "C" void pp_dev_ingest_prefetch_(pp_dev_ingest *D, const char *path, uint64_t second_buffer_bytes);
extern "C" int pp_ingest_fail_cut_(const pp_ingest *I, uint64_t *cut);
extern "C" int pp_ingest_sam_prefix_(pp_ingest *I, const char *path, uint64_t cut, const uint8_t *pass, uint64_t n_pass,
                                     pp_sam_counts *counts, char *err, size_t errlen);

extern "C" int pp_polish_files_filtered_(pp_ctx *ctx, const char *assembly, const char *const *sams, int n_sams,
                                         const pp_polish_options *opt, pp_bytes *fasta,
                                         const uint8_t *const *pass, const uint64_t *n_pass) {
    return polish_files_impl(&ctx, 1, assembly, sams, n_sams, opt, fasta, pass, n_pass);
}

extern "C" int pp_polish_files(pp_ctx *ctx, const char *assembly, const char *const *sams, int n_sams,
                               const pp_polish_options *opt, pp_bytes *fasta) {
    return polish_files_impl(&ctx, 1, assembly, sams, n_sams, opt, fasta, nullptr, nullptr);
}

// One process, several GPUs (polish::polish has no counterpart: src/polish.rs:137-154 is one thread): every context
// uploads and tokenizes its own slice of every SAM file (or the host ingest parses once), the records are partitioned --
// a context is sent the records that reach its units (pp_shard_split) -- the contexts polish side by side on their own
// threads, and every device copies its own share of the polished bytes out for the host to put together -- or, with
// PP_GATHER=rccl, the bytes meet on the first context's GPU in ONE RCCL gather over xGMI (pp_polish_gather: the north
// star's "single RCCL gather for the final FASTA") followed by one device-to-host copy.  The RCCL route of THIS driver is
// opt-in until it has run on a multi-GPU node (see below); it cannot run without librccl or with two contexts on one
// device (PP_SHARE_GPU, tests).  PP_TIMING prints the route that was taken.
extern "C" int pp_polish_files_multi(pp_ctx *const *ctxs, int n_ctx, const char *assembly, const char *const *sams,
                                     int n_sams, const pp_polish_options *opt, pp_bytes *fasta) {
    if (!ctxs || n_ctx < 1) return PP_ERR_ARG;
    for (int i = 0; i < n_ctx; i++)
        if (!ctxs[i]) return PP_ERR_ARG;
    return polish_files_impl(ctxs, n_ctx, assembly, sams, n_sams, opt, fasta, nullptr, nullptr);
}

extern "C" void pp_ctx_enable_peers_(pp_ctx *const *ctxs, int n);
extern "C" int pp_ctx_device_(const pp_ctx *ctx);
extern "C" int pp_shard_split_view_(pp_ctx *ctx, const pp_shard_plan *plan, uint32_t dest, const pp_aln_batch *batch, int mem,
                                    uint32_t wo_idx_base, pp_shard_part **out);
extern "C" int pp_polish_gather_to_host_(pp_ctx *ctx, uint8_t *host_out, uint64_t cap, uint64_t *rank_len, uint64_t *rank_contig_off);
extern "C" int pp_dev_ingest_slice_(pp_dev_ingest *D, const char *path, const char *text, uint64_t size, pp_sam_counts *counts);

namespace {

// ---- several GPUs, ONE copy of the text: every GPU uploads and tokenizes a slice of each SAM file ---------------------
// Where may a file be cut?  At the start of a line that opens a read group (src/alignment.rs:255-263: an aligned record
// joins the group of the aligned record before it when that one's QNAME is empty or equal; header, empty and unaligned
// lines neither join nor close anything).  Returns the first such line start at or after `from`, or `size`.
struct LineView { const char *q; size_t qlen; bool aligned; };
LineView look_at_line(const char *text, size_t ls, size_t le) {
    LineView v{text + ls, 0, false};
    if (le == ls || text[ls] == '@') return v;
    const char *tab = (const char *)memchr(text + ls, '\t', le - ls);
    if (!tab) return v;
    v.qlen = (size_t)(tab - (text + ls));
    const char *f = tab + 1, *end = text + le;
    if (f < end && *f == '+') f++;
    unsigned long long flag = 0;
    const char *d = f;
    while (d < end && *d >= '0' && *d <= '9' && d - f < 12) flag = flag * 10 + (unsigned long long)(*d++ - '0');
    if (d == f || (d < end && *d != '\t')) return v;  // not a number: the tokenizer will say so; no cut here
    v.aligned = (flag & 4ull) == 0;
    return v;
}
size_t group_cut(const char *text, size_t size, size_t from) {
    if (from == 0) return 0;
    if (from >= size) return size;
    // the start of the first line at or after `from`
    size_t p = from;
    if (text[p - 1] != '\n') {
        const char *nl = (const char *)memchr(text + p, '\n', size - p);
        if (!nl) return size;
        p = (size_t)(nl - text) + 1;
    }
    // the aligned record before it (scan back over header / empty / unaligned lines; give up after a while: then the
    // first aligned line at or after p is compared with nothing and the search just moves on one group)
    bool have_prev = false, unknown_prev = false;
    const char *pq = nullptr;
    size_t pql = 0;
    {
        size_t e = p;  // e = one past the '\n' that ends the line being looked at
        for (int tries = 0; tries < 4096 && e > 0; tries++) {
            const size_t le = e - 1;  // the '\n'
            size_t ls = le;
            while (ls > 0 && text[ls - 1] != '\n') ls--;
            const LineView v = look_at_line(text, ls, le);
            if (v.aligned) { have_prev = true; pq = v.q; pql = v.qlen; break; }
            e = ls;
        }
        // A long stretch without aligned records (unmapped reads grouped together): the aligned record before it is out
        // of sight, so the first aligned line from p on may still belong to ITS group.  It is then taken as the record to
        // compare with, and the cut falls on the first QNAME change after it -- a boundary whatever came before.  (Until
        // round 4 this gave up and returned `size`: every later cut then fell on `size` too and one GPU tokenized the rest
        // of the file.)
        unknown_prev = !have_prev && e > 0;
    }
    while (p < size) {
        const char *nl = (const char *)memchr(text + p, '\n', size - p);
        const size_t le = nl ? (size_t)(nl - text) : size;
        const LineView v = look_at_line(text, p, le);
        if (v.aligned) {
            if (unknown_prev) unknown_prev = false;
            else if (!have_prev || (pql != 0 && (pql != v.qlen || memcmp(pq, v.q, pql) != 0))) return p;
            have_prev = true; pq = v.q; pql = v.qlen;
        }
        p = le + 1;
    }
    return size;
}

// one stretch of records handed to a destination context, for turning its rank-local record numbers back into the job's
struct Piece { pp_shard_part *part; int src; uint64_t base, n; };

// the job-wide number of the record a context's device error is about, or ~0 (not a record-level error)
uint64_t job_record_of(pp_ctx *cd, pp_ctx *const *ctxs, const std::vector<Piece> &pieces, uint32_t *kind) {
    uint64_t local = 0;
    if (!pp_polish_error_record(cd, &local, kind)) return ~0ull;
    uint64_t at = 0;
    for (const Piece &pc : pieces) {
        if (local < at + pc.n) {
            const uint32_t *orig = nullptr;
            pp_shard_part_batch(pc.part, nullptr, &orig);
            uint32_t o = 0;
            if (pp_shard_part_mem(pc.part) == PP_MEM_HOST) o = orig[local - at];
            else if (pp_ctx_download(ctxs[pc.src], &o, orig + (local - at), 4) != PP_OK) return ~0ull;
            return pc.base + o;
        }
        at += pc.n;
    }
    return ~0ull;
}

}  // namespace

// (tests) where the multi-GPU driver would cut `text` at or after `from`
extern "C" uint64_t pp_sam_group_cut_(const char *text, uint64_t size, uint64_t from) { return group_cut(text, (size_t)size, (size_t)from); }

// pass / n_pass: optional per-file filter verdicts (pp_ingest_sam_filtered), used by pp_filter_polish_files
static int polish_files_impl(pp_ctx *const *ctxs, int n_ctx, const char *assembly, const char *const *sams, int n_sams,
                             const pp_polish_options *opt, pp_bytes *fasta, const uint8_t *const *pass,
                             const uint64_t *n_pass, int resume_log_at) {
    pp_ctx *const ctx = ctxs[0];  // carries the error text
    if (!ctx || !assembly || !opt || !fasta || (n_sams > 0 && !sams)) return PP_ERR_ARG;
    const bool multi = n_ctx > 1;
    if (multi && opt->debug_path) return set_err(ctx, PP_ERR_ARG, "--debug needs a single GPU");
    fasta->data = nullptr;
    fasta->len = 0;
    // resume_log_at >= 0: the second look at an input the device tokenizer handed back (a defect in the text, or bytes
    // outside ASCII that the host parsers must judge), on the host ingest only.  The banner, the assembly section and the
    // lines of the files before that one are on stderr already: the log resumes at that file.
    const bool host_ingest_only = resume_log_at >= 0;
    Log log{opt->quiet != 0 || host_ingest_only};
    auto t0 = std::chrono::steady_clock::now();
    char err[1024] = "";

    // check_option_values (polish.rs:277-287) is repeated by pp_polish_begin; do it first as the reference does
    if (opt->fraction_valid <= 0.0 || opt->fraction_valid >= 1.0)
        return set_err(ctx, PP_ERR_QUIT, "--fraction_valid must be between 0 and 1 (exclusive)");
    if (opt->fraction_invalid <= 0.0 || opt->fraction_invalid >= 1.0)
        return set_err(ctx, PP_ERR_QUIT, "--fraction_invalid must be between 0 and 1 (exclusive)");
    if (opt->fraction_invalid >= opt->fraction_valid)
        return set_err(ctx, PP_ERR_QUIT, "--fraction_invalid must be less than --fraction_valid");
    // check_inputs_exist, polish.rs:269-274
    if (!exists(assembly)) {
        snprintf(err, sizeof err, "\"%s\" file does not exist", assembly);
        return set_err(ctx, PP_ERR_QUIT, err);
    }
    for (int i = 0; i < n_sams; i++)
        if (!exists(sams[i])) {
            snprintf(err, sizeof err, "\"%s\" file does not exist", sams[i]);
            return set_err(ctx, PP_ERR_QUIT, err);
        }

    // The device tokenizer uploads the SAM text as it is: map the files and pre-fault the mappings NOW, on background
    // threads, while the HIP runtime is still initialising (a copy out of an untouched mapping runs at a quarter of
    // the link's rate).
    // Several contexts: every GPU uploads and tokenizes its own slice of each file (`sharded`), so a byte of text crosses
    // PCIe once; the host ingest (PP_DEVICE_INGEST=0, or a file the tokenizer handed back) parses once and sends every
    // context the records that reach its units.
    const bool dev_ingest = !host_ingest_only && !(getenv("PP_DEVICE_INGEST") && atoi(getenv("PP_DEVICE_INGEST")) == 0) && !opt->debug_path;
    if (dev_ingest)
        for (int i = 0; i < n_sams; i++) pph::prefetch_file(sams[i], ctx);
    struct DropPrefetched { const void *owner; ~DropPrefetched() { pph::prefetch_drop_all(owner); } } drop_prefetched{ctx};

    // starting_message, polish.rs:41-73
    log("\nStarting Polypolish polish\n%s\n\nInput assembly:\n  %s\n\nInput short-read alignments:\n", pp_version(), assembly);
    for (int i = 0; i < n_sams; i++) log("  %s\n", sams[i]);
    log("\nSettings:\n  --fraction_invalid %g\n  --fraction_valid %g\n  --max_errors %u\n  --min_depth %u\n",
        opt->fraction_invalid, opt->fraction_valid, opt->max_errors, opt->min_depth);
    if (opt->careful) log("  --careful\n");
    if (opt->debug_path) log("  --debug %s\n\n", opt->debug_path);
    else log("  not logging debugging information\n\n");

    const bool timing = getenv("PP_TIMING") != nullptr;
    // PP_TIMING=1: "[timing] <stage>  <seconds since the driver was entered>  (<seconds since the process started>)"
    auto lap = [&](const char *what) {
        if (timing) fprintf(stderr, "[timing] %-36s %8.3f s  (process %7.3f s)\n", what,
                            std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(), pph::seconds_since_process_start());
    };
    lap("driver entered");
    // load_assembly, polish.rs:93-106
    log("Loading assembly\n");
    pp_assembly *a = nullptr;
    int rc = pp_assembly_load(assembly, &a, err, sizeof err);
    if (rc) return set_err(ctx, rc, err);
    const uint32_t nc = pp_assembly_n_contigs(a);
    const uint64_t *off = pp_assembly_offsets(a);
    for (uint32_t c = 0; c < nc; c++) log("%s (%s bp)\n", pp_assembly_name(a, c), commas(off[c + 1] - off[c]).c_str());
    log("\n");

    lap("assembly loaded");
    // The FASTA that will be returned: headers + polished bytes.  Its buffer is allocated for an upper bound NOW and touched
    // by a helper thread while the alignments are loaded -- the polished bytes then come off the device straight into their
    // place in it (a fresh 250 MB destination took 30 ms of page faults inside the copy, and assembling the FASTA from a
    // second buffer another 40 ms of memcpy).
    size_t fasta_cap = 0;
    for (uint32_t c = 0; c < nc; c++) fasta_cap += strlen(pp_assembly_name(a, c)) + strlen(pp_assembly_description(a, c)) + 16;
    const size_t header_bytes = fasta_cap;
    fasta_cap += (size_t)(off[nc] + off[nc] / 8) + 2 * (size_t)nc + 65536;  // (every planted insertion adds a byte: far below 1/8)
    uint8_t *out = (uint8_t *)malloc(fasta_cap);
    std::thread out_toucher;
    if (out) out_toucher = std::thread([out, fasta_cap] { for (size_t q = 0; q < fasta_cap; q += 4096) out[q] = 0; });
    // (every way out of this function: the helper joined first, then the buffer released unless it went to the caller)
    struct OutGuard { std::thread &t; uint8_t *&p; ~OutGuard() { if (t.joinable()) t.join(); free(p); } } out_guard{out_toucher, out};
    // load_alignments, polish.rs:109-134 -- by the device tokenizer (pp_tokenize.hip), or on the host (multi-threaded
    // parse) with PP_DEVICE_INGEST=0 and with --debug (the TSV needs the read bytes on the host)
    log("Loading alignments\n");
    pp_ingest *g = nullptr;
    pp_dev_ingest *dg = nullptr;
    std::vector<pp_dev_ingest *> dgs((size_t)(multi && dev_ingest ? n_ctx : 0), nullptr);  // sharded ingest: one per context
    // Host ingest without --debug: one ingest object per SAM file, and (one context) the batch of file i goes to the device
    // (pp_polish_begin + pp_polish_add on a helper thread) while file i+1 is parsed -- the reference streams its files
    // one after the other as well (alignment.rs:238-265).  --debug keeps ONE host batch (the TSV indexes its SEQ bytes).
    // Several contexts: the files' batches wait until the plan is known, then every context is sent its part.
    const bool per_file = !dev_ingest && !opt->debug_path && (multi || !(getenv("PP_STREAM_ADDS") && atoi(getenv("PP_STREAM_ADDS")) == 0));
    const bool stream_adds = per_file && !multi;
    std::vector<pp_ingest *> gs;
    std::vector<std::future<int>> pending;  // the upload of the file before
    std::vector<uint64_t> per_contig(nc, 0);  // alignment records per contig (the planner's weights)
    bool begun = false;
    pp_params prm{opt->min_depth, opt->fraction_valid, opt->fraction_invalid};
    auto wait_pending = [&]() {
        int r = PP_OK;
        for (size_t i = 0; i < pending.size(); i++) {
            const int ri = pending[i].get();
            if (ri && !r) r = ri;
        }
        pending.clear();
        return r;
    };
    std::vector<std::vector<Piece>> pieces((size_t)n_ctx);  // multi: what every context was sent, in file order
    auto free_all = [&]() {
        (void)wait_pending();
        for (auto &v : pieces)
            for (Piece &pc : v) pp_shard_part_free(pc.part);
        pieces.clear();
        for (pp_ingest *x : gs) pp_ingest_free(x);
        pp_ingest_free(g);
        pp_dev_ingest_free(dg);
        for (pp_dev_ingest *x : dgs) pp_dev_ingest_free(x);
        pp_assembly_free(a);
    };
    // run f(d) for every context on its own thread; the first failure (lowest d) is returned, its text put on ctxs[0]
    auto on_all = [&](const std::function<int(int)> &f) {
        std::vector<std::future<int>> jobs;
        for (int d = 0; d < n_ctx; d++) jobs.push_back(std::async(std::launch::async, f, d));
        int r = PP_OK;
        for (int d = 0; d < n_ctx; d++) {
            const int rd = jobs[(size_t)d].get();
            if (rd && !r) {
                r = rd;
                if (d) set_err(ctx, rd, pp_last_error(ctxs[d]));
            }
        }
        return r;
    };
    const bool sharded = multi && dev_ingest;
    if (multi) pp_ctx_enable_peers_(ctxs, n_ctx);
    if (sharded) rc = on_all([&](int d) { return pp_dev_ingest_create(ctxs[d], a, opt->max_errors, opt->careful, &dgs[(size_t)d]); });
    else rc = dev_ingest ? pp_dev_ingest_create(ctx, a, opt->max_errors, opt->careful, &dg)
                         : (per_file ? PP_OK : pp_ingest_create(a, opt->max_errors, opt->careful, &g));
    uint64_t largest_sam = 0;
    if (rc == PP_OK && dev_ingest && !sharded) {
        uint64_t largest = 0, total = 0;
        for (int i = 0; i < n_sams; i++) {
            struct stat st;
            if (stat(sams[i], &st) == 0 && S_ISREG(st.st_mode)) {
                largest = std::max<uint64_t>(largest, (uint64_t)st.st_size);
                total += (uint64_t)st.st_size;
            }
        }
        largest_sam = largest;
        if (largest) rc = pp_dev_ingest_reserve_text_(dg, largest);
        if (rc == PP_OK && n_sams > 1) rc = pp_dev_ingest_expect(dg, total);  // the batch's arrays sized once, for all the files
    }
    if (dev_ingest) lap("device ready, tokenizer created");  // (pp_dev_ingest_create waits for the HIP runtime's start-up)
    uint64_t alignment_total = 0, used_total = 0;
    // sharded: the records of file i's slice on context s are records [slice_end[i-1][s], slice_end[i][s]) of its batch
    std::vector<std::vector<uint64_t>> slice_end;
    for (int i = 0; rc == PP_OK && i < n_sams; i++) {
        pp_sam_counts c{0, 0, 0};
        if (i == resume_log_at) log.quiet = opt->quiet != 0;
        if (sharded) {
            // cut the file into one slice per context at read-group boundaries; every context uploads and tokenizes its own
            pph::FileText F;
            if (!F.open_file(sams[i])) {
                snprintf(err, sizeof err, "unable to load alignments from \"%s\"", sams[i]);
                rc = set_err(ctx, PP_ERR_QUIT, err);
                break;
            }
            std::vector<size_t> cut((size_t)n_ctx + 1, F.size);
            cut[0] = 0;
            for (int d = 1; d < n_ctx; d++) cut[(size_t)d] = std::max(cut[(size_t)d - 1], group_cut(F.text, F.size, F.size / (size_t)n_ctx * (size_t)d));
            std::vector<pp_sam_counts> cs((size_t)n_ctx);
            int rs = on_all([&](int d) {
                return pp_dev_ingest_slice_(dgs[(size_t)d], sams[i], F.text + cut[(size_t)d], cut[(size_t)d + 1] - cut[(size_t)d], &cs[(size_t)d]);
            });
            for (auto &x : cs) { c.alignments += x.alignments; c.used += x.used; c.reads += x.reads; }
            if (rs == PP_OK && c.alignments == 0) rs = PP_ERR_PANIC;  // a file without aligned records: the host path has the message
            if (rs == PP_ERR_QUIT || rs == PP_ERR_PANIC || rs == PP_ERR_NOT_ASCII) {
                free_all();  // as for one context: the host ingest works out what the reference reports first
                return polish_files_impl(ctxs, n_ctx, assembly, sams, n_sams, opt, fasta, pass, n_pass, i);
            }
            if ((rc = rs)) break;
            std::vector<uint64_t> ends((size_t)n_ctx);
            for (int d = 0; d < n_ctx; d++) {
                pp_aln_batch bd;
                pp_dev_ingest_batch(dgs[(size_t)d], &bd);
                ends[(size_t)d] = bd.n_aln;
            }
            slice_end.push_back(ends);
        } else if (dev_ingest) {
            // the text of the NEXT file goes up (second text buffer, upload stream) while this one is tokenized
            if (i + 1 < n_sams) pp_dev_ingest_prefetch_(dg, sams[i + 1], largest_sam);
            rc = pass ? pp_dev_ingest_sam_filtered(dg, sams[i], pass[i], n_pass[i], &c) : pp_dev_ingest_sam(dg, sams[i], &c);
            if (rc == PP_ERR_QUIT || rc == PP_ERR_PANIC || rc == PP_ERR_NOT_ASCII) {
                // A defect in the text.  Which defect the reference reports FIRST also depends on what its CIGAR walk
                // makes of the records before it: the host ingest works that out (below), on this rare path.
                free_all();
                return polish_files_impl(ctxs, n_ctx, assembly, sams, n_sams, opt, fasta, pass, n_pass, i);
            }
            if (rc) break;
        } else {
            pp_ingest *gi = g;
            if (per_file) {
                rc = pp_ingest_create(a, opt->max_errors, opt->careful, &gi);
                if (rc) break;
                gs.push_back(gi);
            }
            rc = pass ? pp_ingest_sam_filtered(gi, sams[i], pass[i], n_pass[i], &c, err, sizeof err)
                      : pp_ingest_sam(gi, sams[i], &c, err, sizeof err);
            if (rc) {
                set_err(ctx, rc, err);
                // The reference streams: every read group before the failing point had already gone through
                // add_alignment (alignment.rs:297-303), so a record there that only the CIGAR walk rejects (unexpected
                // op, CIGAR / SEQ length mismatch, past the contig end) is what it reports.  Run the device over exactly
                // those records: the earlier files and this file up to the group that was pending.
                uint64_t cut = 0;
                if (pp_ingest_fail_cut_(gi, &cut) && wait_pending() == PP_OK) {
                    int rd = begun ? PP_OK : pp_polish_begin(ctx, nc, off, pp_assembly_bases(a), PP_MEM_HOST, &prm);
                    begun = true;
                    pp_aln_batch bi;
                    if (rd == PP_OK && !per_file && g) {
                        pp_ingest_batch(g, &bi);
                        if (bi.n_aln) rd = pp_polish_add(ctx, &bi, PP_MEM_HOST);
                    }
                    if (multi)  // (the earlier files' batches have not gone anywhere yet: the first context takes them whole)
                        for (size_t q = 0; rd == PP_OK && q + 1 < gs.size(); q++) {
                            pp_ingest_batch(gs[q], &bi);
                            if (bi.n_aln) rd = pp_polish_add(ctx, &bi, PP_MEM_HOST);
                        }
                    pp_ingest *gp = nullptr;
                    char err2[256];
                    pp_sam_counts c2;
                    if (rd == PP_OK && cut > 0 && pp_ingest_create(a, opt->max_errors, opt->careful, &gp) == PP_OK &&
                        pp_ingest_sam_prefix_(gp, sams[i], cut, pass ? pass[i] : nullptr, pass ? n_pass[i] : 0, &c2, err2,
                                              sizeof err2) == PP_OK) {
                        pp_ingest_batch(gp, &bi);
                        if (bi.n_aln) rd = pp_polish_add(ctx, &bi, PP_MEM_HOST);
                    }
                    if (rd == PP_OK) rd = pp_polish_finish(ctx);
                    pp_ingest_free(gp);
                    if (rd == PP_ERR_QUIT || rd == PP_ERR_PANIC) rc = rd;  // the device's message stands
                    else set_err(ctx, rc, err);
                }
                break;
            }
            if (stream_adds) {
                if ((rc = wait_pending())) break;  // the upload of the file before
                const bool first = !begun;
                begun = true;
                // room for all files at once: this file's batch scaled by the files' sizes on disk
                double scale = 1.0;
                if (first && n_sams > 1) {
                    struct stat st0;
                    double all_bytes = 0, this_bytes = 0;
                    for (int q = 0; q < n_sams; q++)
                        if (stat(sams[q], &st0) == 0 && S_ISREG(st0.st_mode)) { all_bytes += (double)st0.st_size; if (q == i) this_bytes = (double)st0.st_size; }
                    if (this_bytes > 0) scale = std::min(64.0, all_bytes / this_bytes * 1.02);
                }
                pending.push_back(std::async(std::launch::async, [ctx, gi, first, nc, off, a, prm, scale]() {
                    int r = first ? pp_polish_begin(ctx, nc, off, pp_assembly_bases(a), PP_MEM_HOST, &prm) : PP_OK;
                    pp_aln_batch bi;
                    pp_ingest_batch(gi, &bi);
                    if (r == PP_OK && first && scale > 1.0)
                        r = pp_polish_reserve(ctx, (uint64_t)((double)bi.n_aln * scale) + 1024, (uint64_t)((double)bi.seq_bytes * scale) + 4096,
                                              (uint64_t)((double)bi.n_cig_total * scale) + 1024);
                    if (r == PP_OK) r = pp_polish_add(ctx, &bi, PP_MEM_HOST);
                    return r;
                }));
            }
        }
        log("%s: %s alignments from %s reads\n", sams[i], commas(c.alignments).c_str(), commas(c.reads).c_str());
        if (timing) { char what[64]; snprintf(what, sizeof what, "file %d ingested", i + 1); lap(what); }
        alignment_total += c.alignments;
        used_total += c.used;
    }
    if (rc == PP_OK) rc = wait_pending();
    if (rc) {
        free_all();
        return rc;
    }
    log("\nFiltering for high-quality end-to-end alignments%s:\n  %s alignments kept\n  %s alignments discarded\n\n",
        opt->careful ? " from reads with only one alignment" : "", commas(used_total).c_str(),
        commas(alignment_total - used_total).c_str());

    lap("alignments ingested");
    // polish_sequences, polish.rs:137-154 -- on the device
    log("Polishing assembly sequences\n");
    pp_aln_batch batch;
    memset(&batch, 0, sizeof batch);
    if (dev_ingest && !sharded) pp_dev_ingest_batch(dg, &batch); else if (g) pp_ingest_batch(g, &batch);
    // create_debug_file, polish.rs:230-245: the file is created (and the header written) before polishing
    FILE *dbg = nullptr;
    if (opt->debug_path) {
        dbg = fopen(opt->debug_path, "wb");
        if (!dbg) {
            snprintf(err, sizeof err, "unable to create \"%s\"", opt->debug_path);
            free_all();
            return set_err(ctx, PP_ERR_QUIT, err);
        }
        fputs("name\tpos\tbase\tdepth\tinvalid\tvalid\tpileup\tstatus\tnew_base\n", dbg);
    }
    pp_polish_set_debug(ctx, dbg ? 1 : 0);
    uint64_t total = 0;
    bool direct_fetch = false;
    std::vector<uint8_t> polished(1);
    std::vector<uint64_t> out_off(nc + 1);
    std::vector<pp_contig_stats> stats(nc);
    if (!multi) {
        if (!begun) {  // one batch (device tokenizer, --debug), or no SAM files at all
            rc = pp_polish_begin(ctx, nc, off, pp_assembly_bases(a), PP_MEM_HOST, &prm);
            if (rc == PP_OK && (dev_ingest || g)) rc = pp_polish_add(ctx, &batch, dev_ingest ? PP_MEM_DEVICE : PP_MEM_HOST);
        }
        if (rc == PP_OK) rc = pp_polish_finish(ctx);
        lap("uploaded + polished on device");
        if (rc == PP_OK && dbg) rc = write_debug_tsv(ctx, dbg, a, &batch);
        if (dbg) fclose(dbg);
        pp_polish_set_debug(ctx, 0);
        if (rc == PP_OK) rc = pp_polish_result_size(ctx, &total);
        // offsets and statistics now; the bytes go straight into the FASTA buffer below (contig by contig) when that is a
        // handful of copies, else through one copy of everything
        direct_fetch = rc == PP_OK && out && nc <= 256 && header_bytes + total + nc <= fasta_cap;
        if (!direct_fetch) polished.resize(total ? total : 1);
        if (rc == PP_OK) rc = pp_polish_result(ctx, direct_fetch ? nullptr : polished.data(), PP_MEM_HOST, out_off.data(), stats.data());
    } else {
        // ---- the plan, from the alignment counts per contig ----
        // source batches in file order: sharded -> (file, slice) pieces living on the contexts' GPUs; host ingest -> files
        struct Src { pp_aln_batch view; int mem; int owner; uint64_t base; uint32_t wo_base; std::vector<uint64_t> runs; };
        std::vector<Src> srcs;
        uint64_t base = 0;
        if (sharded) {
            std::vector<pp_aln_batch> whole((size_t)n_ctx);
            for (int d = 0; d < n_ctx; d++) pp_dev_ingest_batch(dgs[(size_t)d], &whole[(size_t)d]);
            for (size_t f = 0; f < slice_end.size(); f++)
                for (int sidx = 0; sidx < n_ctx; sidx++) {
                    const uint64_t lo = f ? slice_end[f - 1][(size_t)sidx] : 0, hi = slice_end[f][(size_t)sidx];
                    pp_aln_batch v = whole[(size_t)sidx];  // seq / cigar: the whole arrays (seq_off / cig_off are absolute)
                    v.n_aln = hi - lo;
                    v.contig += lo; v.ref_start += lo; v.k += lo; v.seq_off += lo; v.seq_len += lo; v.cig_off += lo; v.n_cig += lo;
                    if (v.wo) v.wo += lo;  // (a slice's entries of the window-order mirror are its own stretch; they count from lo)
                    // the slice's runs: the whole batch's, cut to [lo, hi) and counted from lo (the tokenizer ends a run with every
                    // file: one run, the slice itself) -- with them a part of the slice takes the direct path like any other
                    std::vector<uint64_t> runs;
                    const pp_aln_batch &wb = whole[(size_t)sidx];
                    for (uint32_t r = 0; v.wo && wb.wo_run_end && r < wb.wo_n_runs; r++) {
                        const uint64_t e = std::min(std::max(wb.wo_run_end[r], lo), hi) - lo;
                        if (e > (runs.empty() ? 0 : runs.back())) runs.push_back(e);
                    }
                    if (runs.empty() || runs.back() != hi - lo) runs.clear();  // (not known: the bucketing path)
                    v.wo_n_runs = 0;
                    v.wo_run_end = nullptr;
                    srcs.push_back(Src{v, PP_MEM_DEVICE, sidx, base, (uint32_t)lo, std::move(runs)});
                    srcs.back().view.wo_n_runs = (uint32_t)srcs.back().runs.size();
                    srcs.back().view.wo_run_end = srcs.back().runs.empty() ? nullptr : srcs.back().runs.data();
                    base += hi - lo;
                }
            // (one context after the other: a histogram kernel each, they add into the same host array)
            for (int d = 0; rc == PP_OK && d < n_ctx; d++) {
                rc = pp_shard_count(ctxs[d], &whole[(size_t)d], PP_MEM_DEVICE, nc, per_contig.data());
                if (rc && d) set_err(ctx, rc, pp_last_error(ctxs[d]));
            }
        } else {
            for (pp_ingest *gi : gs) {
                pp_aln_batch v;
                pp_ingest_batch(gi, &v);
                srcs.push_back(Src{v, PP_MEM_HOST, -1, base, 0u, {}});
                base += v.n_aln;
                pp_shard_count(nullptr, &v, PP_MEM_HOST, nc, per_contig.data());
            }
        }
        pp_shard_plan *plan = nullptr;
        if (rc == PP_OK) rc = pp_shard_plan_create(nc, off, per_contig.data(), (uint32_t)n_ctx, 0, &plan);
        // ---- every source batch is split once per destination (on the GPU that holds it, or on the host) ----
        std::vector<std::vector<pp_shard_part *>> parts(srcs.size(), std::vector<pp_shard_part *>((size_t)n_ctx, nullptr));
        if (rc == PP_OK) {
            if (sharded)
                rc = on_all([&](int sidx) {  // a context splits the pieces it holds, for every destination
                    for (size_t q = 0; q < srcs.size(); q++) {
                        if (srcs[q].owner != sidx) continue;
                        for (int d = 0; d < n_ctx; d++)
                            if (int r = pp_shard_split_view_(ctxs[sidx], plan, (uint32_t)d, &srcs[q].view, PP_MEM_DEVICE, srcs[q].wo_base, &parts[q][(size_t)d])) return r;
                    }
                    return (int)PP_OK;
                });
            else
                rc = on_all([&](int d) {
                    for (size_t q = 0; q < srcs.size(); q++)
                        if (int r = pp_shard_split(nullptr, plan, (uint32_t)d, &srcs[q].view, PP_MEM_HOST, &parts[q][(size_t)d])) {
                            set_err(ctxs[d], r, "splitting the records failed");
                            return r;
                        }
                    return (int)PP_OK;
                });
        }
        for (size_t q = 0; q < srcs.size(); q++)
            for (int d = 0; d < n_ctx; d++)
                if (parts[q][(size_t)d]) {
                    pp_aln_batch v;
                    pp_shard_part_batch(parts[q][(size_t)d], &v, nullptr);
                    pieces[(size_t)d].push_back(Piece{parts[q][(size_t)d], srcs[q].owner, srcs[q].base, v.n_aln});
                }
        // the tokenizers' batches (and their text buffers) are not needed once every part has been cut out of them
        for (pp_dev_ingest *&x : dgs) { pp_dev_ingest_free(x); x = nullptr; }
        lap("records split");
        // every context: its records, the ranges of its units, finish, its own bytes to the host -- side by side
        // ---- how the polished bytes will reach the host: one RCCL gather to the first context's GPU, or every device by itself ----
        // The RCCL route is OPT-IN (PP_GATHER=rccl) in this one-process driver: it has only ever run with world = 1 -- the boxes
        // this was built on have one GPU, and RCCL refuses two ranks on one device -- and a rank that fails inside
        // ncclCommInitRank leaves the other threads waiting there.  The default is the route every test runs: each device
        // copies its share out and the FASTA is assembled on the host.  (One process per GPU -- python -m
        // polypolish_amd.distributed, bench.py --gpus N -- gathers over RCCL, behind a watchdog.)
        bool use_rccl = getenv("PP_GATHER") && !strcmp(getenv("PP_GATHER"), "rccl");
        for (int d = 0; d < n_ctx && use_rccl && !getenv("PP_RCCL_LIB"); d++)  // (a stand-in library, tests: it takes several ranks on one device)
            for (int e = 0; e < d; e++)
                if (pp_ctx_device_(ctxs[d]) == pp_ctx_device_(ctxs[e])) use_rccl = false;  // RCCL refuses two ranks on one device
        if (use_rccl && rc == PP_OK) {
            uint8_t id[PP_COMM_ID_BYTES];
            if (pp_comm_unique_id(id) != PP_OK) use_rccl = false;  // librccl not loadable
            else if (on_all([&](int d) { return pp_comm_init(ctxs[d], d, n_ctx, id); }) != PP_OK) {
                for (int d = 0; d < n_ctx; d++) pp_comm_destroy(ctxs[d]);
                use_rccl = false;
            }
        }
        if (timing) fprintf(stderr, "[timing] polished bytes -> host: %s\n", use_rccl ? "ONE RCCL gather (ncclSend/ncclRecv over xGMI) to the first GPU + one D2H"
                                                                                    : "every device copies its share out, assembled on the host (the default; PP_GATHER=rccl asks for the RCCL gather -- not with two contexts on one device or without librccl)");
        std::vector<uint8_t> gathered;
        std::vector<uint64_t> r_total((size_t)n_ctx, 0);
        std::vector<std::vector<uint8_t>> r_bytes((size_t)n_ctx);
        std::vector<std::vector<uint64_t>> r_off((size_t)n_ctx, std::vector<uint64_t>(nc + 1, 0));
        std::vector<std::vector<pp_contig_stats>> r_stats((size_t)n_ctx, std::vector<pp_contig_stats>(nc));
        std::vector<int> r_rc((size_t)n_ctx, PP_OK);
        if (rc == PP_OK) {
            (void)on_all([&](int d) {
                pp_ctx *cd = ctxs[d];
                int r = pp_polish_begin(cd, nc, off, pp_assembly_bases(a), PP_MEM_HOST, &prm);
                uint64_t tn = 0, ts = 0, tc = 0;
                for (const Piece &pc : pieces[(size_t)d]) {
                    pp_aln_batch v;
                    pp_shard_part_batch(pc.part, &v, nullptr);
                    tn += v.n_aln; ts += v.seq_bytes; tc += v.n_cig_total;
                }
                if (r == PP_OK) r = pp_polish_reserve(cd, tn + 16, ts + 64, tc + 16);
                bool first = true;
                for (const Piece &pc : pieces[(size_t)d]) {
                    if (r) break;
                    pp_aln_batch v;
                    pp_shard_part_batch(pc.part, &v, nullptr);
                    if (v.n_aln == 0) continue;
                    // (a part on this context's own GPU is device memory; the FIRST batch of a job would be used in place,
                    // which is fine -- the parts live until every context has finished)
                    const int pm = pp_shard_part_mem(pc.part);
                    r = pp_polish_add(cd, &v, pm == PP_MEM_HOST ? PP_MEM_HOST : (pc.src == d && !first ? PP_MEM_DEVICE : PP_MEM_PEER));
                    first = false;
                }
                std::vector<uint64_t> lo(nc), hi(nc);
                if (r == PP_OK) r = pp_shard_emit_ranges(plan, (uint32_t)d, lo.data(), hi.data());
                if (r == PP_OK) r = pp_polish_set_emit(cd, lo.data(), hi.data());
                if (r == PP_OK) r = pp_polish_finish(cd);
                uint64_t t = 0;
                if (r == PP_OK) r = pp_polish_result_size(cd, &t);
                r_total[(size_t)d] = t;
                // offsets and statistics now; the bytes follow over RCCL, or (host route) with this very call
                if (r == PP_OK && !use_rccl) r_bytes[(size_t)d].resize(t ? t : 1);
                if (r == PP_OK) r = pp_polish_result(cd, use_rccl ? nullptr : r_bytes[(size_t)d].data(), PP_MEM_HOST, r_off[(size_t)d].data(), r_stats[(size_t)d].data());
                r_rc[(size_t)d] = r;
                return r;
            });
            bool all_ok = true;
            for (int d = 0; d < n_ctx; d++) all_ok = all_ok && r_rc[(size_t)d] == PP_OK;
            if (all_ok && use_rccl) {
                // the one exchange of the job: every context's bytes to the first context's GPU, then one copy to the host
                uint64_t sum = 0;
                for (uint64_t t : r_total) sum += t;
                gathered.resize(sum ? sum : 1);
                std::vector<uint64_t> lens((size_t)n_ctx, 0);
                const auto tg = std::chrono::steady_clock::now();
                const int rg = on_all([&](int d) {
                    return pp_polish_gather_to_host_(ctxs[d], d == 0 ? gathered.data() : nullptr, d == 0 ? sum : 0, d == 0 ? lens.data() : nullptr, nullptr);
                });
                if (timing) fprintf(stderr, "[timing] RCCL gather of %llu bytes from %d contexts + one D2H: %.3f ms\n", (unsigned long long)sum, n_ctx,
                                    1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - tg).count());
                if (rg) rc = rg;
                else
                    for (int d = 0; d < n_ctx; d++)
                        if (lens[(size_t)d] != r_total[(size_t)d]) rc = set_err(ctx, PP_ERR_HIP, "the RCCL gather delivered a rank's bytes short");
            }
            // The job's error is the one about its FIRST bad record in file order (the reference streams,
            // src/alignment.rs:238-303): a context numbers the records it was sent, the parts know where those came from.
            uint64_t best = ~0ull;
            int best_d = -1;
            uint32_t best_kind = 0;
            for (int d = 0; d < n_ctx; d++) {
                if (r_rc[(size_t)d] == PP_OK) continue;
                uint32_t kind = 0;
                const uint64_t jr = job_record_of(ctxs[d], ctxs, pieces[(size_t)d], &kind);
                if (best_d < 0 || jr < best) { best = jr; best_d = d; best_kind = kind; }
            }
            if (best_d >= 0) {
                if (best != ~0ull) rc = pp_polish_error_text(ctx, best_kind, best);
                else { rc = r_rc[(size_t)best_d]; if (best_d) set_err(ctx, rc, pp_last_error(ctxs[best_d])); }
            }
        }
        lap("uploaded + polished on the devices");
        if (rc == PP_OK) {
            std::vector<const uint8_t *> bp((size_t)n_ctx);
            std::vector<const uint64_t *> op((size_t)n_ctx);
            uint64_t at = 0;
            for (int d = 0; d < n_ctx; d++) {
                bp[(size_t)d] = use_rccl ? gathered.data() + at : r_bytes[(size_t)d].data();
                at += r_total[(size_t)d];
                op[(size_t)d] = r_off[(size_t)d].data();
            }
            rc = pp_shard_assemble(plan, bp.data(), op.data(), nullptr, out_off.data());
            total = out_off[nc];
            polished.resize(total ? total : 1);
            if (rc == PP_OK) rc = pp_shard_assemble(plan, bp.data(), op.data(), polished.data(), out_off.data());
            for (uint32_t c = 0; c < nc; c++) {  // a position is counted by the rank that emits it
                stats[c] = pp_contig_stats{out_off[c + 1] - out_off[c], 0, 0, 0.0};
                for (int d = 0; d < n_ctx; d++) {
                    stats[c].changed += r_stats[(size_t)d][c].changed;
                    stats[c].zero_depth += r_stats[(size_t)d][c].zero_depth;
                    stats[c].depth_sum += r_stats[(size_t)d][c].depth_sum;
                }
            }
        }
        if (use_rccl)
            for (int d = 0; d < n_ctx; d++) pp_comm_destroy(ctxs[d]);
        pp_shard_plan_free(plan);
    }
    if (rc) {
        free_all();
        return rc;
    }

    lap(direct_fetch ? "result: offsets fetched" : "result fetched");
    // print_seq_to_stdout (polish.rs:196-203), one contig after the other in FASTA order
    if (out_toucher.joinable()) out_toucher.join();
    if (!out || header_bytes + total + nc > fasta_cap) {  // (a job whose polished bytes outgrow the bound: a buffer of the exact size)
        free(out);
        out = (uint8_t *)malloc(header_bytes + total + nc + 1);
        if (!out) { free_all(); return set_err(ctx, PP_ERR_HIP, "out of host memory for the FASTA"); }
    }
    const uint8_t *d_polished = direct_fetch ? pp_polish_result_device(ctx) : nullptr;
    size_t w = 0;
    for (uint32_t c = 0; c < nc; c++) {
        const char *name = pp_assembly_name(a, c), *desc = pp_assembly_description(a, c);
        out[w++] = '>';
        memcpy(out + w, name, strlen(name)); w += strlen(name);
        if (desc[0]) {
            out[w++] = ' ';
            memcpy(out + w, desc, strlen(desc)); w += strlen(desc);
        }
        memcpy(out + w, " polypolish\n", 12); w += 12;
        if (direct_fetch) {
            if (int rd = pp_ctx_download(ctx, out + w, d_polished + out_off[c], out_off[c + 1] - out_off[c])) { free_all(); return rd; }
        } else memcpy(out + w, polished.data() + out_off[c], out_off[c + 1] - out_off[c]);
        w += out_off[c + 1] - out_off[c];
        out[w++] = '\n';
        // print_polishing_info, polish.rs:206-227
        const double len = (double)(off[c + 1] - off[c]);
        const double changed_pct = 100.0 * (double)stats[c].changed / len;
        log("Polishing %s (%s bp):\n  mean read depth: %.1fx\n  %s bp %s a depth of zero (%.4f%% coverage)\n"
            "  %s %s changed (%.4f%% of total positions)\n  estimated pre-polishing sequence accuracy: %.4f%% (%s)\n\n",
            name, commas(off[c + 1] - off[c]).c_str(), stats[c].depth_sum / len, commas(stats[c].zero_depth).c_str(),
            stats[c].zero_depth == 1 ? "has" : "have", 100.0 * (len - (double)stats[c].zero_depth) / len,
            commas(stats[c].changed).c_str(), stats[c].changed == 1 ? "position" : "positions", changed_pct,
            100.0 - changed_pct, qscore(100.0 - changed_pct).c_str());
    }
    fasta->data = out;
    fasta->len = w;
    out = nullptr;  // the caller's now (pp_bytes_free)
    lap("FASTA assembled");

    // finished_message, polish.rs:76-90
    log("Finished!\nPolished sequence (to stdout):\n");
    for (uint32_t c = 0; c < nc; c++)
        log("  %s_polypolish (%s bp)\n", pp_assembly_name(a, c), commas(stats[c].polished_len).c_str());
    double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    log("\nTime to run: %s\n\n", format_duration(secs).c_str());
    free_all();
    lap("device and host buffers released");
    return PP_OK;
}


extern "C" int pp_log_text(int what, double value, char *out, size_t cap) {
    if (!out || cap == 0) return PP_ERR_ARG;
    std::string s;
    switch (what) {
    case PP_TEXT_QSCORE: s = pph::qscore(value); break;
    case PP_TEXT_DURATION: s = pph::format_duration_us((uint64_t)value); break;
    case PP_TEXT_PERCENTILE_NAME: s = pph::percentile_name(value); break;
    default: return PP_ERR_ARG;
    }
    if (s.size() + 1 > cap) return PP_ERR_ARG;
    memcpy(out, s.c_str(), s.size() + 1);
    return PP_OK;
}
