// pp_driver.cpp -- whole-command drivers behind the `polypolish` CLI contract.
//   pp_polish_files  = polish::polish   (src/polish.rs:26-38)
// Output bytes (FASTA on stdout, filtered SAMs) are the contract; the stderr log reproduces the
// reference's numbers (counts, depth, changed positions, Q-score) in plain text -- the
// reference's colours, wrapping and timestamps (src/log.rs) are cosmetic and not reproduced.
#include <sys/stat.h>

#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <algorithm>

#include "polypolish_hip.h"

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

// polish.rs:290-300
std::string qscore(double identity) {
    if (identity >= 100.0) return "Q\xE2\x88\x9E";
    if (identity <= 0.0) return "Q0";
    double errors = 1.0 - identity / 100.0;
    char buf[64];
    snprintf(buf, sizeof buf, "Q%.2f", -10.0 * log10(errors));
    return buf;
}

// misc.rs:195-201
std::string format_duration(double seconds) {
    uint64_t us = (uint64_t)(seconds * 1e6);
    char buf[64];
    snprintf(buf, sizeof buf, "%llu:%02llu:%02llu.%06llu", (unsigned long long)(us / 1000000 / 3600),
             (unsigned long long)(us / 1000000 / 60 % 60), (unsigned long long)(us / 1000000 % 60),
             (unsigned long long)(us % 1000000));
    return buf;
}

int set_err(pp_ctx *ctx, int code, const char *msg);

}  // namespace

// pp_ctx is opaque here; error text is stored through a tiny hook exported by pp_kernels.hip
extern "C" int pp_ctx_set_error_(pp_ctx *ctx, int code, const char *msg);
namespace {
int set_err(pp_ctx *ctx, int code, const char *msg) { return pp_ctx_set_error_(ctx, code, msg); }
}  // namespace

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

extern "C" int pp_polish_files(pp_ctx *ctx, const char *assembly, const char *const *sams, int n_sams,
                               const pp_polish_options *opt, pp_bytes *fasta) {
    if (!ctx || !assembly || !opt || !fasta || (n_sams > 0 && !sams)) return PP_ERR_ARG;
    fasta->data = nullptr;
    fasta->len = 0;
    Log log{opt->quiet != 0};
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

    // starting_message, polish.rs:41-73
    log("\nStarting Polypolish polish\n%s\n\nInput assembly:\n  %s\n\nInput short-read alignments:\n", pp_version(), assembly);
    for (int i = 0; i < n_sams; i++) log("  %s\n", sams[i]);
    log("\nSettings:\n  --fraction_invalid %g\n  --fraction_valid %g\n  --max_errors %u\n  --min_depth %u\n",
        opt->fraction_invalid, opt->fraction_valid, opt->max_errors, opt->min_depth);
    if (opt->careful) log("  --careful\n");
    if (opt->debug_path) log("  --debug %s\n\n", opt->debug_path);
    else log("  not logging debugging information\n\n");

    const bool timing = getenv("PP_TIMING") != nullptr;
    auto lap = [&](const char *what) {
        if (timing) fprintf(stderr, "[timing] %-28s %8.3f s\n", what,
                            std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    };
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
    // load_alignments, polish.rs:109-134
    log("Loading alignments\n");
    pp_ingest *g = nullptr;
    rc = pp_ingest_create(a, opt->max_errors, opt->careful, &g);
    uint64_t alignment_total = 0, used_total = 0;
    for (int i = 0; rc == PP_OK && i < n_sams; i++) {
        pp_sam_counts c;
        rc = pp_ingest_sam(g, sams[i], &c, err, sizeof err);
        if (rc) { set_err(ctx, rc, err); break; }
        log("%s: %s alignments from %s reads\n", sams[i], commas(c.alignments).c_str(), commas(c.reads).c_str());
        alignment_total += c.alignments;
        used_total += c.used;
    }
    if (rc) {
        pp_ingest_free(g);
        pp_assembly_free(a);
        return rc;
    }
    log("\nFiltering for high-quality end-to-end alignments%s:\n  %s alignments kept\n  %s alignments discarded\n\n",
        opt->careful ? " from reads with only one alignment" : "", commas(used_total).c_str(),
        commas(alignment_total - used_total).c_str());

    lap("alignments ingested");
    // polish_sequences, polish.rs:137-154 -- on the device
    log("Polishing assembly sequences\n");
    pp_params prm{opt->min_depth, opt->fraction_valid, opt->fraction_invalid};
    pp_aln_batch batch;
    pp_ingest_batch(g, &batch);
    // create_debug_file, polish.rs:230-245: the file is created (and the header written) before polishing
    FILE *dbg = nullptr;
    if (opt->debug_path) {
        dbg = fopen(opt->debug_path, "wb");
        if (!dbg) {
            snprintf(err, sizeof err, "unable to create \"%s\"", opt->debug_path);
            pp_ingest_free(g);
            pp_assembly_free(a);
            return set_err(ctx, PP_ERR_QUIT, err);
        }
        fputs("name\tpos\tbase\tdepth\tinvalid\tvalid\tpileup\tstatus\tnew_base\n", dbg);
    }
    pp_polish_set_debug(ctx, dbg ? 1 : 0);
    rc = pp_polish_begin(ctx, nc, off, pp_assembly_bases(a), PP_MEM_HOST, &prm);
    if (rc == PP_OK) rc = pp_polish_add(ctx, &batch, PP_MEM_HOST);
    if (rc == PP_OK) rc = pp_polish_finish(ctx);
    lap("uploaded + polished on device");
    if (rc == PP_OK && dbg) rc = write_debug_tsv(ctx, dbg, a, &batch);
    if (dbg) fclose(dbg);
    pp_polish_set_debug(ctx, 0);
    uint64_t total = 0;
    if (rc == PP_OK) rc = pp_polish_result_size(ctx, &total);
    std::vector<uint8_t> polished(total ? total : 1);
    std::vector<uint64_t> out_off(nc + 1);
    std::vector<pp_contig_stats> stats(nc);
    if (rc == PP_OK) rc = pp_polish_result(ctx, polished.data(), PP_MEM_HOST, out_off.data(), stats.data());
    if (rc) {
        pp_ingest_free(g);
        pp_assembly_free(a);
        return rc;
    }

    lap("result fetched");
    // print_seq_to_stdout (polish.rs:196-203), one contig after the other in FASTA order
    size_t need = total;
    for (uint32_t c = 0; c < nc; c++)
        need += strlen(pp_assembly_name(a, c)) + strlen(pp_assembly_description(a, c)) + 16;
    uint8_t *out = (uint8_t *)malloc(need ? need : 1);
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
        memcpy(out + w, polished.data() + out_off[c], out_off[c + 1] - out_off[c]); w += out_off[c + 1] - out_off[c];
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

    // finished_message, polish.rs:76-90
    log("Finished!\nPolished sequence (to stdout):\n");
    for (uint32_t c = 0; c < nc; c++)
        log("  %s_polypolish (%s bp)\n", pp_assembly_name(a, c), commas(stats[c].polished_len).c_str());
    double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    log("\nTime to run: %s\n\n", format_duration(secs).c_str());
    pp_ingest_free(g);
    pp_assembly_free(a);
    return PP_OK;
}

// =================================================================================================
//   pp_filter_files = filter::filter   (src/filter.rs:26-37)
// =================================================================================================
#include <algorithm>
#include <unordered_map>

namespace {

struct FilterFile {
    std::vector<char> text;
    // every line: offset + length (without the newline / CR), and the alignment index or -1
    std::vector<uint64_t> line_off;
    std::vector<uint32_t> line_len;
    std::vector<int64_t> line_aln;
    // per aligned record, file order
    std::vector<uint32_t> ref_id, ref_start, flags, n_cig, cigar, read;
    std::vector<uint64_t> cig_off;
    std::vector<uint32_t> grp_off, grp_idx;
};

struct FilterErr {
    int code;
    std::string msg;
};

bool slurp(const char *path, std::vector<char> &out) {
    FILE *f = fopen(path, "rb");
    if (!f) return false;
    char tmp[1 << 16];
    size_t r;
    while ((r = fread(tmp, 1, sizeof tmp, f)) > 0) out.insert(out.end(), tmp, tmp + r);
    bool bad = ferror(f);
    fclose(f);
    return !bad;
}

bool parse_u(const char *s, size_t n, uint64_t max, uint64_t &out) {
    size_t i = 0;
    if (n == 0) return false;
    if (s[0] == '+') { i = 1; if (n == 1) return false; }
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

int cigar_op(char c) {
    switch (c) {
    case 'M': return PP_OP_M; case 'I': return PP_OP_I; case 'D': return PP_OP_D; case 'N': return PP_OP_N;
    case 'S': return PP_OP_S; case 'H': return PP_OP_H; case 'P': return PP_OP_P; case '=': return PP_OP_EQ;
    case 'X': return PP_OP_X; default: return -1;
    }
}

// load_alignments_one_file, filter.rs:110-145 (Alignment::new_quick, alignment.rs:102-128)
void load_filter_file(const char *path, FilterFile &F, std::unordered_map<std::string, uint32_t> &reads,
                      std::unordered_map<std::string, uint32_t> &refs, uint64_t &total_alignments,
                      uint64_t &n_read_names) {
    if (!slurp(path, F.text)) {
        char m[1024];
        snprintf(m, sizeof m, "unable to load alignments from \"%s\"", path);
        throw FilterErr{PP_ERR_QUIT, m};
    }
    const char *p = F.text.data(), *end = p + F.text.size();
    uint64_t line_no = 0;
    std::unordered_map<uint32_t, char> seen;  // read_names of this file (for the stderr count)
    while (p < end) {
        const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
        size_t l = nl ? (size_t)(nl - p) : (size_t)(end - p);
        const char *line = p;
        p += l + (nl ? 1 : 0);
        if (l > 0 && line[l - 1] == '\r') l--;
        line_no++;
        F.line_off.push_back((uint64_t)(line - F.text.data()));
        F.line_len.push_back((uint32_t)l);
        F.line_aln.push_back(-1);
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
        if (nc < 11) {
            char m[1024];
            snprintf(m, sizeof m, "too few columns in \"%s\" (line %llu)", path, (unsigned long long)line_no);
            throw FilterErr{PP_ERR_QUIT, m};
        }
        uint64_t flags, pos;
        if (!parse_u(col[1], len[1], 0xFFFFFFFFull, flags) || !parse_u(col[3], len[3], UINT64_MAX, pos)) {
            char m[1024];
            snprintf(m, sizeof m, "could not parse FLAG or POS in \"%s\" (line %llu)", path, (unsigned long long)line_no);
            throw FilterErr{PP_ERR_PANIC, m};
        }
        if (flags & 4) continue;
        if (pos > 0) pos -= 1;
        if (pos > 0xFFFFFFFFull) {
            char m[1024];
            snprintf(m, sizeof m, "POS beyond 2^32 in \"%s\" (line %llu)", path, (unsigned long long)line_no);
            throw FilterErr{PP_ERR_LIMIT, m};
        }
        F.line_aln.back() = (int64_t)F.flags.size();
        uint32_t rid = reads.emplace(std::string(col[0], len[0]), (uint32_t)reads.size()).first->second;
        uint32_t fid = refs.emplace(std::string(col[2], len[2]), (uint32_t)refs.size()).first->second;
        seen.emplace(rid, 0);
        F.read.push_back(rid);
        F.ref_id.push_back(fid);
        F.ref_start.push_back((uint32_t)pos);
        F.flags.push_back((uint32_t)flags);
        F.cig_off.push_back(F.cigar.size());
        // Regex::find_iter over \d+[MIDNSHP=X] (alignment.rs:140): non-matching text is skipped
        const char *c = col[5];
        size_t cl = len[5], i = 0;
        uint32_t runs = 0;
        while (i < cl) {
            if (c[i] >= '0' && c[i] <= '9') {
                size_t j = i;
                while (j < cl && c[j] >= '0' && c[j] <= '9') j++;
                int op = j < cl ? cigar_op(c[j]) : -1;
                if (op >= 0) {
                    uint64_t num;
                    if (!parse_u(c + i, j - i, UINT64_MAX, num)) throw FilterErr{PP_ERR_PANIC, "CIGAR run length overflow"};
                    while (num > 0) {
                        uint32_t piece = num > 0x0FFFFFFFull ? 0x0FFFFFFFu : (uint32_t)num;
                        F.cigar.push_back((piece << 4) | (uint32_t)op);
                        runs++;
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
        F.n_cig.push_back(runs);
        total_alignments++;
    }
    n_read_names = seen.size();
}

void build_groups(FilterFile &F, uint32_t n_reads) {
    F.grp_off.assign((size_t)n_reads + 1, 0);
    for (uint32_t r : F.read) F.grp_off[r + 1]++;
    for (uint32_t r = 0; r < n_reads; r++) F.grp_off[r + 1] += F.grp_off[r];
    F.grp_idx.resize(F.read.size());
    std::vector<uint32_t> cur(F.grp_off.begin(), F.grp_off.end() - 1);
    for (uint32_t i = 0; i < F.read.size(); i++) F.grp_idx[cur[F.read[i]]++] = i;  // file order inside a group
}

// get_percentile, filter.rs:249-259
uint32_t percentile(const std::vector<uint32_t> &sorted, double p) {
    if (sorted.empty()) return 0;
    double fraction = p / 100.0;
    double r = ceil(fraction * (double)sorted.size());
    size_t rank = r <= 0.0 ? 0 : (r >= 1.8e19 ? SIZE_MAX : (size_t)r);
    if (rank < 1) rank = 1;
    return rank - 1 < sorted.size() ? sorted[rank - 1] : 0;
}

// get_percentile_name, filter.rs:262-270
std::string percentile_name(double p) {
    char b[64];
    snprintf(b, sizeof b, "%g", p);
    std::string s = b;
    const char *suffix = "th";
    if (s.back() == '1' && p != 11.0) suffix = "st";
    else if (s.back() == '2' && p != 12.0) suffix = "nd";
    else if (s.back() == '3' && p != 13.0) suffix = "rd";
    return s + suffix + " percentile";
}

}  // namespace

extern "C" int pp_filter_files(pp_ctx *ctx, const char *in1, const char *in2, const char *out1,
                               const char *out2, const char *orientation, double low, double high, int quiet,
                               pp_filter_report *report) {
    if (!ctx || !in1 || !in2 || !out1 || !out2 || !orientation) return PP_ERR_ARG;
    Log log{quiet != 0};
    auto t0 = std::chrono::steady_clock::now();
    // check_inputs, filter.rs:40-53
    const char *f4[4] = {in1, in2, out1, out2};
    for (int i = 0; i < 4; i++)
        for (int j = i + 1; j < 4; j++)
            if (strcmp(f4[i], f4[j]) == 0)
                return set_err(ctx, PP_ERR_QUIT, "--in1, --in2, --out1 and --out2 must all have unique values");
    if (low <= 0.0 || low >= 50.0) return set_err(ctx, PP_ERR_QUIT, "--low must be greater than 0 and less than 50");
    if (high <= 50.0 || high >= 100.0) return set_err(ctx, PP_ERR_QUIT, "--high must be greater than 50 and less than 100");
    log("\nStarting Polypolish filter\n%s\n\nInput alignments:\n  %s\n  %s\n\nOutput alignments:\n  %s\n  %s\n\n"
        "Settings:\n  --orientation %s\n  --low %g\n  --high %g\n\n", pp_version(), in1, in2, out1, out2, orientation, low, high);

    FilterFile F[2];
    std::unordered_map<std::string, uint32_t> reads, refs;
    uint64_t before = 0;
    log("Loading alignments\n");
    try {
        const char *ins[2] = {in1, in2};
        for (int f = 0; f < 2; f++) {
            uint64_t prev = before, names = 0;
            load_filter_file(ins[f], F[f], reads, refs, before, names);
            log("%s: %s alignments from %s reads\n", ins[f], commas(before - prev).c_str(), commas(names).c_str());
            if (before == 0) {
                char m[1024];
                snprintf(m, sizeof m, "no alignments found in \"%s\"", ins[f]);
                throw FilterErr{PP_ERR_QUIT, m};
            }
        }
    } catch (const FilterErr &e) {
        return set_err(ctx, e.code, e.msg.c_str());
    }
    log("\n");
    const uint32_t n_reads = (uint32_t)reads.size();
    for (int f = 0; f < 2; f++) build_groups(F[f], n_reads);

    pp_filter_input in;
    in.n_reads = n_reads;
    for (int f = 0; f < 2; f++) {
        pp_filter_file &d = in.file[f];
        d.n_aln = F[f].flags.size();
        d.ref_id = F[f].ref_id.data(); d.ref_start = F[f].ref_start.data(); d.flags = F[f].flags.data();
        d.cig_off = F[f].cig_off.data(); d.n_cig = F[f].n_cig.data(); d.cigar = F[f].cigar.data();
        d.n_cig_total = F[f].cigar.size(); d.read = F[f].read.data();
        d.grp_off = F[f].grp_off.data(); d.grp_idx = F[f].grp_idx.data();
    }
    int rc = pp_filter_begin(ctx, &in, PP_MEM_HOST);
    if (rc) return rc;

    // get_insert_size_thresholds, filter.rs:148-186 (samples from the device, reduction on the host)
    log("Finding insert size thresholds\n");
    std::vector<uint8_t> orient(n_reads ? n_reads : 1);
    std::vector<uint32_t> insert(n_reads ? n_reads : 1);
    rc = pp_filter_samples(ctx, orient.data(), insert.data());
    if (rc) return rc;
    uint64_t counts[4] = {0, 0, 0, 0};
    for (uint32_t r = 0; r < n_reads; r++)
        if (orient[r] < 4) counts[orient[r]]++;
    if (counts[0] + counts[1] + counts[2] + counts[3] == 0)
        return set_err(ctx, PP_ERR_QUIT, "no one-alignment-per-read pairs available to determine orientation and "
                                         "insert size thresholds");
    static const char *ONAMES[4] = {"fr", "rf", "ff", "rr"};
    for (int o = 0; o < 4; o++) log("%s: %s pairs\n", ONAMES[o], commas(counts[o]).c_str());
    int correct = -1;
    if (strcmp(orientation, "auto") == 0) {  // auto_determine_orientation, filter.rs:238-246
        uint64_t mx = *std::max_element(counts, counts + 4);
        int n_max = 0;
        for (int o = 0; o < 4; o++)
            if (counts[o] == mx) { n_max++; correct = o; }
        if (n_max != 1) return set_err(ctx, PP_ERR_QUIT, "could not automatically determine read pair orientation");
        log("\nAutomatically determined correct orientation: %s\n\n", ONAMES[correct]);
    } else {
        for (int o = 0; o < 4; o++)
            if (strcmp(orientation, ONAMES[o]) == 0) correct = o;
        log("\nUser-specified correct orientation: %s\n\n", orientation);
    }
    std::vector<uint32_t> sizes;
    if (correct >= 0)
        for (uint32_t r = 0; r < n_reads; r++)
            if (orient[r] == correct) sizes.push_back(insert[r]);
    if (sizes.empty()) return set_err(ctx, PP_ERR_QUIT, "no read pairs available to determine insert size thresholds");
    std::sort(sizes.begin(), sizes.end());
    const uint32_t lo = percentile(sizes, low), hi = percentile(sizes, high);
    log("Low threshold:  %u (%s)\nHigh threshold: %u (%s)\n\n", lo, percentile_name(low).c_str(), hi,
        percentile_name(high).c_str());

    // filter_sams, filter.rs:273-349
    log("Filtering SAM files\n");
    std::vector<uint8_t> pass[2];
    for (int f = 0; f < 2; f++) pass[f].resize(F[f].flags.size() ? F[f].flags.size() : 1);
    rc = pp_filter_pairs(ctx, lo, hi, (uint8_t)correct, pass[0].data(), pass[1].data());
    if (rc) return rc;
    uint64_t after = 0;
    const char *ins[2] = {in1, in2}, *outs[2] = {out1, out2};
    for (int f = 0; f < 2; f++) {
        FILE *o = fopen(outs[f], "wb");
        if (!o) {
            char m[1024];
            snprintf(m, sizeof m, "unable to write alignments to \"%s\"", outs[f]);
            return set_err(ctx, PP_ERR_QUIT, m);
        }
        std::vector<char> buf;
        buf.reserve(F[f].text.size() + F[f].flags.size() * 10 + 16);
        uint64_t n_pass = 0, n_fail = 0;
        for (size_t i = 0; i < F[f].line_off.size(); i++) {
            const char *line = F[f].text.data() + F[f].line_off[i];
            buf.insert(buf.end(), line, line + F[f].line_len[i]);
            int64_t a = F[f].line_aln[i];
            if (a >= 0) {
                if (pass[f][(size_t)a]) n_pass++;
                else {
                    static const char tag[] = "\tZP:Z:fail";
                    buf.insert(buf.end(), tag, tag + 10);
                    n_fail++;
                }
            }
            buf.push_back('\n');
        }
        bool ok = fwrite(buf.data(), 1, buf.size(), o) == buf.size();
        ok = (fclose(o) == 0) && ok;
        if (!ok) {
            char m[1024];
            snprintf(m, sizeof m, "unable to write alignments to \"%s\"", outs[f]);
            return set_err(ctx, PP_ERR_QUIT, m);
        }
        log("Filtering %s:\n  %s pass\n  %s fail\n\n", ins[f], commas(n_pass).c_str(), commas(n_fail).c_str());
        after += n_pass;
    }
    if (report) {
        report->before_count = before;
        report->after_count = after;
        report->low_threshold = lo;
        report->high_threshold = hi;
        report->orientation = correct;
        for (int o = 0; o < 4; o++) report->orientation_counts[o] = counts[o];
    }
    double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    log("Finished!\nAlignments before filtering: %s\nAlignments after filtering:  %s\n\nTime to run: %s\n\n",
        commas(before).c_str(), commas(after).c_str(), format_duration(secs).c_str());
    return PP_OK;
}
