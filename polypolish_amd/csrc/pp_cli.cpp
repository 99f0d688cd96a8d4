// pp_cli.cpp -- `polypolish`, a drop-in for the reference binary's command line
// (src/main.rs:23-126): same subcommands, flag names, short flags, defaults, stdout/stderr split
// and exit codes (0 ok, 1 "Error: ...", 2 usage, 101 where the reference would panic).
// All work happens in libpolypolish_hip.so on an MI355X; there is no CPU path.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <string>
#include <sys/stat.h>
#include <unistd.h>
#include <vector>

#include "polypolish_hip.h"
#include "pp_host.h"
extern "C" void pp_process_leaving_soon_(int yes);

static const char *HELP =
    "Polypolish (MI355X/gfx950 implementation, parity target v0.6.1)\n"
    "short-read polishing of long-read assemblies\n\n"
    "Usage: polypolish <COMMAND>\n\n"
    "Commands:\n"
    "  filter  filter paired-end alignments based on insert size\n"
    "  polish  polish a long-read assembly using short-read alignments\n"
    "  filter-polish  both in one process, without the intermediate SAM files (not in the reference)\n\n"
    "Options:\n  -h, --help     Print help\n  -V, --version  Print version\n";

static const char *HELP_FILTER =
    "filter paired-end alignments based on insert size\n\n"
    "Usage: polypolish filter [OPTIONS] --in1 <IN1> --in2 <IN2> --out1 <OUT1> --out2 <OUT2>\n\n"
    "Options:\n"
    "      --in1 <IN1>                  Input SAM file - first read in pairs\n"
    "      --in2 <IN2>                  Input SAM file - first second in pairs\n"
    "      --out1 <OUT1>                Output SAM file - first read in pairs\n"
    "      --out2 <OUT2>                Output SAM file - first second in pairs\n"
    "      --orientation <ORIENTATION>  Expected pair orientation [default: auto]\n"
    "      --low <LOW>                  Low percentile threshold [default: 0.1]\n"
    "      --high <HIGH>                High percentile threshold [default: 99.9]\n"
    "  -h, --help                       Print help\n  -V, --version                    Print version\n";

static const char *HELP_POLISH =
    "polish a long-read assembly using short-read alignments\n\n"
    "Usage: polypolish polish [OPTIONS] <ASSEMBLY> [SAM]...\n\n"
    "Arguments:\n  <ASSEMBLY>  Assembly to polish (one file in FASTA format)\n"
    "  [SAM]...    Short read alignments (one or more files in SAM format)\n\n"
    "Options:\n"
    "      --debug <DEBUG>                        Optional file to store per-base information for debugging purposes\n"
    "  -i, --fraction_invalid <FRACTION_INVALID>  A base must make up less than this fraction of the read depth to be considered invalid [default: 0.2]\n"
    "  -v, --fraction_valid <FRACTION_VALID>      A base must make up at least this fraction of the read depth to be considered valid [default: 0.5]\n"
    "  -m, --max_errors <MAX_ERRORS>              Ignore alignments with more than this many mismatches and indels [default: 10]\n"
    "  -d, --min_depth <MIN_DEPTH>                A base must occur at least this many times in the pileup to be considered valid [default: 5]\n"
    "      --careful                              Ignore any reads with multiple alignments\n"
    "  -h, --help                                 Print help\n  -V, --version                              Print version\n";

static const char *HELP_FUSED =
    "filter paired-end alignments by insert size and polish with them, in one process\n\n"
    "Usage: polypolish filter-polish [OPTIONS] --in1 <IN1> --in2 <IN2> <ASSEMBLY>\n\n"
    "Options: those of `filter` (--out1/--out2 optional: write the tagged SAMs as well) and of `polish`.\n"
    "The FASTA on stdout is byte-identical to `filter` followed by `polish` on its outputs.\n";

static int usage_error(const char *msg) {
    fprintf(stderr, "error: %s\n\nFor more information, try '--help'.\n", msg);
    return 2;
}

static bool parse_f64(const char *s, double &v) {
    char *e = nullptr;
    v = strtod(s, &e);
    return e && *e == 0 && e != s;
}
static bool parse_u32(const char *s, uint32_t &v) {
    if (!*s) return false;
    char *e = nullptr;
    if (*s == '-') return false;
    unsigned long long x = strtoull(s, &e, 10);
    if (!e || *e != 0 || x > 0xFFFFFFFFull) return false;
    v = (uint32_t)x;
    return true;
}

// ---- clap's argument grammar (the reference derives its parser with clap 4.4, src/main.rs:23-109) ----------------
//   --name value | --name=value | -x value | -xvalue | -x=value | flags clustered (-hV) | "--" ends the options |
//   a lone "-" is a positional | an option given twice is an error | unknown options are errors.
struct OptSpec {
    const char *long_name;   // "--min_depth"
    char short_name;         // 'd', or 0
    bool takes_value;
    const char *value_name;  // "<MIN_DEPTH>" (error texts)
};

struct Parsed {
    std::vector<const char *> value;  // per option: the value (flags: "" when present), nullptr when absent
    std::vector<const char *> positional;
    int help = 0, version = 0;
    std::string error;                // non-empty: usage error (exit 2)
};

static std::string opt_display(const OptSpec &o) {
    std::string d = o.long_name;
    if (o.takes_value) { d += ' '; d += o.value_name; }
    return d;
}

static bool looks_like_option(const char *a) { return a[0] == '-' && a[1] != 0; }

static Parsed parse_args(int argc, char **argv, int first, const std::vector<OptSpec> &specs) {
    Parsed P;
    P.value.assign(specs.size(), nullptr);
    bool only_positional = false;
    auto set = [&](size_t k, const char *v) {
        if (P.value[k]) { P.error = "the argument '" + opt_display(specs[k]) + "' cannot be used multiple times"; return false; }
        P.value[k] = v;
        return true;
    };
    for (int i = first; i < argc && P.error.empty(); i++) {
        const char *a = argv[i];
        if (only_positional || a[0] != '-' || a[1] == 0) { P.positional.push_back(a); continue; }
        if (a[1] == '-') {
            if (a[2] == 0) { only_positional = true; continue; }
            const char *eq = strchr(a, '=');
            const std::string name = eq ? std::string(a, (size_t)(eq - a)) : std::string(a);
            if (name == "--help") { P.help = 1; return P; }
            if (name == "--version") { P.version = 1; return P; }
            size_t k = 0;
            while (k < specs.size() && name != specs[k].long_name) k++;
            if (k == specs.size()) { P.error = "unexpected argument '" + name + "' found"; break; }
            if (!specs[k].takes_value) {
                if (eq) { P.error = "unexpected value '" + std::string(eq + 1) + "' for '" + name + "' found; no more were expected"; break; }
                set(k, "");
                continue;
            }
            // clap takes the next argument as the value unless it looks like an option (a leading '-', other than a lone
            // "-"): `--debug --careful x.fa` is "a value is required", not a file named "--careful"
            const char *v = eq ? eq + 1 : (i + 1 < argc && !looks_like_option(argv[i + 1]) ? argv[++i] : nullptr);
            if (!v) { P.error = "a value is required for '" + opt_display(specs[k]) + "' but none was supplied"; break; }
            set(k, v);
            continue;
        }
        for (const char *c = a + 1; *c && P.error.empty(); c++) {  // a cluster of short options
            if (*c == 'h') { P.help = 1; return P; }
            if (*c == 'V') { P.version = 1; return P; }
            size_t k = 0;
            while (k < specs.size() && specs[k].short_name != *c) k++;
            if (k == specs.size()) { P.error = std::string("unexpected argument '-") + *c + "' found"; break; }
            if (!specs[k].takes_value) { set(k, ""); continue; }
            const char *v = c[1] ? (c[1] == '=' ? c + 2 : c + 1) : (i + 1 < argc && !looks_like_option(argv[i + 1]) ? argv[++i] : nullptr);
            if (!v) { P.error = "a value is required for '" + opt_display(specs[k]) + "' but none was supplied"; break; }
            set(k, v);
            break;  // the rest of the argument was the value
        }
    }
    return P;
}

// value of option k as f64 / u32 (clap's value_parser for the field type); false + error text on a bad value
static bool take_f64(const Parsed &P, const std::vector<OptSpec> &specs, size_t k, double &dst, std::string &err) {
    if (!P.value[k]) return true;
    if (parse_f64(P.value[k], dst)) return true;
    err = std::string("invalid value '") + P.value[k] + "' for '" + opt_display(specs[k]) + "': invalid float literal";
    return false;
}
static bool take_u32(const Parsed &P, const std::vector<OptSpec> &specs, size_t k, uint32_t &dst, std::string &err) {
    if (!P.value[k]) return true;
    if (parse_u32(P.value[k], dst)) return true;
    err = std::string("invalid value '") + P.value[k] + "' for '" + opt_display(specs[k]) + "': invalid digit found in string";
    return false;
}

static int no_device(int device) {
    fprintf(stderr, "\nError: no usable MI355X (HIP) device %d -- this build has no CPU path\n", device);
    return 1;
}

// A finished run has nothing left to save: flush the streams and leave without tearing down the HIP
// runtime and unmapping gigabytes of parsed input page by page (the kernel reclaims both at exit).
static int finish(int code) {
    fflush(stdout);
    if (getenv("PP_TIMING")) fprintf(stderr, "[timing] %-36s %8s    (process %7.3f s)\n", "output written, leaving", "", pph::seconds_since_process_start());
    fflush(stderr);
    _exit(code);
}

// The devices `polish` runs on.  The count comes from the environment or sysfs first: asking HIP initialises the
// runtime (a few 100 ms that the single-GPU path hides behind the host parse), so it is only asked when there may
// be more than one GPU.
static std::vector<int> polish_devices(int device, bool single_only) {
    std::vector<int> one{device};
    if (single_only || getenv("PP_DEVICE")) return one;
    if (const char *sh = getenv("PP_SHARE_GPU")) {
        const int n = atoi(sh);
        return n > 1 ? std::vector<int>((size_t)n, device) : one;
    }
    int guess = 0;
    const char *vis = getenv("HIP_VISIBLE_DEVICES") ? getenv("HIP_VISIBLE_DEVICES") : getenv("ROCR_VISIBLE_DEVICES");
    if (vis && *vis) {
        guess = 1;
        for (const char *p = vis; *p; p++) guess += *p == ',';
    } else {
        for (int node = 0; node < 64; node++) {  // kfd topology: GPU nodes have SIMDs
            char path[128];
            snprintf(path, sizeof path, "/sys/class/kfd/kfd/topology/nodes/%d/properties", node);
            FILE *f = fopen(path, "r");
            if (!f) break;
            char key[64];
            unsigned long long val;
            while (fscanf(f, "%63s %llu", key, &val) == 2)
                if (!strcmp(key, "simd_count") && val > 0) { guess++; break; }
            fclose(f);
        }
    }
    int want = guess;
    if (const char *g = getenv("PP_GPUS")) want = std::min(want, std::max(1, atoi(g)));
    if (want <= 1) return one;
    const int have = pp_device_count();
    want = std::min(want, have);
    if (want <= 1) return one;
    std::vector<int> v;
    for (int d = 0; d < want; d++) v.push_back(d);
    return v;
}

int main(int argc, char **argv) {
    if (argc < 2) {
        fputs(HELP, stderr);
        return 2;
    }
    std::string cmd = argv[1];
    if (cmd == "-h" || cmd == "--help") { fputs(HELP, stdout); return 0; }
    if (cmd == "-V" || cmd == "--version") { printf("Polypolish v0.6.1 (%s)\n", pp_version()); return 0; }
    int device = getenv("PP_DEVICE") ? atoi(getenv("PP_DEVICE")) : 0;
    pp_process_leaving_soon_(1);  // every command ends in finish(): _exit without tearing the mappings and the runtime down

    if (cmd == "polish") {
        pp_polish_options opt{0.2, 0.5, 10, 5, 0, nullptr, 0};
        const std::vector<OptSpec> specs = {{"--debug", 0, true, "<DEBUG>"},
                                            {"--fraction_invalid", 'i', true, "<FRACTION_INVALID>"},
                                            {"--fraction_valid", 'v', true, "<FRACTION_VALID>"},
                                            {"--max_errors", 'm', true, "<MAX_ERRORS>"},
                                            {"--min_depth", 'd', true, "<MIN_DEPTH>"},
                                            {"--careful", 0, false, ""}};
        const Parsed P = parse_args(argc, argv, 2, specs);
        if (P.help) { fputs(HELP_POLISH, stdout); return 0; }
        if (P.version) { printf("polypolish-polish v0.6.1\n"); return 0; }
        std::string perr = P.error;
        if (perr.empty()) {
            opt.debug_path = P.value[0];
            opt.careful = P.value[5] != nullptr;
            (void)(take_f64(P, specs, 1, opt.fraction_invalid, perr) && take_f64(P, specs, 2, opt.fraction_valid, perr) &&
                   take_u32(P, specs, 3, opt.max_errors, perr) && take_u32(P, specs, 4, opt.min_depth, perr));
        }
        if (!perr.empty()) return usage_error(perr.c_str());
        const char *assembly = P.positional.empty() ? nullptr : P.positional[0];
        std::vector<const char *> sams(P.positional.begin() + (P.positional.empty() ? 0 : 1), P.positional.end());
        if (!assembly) return usage_error("the following required arguments were not provided:\n  <ASSEMBLY>");
        // Several GPUs polish (contigs / windows of a large contig shard across them) when PP_GPUS=n asks for them; never
        // with --debug.  On small jobs one GPU is the faster choice end to end: the polish itself is a millisecond per 5 Mbp,
        // what a second context adds is its start-up.  (Until round 4 the CLI switched by itself from 8 GiB of SAM text on;
        // the path has only ever run with several contexts on ONE device -- PP_SHARE_GPU=n, tests on a one-GPU box -- so it
        // stays opt-in until it has been run on a multi-GPU node.)
        const bool few = !getenv("PP_GPUS") && !getenv("PP_SHARE_GPU");
        std::vector<int> devs = polish_devices(device, opt.debug_path != nullptr || few);
        if (devs.size() > 1) {
            std::vector<pp_ctx *> cs(devs.size(), nullptr);
            for (size_t d = 0; d < devs.size(); d++)
                if (pp_ctx_create_async(devs[d], &cs[d])) return no_device(devs[d]);
            pp_bytes fasta{nullptr, 0};
            int rc = pp_polish_files_multi(cs.data(), (int)cs.size(), assembly, sams.data(), (int)sams.size(), &opt, &fasta);
            for (size_t d = 0; d < devs.size(); d++)
                if (pp_ctx_wait(cs[d]) != PP_OK) return no_device(devs[d]);
            if (rc) {
                fprintf(stderr, "\nError: %s\n", pp_last_error(cs[0]));
                for (pp_ctx *c : cs) pp_ctx_destroy(c);
                return rc == PP_ERR_PANIC ? 101 : 1;
            }
            fwrite(fasta.data, 1, fasta.len, stdout);
            return finish(0);
        }
        // the device initialises on a helper thread while the host loads and parses the inputs
        pp_ctx *ctx = nullptr;
        int rc = pp_ctx_create_async(device, &ctx);
        if (rc) return no_device(device);
        pp_bytes fasta{nullptr, 0};
        rc = pp_polish_files(ctx, assembly, sams.data(), (int)sams.size(), &opt, &fasta);
        if (pp_ctx_wait(ctx) != PP_OK) return no_device(device);
        if (rc) {
            fprintf(stderr, "\nError: %s\n", pp_last_error(ctx));
            pp_ctx_destroy(ctx);
            return rc == PP_ERR_PANIC ? 101 : 1;
        }
        fwrite(fasta.data, 1, fasta.len, stdout);
        return finish(0);
    }

    if (cmd == "filter-polish") {
        pp_polish_options opt{0.2, 0.5, 10, 5, 0, nullptr, 0};
        const char *orientation = "auto";
        double low = 0.1, high = 99.9;
        const std::vector<OptSpec> specs = {{"--in1", 0, true, "<IN1>"}, {"--in2", 0, true, "<IN2>"},
                                            {"--out1", 0, true, "<OUT1>"}, {"--out2", 0, true, "<OUT2>"},
                                            {"--orientation", 0, true, "<ORIENTATION>"}, {"--low", 0, true, "<LOW>"},
                                            {"--high", 0, true, "<HIGH>"}, {"--debug", 0, true, "<DEBUG>"},
                                            {"--fraction_invalid", 'i', true, "<FRACTION_INVALID>"},
                                            {"--fraction_valid", 'v', true, "<FRACTION_VALID>"},
                                            {"--max_errors", 'm', true, "<MAX_ERRORS>"},
                                            {"--min_depth", 'd', true, "<MIN_DEPTH>"},
                                            {"--careful", 0, false, ""}};
        const Parsed P = parse_args(argc, argv, 2, specs);
        if (P.help || P.version) { fputs(HELP_FUSED, stdout); return 0; }
        std::string perr = P.error;
        if (perr.empty()) {
            if (P.value[4]) orientation = P.value[4];
            opt.debug_path = P.value[7];
            opt.careful = P.value[12] != nullptr;
            (void)(take_f64(P, specs, 5, low, perr) && take_f64(P, specs, 6, high, perr) &&
                   take_f64(P, specs, 8, opt.fraction_invalid, perr) && take_f64(P, specs, 9, opt.fraction_valid, perr) &&
                   take_u32(P, specs, 10, opt.max_errors, perr) && take_u32(P, specs, 11, opt.min_depth, perr));
            if (perr.empty() && P.positional.size() > 1) perr = std::string("unexpected argument '") + P.positional[1] + "' found";
        }
        if (!perr.empty()) return usage_error(perr.c_str());
        const char *in1 = P.value[0], *in2 = P.value[1], *out1 = P.value[2], *out2 = P.value[3];
        const char *assembly = P.positional.empty() ? nullptr : P.positional[0];
        if (!assembly || !in1 || !in2)
            return usage_error("the following required arguments were not provided:\n  --in1 <IN1> --in2 <IN2> <ASSEMBLY>");
        pp_ctx *ctx = nullptr;
        int rc = pp_ctx_create_async(device, &ctx);
        if (rc) return no_device(device);
        pp_bytes fasta{nullptr, 0};
        rc = pp_filter_polish_files(ctx, assembly, in1, in2, out1, out2, orientation, low, high, &opt, nullptr, &fasta);
        if (pp_ctx_wait(ctx) != PP_OK) return no_device(device);
        if (rc) {
            fprintf(stderr, "\nError: %s\n", pp_last_error(ctx));
            pp_ctx_destroy(ctx);
            return rc == PP_ERR_PANIC ? 101 : 1;
        }
        fwrite(fasta.data, 1, fasta.len, stdout);
        return finish(0);
    }

    if (cmd == "filter") {
        const char *orientation = "auto";
        double low = 0.1, high = 99.9;
        const std::vector<OptSpec> specs = {{"--in1", 0, true, "<IN1>"}, {"--in2", 0, true, "<IN2>"},
                                            {"--out1", 0, true, "<OUT1>"}, {"--out2", 0, true, "<OUT2>"},
                                            {"--orientation", 0, true, "<ORIENTATION>"}, {"--low", 0, true, "<LOW>"},
                                            {"--high", 0, true, "<HIGH>"}};
        const Parsed P = parse_args(argc, argv, 2, specs);
        if (P.help) { fputs(HELP_FILTER, stdout); return 0; }
        if (P.version) { printf("polypolish-filter v0.6.1\n"); return 0; }
        std::string perr = P.error;
        if (perr.empty()) {
            if (P.value[4]) orientation = P.value[4];
            (void)(take_f64(P, specs, 5, low, perr) && take_f64(P, specs, 6, high, perr));
            if (perr.empty() && !P.positional.empty()) perr = std::string("unexpected argument '") + P.positional[0] + "' found";
        }
        if (!perr.empty()) return usage_error(perr.c_str());
        const char *in1 = P.value[0], *in2 = P.value[1], *out1 = P.value[2], *out2 = P.value[3];
        if (!in1 || !in2 || !out1 || !out2)
            return usage_error("the following required arguments were not provided:\n  --in1 <IN1> --in2 <IN2> --out1 <OUT1> --out2 <OUT2>");
        pp_ctx *ctx = nullptr;
        int rc = pp_ctx_create_async(device, &ctx);
        if (rc) return no_device(device);
        rc = pp_filter_files(ctx, in1, in2, out1, out2, orientation, low, high, 0, nullptr);
        if (pp_ctx_wait(ctx) != PP_OK) return no_device(device);
        if (rc) {
            fprintf(stderr, "\nError: %s\n", pp_last_error(ctx));
            pp_ctx_destroy(ctx);
            return rc == PP_ERR_PANIC ? 101 : 1;
        }
        return finish(0);
    }
    return usage_error((std::string("unrecognized subcommand '") + cmd + "'").c_str());
}
