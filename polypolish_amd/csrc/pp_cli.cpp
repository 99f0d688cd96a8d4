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

// returns the value of `--name value` / `--name=value` / `-x value`; advances i
static const char *opt_value(int argc, char **argv, int &i, const char *arg, const char *long_name, const char *short_name) {
    size_t ln = strlen(long_name);
    if (strncmp(arg, long_name, ln) == 0 && arg[ln] == '=') return arg + ln + 1;
    if (strcmp(arg, long_name) == 0 || (short_name && strcmp(arg, short_name) == 0)) {
        if (i + 1 >= argc) return nullptr;
        return argv[++i];
    }
    return nullptr;
}
static bool is_opt(const char *arg, const char *long_name, const char *short_name) {
    size_t ln = strlen(long_name);
    return strcmp(arg, long_name) == 0 || (strncmp(arg, long_name, ln) == 0 && arg[ln] == '=') ||
           (short_name && strcmp(arg, short_name) == 0);
}

static int no_device(int device) {
    fprintf(stderr, "\nError: no usable MI355X (HIP) device %d -- this build has no CPU path\n", device);
    return 1;
}

// A finished run has nothing left to save: flush the streams and leave without tearing down the HIP
// runtime and unmapping gigabytes of parsed input page by page (the kernel reclaims both at exit).
static int finish(int code) {
    fflush(stdout);
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

    if (cmd == "polish") {
        pp_polish_options opt{0.2, 0.5, 10, 5, 0, nullptr, 0};
        const char *assembly = nullptr;
        std::vector<const char *> sams;
        for (int i = 2; i < argc; i++) {
            const char *a = argv[i];
            if (!strcmp(a, "-h") || !strcmp(a, "--help")) { fputs(HELP_POLISH, stdout); return 0; }
            if (!strcmp(a, "-V") || !strcmp(a, "--version")) { printf("polypolish-polish v0.6.1\n"); return 0; }
            if (!strcmp(a, "--careful")) { opt.careful = 1; continue; }
            if (is_opt(a, "--debug", nullptr)) {
                const char *v = opt_value(argc, argv, i, a, "--debug", nullptr);
                if (!v) return usage_error("a value is required for '--debug <DEBUG>' but none was supplied");
                opt.debug_path = v;
                continue;
            }
            struct { const char *l, *s; int kind; void *dst; } table[] = {
                {"--fraction_invalid", "-i", 0, &opt.fraction_invalid}, {"--fraction_valid", "-v", 0, &opt.fraction_valid},
                {"--max_errors", "-m", 1, &opt.max_errors}, {"--min_depth", "-d", 1, &opt.min_depth}};
            bool matched = false;
            for (auto &t : table) {
                if (!is_opt(a, t.l, t.s)) continue;
                const char *v = opt_value(argc, argv, i, a, t.l, t.s);
                if (!v) return usage_error((std::string("a value is required for '") + t.l + "' but none was supplied").c_str());
                bool ok = t.kind == 0 ? parse_f64(v, *(double *)t.dst) : parse_u32(v, *(uint32_t *)t.dst);
                if (!ok) return usage_error((std::string("invalid value '") + v + "' for '" + t.l + "'").c_str());
                matched = true;
                break;
            }
            if (matched) continue;
            if (a[0] == '-' && a[1] != 0) return usage_error((std::string("unexpected argument '") + a + "' found").c_str());
            if (!assembly) assembly = a; else sams.push_back(a);
        }
        if (!assembly) return usage_error("the following required arguments were not provided:\n  <ASSEMBLY>");
        // Several GPUs polish (contigs / windows of a large contig shard across them) when PP_GPUS=n asks for them, or
        // by themselves from 8 GiB of SAM text on; never with --debug.  Below that one GPU is the faster choice end to
        // end: the polish itself is a millisecond per 5 Mbp, what a second context adds is its start-up and another
        // copy of the records over PCIe, and only the single-GPU path tokenizes on the device.  PP_SHARE_GPU=n (tests
        // on a one-GPU box) runs n contexts on the one device.
        unsigned long long sam_bytes = 0;
        for (const char *sp : sams) {
            struct stat st;
            if (stat(sp, &st) == 0 && S_ISREG(st.st_mode)) sam_bytes += (unsigned long long)st.st_size;
        }
        const bool few = sam_bytes < (8ull << 30) && !getenv("PP_GPUS") && !getenv("PP_SHARE_GPU");
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
        const char *assembly = nullptr, *in1 = nullptr, *in2 = nullptr, *out1 = nullptr, *out2 = nullptr, *orientation = "auto";
        double low = 0.1, high = 99.9;
        for (int i = 2; i < argc; i++) {
            const char *a = argv[i];
            if (!strcmp(a, "-h") || !strcmp(a, "--help")) { fputs(HELP_FUSED, stdout); return 0; }
            if (!strcmp(a, "--careful")) { opt.careful = 1; continue; }
            struct { const char *l; const char **dst; } paths[] = {{"--in1", &in1}, {"--in2", &in2}, {"--out1", &out1},
                                                                  {"--out2", &out2}, {"--orientation", &orientation},
                                                                  {"--debug", &opt.debug_path}};
            bool matched = false;
            for (auto &t : paths) {
                if (!is_opt(a, t.l, nullptr)) continue;
                const char *v = opt_value(argc, argv, i, a, t.l, nullptr);
                if (!v) return usage_error((std::string("a value is required for '") + t.l + "' but none was supplied").c_str());
                *t.dst = v;
                matched = true;
                break;
            }
            if (matched) continue;
            struct { const char *l, *s; int kind; void *dst; } table[] = {
                {"--fraction_invalid", "-i", 0, &opt.fraction_invalid}, {"--fraction_valid", "-v", 0, &opt.fraction_valid},
                {"--max_errors", "-m", 1, &opt.max_errors}, {"--min_depth", "-d", 1, &opt.min_depth},
                {"--low", nullptr, 0, &low}, {"--high", nullptr, 0, &high}};
            for (auto &t : table) {
                if (!is_opt(a, t.l, t.s)) continue;
                const char *v = opt_value(argc, argv, i, a, t.l, t.s);
                if (!v) return usage_error((std::string("a value is required for '") + t.l + "' but none was supplied").c_str());
                bool ok = t.kind == 0 ? parse_f64(v, *(double *)t.dst) : parse_u32(v, *(uint32_t *)t.dst);
                if (!ok) return usage_error((std::string("invalid value '") + v + "' for '" + t.l + "'").c_str());
                matched = true;
                break;
            }
            if (matched) continue;
            if (a[0] == '-' && a[1] != 0) return usage_error((std::string("unexpected argument '") + a + "' found").c_str());
            if (assembly) return usage_error((std::string("unexpected argument '") + a + "' found").c_str());
            assembly = a;
        }
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
        const char *in1 = nullptr, *in2 = nullptr, *out1 = nullptr, *out2 = nullptr, *orientation = "auto";
        double low = 0.1, high = 99.9;
        for (int i = 2; i < argc; i++) {
            const char *a = argv[i];
            if (!strcmp(a, "-h") || !strcmp(a, "--help")) { fputs(HELP_FILTER, stdout); return 0; }
            if (!strcmp(a, "-V") || !strcmp(a, "--version")) { printf("polypolish-filter v0.6.1\n"); return 0; }
            struct { const char *l; const char **dst; } paths[] = {{"--in1", &in1}, {"--in2", &in2}, {"--out1", &out1},
                                                                  {"--out2", &out2}, {"--orientation", &orientation}};
            bool matched = false;
            for (auto &t : paths) {
                if (!is_opt(a, t.l, nullptr)) continue;
                const char *v = opt_value(argc, argv, i, a, t.l, nullptr);
                if (!v) return usage_error((std::string("a value is required for '") + t.l + "' but none was supplied").c_str());
                *t.dst = v;
                matched = true;
                break;
            }
            if (matched) continue;
            if (is_opt(a, "--low", nullptr) || is_opt(a, "--high", nullptr)) {
                bool is_low = is_opt(a, "--low", nullptr);
                const char *v = opt_value(argc, argv, i, a, is_low ? "--low" : "--high", nullptr);
                if (!v || !parse_f64(v, is_low ? low : high)) return usage_error("invalid value for '--low/--high'");
                continue;
            }
            return usage_error((std::string("unexpected argument '") + a + "' found").c_str());
        }
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
