/*
 * pp_oracle_main.c -- command-line front end of the CPU oracle (TEST
 * INFRASTRUCTURE, see pp_oracle.h).  Accepts the reference's argv
 * (main.rs:44-109) so shell-level comparisons against bin/polypolish read
 *   pp_oracle polish  [--debug F] [-i X] [-v X] [-m N] [-d N] [--careful] ASM [SAM...]
 *   pp_oracle filter  --in1 A --in2 B --out1 C --out2 D [--orientation o] [--low x] [--high y]
 */
#include "pp_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static int usage(void) {
    fprintf(stderr, "usage: pp_oracle polish|filter ... (same flags as polypolish v0.6.1)\n");
    return 2;
}

int main(int argc, char **argv) {
    if (argc < 2) return usage();
    char err[1024] = "";
    if (strcmp(argv[1], "polish") == 0) {
        orc_polish_params p = {0.2, 0.5, 10, 5, 0};
        const char *debug = NULL, *assembly = NULL;
        const char *sams[1024];
        int n_sams = 0;
        for (int i = 2; i < argc; i++) {
            const char *a = argv[i];
            if (!strcmp(a, "--debug") && i + 1 < argc) debug = argv[++i];
            else if ((!strcmp(a, "-i") || !strcmp(a, "--fraction_invalid")) && i + 1 < argc) p.fraction_invalid = atof(argv[++i]);
            else if ((!strcmp(a, "-v") || !strcmp(a, "--fraction_valid")) && i + 1 < argc) p.fraction_valid = atof(argv[++i]);
            else if ((!strcmp(a, "-m") || !strcmp(a, "--max_errors")) && i + 1 < argc) p.max_errors = (uint32_t)strtoul(argv[++i], NULL, 10);
            else if ((!strcmp(a, "-d") || !strcmp(a, "--min_depth")) && i + 1 < argc) p.min_depth = (uint32_t)strtoul(argv[++i], NULL, 10);
            else if (!strcmp(a, "--careful")) p.careful = 1;
            else if (!assembly) assembly = a;
            else if (n_sams < 1024) sams[n_sams++] = a;
        }
        if (!assembly) return usage();
        orc_buf fasta = {0}, dbg = {0};
        int rc = orc_polish_files(assembly, sams, n_sams, &p, &fasta, debug ? &dbg : NULL, NULL,
                                  NULL, err, sizeof err);
        if (rc != ORC_OK) {
            fprintf(stderr, "\nError: %s\n", err);
            return rc;
        }
        fwrite(fasta.data, 1, fasta.len, stdout);
        if (debug) {
            FILE *f = fopen(debug, "wb");
            if (!f) {
                fprintf(stderr, "\nError: unable to create \"%s\"\n", debug);
                return 1;
            }
            fwrite(dbg.data, 1, dbg.len, f);
            fclose(f);
        }
        orc_buf_free(&fasta);
        orc_buf_free(&dbg);
        return 0;
    }
    if (strcmp(argv[1], "filter") == 0) {
        const char *in1 = NULL, *in2 = NULL, *out1 = NULL, *out2 = NULL, *orientation = "auto";
        double low = 0.1, high = 99.9;
        for (int i = 2; i + 1 < argc; i += 2) {
            const char *a = argv[i], *v = argv[i + 1];
            if (!strcmp(a, "--in1")) in1 = v;
            else if (!strcmp(a, "--in2")) in2 = v;
            else if (!strcmp(a, "--out1")) out1 = v;
            else if (!strcmp(a, "--out2")) out2 = v;
            else if (!strcmp(a, "--orientation")) orientation = v;
            else if (!strcmp(a, "--low")) low = atof(v);
            else if (!strcmp(a, "--high")) high = atof(v);
            else return usage();
        }
        if (!in1 || !in2 || !out1 || !out2) return usage();
        orc_filter_report rep;
        int rc = orc_filter_files(in1, in2, out1, out2, orientation, low, high, &rep, err, sizeof err);
        if (rc != ORC_OK) {
            fprintf(stderr, "\nError: %s\n", err);
            return rc;
        }
        fprintf(stderr, "Alignments before filtering: %llu\nAlignments after filtering:  %llu\n",
                (unsigned long long)rep.before_count, (unsigned long long)rep.after_count);
        return 0;
    }
    return usage();
}
