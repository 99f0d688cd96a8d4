/* A host in plain C99 over the C ABI, the way the reference's Rust would bind it:
 * the two host halves (load_assembly, add_to_pileup's streaming loop) fill the SoA, seam B polishes it on the
 * MI355X, the caller prints the FASTA.  No Python, no torch, no C++ on this side of the boundary.
 *
 *   gcc -std=c99 -Iinclude examples/polish_min.c -Lpolypolish_amd/_build -lpolypolish_hip \
 *       -Wl,-rpath,$PWD/polypolish_amd/_build -o polish_min
 *   ./polish_min assembly.fasta alignments_1.sam [alignments_2.sam ...] > polished.fasta
 */
#include <inttypes.h>
#include <stdio.h>
#include <stdlib.h>

#include "polypolish_hip.h"

static int fail(const char *what, const char *msg) {
    fprintf(stderr, "Error: %s%s%s\n", what, msg && *msg ? ": " : "", msg ? msg : "");
    return 1;
}

int main(int argc, char **argv) {
    if (argc < 3) return fail("usage: polish_min assembly.fasta a.sam [b.sam ...]", "");
    char err[1024] = "";
    pp_ctx *ctx = NULL;
    if (pp_ctx_create(0, &ctx)) return fail("no MI355X context (there is no CPU fallback)", "");

    /* load_assembly (src/polish.rs:89-101) */
    pp_assembly *a = NULL;
    if (pp_assembly_load(argv[1], &a, err, sizeof err)) return fail("assembly", err);
    const uint32_t nc = pp_assembly_n_contigs(a);

    /* add_to_pileup for every SAM, in argv order (src/polish.rs:104-117) */
    pp_ingest *g = NULL;
    if (pp_ingest_create(a, 10, 0, &g)) return fail("ingest", "");
    for (int i = 2; i < argc; i++) {
        pp_sam_counts c;
        if (pp_ingest_sam(g, argv[i], &c, err, sizeof err)) return fail(argv[i], err);
        fprintf(stderr, "%s: %" PRIu64 " alignments, %" PRIu64 " used, %" PRIu64 " reads\n", argv[i], c.alignments,
                c.used, c.reads);
    }
    pp_aln_batch batch;
    pp_ingest_batch(g, &batch);

    /* seam B: pileup + vote on the device */
    pp_params params;
    params.min_depth = 5;
    params.fraction_valid = 0.5;
    params.fraction_invalid = 0.2;
    if (pp_polish_begin(ctx, nc, pp_assembly_offsets(a), pp_assembly_bases(a), PP_MEM_HOST, &params) ||
        pp_polish_add(ctx, &batch, PP_MEM_HOST) || pp_polish_finish(ctx))
        return fail("polish", pp_last_error(ctx));
    uint64_t total = 0;
    if (pp_polish_result_size(ctx, &total)) return fail("result", pp_last_error(ctx));
    uint8_t *out = (uint8_t *)malloc(total ? total : 1);
    uint64_t *off = (uint64_t *)malloc((nc + 1) * sizeof *off);
    pp_contig_stats *st = (pp_contig_stats *)malloc((nc ? nc : 1) * sizeof *st);
    if (!out || !off || !st) return fail("out of memory", "");
    if (pp_polish_result(ctx, out, PP_MEM_HOST, off, st)) return fail("result", pp_last_error(ctx));

    /* polish_one_sequence's printing (src/polish.rs:170-203) */
    for (uint32_t c = 0; c < nc; c++) {
        const char *d = pp_assembly_description(a, c);
        printf(">%s%s%s polypolish\n", pp_assembly_name(a, c), d && *d ? " " : "", d ? d : "");
        fwrite(out + off[c], 1, (size_t)(off[c + 1] - off[c]), stdout);
        putchar('\n');
        fprintf(stderr, "%s: %" PRIu64 " bp, %" PRIu64 " changed\n", pp_assembly_name(a, c), st[c].polished_len,
                st[c].changed);
    }
    free(st); free(off); free(out);
    pp_ingest_free(g);
    pp_assembly_free(a);
    pp_ctx_destroy(ctx);
    return 0;
}
