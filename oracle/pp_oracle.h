/*
 * pp_oracle.h -- CPU oracle for the Polypolish filter + pileup/vote hot path.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library.  The shipped path
 * (polypolish_amd/, bin/polypolish) never links, imports or executes it.
 *
 * What it is: a single-threaded plain-C restatement of the algorithm of
 * rrwick/Polypolish v0.6.1 (Rust), written from a reading of the reference
 * sources; every function cites the reference file:line it follows
 * (paths are relative to /root/reference/).
 *
 * Parity pinning: the Rust reference cannot be compiled in this image (no
 * cargo/rustc), so the oracle is pinned against every known-answer vector the
 * reference's own unit tests hold (committed as data
 * in tests/golden/reference_unit_vectors.json).  Functions the reference
 * itself never tests (read grouping, the CIGAR walk + homopolymer trim, the
 * pair rule, whole-program FASTA bytes) are pinned only by source reading
 * plus an independent second restatement (oracle/pyref.py) that must agree
 * byte-for-byte on randomised inputs -- for those the parity status is
 * "restated, cross-checked, not reference-pinned".
 */
#ifndef PP_ORACLE_H
#define PP_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Return codes mirror the reference's process exit codes. */
#define ORC_OK 0
#define ORC_QUIT 1    /* misc.rs:29-33 quit_with_error -> exit(1) */
#define ORC_PANIC 101 /* Rust panic (unwrap on None/Err, index out of bounds) */

/* BaseStatus (pileup.rs:18-25), numbered in the order of the debug strings
 * at pileup.rs:156-163. */
enum {
    ORC_ST_KEPT = 0,      /* OriginalBaseKept     "kept"      */
    ORC_ST_CHANGED = 1,   /* Changed              "changed"   */
    ORC_ST_LOW_DEPTH = 2, /* DepthTooLow          "low_depth" */
    ORC_ST_NONE = 3,      /* NoValidOptions       "none"      */
    ORC_ST_MULTIPLE = 4,  /* MultipleValidOptions "multiple"  */
    ORC_ST_TOO_CLOSE = 5  /* TooClose             "too_close" */
};

/* Growable byte buffer used for all variable-length outputs. */
typedef struct {
    char *data;
    size_t len;
    size_t cap;
} orc_buf;
void orc_buf_free(orc_buf *b);

/* ---- scalar helpers pinned by the reference's unit tests ---------------- */
uint32_t orc_bankers_rounding(double x);                          /* misc.rs:208-215 */
void orc_reverse_complement(const char *in, size_t n, char *out);  /* misc.rs:170-191 */
/* returns 0 and a malloc'd expanded string (caller frees) or -1 on invalid CIGAR */
int orc_get_expanded_cigar(const char *cigar, char **expanded, size_t *exp_len); /* alignment.rs:325-346 */
/* ref_end of a CIGAR placed at 0-based ref_start */
uint64_t orc_get_ref_end(uint64_t ref_start, const char *cigar);   /* alignment.rs:138-149 */
/* 0 fr, 1 rf, 2 ff, 3 rr */
int orc_get_orientation(uint32_t flags1, uint64_t start1, const char *cigar1, uint32_t flags2,
                        uint64_t start2, const char *cigar2);      /* filter.rs:189-209 */
uint32_t orc_get_insert_size(uint64_t start1, const char *cigar1, uint64_t start2,
                             const char *cigar2);                  /* filter.rs:212-218 */
uint32_t orc_get_percentile(const uint32_t *sorted, size_t n, double percentile); /* filter.rs:249-259 */
/* counts[4] in order fr, rf, ff, rr -> index of the unique maximum, or -1 */
int orc_auto_determine_orientation(const uint64_t counts[4]);      /* filter.rs:238-246 */

/* Parse one SAM line the way Alignment::new does (alignment.rs:49-98) and
 * report ref_start / ref_end (used for T3).  Returns ORC_OK / ORC_QUIT / ORC_PANIC. */
int orc_parse_positions(const char *sam_line, uint64_t *ref_start, uint64_t *ref_end);

/* ---- one PileupBase driven directly (pileup.rs:29-134; T8) -------------- */
typedef struct orc_pileup_base orc_pileup_base;
orc_pileup_base *orc_pb_new(char original);
void orc_pb_free(orc_pileup_base *b);
void orc_pb_add_seq(orc_pileup_base *b, const char *seq, size_t n, double depth_contribution);
/* new_base is copied into out (NUL-terminated, cap bytes); returns the status */
int orc_pb_get_polished_seq(const orc_pileup_base *b, uint32_t min_depth, double fraction_valid,
                            double fraction_invalid, char *out, size_t cap);
void orc_pb_get_count_str(const orc_pileup_base *b, orc_buf *out); /* pileup.rs:137-148 */

/* ---- CIGAR walk + trim on one alignment (alignment.rs:175-201,364-378) -- */
/* On success fills starts/ends (malloc'd, caller frees) with the kept
 * (start,end) read slices, one per covered reference position. */
int orc_read_bases_for_each_target_base(const char *cigar, const char *read_seq, size_t seq_len,
                                        uint32_t **starts, uint32_t **ends, size_t *n,
                                        char *err, size_t errlen);

/* ---- FASTA loader (misc.rs:38-167) -------------------------------------- */
typedef struct {
    size_t n;
    char **name;
    char **desc;
    char **seq;
    size_t *len;
} orc_fasta;
int orc_load_fasta(const char *path, orc_fasta *out, char *err, size_t errlen);
void orc_fasta_free(orc_fasta *f);

/* ---- whole-program drivers ---------------------------------------------- */
typedef struct {
    double fraction_invalid; /* -i, default 0.2 (main.rs:85-87) */
    double fraction_valid;   /* -v, default 0.5 (main.rs:89-91) */
    uint32_t max_errors;     /* -m, default 10  (main.rs:93-95) */
    uint32_t min_depth;      /* -d, default 5   (main.rs:97-99) */
    int careful;             /* --careful       (main.rs:101-103) */
} orc_polish_params;

/* Optional per-position dump, concatenated over contigs in FASTA order. */
typedef struct {
    size_t n_positions;
    double *depth;      /* PileupBase.depth                */
    uint32_t *count_a;  /* count_a/c/g/t                    */
    uint32_t *count_c;
    uint32_t *count_g;
    uint32_t *count_t;
    uint32_t *count_other; /* sum over the HashMap (incl. "-") */
    uint32_t *valid_thr;
    uint32_t *invalid_thr;
    uint8_t *status;    /* ORC_ST_*                         */
    uint32_t *emit_len; /* bytes this position put into the polished string (new_base minus '-', polish.rs:185-188) */
} orc_positions;
void orc_positions_free(orc_positions *p);

typedef struct {
    uint64_t alignment_total; /* polish.rs:113-121 */
    uint64_t used_total;
    uint64_t read_total;
} orc_polish_counts;

/* polish.rs:26-38.  fasta receives exactly the bytes the reference prints to
 * stdout; debug (may be NULL) receives the --debug TSV bytes. */
int orc_polish_files(const char *assembly, const char *const *sams, int n_sams,
                     const orc_polish_params *p, orc_buf *fasta, orc_buf *debug,
                     orc_positions *positions, orc_polish_counts *counts, char *err,
                     size_t errlen);

/* filter.rs:26-37.  Writes out1/out2 exactly as the reference does. */
typedef struct {
    uint64_t before_count;
    uint64_t after_count;
    uint32_t low_threshold;
    uint32_t high_threshold;
    int orientation; /* 0 fr 1 rf 2 ff 3 rr, -1 = user string matching none */
    uint64_t orientation_counts[4];
} orc_filter_report;
int orc_filter_files(const char *in1, const char *in2, const char *out1, const char *out2,
                     const char *orientation, double low, double high, orc_filter_report *rep,
                     char *err, size_t errlen);

/* ---- record-level entry (same SoA the product's C ABI takes) ------------
 * Runs Pileup::add_alignment (pileup.rs:189-200) for every record in array
 * order with depth contribution 1.0/k, then polish_one_sequence
 * (polish.rs:157-193) per contig.  The host-side gates of process_one_read
 * (alignment.rs:275-305) are assumed to have been applied by the caller, as at
 * the product boundary.  CIGAR ops are packed len<<4|op with op codes
 * M=0 I=1 D=2 N=3 S=4 H=5 P=6 '='=7 X=8. */
typedef struct {
    uint64_t n_aln;
    const uint32_t *contig;
    const uint32_t *ref_start;
    const uint32_t *k;
    const uint64_t *seq_off;
    const uint32_t *seq_len;
    const uint64_t *cig_off;
    const uint32_t *n_cig;
    const uint8_t *seq;
    const uint32_t *cigar;
} orc_records;

int orc_polish_records(uint32_t n_contigs, const uint64_t *contig_off, const uint8_t *bases,
                       const orc_records *recs, uint32_t min_depth, double fraction_valid,
                       double fraction_invalid, orc_buf *polished /* concatenated, no headers */,
                       uint64_t *polished_off /* n_contigs+1 */, orc_positions *positions,
                       char *err, size_t errlen);

#ifdef __cplusplus
}
#endif
#endif
