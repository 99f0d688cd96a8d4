/*
 * polypolish_hip.h -- C ABI of libpolypolish_hip.so, the MI355X (gfx950) implementation of
 * Polypolish's alignment-filter + per-base pileup/vote hot path.
 *
 * The reference (rrwick/Polypolish v0.6.1, Rust) has no FFI of its own; these entry points sit
 * at the two seams where its drivers call into the hot path, and take exactly what those calls
 * take, as structure-of-arrays in SAM file order (citations are file:line into the reference):
 *
 *   seam B (polish): process_one_read -> pileup.add_alignment(a, depth_contribution)
 *                    (src/alignment.rs:297-303) and polish_one_sequence -> get_polished_seq
 *                    (src/polish.rs:170-187)              => pp_polish_begin / _add / _finish
 *   seam A (filter): filter_sam -> alignment_pass_qc (src/filter.rs:334) and the sampling loop of
 *                    get_insert_size_thresholds (src/filter.rs:155-167)
 *                                                         => pp_filter_samples / pp_filter_pairs
 *
 * plus the ingest either side of them (FASTA/SAM text -> the SoA above; src/misc.rs:38-167,
 * src/alignment.rs:49-128,225-322, src/filter.rs:91-145,309-349) that the `polypolish` CLI and any
 * binding share: as multi-threaded host code (pp_ingest_*, pp_filter_load / pp_filter_write) and as
 * device tokenizers over the uploaded text (pp_dev_ingest_*, pp_filter_load_device), and the whole
 * commands (pp_polish_files, pp_filter_files, pp_filter_polish_files).
 *
 * Conventions: plain pointers and sizes, no C++/torch types.  Every function returns PP_OK or a
 * PP_ERR_* code and never exits the process; pp_last_error() holds the message the reference
 * would have printed after "Error: ".  A pp_ctx is bound to one HIP device and one stream and is
 * not thread-safe; distinct contexts are independent.  There is no CPU fallback: without a
 * usable gfx950 device pp_ctx_create fails with PP_ERR_HIP.
 */
#ifndef POLYPOLISH_HIP_H
#define POLYPOLISH_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes ------------------------------------------------------------------------- */
#define PP_OK 0
#define PP_ERR_QUIT 1    /* the reference would quit_with_error (src/misc.rs:29-33): exit code 1 */
#define PP_ERR_HIP 3     /* HIP runtime failure, or no usable device                             */
#define PP_ERR_ARG 4     /* the caller broke this header's contract                               */
#define PP_ERR_LIMIT 5   /* input exceeds a documented implementation limit                       */
#define PP_ERR_NOT_ASCII 6 /* device text front ends (pp_dev_ingest_sam*, pp_filter_load_device) only: the file holds
                              bytes outside ASCII.  Whether all of its lines are valid UTF-8 -- the reference refuses
                              the others (BufRead::lines) -- is decided by the host parsers: load it with those.  The
                              file drivers (pp_polish_files, pp_filter_files, ...) do that by themselves           */
#define PP_ERR_PANIC 101 /* the reference would panic (unwrap / index out of bounds): exit 101    */

#define PP_MEM_HOST 0   /* pointer is host memory: the library copies it to the device             */
#define PP_MEM_DEVICE 1 /* pointer is device memory on the context's device: borrowed, not copied  */
#define PP_MEM_PEER 2   /* pp_polish_add only: device memory of ANOTHER GPU of this process (a pp_shard_part made on that
                           GPU's context): copied over the fabric, the source may be released when the call returns   */

/* CIGAR runs are packed (length << 4) | op with these op codes (SAM order). */
enum { PP_OP_M = 0, PP_OP_I = 1, PP_OP_D = 2, PP_OP_N = 3, PP_OP_S = 4, PP_OP_H = 5, PP_OP_P = 6,
       PP_OP_EQ = 7, PP_OP_X = 8 };
/* Filter input only: a run with this op code marks an alignment whose CIGAR holds a length that does not fit 64 bits.
   The reference parses run lengths when it needs an alignment's end (get_ref_end, src/alignment.rs:138-149, called from
   get_insert_size / get_orientation, src/filter.rs:189-218) and panics there, so such an alignment is only fatal if it
   takes part in a pair comparison: its ref_end is PP_REF_END_UNPARSEABLE and pp_filter_samples / pp_filter_pairs
   return PP_ERR_PANIC if they have to evaluate it (a precomputed ref_end array uses the same value). */
#define PP_OP_UNPARSEABLE 15
#define PP_REF_END_UNPARSEABLE 0xFFFFFFFFFFFFFFFFull

/* BaseStatus (src/pileup.rs:18-25) in the order of its debug strings (src/pileup.rs:156-163). */
enum { PP_ST_KEPT = 0, PP_ST_CHANGED = 1, PP_ST_LOW_DEPTH = 2, PP_ST_NONE = 3, PP_ST_MULTIPLE = 4,
       PP_ST_TOO_CLOSE = 5 };

typedef struct pp_ctx pp_ctx;

/* ---- context ------------------------------------------------------------------------------ */
int pp_device_count(void);             /* usable HIP devices (0 if there is none); initialises the HIP runtime */
int pp_ctx_create(int device, pp_ctx **out);
/* Same, but the HIP runtime / device initialisation (a few 100 ms in a fresh process) runs on a helper
 * thread so that host-side work (pp_assembly_load, pp_ingest_sam, the host half of pp_polish_files /
 * pp_filter_files) overlaps it.  Every entry point that needs the device waits for it first;
 * pp_ctx_wait() waits explicitly and returns PP_OK or PP_ERR_HIP / PP_ERR_ARG (no usable device). */
int pp_ctx_create_async(int device, pp_ctx **out);
int pp_ctx_wait(pp_ctx *ctx);
void pp_ctx_destroy(pp_ctx *ctx);
const char *pp_last_error(const pp_ctx *ctx);
int pp_ctx_sync(pp_ctx *ctx);          /* hipStreamSynchronize on the context's stream          */
void *pp_ctx_stream(pp_ctx *ctx);      /* the hipStream_t every kernel of this context runs on  */
/* device -> host copy on the context's stream, synchronous (for bindings that have no HIP of their own) */
int pp_ctx_download(pp_ctx *ctx, void *host_dst, const void *dev_src, uint64_t bytes);
const char *pp_version(void);          /* "polypolish-mi355x <ver> (parity target v0.6.1)"       */

/* Three strings of the run log that the reference pins with unit tests, exported so that those vectors can be
 * checked against the product's own text: PP_TEXT_QSCORE (qscore, src/polish.rs:290-300; value = identity in %),
 * PP_TEXT_DURATION (format_duration, src/misc.rs:195-201; value = microseconds), PP_TEXT_PERCENTILE_NAME
 * (get_percentile_name, src/filter.rs:262-270).  PP_OK and a NUL-terminated string in out, or PP_ERR_ARG (unknown `what`, text does not fit). */
enum { PP_TEXT_QSCORE = 0, PP_TEXT_DURATION = 1, PP_TEXT_PERCENTILE_NAME = 2 };
int pp_log_text(int what, double value, char *out, size_t cap);

/* ---- seam B: pileup accumulate + per-position vote ------------------------------------------ */
typedef struct {
    uint32_t min_depth;      /* -d, src/main.rs:97-99   */
    double fraction_valid;   /* -v, src/main.rs:89-91   */
    double fraction_invalid; /* -i, src/main.rs:85-87   */
} pp_params;

/* One batch of GOOD alignments (the gates of process_one_read, src/alignment.rs:282-287, already
 * applied; SEQ "*" already filled, src/alignment.rs:290-295), in global SAM order: array index i
 * is the order in which the reference would call add_alignment.  Contract per record:
 *   contig     index into the assembly's contigs (FASTA order)
 *   ref_start  0-based leftmost reference position (POS-1; src/alignment.rs:58-61)
 *   k          number of good alignments of the read (depth contribution is 1.0/k, :288), >= 1
 *   seq        SEQ bytes after to_ascii_uppercase (src/alignment.rs:94), seq_len[i] of them at
 *              seq + seq_off[i]
 *   cigar      n_cig[i] packed runs at cigar + cig_off[i]; every run length >= 1; first and last
 *              run M or '=' (starts_and_ends_with_match, src/alignment.rs:155-159)
 * The device validates all of it (runs other than M,=,X,I,D; CIGAR/SEQ length mismatch; reads
 * running past the contig end) and reports the FIRST offending record in file order, as the
 * reference would. */
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
    uint64_t seq_bytes;
    const uint32_t *cigar;
    uint64_t n_cig_total;
    /* Optional (NULL = none): a 4-bit mirror of seq, two bases per byte -- base seq[i] in bits 4*(i&1).. of seq4[i >> 1]
     * (the mirror is a function of the whole seq ARRAY, position by position, not of the records), (seq_bytes + 1) / 2
     * bytes followed by at least 32 readable bytes; codes PP_SEQ4_*.  With it the pileup kernel fetches the reads without
     * indels at half the bytes, one lane per read.  The library's own producers bring one (pp_dev_ingest_batch,
     * device parts of pp_shard_split); pp_polish_add copies it along, and packs one on the device for a batch that comes
     * without (host batches: the host ingest's, anybody's) whenever the batch is copied into the library's arrays anyway --
     * only a foreign PP_MEM_DEVICE batch that is polished in place (the only batch of its job) runs without.  Every result
     * is the same with and without it.  seq must be there all the same: everything that needs a byte as it was delivered
     * (string-keyed tallies, trims through bytes other than A/C/G/T/N/-) reads seq. */
    const uint8_t *seq4;
    /* Optional (NULL = none): the records once more, in WINDOW ORDER -- n_aln entries of 32 bytes, a permutation of the batch's
     * records in which the records that start in one 2048-position window of the assembly are adjacent (per SAM file, as the
     * SEQ bytes of PP_SEQ_WINDOW_GROUPED; inside a window in any order).  Like seq4 it is a mirror, a function of the arrays
     * above, and a hint: every result is the same with and without it.  The polish reads the records through it when it
     * is there: a workgroup's records then fall into a handful of windows and its work items are written next to each
     * other, where the file order scatters them over the whole assembly (the bucketing kernels take 0.15 instead of 0.25 ms
     * on the 5 Mbp / 200x job).  The library's ingests bring one; pp_polish_add carries it along while every batch of the
     * job has one.
     * The hint is ENFORCED (round 6): a mirror that is not one of the library's own (pp_ingest_batch, pp_dev_ingest_batch, the
     * parts of pp_shard_split, or a part of one of those) is compared with the arrays on the device before anything reads the
     * records through it -- every entry a record of the batch, none twice, contig / ref_start / k / seq_off / seq_len / first
     * run the record's -- and one that does not stand the comparison is left aside: the job runs without it and gives the
     * same bytes.  The comparison is a scattered read of the arrays (about 1 ms per 6.7 M records): a caller whose mirror is
     * not worth that should not pass one. */
    const struct pp_wo_rec *wo;
    /* Optional with wo (0 / NULL = not known): the mirror as RUNS.  wo_n_runs stretches of entries, one behind the other
     * (one per SAM file, as the library's ingests write it), each of them window-grouped IN ASCENDING WINDOW ORDER;
     * wo_run_end[r] = the index one past the last entry of run r (ascending; the last one = n_aln).  HOST memory whatever
     * the batch's `mem` (a handful of numbers; copied by pp_polish_add).  With it -- a sharded job (pp_polish_set_emit)
     * included: pp_shard_split restricts the table along with the mirror -- the pileup kernel takes the records that are one short M run inside their contig STRAIGHT from
     * the mirror, window by window (round 5: no bucketing pass, no work items for them); one streaming pass over the
     * mirror validates every record as before, finds where each window's entries start in every run and cuts work items
     * only for the others (indels, long reads) and for the reads that reach into the next window.  A hint like the
     * mirror itself: where the entries turn out not to be in that order (any permutation is still a valid mirror) the
     * job silently takes the bucketing path; every result is the same. */
    uint32_t wo_n_runs;
    const uint64_t *wo_run_end;
} pp_aln_batch;
#define PP_WO_MAX_RUNS 16  /* more runs than this: the bucketing path */
/* one record of pp_aln_batch.wo: the fields the bucketing reads, file_idx = the record's index in the batch's arrays (its
 * place in file order), op0 = its only CIGAR run (packed as in `cigar`) or PP_WO_MULTI_RUN for a record of several runs
 * (those are read from n_cig / cig_off / cigar by file_idx) */
typedef struct pp_wo_rec {
    uint32_t contig, ref_start, k, seq_len;
    uint64_t seq_off;
    uint32_t op0, file_idx;
} pp_wo_rec;
#define PP_WO_MULTI_RUN 0xFFFFFFFFu
/* The library's own producers of batches (pp_ingest_*, pp_dev_ingest_*, pp_shard_split) start every record's SEQ on a multiple
 * of PP_SEQ_ALIGN bytes of the seq array, the bytes in between zero, and lay the SEQ bytes of a SAM file out WINDOW-GROUPED
 * (PP_SEQ_WINDOW_GROUPED below: the default since round 4): the pileup kernel waits for the 128-byte lines a read touches,
 * and fetches a window's reads from one stretch of memory.  A batch from elsewhere may place its SEQ bytes anywhere (seq_off). */
#define PP_SEQ_ALIGN 32
/* codes of seq4: the four bases as their counter rows, N, '-' (the deletion key, src/pileup.rs:194-197), anything else */
#define PP_SEQ4_A 0
#define PP_SEQ4_C 1
#define PP_SEQ4_T 2
#define PP_SEQ4_G 3
#define PP_SEQ4_N 4
#define PP_SEQ4_DASH 5
#define PP_SEQ4_OTHER 15

/* Per-contig figures the reference prints to stderr (src/polish.rs:206-227). */
typedef struct {
    uint64_t polished_len;     /* bytes of the polished sequence                      */
    uint64_t changed;          /* positions with status Changed                       */
    uint64_t zero_depth;       /* positions with depth == 0.0                         */
    double depth_sum;          /* sum of per-position depth (cosmetic: the "mean read depth" line of the log, polish.rs:206-227).
                                * Exact where every depth share 1/k is a power of two; a position with other shares adds its depth
                                * rounded to 2^-10 (its VOTE is exact -- interval test or ordered replay -- its contribution to this
                                * sum is not replayed): |depth_sum - reference| <= polished_len * 2^-11 + 1e-6 * depth_sum, i.e. the
                                * mean depth printed to one decimal can differ in that digit only when it lies within 0.0005 of a
                                * rounding boundary (tests/test_gpu_parity.py compares the sums under this bound). */
} pp_contig_stats;

/* Optional, between pp_polish_begin and pp_polish_finish: restrict what contig c EMITS to its positions
 * [emit_lo[c], emit_hi[c]) (0-based, relative to the contig; arrays of n_contigs, host memory).  Positions outside
 * the range contribute no polished bytes and no statistics, records that do not reach the range are validated and
 * then dropped, windows outside it are not worked on.  This is how one rank of a sharded job (SURVEY 8e, configs
 * C4 / C5) polishes its contigs or its window of a large contig: it is given the records that reach its ranges
 * (pp_shard_split -- or simply all records) and emits only what it owns; an owned position sees all of its
 * alignments in file order either way, so the order-dependent f64 depth is exact.  NULL, NULL = everything. */
int pp_polish_set_emit(pp_ctx *ctx, const uint64_t *emit_lo, const uint64_t *emit_hi);

/* Start a polish job.  contig_off is a HOST array of n_contigs+1 offsets into `bases` (the
 * concatenated, ASCII-uppercased assembly, src/misc.rs:114,129; total length < 2^32-4096). */
int pp_polish_begin(pp_ctx *ctx, uint32_t n_contigs, const uint64_t *contig_off,
                    const uint8_t *bases, int bases_mem, const pp_params *params);
/* Provide alignments, one batch after the other as the reference streams its SAM files (src/alignment.rs:238-265):
 * every record of a later batch follows the records of the earlier ones in file order; seq_off / cig_off are
 * relative to the batch's own seq / cigar arrays.  `mem` applies to every pointer in the batch.  Host batches are
 * copied before the call returns; PP_MEM_DEVICE batches must stay valid and unchanged until pp_polish_finish returns
 * (a job's ONLY batch is used in place; with more than one batch everything is gathered into library-owned arrays
 * with device-to-device copies on the context's stream).  Limits per job: < 2^32-1 records, < 2^40 SEQ bytes. */
int pp_polish_add(pp_ctx *ctx, const pp_aln_batch *batch, int mem);
/* Optional, after pp_polish_begin: room for everything the job's batches will hold together (a guess is fine). */
int pp_polish_reserve(pp_ctx *ctx, uint64_t n_aln, uint64_t seq_bytes, uint64_t n_cig_total);
/* Run the kernels: CIGAR walk + homopolymer trim (src/alignment.rs:175-201,364-378), pileup
 * accumulation (src/pileup.rs:56-65,189-200), vote (src/pileup.rs:67-134), '-' removal and
 * concatenation (src/polish.rs:185-188).  Results stay on the device until fetched. */
int pp_polish_finish(pp_ctx *ctx);
/* Total polished bytes over all contigs (no headers, no newlines). */
int pp_polish_result_size(pp_ctx *ctx, uint64_t *total_bytes);
/* Copy the polished bytes to `out` (host or device, >= result_size bytes); contig c occupies
 * [contig_out_off[c], contig_out_off[c+1]).  contig_out_off (HOST, n_contigs+1) and stats (HOST,
 * n_contigs) may be NULL. */
int pp_polish_result(pp_ctx *ctx, uint8_t *out, int out_mem, uint64_t *contig_out_off,
                     pp_contig_stats *stats);
/* Device pointer to the polished bytes (valid until the next pp_polish_begin); the context's stream is idle when this returns
 * (pp_polish_finish itself may return a few microseconds before its last kernel's end has been signalled: it has the job's
 * results from that kernel's last workgroup -- readers on other streams come through here). */
const uint8_t *pp_polish_result_device(pp_ctx *ctx);

/* Optional per-position record (what the --debug TSV is made of, src/pileup.rs:150-166):
 * enable before pp_polish_finish, fetch after.  Arrays are HOST, one element per assembly
 * position in concatenated order; any pointer may be NULL. */
typedef struct {
    double *depth;
    uint32_t *count_a, *count_c, *count_g, *count_t;
    uint32_t *count_other;  /* sum over the string-keyed table, deletions ("-") included */
    uint32_t *valid_thr, *invalid_thr;
    uint8_t *status;        /* PP_ST_* */
} pp_positions;
int pp_polish_set_debug(pp_ctx *ctx, int enable);
int pp_polish_positions(pp_ctx *ctx, const pp_positions *out);

/* The rest of a --debug line (src/pileup.rs:137-166): the string-keyed tallies and the winning
 * sequence of every position.  Valid after pp_polish_finish with debug enabled; the arrays are
 * malloc'd by the library (release with pp_debug_extra_free).
 *   emit[p]      0 = nothing is emitted (a deletion won, or the assembly byte is '-'), 1..127 = that
 *                byte, >= 128 = a multi-byte winner listed in multi_* (raw bytes at seq + multi_off)
 *   key_*        one record per (position, distinct key other than A/C/G/T): the key is the len bytes
 *                at seq + off (the batch's seq array); len 0 = the deletion key "-"            */
typedef struct {
    uint8_t *emit;         /* one per assembly position */
    uint64_t n_multi;
    uint32_t *multi_pos, *multi_len;
    uint64_t *multi_off;
    uint64_t n_keys;
    uint32_t *key_pos, *key_len, *key_count;
    uint64_t *key_off;
} pp_debug_extra;
int pp_polish_debug_extra(pp_ctx *ctx, pp_debug_extra *out);
void pp_debug_extra_free(pp_debug_extra *d);

/* Per-kernel device time of the last pp_polish_finish, measured with HIP events on the context's
 * stream when profiling is enabled.  names[i] are static strings. */
#define PP_MAX_KERNELS 16
typedef struct {
    int n;
    const char *name[PP_MAX_KERNELS];
    float ms[PP_MAX_KERNELS];
    uint64_t n_entries;   /* (alignment, window) work items the tile kernel consumed */
    uint64_t n_flagged;   /* positions resolved by the exact (string-keyed, ordered-f64) kernel */
    uint64_t n_passes;    /* passes over the pipeline: 1, or more when a device buffer had to grow (first job) */
} pp_kernel_times;
int pp_ctx_set_profiling(pp_ctx *ctx, int enable); /* 0 off, 1 every kernel group, 2 only the dominant kernel ("tile") */
int pp_polish_kernel_times(pp_ctx *ctx, pp_kernel_times *out);
/* Which way the last pp_polish_finish of this context went: 1 = the direct path (the window-order mirror and its run table:
 * pp_aln_batch.wo_run_end -- a rank of a sharded job included: pp_shard_split restricts the table with the mirror), 0 = the
 * bucketing path (no mirror, no run table, more than PP_WO_MAX_RUNS runs -- one per SAM file, or per file and device in the
 * one-process multi-GPU driver --, a
 * mirror that turned out not to be in run order or not to mirror its records, a window that needs more room for its extras
 * than the windows can be given).  The results are the same either way; for reports (bench.py names the kernel it timed). */
int pp_polish_took_direct_path(const pp_ctx *ctx);

/* ---- multi-GPU: contigs (and windows of a large contig) shard across ranks -----------------------------------------
 * The reference is one thread on one CPU; the partition is SURVEY.md 8(e) / BASELINE.json configs[3], [4].
 * A rank polishes the records that reach its units (pp_shard_split picks them out of a batch; giving a rank every
 * record is allowed too) with pp_polish_set_emit(the ranges of its units): the device works on its windows only, and
 * every owned position sees all of its alignments in file order.  The only exchange of the polish itself is
 * pp_polish_gather.
 *   plan    whole contigs by longest-processing-time on their alignment counts; a contig with more than one rank's
 *           share of the alignments is cut into up to `world` windows on 2048-bp boundaries (>= min_window bp each,
 *           0 = 65536), one per rank.  Units are listed contig by contig, windows in position order. */
typedef struct {
    uint32_t n_units, world, n_contigs;
    uint32_t *contig;      /* per unit */
    uint64_t *lo, *hi;     /* [lo, hi) of the contig */
    uint32_t *rank;        /* the rank that polishes it */
} pp_shard_plan;
int pp_shard_plan_create(uint32_t n_contigs, const uint64_t *contig_off, const uint64_t *aln_per_contig, uint32_t world,
                         uint64_t min_window, pp_shard_plan **out);
void pp_shard_plan_free(pp_shard_plan *plan);
/* the ranges one rank emits, as pp_polish_set_emit takes them (arrays of n_contigs; empty range = not its contig) */
int pp_shard_emit_ranges(const pp_shard_plan *plan, uint32_t rank, uint64_t *emit_lo, uint64_t *emit_hi);
/* The ranks' polished bytes back in assembly order (HOST memory): rank_bytes[r] / rank_contig_off[r] are what rank
 * r's pp_polish_result returned (out, contig_out_off); out may be NULL to get the offsets only. */
int pp_shard_assemble(const pp_shard_plan *plan, const uint8_t *const *rank_bytes, const uint64_t *const *rank_contig_off,
                      uint8_t *out, uint64_t *contig_out_off);
/* The records of `batch` (file order) that rank `dest` needs under `plan`: a record goes to the rank of every unit its
 * reference span [ref_start, ref_start + sum of its M/=/X/D/N run lengths) touches -- its contig's owner, or, on a tiled
 * contig, every window it overlaps (src/alignment.rs:297-303, src/pileup.rs:189-200: an alignment only ever touches
 * its own contig's positions ref_start + j).  A record that touches no unit (bad contig index, start beyond the
 * contig's end) goes to one fixed rank, which reports it.  mem = where `batch` lives: PP_MEM_HOST -> the part is host
 * memory (ctx may be NULL), PP_MEM_DEVICE -> the part is made by kernels on ctx's stream and lives on ctx's GPU.
 * The part is a pp_aln_batch of its own (offsets relative to its own seq / cigar arrays); orig[i] is the index of its
 * record i in `batch` (same memory kind as the part): a record number reported by a rank's pp_polish_finish is
 * rank-local, orig turns it back into the job's, and the job's first bad record is the minimum over the ranks. */
typedef struct pp_shard_part pp_shard_part;
int pp_shard_split(pp_ctx *ctx, const pp_shard_plan *plan, uint32_t dest, const pp_aln_batch *batch, int mem,
                   pp_shard_part **out);
void pp_shard_part_batch(const pp_shard_part *part, pp_aln_batch *out, const uint32_t **orig); /* borrowed views */
int pp_shard_part_mem(const pp_shard_part *part);
void pp_shard_part_free(pp_shard_part *part);
/* Alignment records per contig -- the planner's weights -- ADDED to aln_per_contig (HOST, n_contigs). */
int pp_shard_count(pp_ctx *ctx, const pp_aln_batch *batch, int mem, uint32_t n_contigs, uint64_t *aln_per_contig);
/* The record and the kind of the device error pp_polish_finish last returned on this context (PP_ERR_QUIT / PP_ERR_PANIC
 * from the CIGAR walk): *record = its index over the batches added to THIS context; returns 0 when there was none.
 * pp_polish_error_text writes the message for that kind with another record number (the job-wide one). */
int pp_polish_error_record(const pp_ctx *ctx, uint64_t *record, uint32_t *kind);
int pp_polish_error_text(pp_ctx *ctx, uint32_t kind, uint64_t record);

/* RCCL communicator of one rank (librccl is loaded at run time; the process's own copy is used if it has one):
 * rank 0 calls pp_comm_unique_id, the launcher hands the PP_COMM_ID_BYTES to every rank, every rank calls
 * pp_comm_init on the context of its GPU. */
#define PP_COMM_ID_BYTES 128
int pp_comm_unique_id(void *id);
int pp_comm_init(pp_ctx *ctx, int rank, int world, const void *id);
void pp_comm_destroy(pp_ctx *ctx);
/* After pp_polish_finish, on every rank: the polished bytes of all ranks go to rank 0 (ncclAllGather of the byte
 * counts and per-contig output offsets, then one group of ncclSend / ncclRecv into exclusive-scan offsets).
 * gathered: DEVICE buffer on rank 0 (>= the sum of the counts, `cap` bytes; NULL elsewhere), filled rank by rank;
 * rank_len (world) and rank_contig_off (world x (n_contigs + 1)): HOST, filled on every rank, either may be NULL. */
int pp_polish_gather(pp_ctx *ctx, uint8_t *gathered, uint64_t cap, uint64_t *rank_len, uint64_t *rank_contig_off);

/* ---- seam A: paired-read insert-size filter --------------------------------------------------
 * Alignments of both SAM files as SoA in file order (Alignment::new_quick, src/alignment.rs:
 * 102-128), plus the read-name grouping the reference builds in its HashMap (src/filter.rs:
 * 91-145): reads are numbered 0..n_reads-1; for file f, the alignments of read r are
 * grp_idx[f][grp_off[f][r] .. grp_off[f][r+1]) (indices into file f's arrays, file order). */
typedef struct {
    uint64_t n_aln;
    const uint32_t *ref_id;     /* RNAME interned to an integer (equal names <=> equal ids) */
    const uint32_t *ref_start;  /* POS-1 */
    const uint32_t *flags;      /* FLAG */
    const uint64_t *cig_off;
    const uint32_t *n_cig;
    const uint32_t *cigar;      /* packed runs; ops outside MIDNSHP=X never appear (regex) */
    uint64_t n_cig_total;
    const uint32_t *read;       /* read number of each alignment */
    const uint32_t *grp_off;    /* n_reads+1 */
    const uint32_t *grp_idx;    /* n_aln */
    const uint64_t *ref_end;    /* optional: precomputed get_ref_end per alignment; then cig_off / n_cig / cigar are not read */
} pp_filter_file;

typedef struct {
    uint32_t n_reads;
    pp_filter_file file[2];
} pp_filter_input;

/* Upload (or adopt) the input and compute ref_end for every alignment on the device
 * (Alignment::get_ref_end, src/alignment.rs:138-149). */
int pp_filter_begin(pp_ctx *ctx, const pp_filter_input *in, int mem);
/* For every read with exactly one alignment in each file on the same reference
 * (src/filter.rs:155-167): orientation (0 fr, 1 rf, 2 ff, 3 rr; get_orientation,
 * src/filter.rs:189-209) and insert size (src/filter.rs:212-218).  orient[r] = 255 for reads that
 * are not sampled.  HOST arrays of n_reads. */
int pp_filter_samples(pp_ctx *ctx, uint8_t *orient, uint32_t *insert);
/* alignment_pass_qc (src/filter.rs:352-377) for every alignment of both files; pass1/pass2 are
 * HOST arrays (1 = pass, 0 = append ZP:Z:fail). */
int pp_filter_pairs(pp_ctx *ctx, uint32_t low, uint32_t high, uint8_t orientation,
                    uint8_t *pass1, uint8_t *pass2);
int pp_filter_kernel_times(pp_ctx *ctx, pp_kernel_times *out);

/* Host half of the filter (no device needed): both SAM files -> the pp_filter_input above, and the
 * re-emission of a file with "\tZP:Z:fail" appended where pass == 0.  Multi-threaded; the result does
 * not depend on the thread count.
 *   pp_filter_load    load_alignments, src/filter.rs:91-145 (Alignment::new_quick, src/alignment.rs:102-128):
 *                     file 1 then file 2; counts[f].loaded tells which files were read completely when
 *                     an error is returned ("unable to load", "too few columns ... (line N)", ...)
 *   pp_filter_write   filter_sam, src/filter.rs:309-349: header and unaligned lines verbatim, every line
 *                     ends in "\n"; pass = HOST array over the alignments of file f (0/1) */
typedef struct pp_filter_loaded pp_filter_loaded;
typedef struct {
    uint64_t alignments; /* aligned records in the file        (filter.rs:105) */
    uint64_t reads;      /* distinct QNAMEs among them                          */
    int loaded;
} pp_filter_file_counts;
int pp_filter_load(const char *in1, const char *in2, pp_filter_loaded **out, pp_filter_file_counts counts[2],
                   char *err, size_t errlen);
void pp_filter_loaded_input(const pp_filter_loaded *loaded, pp_filter_input *in); /* borrows from `loaded` */
int pp_filter_write(const pp_filter_loaded *loaded, int file, const uint8_t *pass, const char *out_path,
                    uint64_t *pass_count, uint64_t *fail_count, char *err, size_t errlen);
void pp_filter_loaded_free(pp_filter_loaded *loaded);

/* The same load on the DEVICE: both texts are uploaded; quick parse, get_ref_end, QNAME / RNAME interning
 * (device hash tables) and the per-file group index run as kernels.  pp_filter_dev_input gives the
 * pp_filter_input view in DEVICE memory (ref_end filled in, no CIGAR arrays) for pp_filter_begin(..., PP_MEM_DEVICE);
 * it equals pp_filter_load's result except that RNAME ids are arbitrary.  The host mappings of the two files
 * stay open inside the object (pp_filter_dev_text) for pp_filter_write_text. */
typedef struct pp_filter_dev pp_filter_dev;
int pp_filter_load_device(pp_ctx *ctx, const char *in1, const char *in2, pp_filter_dev **out,
                          pp_filter_file_counts counts[2]);
void pp_filter_dev_input(const pp_filter_dev *loaded, pp_filter_input *in);
const char *pp_filter_dev_text(const pp_filter_dev *loaded, int file, uint64_t *size);
void pp_filter_dev_free(pp_filter_dev *loaded);
/* filter_sam (src/filter.rs:309-349) straight from a SAM text in memory: pass = HOST array over its aligned
 * records (lines that are neither headers nor FLAG & 4), in file order. */
int pp_filter_write_text(const char *text, uint64_t size, const uint8_t *pass, uint64_t n_pass, const char *out_path,
                         uint64_t *pass_count, uint64_t *fail_count, char *err, size_t errlen);

/* ---- host ingest (text -> SoA) -------------------------------------------------------------- */
typedef struct pp_assembly pp_assembly;
/* load_fasta + check_load_fasta (src/misc.rs:38-75): plain or gzip, uppercased. */
int pp_assembly_load(const char *path, pp_assembly **out, char *err, size_t errlen);
void pp_assembly_free(pp_assembly *a);
uint32_t pp_assembly_n_contigs(const pp_assembly *a);
const char *pp_assembly_name(const pp_assembly *a, uint32_t i);
const char *pp_assembly_description(const pp_assembly *a, uint32_t i);
const uint64_t *pp_assembly_offsets(const pp_assembly *a); /* n_contigs+1 */
const uint8_t *pp_assembly_bases(const pp_assembly *a);

typedef struct {
    uint64_t alignments; /* aligned records seen            (src/alignment.rs:252) */
    uint64_t used;       /* good alignments kept            (src/alignment.rs:304) */
    uint64_t reads;      /* read groups                     (src/alignment.rs:260,266) */
} pp_sam_counts;

typedef struct pp_ingest pp_ingest;
int pp_ingest_create(const pp_assembly *a, uint32_t max_errors, int careful, pp_ingest **out);
/* add_to_pileup's streaming half (src/alignment.rs:225-272): parse, group adjacent QNAMEs, apply
 * process_one_read's gates, append the good alignments to the batch.  Files in argv order. */
int pp_ingest_sam(pp_ingest *g, const char *path, pp_sam_counts *counts, char *err, size_t errlen);
/* Same, with the filter's verdicts handed over in memory instead of as "ZP:Z:fail" tags in a rewritten
 * file (fused filter -> polish): pass[i] == 0 fails the i-th ALIGNED record of the file (file order, the
 * numbering of pp_filter_file), exactly as if the tag had been appended to its line (src/alignment.rs:72-74).
 * n_pass must equal the number of aligned records. */
int pp_ingest_sam_filtered(pp_ingest *g, const char *path, const uint8_t *pass, uint64_t n_pass,
                           pp_sam_counts *counts, char *err, size_t errlen);
void pp_ingest_batch(const pp_ingest *g, pp_aln_batch *out); /* borrowed view, host memory */
/* QNAME of batch record i (for error messages). */
const char *pp_ingest_read_name(const pp_ingest *g, uint64_t i);
void pp_ingest_free(pp_ingest *g);

/* ---- whole-command drivers (what bin/polypolish calls) --------------------------------------- */
typedef struct {
    double fraction_invalid; /* default 0.2  */
    double fraction_valid;   /* default 0.5  */
    uint32_t max_errors;     /* default 10   */
    uint32_t min_depth;      /* default 5    */
    int careful;
    const char *debug_path;  /* NULL = no --debug file */
    int quiet;               /* suppress the stderr log */
} pp_polish_options;

typedef struct {
    uint8_t *data; /* malloc'd by the library, release with pp_bytes_free */
    uint64_t len;
} pp_bytes;
void pp_bytes_free(pp_bytes *b);

/* polish::polish (src/polish.rs:26-38): FASTA text exactly as the reference prints to stdout. */
int pp_polish_files(pp_ctx *ctx, const char *assembly, const char *const *sams, int n_sams,
                    const pp_polish_options *opt, pp_bytes *fasta);

/* polish::polish on SEVERAL GPUs from one process: the host ingest runs once, every context (one per device, see
 * pp_ctx_create) gets the full batches and the emit ranges of its units (pp_shard_plan_create), the contexts polish
 * side by side and the FASTA is assembled on the host.  Same bytes as pp_polish_files; --debug is not available.
 * The error text is on ctxs[0]. */
int pp_polish_files_multi(pp_ctx *const *ctxs, int n_ctx, const char *assembly, const char *const *sams, int n_sams,
                          const pp_polish_options *opt, pp_bytes *fasta);

/* The same ingest on the DEVICE: the raw SAM text is uploaded and tokenized by kernels
 * (newline index, field split, number / CIGAR / tag validation, contig lookup, read groups, gates, 1/k,
 * "*" fill, upper-casing, CIGAR packing).  The batch is bit-identical to pp_ingest_sam's and lives in HBM:
 * hand it to pp_polish_add with PP_MEM_DEVICE and keep the object alive until pp_polish_finish.  Errors
 * (same codes and messages as the host ingest) are reported through pp_last_error(ctx). */
typedef struct pp_dev_ingest pp_dev_ingest;
int pp_dev_ingest_create(pp_ctx *ctx, const pp_assembly *a, uint32_t max_errors, int careful, pp_dev_ingest **out);
int pp_dev_ingest_sam(pp_dev_ingest *g, const char *path, pp_sam_counts *counts);
/* with the filter's verdicts, as pp_ingest_sam_filtered (pass: HOST array over the file's aligned records) */
int pp_dev_ingest_sam_filtered(pp_dev_ingest *g, const char *path, const uint8_t *pass, uint64_t n_pass,
                               pp_sam_counts *counts);
/* How the batch's SEQ bytes are laid out (seq_off may point anywhere, so this is the producer's choice, not a format):
 * PP_SEQ_WINDOW_GROUPED (the default since round 4) = the reads that start in one 2048-position window of the assembly are
 * adjacent (per SAM file; inside a window in no particular order) -- the pileup kernel then fetches a window's reads from
 * one stretch of memory instead of all over the array; PP_SEQ_FILE_ORDER = in the order of the records.  The window layout
 * costs the tokenizer a count / scan / placement pass over eight bytes per record (no second look at the text or the parse
 * records).  Every other array, the order of the records and every result are the same.  Set before the first
 * pp_dev_ingest_sam* / pp_ingest_sam*; `PP_SEQ_LAYOUT=file` in the environment selects file order for the file drivers.
 * Both ingests write the same layout (the host ingest in file order inside a window); the device tokenizer's batch also
 * brings the 4-bit mirror of its seq array (pp_aln_batch.seq4; `PP_SEQ4=0`: none). */
#define PP_SEQ_FILE_ORDER 0
#define PP_SEQ_WINDOW_GROUPED 1
int pp_dev_ingest_set_seq_layout(pp_dev_ingest *g, int layout);
int pp_ingest_set_seq_layout(pp_ingest *g, int layout);
/* Optional hint, before the first file: the bytes of SAM text of ALL the files that will be handed to this ingest.  The
 * batch's arrays are then sized once, for all of them, off the first file (without it the second file makes every array
 * grow: a device allocation and a copy of what the first one filled -- 6 ms on a 2 x 1.2 GB pair). */
int pp_dev_ingest_expect(pp_dev_ingest *g, uint64_t total_text_bytes);
void pp_dev_ingest_batch(const pp_dev_ingest *g, pp_aln_batch *out); /* borrowed view, DEVICE memory */
void pp_dev_ingest_free(pp_dev_ingest *g);

typedef struct {
    uint64_t before_count, after_count;
    uint32_t low_threshold, high_threshold;
    int orientation; /* 0 fr 1 rf 2 ff 3 rr */
    uint64_t orientation_counts[4];
} pp_filter_report;
/* filter::filter (src/filter.rs:26-37). */
int pp_filter_files(pp_ctx *ctx, const char *in1, const char *in2, const char *out1,
                    const char *out2, const char *orientation, double low, double high, int quiet,
                    pp_filter_report *report);

/* filter + polish in one process: the filter's verdicts go straight into the polish ingest,
 * the tagged SAMs are written only if out1/out2 are given (both or neither).  The FASTA is byte-identical
 * to `filter` followed by `polish` on its outputs. */
int pp_filter_polish_files(pp_ctx *ctx, const char *assembly, const char *in1, const char *in2,
                           const char *out1, const char *out2, const char *orientation, double low, double high,
                           const pp_polish_options *opt, pp_filter_report *report, pp_bytes *fasta);

#ifdef __cplusplus
}
#endif
#endif
