// pp_internal.h -- shared between the translation units of libpolypolish_hip.so (not installed).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <string>
#include <thread>
#include <vector>

#include "polypolish_hip.h"

#ifndef PP_MIRROR_REGISTRY_DECLARED
#define PP_MIRROR_REGISTRY_DECLARED
// The window-order mirrors the library's own producers hand out (pp_ingest_batch, pp_dev_ingest_batch, pp_shard_split's parts),
// as address ranges keyed by their owner: a mirror inside one of them (a part of one: the multi-GPU driver's slice views) is
// taken as it is, any other mirror is compared with the arrays it mirrors before the kernels read the records through it
// (pp_polish_add, run_pipeline).  Implemented in pp_shard.cpp.
void pp_mirror_register_(const void *owner, const void *p, size_t bytes);
void pp_mirror_forget_(const void *owner);
bool pp_mirror_trusted_(const void *p, size_t bytes);
#endif

namespace pp {

// ---- geometry of the pileup tile kernel ----------------------------------------------------
constexpr int TILE = 2048;           // assembly positions owned by one workgroup ("window")
constexpr int TILE_THREADS = 1024;   // 16 waves; two workgroups per CU (57 KiB LDS each)
constexpr int COARSE_WINDOWS = 8;     // windows per coarse bucket of the two-level multisplit ...
constexpr int COARSE_WINDOWS_BIG = 64;  // ... and beyond COARSE_BIG_FROM windows (measured on 250 Mbp: 2.8 vs 3.1 ms)
constexpr uint32_t COARSE_BIG_FROM = 65536;
constexpr int COUNT_RANGE = 16384;   // windows histogrammed per LDS pass of the bucketing kernels
// Depth in fixed point.  A window tallies the depth shares 1/k of its reads as DEFICITS against 1 in units of 2^-b, b =
// win_fx_bits(items of the window) <= DEPTH_FX_BITS: a share 1/2^j with j <= b is exact in any order (the reference's f64
// sum of such shares is exact too), any other share is rounded to the nearest unit and marks the positions it covers as
// INEXACT -- their depth is then known to within (reads there) * 2^-(b+1), and only the positions where that interval
// leaves one of the vote's three tests open (pileup.rs:70-72,114) are replayed in file order (k_exact2).  b is chosen so
// that a position's deficit cannot pass 2^31 however many of the window's items cover it.  The per-contig depth sums
// (the log's mean read depth) are kept in 2^-DEPTH_FX_BITS units.
constexpr int DEPTH_FX_BITS = 20;
constexpr uint32_t MAX_BUCKET = 1u << 21;  // (alignment, window) items per window on the fast path
constexpr uint32_t SORT_MAX = 16384;      // items per window the ordered-depth replay kernel sorts in LDS (128 KiB)
constexpr uint32_t SORT_SMALL = 10112;    // ... its small instance (80 KiB of LDS: two workgroups per CU)
constexpr uint32_t SORT_BUCKETS = 4096;   // its counting sort: buckets over the window's range of record indices ...
constexpr uint32_t SORT_BUCKET_MAX = 48;  // ... finished by insertion sorts unless one holds more than this (-> bitonic)
// Heavy windows (collapsed repeats, BASELINE configs[2]): a window with several times the average number of items would
// keep ONE workgroup busy several times as long while the chip drains.  Up to HEAVY_SLOTS of them per job are listed
// (k_scan_cols / k_heavy); k_tile splits the items of a listed window over HEAVY_PARTS helper blocks that are dispatched
// first (their partial tallies meet in a global slab, the last arrival adds them up and votes), k_exact2 replays it with one block per
// TILE / HEAVY_SUB positions.  Windows past the end of the list simply take the ordinary path.
constexpr uint32_t HEAVY_SLOTS = 32;  // x HEAVY_PARTS = 256 helper blocks: at most one per CU (see k_tile)
constexpr uint32_t HEAVY_PARTS = 8;
constexpr uint32_t HEAVY_SUB = 8;
constexpr uint32_t HEAVY_MIN_ITEMS = 3072;  // ... and never a window below this many items
constexpr uint32_t HEAVY_WORDS = 1 + 2 * HEAVY_SLOTS;  // u32: count | listed windows | arrival tickets (in the metadata block)

// entry flags (entA.y bits 24..31)
constexpr uint32_t ENT_COMPLEX = 1u;   // CIGAR contains I or D runs: walked run by run, trim done by k_prep
constexpr uint32_t ENT_PRETRIM = 2u;   // no indel, but too long / overhanging for the fast path: trim done by k_prep
// A read with ONE 1-base indel (runs aM 1I bM / aM 1D bM: a read across a planted assembly deletion / insertion, or with a
// sequencing indel) is not walked at all: k_fill cuts it into three work items -- the bases in front of the indel
// (ENT_NOTRIM: a fast-class item whose end is not the read's end), the entry AT the indel (ENT_POINT: the two-byte key of
// an insertion, or the empty slot of a deletion: one tally) and the bases behind it (an ordinary fast-class item, trimmed
// like any read) -- get_read_bases_for_each_target_base's output for such a CIGAR (alignment.rs:175-201), piece by piece.
// These two flags live in bits 30-31 of the item's z word (the window-relative start below them, 30 bits signed).
constexpr uint32_t ENT_NOTRIM = 4u;
constexpr uint32_t ENT_POINT = 8u;
constexpr uint32_t NKW_INDEL1 = 3u;    // class bits (30-31) of the nkeep word: span | a << 9 | is_deletion << 17 below them
// both flanks at least this long, else the general walk.  1 since round 6 (8 = PLAIN_MIN_LEN before): a flank of fewer than
// eight bases is a plain item on the direct path (DirectBulk::loadable) and a fast-class item of the one-at-a-time kind elsewhere;
// as slow items -- a wave each, three dependent round trips -- the reads with their indel within eight bases of an end (a
// tenth of the ~200 reads over every planted indel) were 17 us of k_tile_direct's 181 on configs[1] (profiles/r6y_*)
constexpr uint32_t INDEL1_MIN_SEG = 1;
constexpr uint32_t FAST_MAX_LEN = 252; // a read of <= 252 bases is one dword-per-lane wave load
// depth-share class of a work item (8 bits; kclass_of / k_of_class in pp_k_common.h): 0: k = 1 | 1..20: k = 2^class |
// 21..254: k = class - 18 (3..236, the other small k: all-hits reads in a handful of copies) | 255: any other k (looked up)
constexpr uint32_t KCLASS_DYADIC_MAX = 20u;
constexpr uint32_t KCLASS_SMALL_BASE = 18u, KCLASS_SMALL_MAX_K = 236u;
constexpr uint32_t KCLASS_OTHER = 255u;

// device-side error codes, packed as (record index << 8 | code) and combined with atomicMin so
// that the first offending record in file order wins, as in the reference's streaming loop
enum DevErr : uint32_t {
    DE_UNEXPECTED_OP = 1,   // alignment.rs:188-193 quit
    DE_LEN_MISMATCH = 2,    // alignment.rs:196-198 quit
    DE_OUT_OF_BOUNDS = 3,   // pileup.rs:194-196 panic
    DE_BAD_CONTIG = 4,      // alignment.rs:298-300 quit
    DE_BAD_K = 5,
    DE_BAD_RUN = 6,         // zero-length run / op code > 8 / empty CIGAR
    DE_BAD_ENDS = 7,        // first or last run not M/=  (gate of alignment.rs:155-159 not applied)
    DE_NON_ASCII = 8,
    DE_INTERNAL = 9,
    DE_TOO_DEEP = 10,
    DE_OVERFLOW = 11,
    DE_CAPACITY = 12,       // an optimistic device buffer was too small: the host grows it and reruns
    DE_HALO = 14,           // compact run (run_pipeline): a record reaches an owned stretch from outside its slice -> rerun uncompacted
    DE_CAPACITY_LATE = 13,  // the same, raised from k_tile on: the work items are valid, so k_tile and k_exact2 keep
                            // COUNTING what they would need (guarded writes) and one rerun is enough
    DE_BAD_MIRROR = 15,     // an entry of the window-order mirror names a record the batch does not have (pp_aln_batch.wo)
    DE_MIRROR_ORDER = 16,   // direct path (pp_k_direct.h): the mirror's entries are not in the order its run table promises
                            // -> the host runs the job over the bucketing path (always reported at the largest index, so
                            // that the error of any record wins)
    DE_SEQ_RANGE = 18,      // a record's SEQ bytes do not lie inside the batch's seq array (seq_off + seq_len > seq_bytes)
    DE_GW_HINT = 17,        // k_tile: the instance the host launched does not take the job's longest fast-class read (meta
                            // word 9) -> the host reruns with the one that does (same index as DE_MIRROR_ORDER)
};


struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
};

struct KernelTimer {
    const char *name;
    hipEvent_t start, stop;
};

struct ContigStatsDev {  // per contig, accumulated with atomics
    unsigned long long changed;
    unsigned long long zero_depth;
    unsigned long long depth_fx;  // sum of depth in 2^-DEPTH_FX_BITS units (inexact shares as rounded, see DEPTH_FX_BITS)
};

}  // namespace pp

struct pp_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    std::thread init_thread;  // pp_ctx_create_async: device initialisation in flight
    bool init_pending = false;
    int init_rc = 0;
    int profiling = 0;  // 0 off, 1 every kernel group, 2 the dominant kernel only
    bool timer_open = false;
    std::vector<hipEvent_t> event_pool;
    bool debug = false;
    int debug_level = 0;  // what pp_polish_set_debug was given (3: see run_pipeline)
    std::vector<pp::KernelTimer> timers;
    pp_kernel_times last_times{};

    // ---- polish job ----
    bool job_open = false, job_done = false;
    uint32_t n_contigs = 0;
    std::vector<uint64_t> contig_off;
    uint64_t G = 0;
    pp_params params{};
    const uint8_t *d_bases = nullptr;  // device pointer (owned copy or borrowed)
    pp_aln_batch dbatch{};             // device pointers
    bool have_batch = false;
    bool batch_borrowed = false;       // the one batch so far is caller-owned device memory, used in place
    uint64_t acc_n = 0, acc_seq = 0, acc_cig = 0;  // records / SEQ bytes / CIGAR runs accumulated in b_in[]
    uint64_t total_out = 0, n_multi = 0, n_keys = 0;
    uint64_t last_dev_error = ~0ull;   // (record index << 8 | DevErr) of the last pp_polish_finish that failed on the device
    std::vector<uint64_t> contig_out_off;
    std::vector<pp_contig_stats> stats;

    // owned device buffers (grow-only, reused across jobs)
    pp::DevBuf b_bases, b_contig_off, b_status;
    pp::DevBuf b_in[11];  // the accumulated batch arrays (uploaded or gathered); [9] = the 4-bit mirror of [7], the seq array; [10] = the window-order mirror
    bool acc_wo = false;  // every batch accumulated so far brought a window-order mirror (pp_aln_batch.wo)
    pp::DevBuf b_gstart, b_nkeep, b_aflag, b_hist, b_wincnt, b_winoff, b_entA, b_entB, b_ccnt, b_coff;
    pp::DevBuf b_code, b_winlen, b_winout, b_flag_pos, b_flag_cov, b_flag_scr, b_scratch;
    pp::DevBuf b_vote_tab;  // the job's vote thresholds per integer depth (k_meta_init)
    pp::DevBuf b_multi, b_meta, b_out, b_flag_bits, b_win_nflag, b_win_slab, b_slab_win, b_slabs, b_ents, b_keys, b_own;
    pp::DevBuf b_win_heavy, b_hslab;  // heavy windows: slot + 1 per window (u8) | the helpers' partial tallies
    // ---- the direct path (pp_k_direct.h) ----
    std::vector<uint64_t> wo_runs;      // ends of the runs of the job's window-order mirror (pp_aln_batch.wo_run_end, rebased); empty = not known
    pp::DevBuf b_runs, b_first, b_xcnt, b_xent, b_need_win, b_win_lo, b_win_hi, b_later;
    std::vector<uint32_t> runs_on_dev;  // what b_runs holds (identical tables are not uploaded again)
    size_t xcap = 0;                    // room for extras per window (grow-only)
    bool no_direct = false;             // this job is being rerun over the bucketing path (DE_MIRROR_ORDER)
    bool wo_untrusted = false;          // a batch of this job brought a window-order mirror that is not one of the library's own (pp_mirror_trusted_): checked before it is used
    bool trust_all = false;             // pp_ctx_trust_mirrors_: every mirror is taken as the library's own (bench / tests that lay a batch out as the ingests do)
    bool no_wo = false;                 // this job is being rerun without its mirror (it did not stand the check)
    size_t xcap_limit = ~(size_t)0;     // the most room for extras a window of this job can be given (run_pipeline): beyond it, the bucketing path
    uint32_t maxlen_hint = 0;           // the longest fast-class read of the context's last job (meta word 9): picks k_tile's instance for the next one (DE_GW_HINT)
    bool last_direct = false;           // the last pass over the pipeline took the direct path
    bool nothing_flagged_last = false;  // the job before had no position flagged for the exact replays (run_pipeline: their launches are then left out until this job's metadata say otherwise)
    std::vector<uint32_t> emit;  // pp_polish_set_emit: (lo, hi) per contig, empty = everything
    std::vector<uint32_t> run_full_of;  // compact run (pp_kernels.hip, run_pipeline): the job's contig behind each contig of the run
    uint32_t run_nc = 0;                // contigs of the last run
    uint32_t last_multi = ~0u;          // multi-byte winners of the last job (sizes k_emit's grid for the next one)
    uint32_t last_listed = ~0u;         // positions the last job listed for k_exact (sizes its grid for the next one)
    bool no_compact = false;            // this job is being rerun over the whole assembly (DE_HALO)
    uint64_t *h_meta = nullptr;         // pinned host copy of the job's metadata block (+ 2 words: k_emit's EmitTail)
    pp::DevBuf b_wincoarse;             // output bytes per WIN_COARSE consecutive windows (k_emit's offsets)
    pp::DevBuf b_emit_done;             // k_emit's counters of finished workgroups (EmitTail::done)
    bool emit_done_clean = false;       // ... known to be zero
    uint64_t *d_hmeta = nullptr;        // ... as the device sees it
    bool stream_may_be_busy = false;    // the last pp_polish_finish returned on the polled serial, not on the stream's end
    uint64_t emit_serial = 0;           // launches of k_emit so far (what its last workgroup writes behind the copy)
    // what k_meta_init has already set up, on the stream, for the next job (run_pipeline): valid while nothing else touched it
    struct MetaReady {
        const void *meta; uint32_t words; const void *za, *zb, *zc; uint32_t nwin; const void *tab; double fv, fi; const void *ze;
        bool operator==(const MetaReady &o) const {
            return meta == o.meta && words == o.words && za == o.za && zb == o.zb && zc == o.zc && nwin == o.nwin && tab == o.tab && fv == o.fv && fi == o.fi && ze == o.ze;
        }
    };
    MetaReady meta_ready{};
    bool meta_ready_valid = false;
    size_t h_meta_words = 0;
    std::vector<uint8_t> own_blob;      // what b_own holds (emit ranges, window ranges, compact tables), to skip identical uploads
    pp::DevBuf b_sub_bases;             // ... and its assembly bytes
    size_t cap_ent = 0, cap_scr = 0, cap_multi = 0, cap_out = 0, cap_flag = 0, cap_slabs = 0, cap_ents = 0, cap_keys = 0;  // element capacities of the optimistic buffers
    pp::DevBuf b_dbg_depth, b_dbg_counts, b_dbg_status;

    // ---- multi-GPU gather (pp_comm.hip) ----
    void *comm = nullptr;  // ncclComm_t
    int comm_rank = 0, comm_world = 1;
    pp::DevBuf b_comm, b_gather;  // the gather's size exchange | rank 0's receive buffer (pp_polish_gather_to_host_)
    pp::DevBuf b_split[13];  // pp_shard_split's scratch (pp_shard_dev.hip)

    // ---- filter job ----
    pp::DevBuf f_in[2][9], f_refend[2], f_pass[2], f_orient, f_insert, f_poisoned, f_list, f_blkcnt;
    pp_filter_input fdev{};
    const uint64_t *f_refend_ptr[2] = {nullptr, nullptr};
    bool filter_open = false;
    bool filter_reads_done = false;  // the pass over the reads (k_filter_reads) of the open filter job has run
    int64_t filter_n_listed = -1;    // whether it listed any read for k_filter_listed, once read back (-1: not known on the host)

    int fail(int code, const char *fmt, ...) {
        char buf[1024];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        err = buf;
        return code;
    }
};

namespace pp {

#define PP_HIPCHK(ctx, expr)                                                                   \
    do {                                                                                       \
        hipError_t e__ = (expr);                                                               \
        if (e__ != hipSuccess)                                                                 \
            return (ctx)->fail(PP_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e__));    \
    } while (0)

// grow-only device allocation
int dev_ensure(pp_ctx *ctx, DevBuf &b, size_t bytes);
int dev_grow_keep(pp_ctx *ctx, DevBuf &b, size_t bytes, size_t used);
void dev_free(DevBuf &b);
// start/stop a named kernel timer on the context's stream (no-ops unless profiling)
void timer_begin(pp_ctx *ctx, const char *name);
void timer_end(pp_ctx *ctx);
int timers_collect(pp_ctx *ctx, pp_kernel_times *out);
void timers_release(pp_ctx *ctx);

}  // namespace pp
