// pp_k_direct.h -- the DIRECT path (round 5): k_tile takes its bulk straight from the window-order mirror of the records.
// Part of pp_kernels.hip (included there, in this order, and nowhere else: it defines __global__ kernels).
//
// With a mirror whose runs are known (pp_aln_batch.wo_run_end: one run per SAM file, window-grouped in ascending window
// order) the bucketing of rounds 1-4 -- histogram, column scan, scatter of one 16-byte work item per (record, window),
// 0.195 of the 5 Mbp / 200x job's 0.469 ms and 0.8 GB of its 1.5 GB of traffic -- has nothing left to do for the records
// that are ONE short M run inside their contig (98 % of a short-read job): the entries of window w ARE the stretch
// [first[r][w], first[r][w + 1]) of every run r, and such an entry holds everything its work item would.
//   k_prepd    ONE streaming pass over the mirror (32 B per record): validates every record exactly as k_prep / k_fill did
//              (first offending record in file order), finds first[r][w] where the home window changes between neighbours,
//              and cuts 16-byte work items ("extras", the format of k_fill) only for what k_tile cannot take from the
//              mirror: the part of a bulk read that reaches into the next window (7 % of the reads at 150 bases), and
//              every piece of the other records (indels, long reads, overhangs -- prep_general as before).  Extras go to a
//              fixed room of `xcap` items per window (staged in LDS; one returning atomic per block and window).
//   k_winplan  per window: items = mirror entries + extras -> the heavy-window list, the depth limit, the job's item count
//   k_tile     (pp_k_tile.h, DIRECT) the window's mirror entries through the plain class, then its extras like any items
//   k_xmat     the few windows with positions left for the exact replays get their items written out (k_exact / k_exact2
//              read items, as before)
// A rank of a sharded job (pp_polish_set_emit) takes this path as well, over the job's own coordinates: k_prepd / k_prepg /
// k_winplan see all windows (a few words each), k_tile works on the rank's.
// An entry order that is not what the run table promises (any permutation is a valid mirror) makes k_prepd raise
// DE_MIRROR_ORDER: every later kernel returns at once and the host runs the job over the bucketing path.
#pragma once

namespace pp {

constexpr u32 NOHOME = 0xFFFFFFFFu;
// the window a record starts in -- the producers' rule (pp_ingest.cpp window_of, pp_tokenize.hip k_tok_meta): clamped
__device__ __forceinline__ u32 wo_home(u64 c_lo, u32 ref_start, u32 nwin) {
    return (u32)min((c_lo + ref_start) / (u64)TILE, (u64)(nwin - 1u));
}
// "bulk": one M run over the whole read, 1..FAST_MAX_LEN bases, inside its contig (clen = the contig's length).  k_prepd
// and k_tile must agree on it: a bulk record is tallied from the mirror, every other record through extras.
__device__ __forceinline__ bool wo_bulk(bool contig_ok, u32 ref_start, u32 seq_len, u32 op0, u64 clen) {
    return contig_ok && op0 == ((seq_len << 4) | (u32)PP_OP_M) && seq_len > 0 && seq_len <= FAST_MAX_LEN && (u64)ref_start + seq_len <= clen;
}
// the work item of a bulk record (or of a piece of it) in window w: k_fill's format, see there
__device__ __forceinline__ uint4 wo_item(u64 so, u32 len, u32 kc, u64 g, u32 w, u32 file_idx) {
    uint4 e;
    e.x = (u32)so;
    e.y = ((u32)(so >> 32) & 0xFFu) | (kc << 8) | (len << 24);
    e.z = (u32)(int)((long long)g - (long long)w * TILE) & 0x3FFFFFFFu;
    e.w = file_idx;
    return e;
}

// ---------------------------------------------------------------------------------------------------------------------
// k_prepd / k_prepg
// ---------------------------------------------------------------------------------------------------------------------
// A workgroup's extras are STAGED in LDS and get their slots in the windows' rooms at the end, one returning global atomic
// per window of the workgroup instead of one per wave, window and trip through the loop (a wave waited out thirteen of
// those round trips, one after the other: k_prepd 0.16 ms where k_prep + k_fill took 0.19).  A workgroup's entries lie in
// a handful of consecutive windows (the mirror is in window order): XLOCAL counters from the window of its first entry
// on; an extra for a window outside that range (a workgroup across the end of a run, a long read), or one more than the
// stage holds, takes its slot from the global counter on the spot.
constexpr u32 XLOCAL = 64, CTG_LDS = 1024;
// A contig's offset out of the table in LDS (up to CTG_LDS contigs) or out of memory.  As `in_lds ? lds[i] : mem[i]` the compiler
// made ONE flat load of a selected address, and a flat load is waited for with vmcnt(0): every lookup drained all the loads the
// wave had in flight (k_prepd's next entries).  A branch per kind, each waiting for its own load (the empty asm uses the value
// inside the branch): the usual case then waits for LDS alone.
struct CtgTab {
    const u64 *lds, *mem;
    bool in_lds;  // (uniform)
    __device__ __forceinline__ u64 operator()(u32 i) const {
        u32 x, y;
        if (in_lds) {
            const u64 v = lds[i];
            x = (u32)v; y = (u32)(v >> 32);
            asm volatile("" : "+v"(x), "+v"(y));
        } else {
            const u64 v = mem[i];
            x = (u32)v; y = (u32)(v >> 32);
            asm volatile("" : "+v"(x), "+v"(y));
        }
        return (u64)x | ((u64)y << 32);
    }
    // offsets i and i + 1 (one wait)
    __device__ __forceinline__ void pair(u32 i, u64 &a, u64 &b) const {
        u32 x, y, z, w;
        if (in_lds) {
            const u64 v = lds[i], v1 = lds[i + 1u];
            x = (u32)v; y = (u32)(v >> 32); z = (u32)v1; w = (u32)(v1 >> 32);
            asm volatile("" : "+v"(x), "+v"(y), "+v"(z), "+v"(w));
        } else {
            const u64 v = mem[i], v1 = mem[i + 1u];
            x = (u32)v; y = (u32)(v >> 32); z = (u32)v1; w = (u32)(v1 >> 32);
            asm volatile("" : "+v"(x), "+v"(y), "+v"(z), "+v"(w));
        }
        a = (u64)x | ((u64)y << 32);
        b = (u64)z | ((u64)w << 32);
    }
};
struct XSink {
    uint4 *st_item;  // LDS: [xstage] staged items ...
    u32 *st_key;     //      ... their (local window << 16 | rank among the workgroup's extras of that window)
    u32 *l_cnt, *l_base, *l_nb;  // LDS [XLOCAL]: extras / their first slot / entries that are not bulk, per local window
    u32 *n_st;
    u32 xstage, wbase, xcap;
    u32 *x_cnt, *x_nb;
    uint4 *xent;
    u64 *x_need, *status;
    // one extra into its slot of its window's room, or a capacity overflow (the host gives the windows more room and reruns)
    __device__ __forceinline__ void put(u32 w, u32 slot, const uint4 &e) const {
        if (slot < xcap) xent[(u64)w * xcap + slot] = e;
        else {
            atomicMax(x_need, (u64)slot + 1ull);
            report(status, slot, DE_CAPACITY);
        }
    }
    // one extra of window w: staged, or straight to its window
    __device__ __forceinline__ void emit(u32 w, const uint4 &e) const {
        const u32 wl = w - wbase;
        if (wl < XLOCAL) {
            const u32 pos = atomicAdd(n_st, 1u);
            if (pos < xstage) {
                st_key[pos] = (wl << 16) | atomicAdd(&l_cnt[wl], 1u);
                st_item[pos] = e;
                return;
            }
        }
        put(w, atomicAdd(&x_cnt[w], 1u), e);
    }
    // an entry of window h that is not bulk (k_tile passes it over: k_winplan takes it off the window's count)
    __device__ __forceinline__ void not_bulk(u32 h) const {
        if (h - wbase < XLOCAL) atomicAdd(&l_nb[h - wbase], 1u); else atomicAdd(&x_nb[h], 1u);
    }
    __device__ __forceinline__ void clear() const {  // (before a barrier)
        if (threadIdx.x < XLOCAL) { l_cnt[threadIdx.x] = 0; l_nb[threadIdx.x] = 0; }
        if (threadIdx.x == 0) *n_st = 0;
    }
    // every thread of the workgroup, when all extras are in: a stretch of slots per window, then every item to its slot
    __device__ __forceinline__ void flush() const {
        __syncthreads();
        if (threadIdx.x < XLOCAL) {
            const u32 c = l_cnt[threadIdx.x], nb = l_nb[threadIdx.x];
            l_base[threadIdx.x] = c ? atomicAdd(&x_cnt[wbase + threadIdx.x], c) : 0u;  // (c > 0: a window of the assembly)
            if (nb) atomicAdd(&x_nb[wbase + threadIdx.x], nb);
        }
        __syncthreads();
        for (u32 i = threadIdx.x; i < min(*n_st, xstage); i += blockDim.x) {
            const u32 key = st_key[i], wl = key >> 16;
            put(wbase + wl, l_base[wl] + (key & 0xFFFFu), st_item[i]);
        }
    }
};

struct PrepdArgs {
    u64 n;
    const pp_wo_rec *wo;
    const u64 *cig_off;
    const u32 *n_cig, *cigar;
    const u8 *seq;
    const u64 *contig_off;
    u32 n_contigs, nwin;
    const u32 *run_end;  // ends of the mirror's runs (ascending, the last one = n)
    u32 n_runs;
    u32 *first, *x_cnt, *x_nb;
    uint4 *xent;
    u32 xcap;
    u32 *maxlen;
    u64 *x_need;
    uint4 *g_later;      // the entries that are not bulk, for k_prepg: copies of them, two 16-byte words each (file index NOIDX: none) ...
    u64 *g_nlater;       // ... how many (counted past the capacity too) ...
    u64 cap_later;       // ... and the room
    u64 seq_bytes;       // bytes of the seq array: a record's SEQ has to lie inside it
    u64 *status;
};

// A record that is NOT bulk: prep_general's verdict (CIGAR validity, spans, the trim of the slow classes), its pieces window
// by window as extras, and one more entry of its home window that k_tile will pass over.  One lane per record.
// Two round trips: (1) the run count and where the runs are, and the read's last eight bytes (the trims look at the read
// from its end and rarely further); (2) the first eight runs.  (A record of one run has it in its mirror entry.)  Anything
// beyond is read where it is needed, as in k_prep.
// RUNS_IN_REGS = false (k_prepd's own fallback, a workgroup whose noted records do not fit the lists): the runs are read where
// they are needed -- the eight-run cache is an indexed array, and in k_prepd's loop it went to scratch memory (48 bytes a
// lane reserved for every wave of the streaming kernel).
template <bool RUNS_IN_REGS = true, typename CTG>
__device__ __forceinline__ void general_record(const pp_wo_rec &r, const PrepdArgs &P, CTG ctg, const XSink &X) {
    u32 g_out = 0, nk_out = 0;
    u8 fl_out = 0;
    const u32 fi = r.file_idx;
    if (fi >= P.n) return;  // (reported by k_prepd's loop: the arrays cannot be read for it)
    if (r.contig >= P.n_contigs) { report(P.status, fi, DE_BAD_CONTIG); return; }
    const u64 c_lo = ctg(r.contig), c_hi = ctg(r.contig + 1u);
    X.not_bulk(wo_home(c_lo, r.ref_start, P.nwin));
    {
        const bool multi = r.op0 == PP_WO_MULTI_RUN;
        const u32 sl = r.seq_len;
        const u8 *const sq = P.seq + r.seq_off;
        u64 tail8 = 0;
        if (sl >= 8u) __builtin_memcpy(&tail8, sq + (sl - 8u), 8);
        u32 nc = 1;
        u64 co = 0;
        if (multi) { nc = P.n_cig[fi]; co = P.cig_off[fi]; }
#ifdef PP_PREP_STAMPS
        if constexpr (RUNS_IN_REGS) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); PP_STAMP(1, 5); }
#endif
        if (nc == 0) report(P.status, fi, DE_BAD_RUN);
        else {
            const u32 *const cg = P.cigar + co;
            // (eight scalars, not an array: indexed, the array went to scratch memory -- a memory round trip per run in the
            // kernel whose time is its chain of round trips)
            u32 r0 = r.op0, r1 = 0, r2 = 0, r3 = 0, r4 = 0, r5 = 0, r6 = 0, r7 = 0;
            if (RUNS_IN_REGS && multi) {
                r0 = cg[0];
                r1 = cg[min(1u, nc - 1u)]; r2 = cg[min(2u, nc - 1u)]; r3 = cg[min(3u, nc - 1u)]; r4 = cg[min(4u, nc - 1u)];
                r5 = cg[min(5u, nc - 1u)]; r6 = cg[min(6u, nc - 1u)]; r7 = cg[min(7u, nc - 1u)];
            }
#ifdef PP_PREP_STAMPS
            if constexpr (RUNS_IN_REGS) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); PP_STAMP(1, 6); }
#endif
            prep_general_t(fi, r.ref_start, sl, nc,
                           [&](u32 i) -> u32 {
                               if (!RUNS_IN_REGS) return multi ? cg[i] : r.op0;
                               if (i >= 8u) return cg[i];
                               const u32 a = (i & 1u) ? r1 : r0, b = (i & 1u) ? r3 : r2, c = (i & 1u) ? r5 : r4, d = (i & 1u) ? r7 : r6;
                               const u32 ab = (i & 2u) ? b : a, cd = (i & 2u) ? d : c;
                               return (i & 4u) ? cd : ab;
                           },
                           [&](u32 i) -> u8 { return sl >= 8u && i + 8u >= sl ? (u8)(tail8 >> (8u * (i + 8u - sl))) : sq[i]; },
                           c_lo, c_hi - c_lo, &g_out, &nk_out, &fl_out, P.status);
        }
    }
    const u32 word = nk_out | ((u32)fl_out << 30);
#ifdef PP_PREP_STAMPS
    if constexpr (RUNS_IN_REGS) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); PP_STAMP(1, 7); }
#endif
    if (!word) return;
    const u32 kc = kclass_of(r.k);
    const u32 cls = word >> 30, ia = (word >> 9) & 0xFFu, idel = (word >> 17) & 1u;
    for_each_piece(g_out, word, [&](u32 piece, u32 g, u32 sp) {
        if (!sp) return;
        const u32 wa = g / (u32)TILE, wb = min((g + sp - 1u) / (u32)TILE, P.nwin - 1u);
        u64 pso = r.seq_off;
        u32 len = sp, fl = 0, zf = 0;
        if (cls == NKW_INDEL1) {
            if (piece == 0u) zf = 1u;
            else if (piece == 1u) { zf = 2u; pso += ia - 1u + idel; len = idel ? 0u : 2u; }
            else pso += idel ? ia : ia + 1u;
        } else fl = cls;
        for (u32 w = wa; w <= wb && w >= wa; w++) {
            uint4 e;
            e.x = fl ? sp : (u32)pso;
            e.y = (fl ? 0u : (((u32)(pso >> 32) & 0xFFu) | (len << 24))) | (kc << 8) | (fl << 16);
            e.z = ((u32)(int)((long long)g - (long long)w * TILE) & 0x3FFFFFFFu) | (zf << 30);
            e.w = fi;
            X.emit(w, e);
        }
    });
}

constexpr u32 NOIDX = 0xFFFFFFFFu;
// PP_PREPD_TAIL (the default since the end of round 6): k_prepd works its noted records off ITSELF, one lane each out of the list in
// LDS, behind its loop -- no list in memory, no k_prepg.  Round 5 had moved them out ("their chain of round trips kept the chip from
// streaming"): then a workgroup in its chain held a slot that a waiting workgroup would have streamed in.  With 1.5 rounds of
// workgroups or more (768 for the 5 Mbp job, 8,700 entries each) the chains of one round run under the streaming of the next, and
// a latency-bound kernel of 27 us (k_prepg: 130 k records, a chain of five round trips each) is gone: prep 0.072 -> 0.069 on
// configs[1], 0.30 -> 0.26 on configs[3], 0.74 -> 0.62-0.68 on configs[4] (`profiles/r6zz_prepd_tail_ab.txt`).  With ONE round
// (512 workgroups) it loses (0.079), and the 512-thread instance is erratic (0.066-0.15 from one workgroup count to the next).
// -DPP_PREPD_TAIL=0: the two kernels.
#ifndef PP_PREPD_TAIL
#define PP_PREPD_TAIL 1
#endif

// k_prepd: the streaming pass.  The records that are not bulk are only NOTED -- first in LDS, then, one stretch per
// workgroup, in a list in memory that k_prepg works off with a lane per record: here, between the loop and the end of a
// workgroup, their chain of round trips kept the chip from streaming (a workgroup: 21 us of loop, 15 us of waiting).
// A workgroup whose records do not fit the lists (a job of reads with indels throughout) handles them itself.
template <int THREADS>
__global__ __launch_bounds__(THREADS, PP_PREP_WAVES) void k_prepd(u64 chunk, PrepdArgs P) {
    constexpr u32 LATER_MAX = THREADS >= 1024 ? 768 : 320, XSTAGE = 2 * THREADS;  // (73 KB / 40 KB of LDS in all: two / four workgroups per CU)
    __shared__ uint4 later[2 * LATER_MAX];  // the noted entries themselves (k_prepg then starts from the entry, not from its index)
    __shared__ u32 n_later, s_later_at, s_unit;
    __shared__ uint4 st_item[XSTAGE];
    __shared__ u32 st_key[XSTAGE], l_cnt[XLOCAL], l_base[XLOCAL], l_nb[XLOCAL], n_st, s_wbase;
    __shared__ u64 s_ctg[CTG_LDS + 1];  // the contig table when it has up to CTG_LDS contigs: a record's two offsets are then LDS reads, not a dependent trip to memory
    PP_STAMP(0, 0);
    const u64 n = P.n;
    const pp_wo_rec *const wo = P.wo;
    const u32 n_contigs = P.n_contigs, nwin = P.nwin, n_runs = P.n_runs;
    // (the ends of the runs out of LDS: read from memory inside the loop -- every trip of a wave that is not in the last run --
    // each was a vector load waited for with vmcnt(0), which drains the entries asked for ahead)
    __shared__ u32 s_run_end[PP_WO_MAX_RUNS];
    if (threadIdx.x < P.n_runs) s_run_end[threadIdx.x] = P.run_end[threadIdx.x];  // (in front of the barrier below)
    const u32 *const run_end = s_run_end;
    u64 *const status = P.status;
    const u64 lo = (u64)blockIdx.x * chunk, hi = min(n, lo + chunk);
    const bool ctg_lds = n_contigs <= CTG_LDS;
    if (ctg_lds)
        for (u32 i = threadIdx.x; i <= n_contigs; i += blockDim.x) s_ctg[i] = P.contig_off[i];
    if (threadIdx.x == 0) {
        n_later = 0;
        s_unit = blockDim.x >> 6;  // (the next unit of the loop to be taken: every wave starts with the one of its own number)
        u32 wb = 0;
        if (lo < hi) {
            const pp_wo_rec r0 = wo[lo];
            if (r0.contig < n_contigs) wb = wo_home(P.contig_off[r0.contig], r0.ref_start, nwin);
        }
        s_wbase = wb;
    }
    XSink X{st_item, st_key, l_cnt, l_base, l_nb, &n_st, XSTAGE, 0u, P.xcap, P.x_cnt, P.x_nb, P.xent, P.x_need, status};
    X.clear();
    __syncthreads();
    PP_STAMP(0, 1);
    X.wbase = s_wbase;
    const CtgTab ctg{s_ctg, P.contig_off, ctg_lds};
    const u32 lane = threadIdx.x & 63u;
    const u32 stride = nwin + 1u;
    u32 *const first = P.first;
    // first[run][from .. to] = val; short ranges by the lane itself (one entry when a window begins: the usual case), long
    // ones (windows without a record: an uncovered contig) by the whole wave -- every lane of the wave calls this
    auto fill = [&](bool want, u32 run, u32 from, u32 to, u32 val) {
        if (want && to - from < 4u) {
            for (u32 w = from; w <= to; w++) first[(u64)run * stride + w] = val;
            want = false;
        }
        u64 m = __ballot(want);
        while (m) {
            const int lead = __ffsll((long long)m) - 1;
            m &= m - 1;
            const u32 rr = (u32)__builtin_amdgcn_readlane((int)run, lead), ff = (u32)__builtin_amdgcn_readlane((int)from, lead),
                      tt = (u32)__builtin_amdgcn_readlane((int)to, lead), vv = (u32)__builtin_amdgcn_readlane((int)val, lead);
            for (u64 w = (u64)ff + lane; w <= tt; w += 64) first[(u64)rr * stride + w] = vv;
        }
    };
    // The loop is what the kernel's time consists of, and with eight waves to a SIMD its instruction count more than its
    // 32 bytes per record (its first version: ~300 instructions per record, 59 us per block for 213 MB): everything that
    // is rare -- a window begins, a run ends, an entry out of order, a record the checks refuse -- sits behind ONE ballot; the
    // window of the entry in front comes from the neighbouring lane (the wave's first lane: one look at the mirror).
    u32 runw = 0, runw_lo = 0;  // the run of the current wave's first entry (wave-uniform, only ever moves on)
    u32 fast_len = 0;
    constexpr int WU = PP_WO_UNROLL;
    const uint4 *wq = (const uint4 *)wo;
    const u64 trip = (u64)WU * blockDim.x;
    const u64 span = (hi - lo + trip - 1) / trip * trip;  // whole waves and whole trips: the ballots below need every lane
#ifdef PP_EXP_PREPD_EXTRA_STREAM
    // (experiment build: one more pass over the block's entries that only loads them -- what the streaming alone costs)
    {
        u32 acc = 0;
        for (u64 a0 = lo + threadIdx.x; a0 < lo + span; a0 += trip) {
            uint4 xa[WU], xb[WU];
#pragma unroll
            for (int u = 0; u < WU; u++) {
                const u64 a = min(a0 + (u64)u * blockDim.x, n - 1);
                xa[u] = wq[2 * a];
                xb[u] = wq[2 * a + 1];
            }
#pragma unroll
            for (int u = 0; u < WU; u++) acc ^= xa[u].x ^ xa[u].y ^ xa[u].z ^ xa[u].w ^ xb[u].x ^ xb[u].y ^ xb[u].z ^ xb[u].w;
        }
        if (acc == 0x12345679u && n == 1) first[0] = acc;
    }
#endif
    // The loads run AHEAD of the work, in the registers the work has just left (PP_PREPD_ROLL, the default): a trip's WU sets of
    // loads are asked for one by one, set u of the NEXT trip as soon as set u of this one has been worked off -- while a wave
    // computes, one set is always on its way.  Asked for all at the top of a trip and waited for together, nothing of the wave's
    // was in flight while it computed: 23 % of the issue slots used, 61 % of the wave-cycles waiting, 213 MB in 42 us where a
    // loop that only loads them takes 27 (profiles/r6y_*).
#ifndef PP_PREPD_DYN
#define PP_PREPD_DYN 1
#endif
    const u32 ustride = PP_PREPD_DYN ? 64u : blockDim.x;  // between the WU sets of entries a wave works on in one trip
    uint4 qa[WU], qb[WU];
    u32 pc[WU], pr[WU];
    // (`real` = false, behind the block's last trip: the block's first entry once more, one line for the whole wave.  The loads
    // are UNCONDITIONAL: with a branch around them the compiler has to wait for the loads of the path that did not take it --
    // the set asked for ahead included, vmcnt(0) where vmcnt(3) would do)
    auto ask = [&](int u, u64 a0, bool real) __attribute__((always_inline)) {
        const u64 a = real ? min(a0 + (u64)u * ustride, n - 1) : lo;  // clamped
        qa[u] = wq[2 * a];
        qb[u] = wq[2 * a + 1];
        // (contig, ref_start) of the entry in front of the wave's first one
        const u64 wf = a0 - lane + (u64)u * ustride;  // (the same for the whole wave)
        const u64 pf = real ? min(wf ? wf - 1 : 0, n - 1) : lo;
        pc[u] = wo[pf].contig;
        pr[u] = wo[pf].ref_start;
    };
#ifndef PP_PREPD_ROLL
#define PP_PREPD_ROLL 1
#endif
    // The waves of a workgroup take their trips from a counter in LDS (PP_PREPD_DYN, the default), not every wave the same
    // number: with equal shares thread 0's wave was done 5 us before the workgroup's last one (of 27 us of loop: the waves'
    // luck with the memory system), and the kernel ends with its slowest workgroup.  Unit j = the 64 WU entries from
    // lo + 64 WU j on: a wave's entries come in ascending order, whatever units it gets (what the run tracking below counts on).
    const u32 n_units = (u32)(span / (64u * WU));
    auto unit_a0 = [&](u32 j) -> u64 { return lo + (u64)j * (64u * WU) + lane; };
    u32 unit = (u32)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // (the first one: the wave's own number)
    if (PP_PREPD_ROLL) {
#pragma unroll
        for (int u = 0; u < WU; u++) ask(u, PP_PREPD_DYN ? unit_a0(unit) : lo + threadIdx.x, true);  // (lo < n, or the grid would be smaller)
    }
    for (u64 a0 = lo + threadIdx.x; PP_PREPD_DYN ? unit < n_units : a0 < lo + span; a0 += trip) {
        u32 unit_next = 0;
        if (PP_PREPD_DYN) {
            a0 = unit_a0(unit);
            u32 t = 0;
            if (lane == 0) t = atomicAdd(&s_unit, 1u);
            unit_next = (u32)__builtin_amdgcn_readfirstlane((int)t);
        }
        if (!PP_PREPD_ROLL) {
#pragma unroll
            for (int u = 0; u < WU; u++) ask(u, a0, true);
        }
        const bool more = PP_PREPD_DYN ? unit_next < n_units : a0 + trip < lo + span;  // (uniform)
        const u64 a0_next = PP_PREPD_DYN ? (more ? unit_a0(unit_next) : lo) : a0 + trip;
#pragma unroll
        for (int u = 0; u < WU; u++) {
            const u64 a = a0 + (u64)u * ustride;
            const bool in = a < hi;
            const u32 contig = qa[u].x, ref_start = qa[u].y, k = qa[u].z, seq_len = qa[u].w, op0 = qb[u].z, file_idx = qb[u].w;
            const u64 seq_off = (u64)qb[u].x | ((u64)qb[u].y << 32);
            const bool c_ok = contig < n_contigs;
            const u32 cc = min(contig, n_contigs - 1u);
            u64 c_lo, c_hi;
            ctg.pair(cc, c_lo, c_hi);
            const bool bulk = in && wo_bulk(c_ok, ref_start, seq_len, op0, c_hi - c_lo);
            const u64 g = c_lo + ref_start;
            const u32 h = in && c_ok ? (u32)min(g / (u64)TILE, (u64)(nwin - 1u)) : NOHOME;
            // the window of the entry in front
            u32 hp = (u32)__shfl_up((int)h, 1, 64);
            if (lane == 0) hp = pc[u] < n_contigs ? wo_home(ctg(pc[u]), pr[u], nwin) : NOHOME;
            // the wave's run (uniform); a lane behind its end (a wave across the end of a run) finds its own
            const u32 wf32 = (u32)(a - lane);
            while (runw + 1u < n_runs && wf32 >= run_end[runw]) { runw_lo = run_end[runw]; runw++; }
            u32 run = runw, run_lo = runw_lo, run_hi = run_end[runw];
            if ((u32)a >= run_hi && runw + 1u < n_runs)
                while (run + 1u < n_runs && (u32)a >= run_end[run]) { run_lo = run_end[run]; run++; run_hi = run_end[run]; }
            const bool valid = h != NOHOME, starts = (u32)a == run_lo;
            const bool begins = valid && (starts || (hp != NOHOME && h > hp));
            const bool descent = valid && !starts && hp != NOHOME && h < hp;
            const bool ends = valid && (u32)a + 1u == run_hi;
            const bool bad = in && (file_idx >= n || k == 0 || seq_off + seq_len > (1ull << 40) || seq_off + seq_len > P.seq_bytes);
            if (__ballot(begins || descent || ends || bad)) {
                if (bad) {  // the checks every record gets (k_fill's, and the mirror's own)
                    if (file_idx >= n) report(status, a, DE_BAD_MIRROR);
                    else if (k == 0) report(status, file_idx, DE_BAD_K);
                    else if (seq_off + seq_len > (1ull << 40)) report(status, file_idx, DE_OVERFLOW);
                    else report(status, file_idx, DE_SEQ_RANGE);
                }
                if (descent) report(status, (1ull << 40) - 1ull, DE_MIRROR_ORDER);
                fill(begins, run, starts ? 0u : hp + 1u, h, (u32)a);
                fill(ends, run, h + 1u, nwin, (u32)a + 1u);
            }
            // a bulk read that reaches into the next window: one extra there
            const u32 w1 = bulk ? (u32)((g + seq_len - 1u) / (u64)TILE) : 0u;
#ifndef PP_EXP_PREPD_NOEMIT
            if (bulk && w1 > h) X.emit(w1, wo_item(seq_off, seq_len, kclass_of(k), g, w1, file_idx));
#endif
            if (bulk) fast_len = max(fast_len, seq_len);
#ifndef PP_EXP_PREPD_NOLATER
            else if (in) {
                const u32 slot = atomicAdd(&n_later, 1u);
                if (slot < LATER_MAX) { later[2u * slot] = qa[u]; later[2u * slot + 1u] = qb[u]; }
            }
#endif
            if (PP_PREPD_ROLL) ask(u, a0_next, more);
        }
        unit = unit_next;
    }
    PP_STAMP(0, 2);
    __syncthreads();
    PP_STAMP(0, 3);
    // the noted records: a stretch of the list in memory -- or, when there are more of them than either list holds, worked
    // off here, from the block's entries once more (then the stretch, as far as it lies inside the list, is marked empty)
    const u32 n_noted = n_later;
#if PP_PREPD_TAIL
    // the noted records worked off here, one lane each out of the list in LDS (see PP_PREPD_TAIL above)
    const bool listed = n_noted <= LATER_MAX;
    if (listed) {
        for (u32 i = threadIdx.x; i < n_noted; i += blockDim.x) {
            const uint4 qa = later[2u * i], qb = later[2u * i + 1u];
            pp_wo_rec r;
            r.contig = qa.x; r.ref_start = qa.y; r.k = qa.z; r.seq_len = qa.w;
            r.seq_off = (u64)qb.x | ((u64)qb.y << 32); r.op0 = qb.z; r.file_idx = qb.w;
            general_record(r, P, ctg, X);
        }
    }
#else
    if (threadIdx.x == 0) s_later_at = n_noted ? (u32)min(atomicAdd(P.g_nlater, (u64)n_noted), (u64)NOIDX) : 0u;
    __syncthreads();
    const u64 at = s_later_at;
    const bool listed = n_noted <= LATER_MAX && at + n_noted <= P.cap_later;
    for (u32 i = threadIdx.x; i < 2u * n_noted; i += blockDim.x)
        if (at + (i >> 1) < P.cap_later) P.g_later[2ull * at + i] = listed ? later[i] : make_uint4(NOIDX, NOIDX, NOIDX, NOIDX);
#endif
    if (!listed && n_noted) {
        for (u64 a = lo + threadIdx.x; a < hi; a += blockDim.x) {
            const pp_wo_rec r = wo[a];
            const u32 cc = min(r.contig, n_contigs - 1u);
            if (!wo_bulk(r.contig < n_contigs, r.ref_start, r.seq_len, r.op0, ctg(cc + 1u) - ctg(cc))) general_record<PP_PREPD_TAIL != 0>(r, P, ctg, X);  // (runs in registers where the tail above has them anyway: with every run read where it is needed a workgroup of a job with indels in 30 % of its reads took 0.8 ms here)
        }
    }
    PP_STAMP(0, 4);
    X.flush();
    PP_STAMP(0, 5);
    // the job's longest fast-class read (as k_prep)
    if (__ballot(fast_len > PLAIN_NARROW_MAX)) {
        for (int o = 32; o > 0; o >>= 1) fast_len = max(fast_len, (u32)__shfl_xor((int)fast_len, o, 64));
        if (lane == 0 && fast_len > __hip_atomic_load(P.maxlen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(P.maxlen, fast_len);
    }
    PP_STAMP(0, 6);
}

// k_prepg: the records k_prepd noted, one lane each (a few per cent of a short-read job: all of their round trips at once)
// LDS: 13 KB a workgroup -- a stage of two extras a thread (a workgroup has a record for two of three threads and most records
// are three pieces; what does not fit takes its slot on the spot), a contig table of PREPG_CTG_LDS entries.  With four a thread and
// k_prepd's table of 1,024 contigs (29 KB) five workgroups fit a CU, 1,280 the chip, and the 1,628 of a 5 Mbp job started over
// 16 us (block timeline, `profiles/r6zz_prepg_trips_and_rounds.txt`); with all of them resident at once the kernel takes the
// same 27 us (A/B in the same file: the chains of round trips are slower the more of them run at a time) -- kept for the room it
// leaves, not for a gain.
#ifndef PP_PREPG_XSTAGE
#define PP_PREPG_XSTAGE 2
#endif
#ifndef PP_PREPG_CTG
#define PP_PREPG_CTG 256
#endif
constexpr u32 PREPG_CTG_LDS = PP_PREPG_CTG;
template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_prepg(PrepdArgs P) {
    constexpr u32 XSTAGE = PP_PREPG_XSTAGE * THREADS;
    __shared__ uint4 st_item[XSTAGE];
    __shared__ u32 st_key[XSTAGE], l_cnt[XLOCAL], l_base[XLOCAL], l_nb[XLOCAL], n_st;
    __shared__ u64 s_ctg[PREPG_CTG_LDS + 1];
    PP_STAMP(1, 0);
    if (*P.status != ~0ull) return;
    const u64 total = min(*P.g_nlater, P.cap_later);
    const u64 per = (total + gridDim.x - 1) / gridDim.x;
    const u64 i0 = min(total, (u64)blockIdx.x * per), i1 = min(total, i0 + per);
    if (i0 >= i1) return;
    const bool ctg_lds = P.n_contigs <= PREPG_CTG_LDS;
    // One round trip for everything that is known now: the contig table, the stretch's first entry (every thread asks for the
    // same one: the window the local counters start at), this thread's own first entry.  (Thread 0 alone looking at the first
    // entry and then at its contig's offset, and every thread asking for its entry behind the barrier, were three trips.)
    if (ctg_lds)
        for (u32 i = threadIdx.x; i <= P.n_contigs; i += blockDim.x) s_ctg[i] = P.contig_off[i];
    const uint4 q0 = P.g_later[2ull * i0];
    const u64 i_first = i0 + threadIdx.x, i_mine = min(i_first, i1 - 1);
    uint4 qa = P.g_later[2ull * i_mine], qb = P.g_later[2ull * i_mine + 1];
    XSink X{st_item, st_key, l_cnt, l_base, l_nb, &n_st, XSTAGE, 0u, P.xcap, P.x_cnt, P.x_nb, P.xent, P.x_need, P.status};
    X.clear();
    __syncthreads();
    PP_STAMP(1, 1);
    const CtgTab ctg{s_ctg, P.contig_off, ctg_lds};
    X.wbase = q0.x < P.n_contigs ? wo_home(ctg(q0.x), q0.y, P.nwin) : 0u;  // (the same in every thread)
    for (u64 i = i_first; i < i1; i += blockDim.x) {
        if (i != i_first) { qa = P.g_later[2ull * i]; qb = P.g_later[2ull * i + 1]; }
        pp_wo_rec r;
        r.contig = qa.x; r.ref_start = qa.y; r.k = qa.z; r.seq_len = qa.w;
        r.seq_off = (u64)qb.x | ((u64)qb.y << 32); r.op0 = qb.z; r.file_idx = qb.w;
        if (!(qa.x == NOIDX && qb.w == NOIDX)) general_record(r, P, ctg, X);
    }
    PP_STAMP(1, 2);
    __syncthreads();
    PP_STAMP(1, 3);
    X.flush();
    PP_STAMP(1, 4);
}

// ---------------------------------------------------------------------------------------------------------------------
// k_winplan: what a window holds -> heavy-window list, depth limit, the job's item count; the run tables must ascend
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_winplan(u32 nwin, u32 n_runs, const u32 *__restrict__ first, const u32 *__restrict__ x_cnt,
                                                 const u32 *__restrict__ x_nb, u32 xcap, u32 heavy_min, u32 *__restrict__ heavy, u8 *__restrict__ win_heavy,
                                                 u64 *__restrict__ n_items, u64 *status) {
    const u32 w = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = w < nwin && *status == ~0ull;
    u32 cnt = 0;
    bool bad = false;
    if (live) {
        cnt = min(x_cnt[w], xcap);
        for (u32 r = 0; r < n_runs; r++) {
            const u32 a = first[(u64)r * (nwin + 1u) + w], b = first[(u64)r * (nwin + 1u) + w + 1u];
            if (b < a) bad = true; else cnt += b - a;
        }
        // (an entry that is not bulk is passed over by k_tile -- its pieces are among the extras: what is left is what the
        // bucketing would have counted for the window)
        if (!bad) cnt -= min(cnt, x_nb[w]);
        if (bad) report(status, (1ull << 40) - 1ull, DE_MIRROR_ORDER);
        else {
            if (cnt >= MAX_BUCKET) report(status, w, DE_TOO_DEEP);
            note_heavy(w, cnt, heavy_min, heavy, win_heavy);
        }
    } else if (w < nwin) win_heavy[w] = 0;
    const u32 tot = wave_sum(bad ? 0u : cnt);
    if ((threadIdx.x & 63u) == 0 && tot) atomicAdd(n_items, (u64)tot);
}

// ---------------------------------------------------------------------------------------------------------------------
// k_xmat: work items of the windows that the exact replays will read (k_tile listed them: need_win[0 .. *n_need))
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_xmat(const u32 *__restrict__ need_win, const u64 *__restrict__ n_need, u32 nwin,
                                               const pp_wo_rec *__restrict__ wo, u32 n_runs, const u32 *__restrict__ first,
                                               const u32 *__restrict__ x_cnt, const uint4 *__restrict__ xent, u32 xcap,
                                               const u64 *__restrict__ contig_off, u32 n_contigs, uint4 *__restrict__ ent,
                                               u64 cap_ent, u64 *__restrict__ cursor, u32 *__restrict__ win_lo,
                                               u32 *__restrict__ win_hi, u64 *status) {
    __shared__ u64 s_base;
    __shared__ u32 s_cnt;
    if (job_state(status) == 2) return;
    const u32 todo = (u32)min(*n_need, (u64)nwin);
    const uint4 *wq = (const uint4 *)wo;
    for (u32 i = blockIdx.x; i < todo; i += gridDim.x) {
        const u32 w = need_win[i];
        const u32 xn = min(x_cnt[w], xcap);
        u32 ub = xn;
        for (u32 r = 0; r < n_runs; r++) ub += first[(u64)r * (nwin + 1u) + w + 1u] - first[(u64)r * (nwin + 1u) + w];
        __syncthreads();
        if (threadIdx.x == 0) {
            s_base = atomicAdd(cursor, (u64)ub);
            s_cnt = 0;
        }
        __syncthreads();
        const u64 base = s_base;
        if (base + ub > cap_ent || base + ub > 0xFFFFFFFFull) {  // the host grows the buffer (the cursor keeps counting) and reruns
            if (threadIdx.x == 0) {
                report(status, base + ub, DE_CAPACITY_LATE);
                win_lo[w] = 0;
                win_hi[w] = 0;
            }
            continue;
        }
        for (u32 r = 0; r < n_runs; r++) {
            const u32 a0 = first[(u64)r * (nwin + 1u) + w], a1 = first[(u64)r * (nwin + 1u) + w + 1u];
            for (u32 a = a0 + threadIdx.x; a < a1; a += blockDim.x) {
                const uint4 qa = wq[2ull * a], qb = wq[2ull * a + 1];
                const u32 cc = min(qa.x, n_contigs - 1u);
                const u64 c_lo = contig_off[cc], c_hi = contig_off[cc + 1];
                if (!wo_bulk(qa.x < n_contigs, qa.y, qa.w, qb.z, c_hi - c_lo)) continue;  // (its pieces are among the extras)
                const u64 so = (u64)qb.x | ((u64)qb.y << 32);
                ent[base + atomicAdd(&s_cnt, 1u)] = wo_item(so, qa.w, kclass_of(qa.z), c_lo + qa.y, w, qb.w);
            }
        }
        for (u32 j = threadIdx.x; j < xn; j += blockDim.x) ent[base + atomicAdd(&s_cnt, 1u)] = xent[(u64)w * xcap + j];
        __syncthreads();
        if (threadIdx.x == 0) {
            win_lo[w] = (u32)base;
            win_hi[w] = (u32)base + s_cnt;
        }
    }
}

}  // namespace pp
