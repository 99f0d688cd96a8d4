// pp_k_direct.h -- k_direct: records in window order -> work items, in ONE pass (no histogram, no second read).
// Part of pp_kernels.hip (included there, after pp_k_prep.h and pp_k_bucket.h, and nowhere else: it defines __global__ kernels).
#pragma once

namespace pp {

// =============================================================================================
// k_direct
// =============================================================================================
// With the batch's window-order mirror (pp_aln_batch.wo) the records of one 2048-position window are adjacent, the windows
// in order.  k_prep + k_scan_cols + k_scan + k_fill then do a multisplit of something that is sorted already, and both big
// kernels run at memory speed over the same 32 bytes per record (profiles/r4_final_prep_fill_block_timeline.txt).  Here
// the work items are written where the order of the records puts them:
//   * an item that lies in the window its record starts in (the record's HOME window) goes to ent[A++], A counting the
//     records' items in mirror order: the items of window w are one stretch ent[win_off[w] .. win_off[w + 1]);
//   * an item in the window after it (a read across a window boundary, a piece of a one-indel read behind it) goes to
//     ent[cap - 1 - B++]: what the records of window w spill into w + 1 is the stretch B in [spill_off[w], spill_off[w + 1]).
//   k_tile / k_exact / k_exact2 take a window's items from both (WinItems, pp_k_common.h).
//   * a job of several SAM files brings its records in several such RUNS, one behind the other (every file's windows once):
//     a record whose home window lies before that of the record in front of it starts a run, every run has window tables
//     of its own (win_off / spill_off: DIRECT_MAX_RUNS x (windows + 1) words), and a window's items are the stretches of
//     all runs.
// A and B are exclusive prefix sums over ALL records before one: a chained scan with decoupled look-back over chunks of
// DIRECT_CHUNK records.  The workgroups take the chunks round robin (chunk = blockIdx.x + round * gridDim.x; the grid is
// never larger than what is resident at once: run_pipeline asks the occupancy API), a chunk publishes its own counts as
// soon as it has them and its inclusive prefix once a wave has added up the published words of the chunks before it back
// to the nearest inclusive one.  One 64-bit word per chunk carries flag and both counts, written and read as relaxed
// agent-scope atomics: no fence, nothing else is communicated.  A wait that does not end (the workgroups are not all
// resident after all: a second process on the same GPU) gives up after DIRECT_SPIN_LIMIT polls with DE_DIRECT, and so does
// anything this layout cannot hold -- a record that is not in window order, an item two windows away from its record's
// start (long reads): the host reruns the job through k_prep / k_fill (after a wait that did not end it keeps to them for the context's lifetime).
// Everything k_prep and k_fill check of a record is checked here, with the same codes.
// Not for sharded jobs (pp_polish_set_emit: records are dropped by range there, and a compact run renumbers the windows).
constexpr u32 DIRECT_THREADS = 512;                 // (four waves per SIMD: 128 VGPRs -- the noted records' pass calls prep_general, and at 64 most of what is live around the call was spilt)
constexpr u32 DIRECT_RPT = 4;                       // records per thread and round
constexpr u32 DIRECT_CHUNK = DIRECT_THREADS * DIRECT_RPT;
constexpr u32 DIRECT_LATER_MAX = 1024;              // noted (non-bulk) records of a chunk, done one per lane; more of them: DE_DIRECT
constexpr u32 DIRECT_SPIN_LIMIT = 1u << 18;
constexpr u64 LOOK_AGG = 1ull << 62, LOOK_INC = 2ull << 62, LOOK_VAL = (1ull << 62) - 1ull;  // a look-back word: flag | value
// per record, chunk and prefix: A (items in the home window), B (items in the window after it), R (runs that START here: a
// record whose home window lies before that of the record in front of it -- the next SAM file's records begin)

// the pieces of a record as (window, item) pairs: f(window, item)
template <typename F>
__device__ __forceinline__ void direct_items(u32 g, u32 word, u64 so_rec, u32 kc, u32 fi, u32 nwin, F f) {
    const u32 cls = word >> 30;
    const u32 ia = (word >> 9) & 0xFFu, idel = (word >> 17) & 1u;  // a one-indel read: run length in front, kind
    for_each_piece(g, word, [&](u32 piece, u32 gp, u32 sp) {
        if (!sp) return;
        const u32 w0 = gp / (u32)TILE, w1 = min((gp + sp - 1u) / (u32)TILE, nwin - 1u);
        u64 so = so_rec;
        u32 len = sp, fl = 0, zf = 0;
        if (cls == NKW_INDEL1) {  // (the same words as k_fill's, see there)
            if (piece == 0u) zf = 1u;
            else if (piece == 1u) { zf = 2u; so += ia - 1u + idel; len = idel ? 0u : 2u; }
            else so += idel ? ia : ia + 1u;
        } else fl = cls;
        for (u32 w = w0; w <= w1 && w >= w0; w++) {
            uint4 e;
            e.x = fl ? sp : (u32)so;
            e.y = (fl ? 0u : (((u32)(so >> 32) & 0xFFu) | (len << 24))) | (kc << 8) | (fl << 16);
            e.z = ((u32)(int)((long long)gp - (long long)w * TILE) & 0x3FFFFFFFu) | (zf << 30);
            e.w = fi;
            f(w, e);
        }
    });
}

// look[0 .. nchunks): A << 31 | B;  look[nchunks .. 2 nchunks): R  (both with a flag; a pair counts when the flags agree)
__global__ __launch_bounds__(DIRECT_THREADS, 4) void k_direct(u64 n, const pp_wo_rec *__restrict__ wo, const u64 *__restrict__ cig_off,
                                                    const u32 *__restrict__ n_cig, const u32 *__restrict__ cigar,
                                                    const u8 *__restrict__ seq, const u64 *__restrict__ contig_off, u32 n_contigs,
                                                    u64 G, u32 nwin, u64 *__restrict__ look, uint4 *__restrict__ ent, u32 ent_cap,
                                                    u32 *__restrict__ win_off, u32 *__restrict__ spill_off, u32 *__restrict__ maxlen,
                                                    u64 *__restrict__ total_out, u32 *__restrict__ runs_out, u64 *__restrict__ late,
                                                    u64 *status) {
    __shared__ u32 s_hw[DIRECT_CHUNK + 1];  // home windows: [0] of the record in front of the chunk (~0: there is none), [1 + i] of record i of it
    __shared__ u32 s_later[DIRECT_LATER_MAX], s_lg[DIRECT_LATER_MAX], s_lw[DIRECT_LATER_MAX], s_nlater;
    __shared__ u64 s_wsum[DIRECT_RPT][DIRECT_THREADS / 64];
    __shared__ u64 s_excl[2];
    const u32 tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const u64 nchunks = (n + DIRECT_CHUNK - 1) / DIRECT_CHUNK;
    const uint4 *wq = (const uint4 *)wo;
    const u32 stride = nwin + 1u;
    constexpr u32 NONE = 0xFFFFFFFFu;
    u32 fast_len = 0;
    // every record that is not a single short M run inside its contig: (g, word) as k_prep's finish() leaves them
    auto general = [&](const pp_wo_rec &r, u32 *g_out, u32 *word_out) {
        u32 g = 0, nk = 0;
        u8 fl = 0;
        const u32 fi = r.file_idx;
        if (r.contig >= n_contigs) report(status, fi, DE_BAD_CONTIG);
        else {
            const u32 nc = r.op0 == PP_WO_MULTI_RUN ? n_cig[fi] : 1u;
            if (nc == 0) report(status, fi, DE_BAD_RUN);
            else
                prep_general(fi, r.ref_start, r.seq_len, r.seq_off, cigar + cig_off[fi], nc, seq, contig_off[r.contig],
                             contig_off[r.contig + 1] - contig_off[r.contig], &g, &nk, &fl, status);
        }
        *g_out = g;
        *word_out = nk | ((u32)fl << 30);
    };
    for (u64 ch = blockIdx.x; ch < nchunks; ch += gridDim.x) {
        const u64 c0 = ch * DIRECT_CHUNK;
        const bool first_round = ch == blockIdx.x;
        if (first_round) PP_STAMP(2, 0);
        if (tid == 0) s_nlater = 0;
        uint4 qa[DIRECT_RPT], qb[DIRECT_RPT];
#pragma unroll
        for (u32 u = 0; u < DIRECT_RPT; u++) {
            const u64 a = min(c0 + u * DIRECT_THREADS + tid, n - 1);  // clamped: the loads are unconditional
            qa[u] = wq[2 * a];
            qb[u] = wq[2 * a + 1];
        }
        if (tid == 0) {  // the home window of the record in front of the chunk (from the mirror alone, as below)
            u32 before = NONE;
            if (c0) {
                const uint4 p = wq[2 * (c0 - 1)];
                before = (u32)(min(contig_off[min(p.x, n_contigs - 1u)] + p.y, G - 1) / (u64)TILE);
            }
            s_hw[0] = before;
        }
        __syncthreads();
        // ---- per record: home window (the window of its start, clamped -- whatever becomes of the record); (g, word) of
        // the bulk on the spot, the others noted ----
        u32 g[DIRECT_RPT], word[DIRECT_RPT], slot[DIRECT_RPT], kc[DIRECT_RPT], fidx[DIRECT_RPT];
        u64 sof[DIRECT_RPT];
#pragma unroll
        for (u32 u = 0; u < DIRECT_RPT; u++) {
            const u32 i = u * DIRECT_THREADS + tid;
            const bool in = c0 + i < n;
            pp_wo_rec r;
            r.contig = qa[u].x; r.ref_start = qa[u].y; r.k = qa[u].z; r.seq_len = qa[u].w;
            r.seq_off = (u64)qb[u].x | ((u64)qb[u].y << 32); r.op0 = qb[u].z; r.file_idx = qb[u].w;
            const u32 cc = min(r.contig, n_contigs - 1u);
            const u64 c_lo = contig_off[cc], c_hi = contig_off[cc + 1];
            s_hw[1 + i] = in ? (u32)(min(c_lo + r.ref_start, G - 1) / (u64)TILE) : NONE;
            g[u] = 0; word[u] = 0; slot[u] = NONE;
            kc[u] = kclass_of(r.k); fidx[u] = r.file_idx; sof[u] = r.seq_off;
            if (!in) continue;
            // k_fill's checks: they only count when k_prep's found nothing (k_fill does not run then) -- kept apart in
            // *late (the smallest key as the largest complement: zero = none) and merged by k_heavy_direct
            if (r.k == 0) atomicMax(late, ~(((u64)r.file_idx << 8) | (u64)DE_BAD_K));
            else if (r.seq_off + r.seq_len > (1ull << 40)) atomicMax(late, ~(((u64)r.file_idx << 8) | (u64)DE_OVERFLOW));
            const bool bulk = r.contig < n_contigs && r.op0 != PP_WO_MULTI_RUN && (r.op0 & 15u) == PP_OP_M && (r.op0 >> 4) == r.seq_len &&
                              r.seq_len > 0 && r.seq_len <= FAST_MAX_LEN && (u64)r.ref_start + r.seq_len <= c_hi - c_lo;
            if (bulk) {
                fast_len = max(fast_len, r.seq_len);
                g[u] = (u32)(c_lo + r.ref_start);
                word[u] = r.seq_len;
            } else {
                const u32 s = atomicAdd(&s_nlater, 1u);
                if (s < DIRECT_LATER_MAX) { s_later[s] = i; slot[u] = s; }
                else report(status, 0, DE_DIRECT);  // half of a chunk's records with indels, long, or at a contig's end: not a job for this kernel
            }
        }
        __syncthreads();
        if (first_round) PP_STAMP(2, 1);
        // ---- the noted records, one per lane (their round trips are the tail of the round: most lanes idle) ----
        for (u32 s = tid; s < min(s_nlater, DIRECT_LATER_MAX); s += DIRECT_THREADS) {
            const pp_wo_rec r = wo[c0 + s_later[s]];
            u32 gg, ww;
            general(r, &gg, &ww);
            s_lg[s] = gg;
            s_lw[s] = ww;
        }
        __syncthreads();
        if (first_round) PP_STAMP(2, 2);
        // ---- per record: items in its home window (A) and in the one after it (B); does a run start here (R) ----
        u32 cnt[DIRECT_RPT], starts[DIRECT_RPT];  // A | B << 16 (a chunk holds < 2^16 of either); 1: a run starts with this record
        bool bad = false;
#pragma unroll
        for (u32 u = 0; u < DIRECT_RPT; u++) {
            const u32 i = u * DIRECT_THREADS + tid;
            if (slot[u] != NONE) { g[u] = s_lg[slot[u]]; word[u] = s_lw[slot[u]]; }
            const u32 hw = s_hw[1 + i], prev = s_hw[i];
            u32 na = 0, nb = 0;
            if (word[u])
                direct_items(g[u], word[u], 0, 0, 0, nwin, [&](u32 w, const uint4 &) {
                    if (w == hw) na++;
                    else if (w == hw + 1u) nb++;
                    else bad = true;
                });
            cnt[u] = na | (nb << 16);
            starts[u] = (c0 + i < n && prev != NONE && hw < prev) ? 1u : 0u;
        }
        if (bad) report(status, 0, DE_DIRECT);
        // ---- exclusive prefix of the counts in record order: rows of DIRECT_THREADS records, row after row ----
        u32 inc[DIRECT_RPT], rbefore[DIRECT_RPT];
#pragma unroll
        for (u32 u = 0; u < DIRECT_RPT; u++) {
            u32 v = cnt[u];
            for (int o = 1; o < 64; o <<= 1) {
                const u32 x = (u32)__shfl_up((int)v, o, 64);
                if ((int)lane >= o) v += x;
            }
            inc[u] = v;
            const u64 rs = __ballot(starts[u] != 0);  // (runs start once per SAM file: counted by ballot, no scan)
            rbefore[u] = (u32)__popcll(rs & ((1ull << lane) - 1ull));
            if (lane == 63) s_wsum[u][wave] = (u64)v | ((u64)__popcll(rs) << 32);
        }
        __syncthreads();
        u32 before[DIRECT_RPT];
        u64 chunk_total = 0;  // A | B << 16 | R << 32
        {
            u64 run = 0;
#pragma unroll
            for (u32 u = 0; u < DIRECT_RPT; u++) {
                u64 b = run;
                for (u32 i = 0; i < DIRECT_THREADS / 64; i++) {
                    const u64 ws = s_wsum[u][i];
                    if (i < wave) b += ws;
                    run += ws;
                }
                before[u] = (u32)b + inc[u] - cnt[u];
                rbefore[u] += (u32)(b >> 32);
            }
            chunk_total = run;
        }
        if (first_round) PP_STAMP(2, 3);
        // ---- the chunk's place among all chunks: decoupled look-back (wave 0) ----
        if (wave == 0) {
            const u64 agg_ab = ((chunk_total & 0xFFFFull) << 31) | ((chunk_total >> 16) & 0xFFFFull), agg_r = chunk_total >> 32;
            if (lane == 0 && ch) {
                __hip_atomic_store(&look[ch], LOOK_AGG | agg_ab, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&look[nchunks + ch], LOOK_AGG | agg_r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            u64 ex_ab = 0, ex_r = 0;
            if (ch) {
                long long pos = (long long)ch - 1;
                u32 spins = 0;
                for (;;) {
                    const long long j = pos - (long long)lane;
                    u64 v = LOOK_INC, v2 = LOOK_INC;  // (in front of the first chunk: an inclusive prefix of zero)
                    if (j >= 0) {
                        v = __hip_atomic_load(&look[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        v2 = __hip_atomic_load(&look[nchunks + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    const u32 f1 = (u32)(v >> 62), f2 = (u32)(v2 >> 62);
                    const u32 fl = f1 == f2 ? f1 : 0u;  // (a pair half way through its update is not there yet)
                    const u64 ready = __ballot(fl != 0), incl = __ballot(fl == 2u);
                    const u32 upto = incl ? (u32)__ffsll((long long)incl) : 64u;  // lanes [0, upto) are wanted
                    const u64 want = upto == 64u ? ~0ull : ((1ull << upto) - 1ull);
                    if ((ready & want) == want) {
                        ex_ab += wave_sum64(lane < upto ? (v & LOOK_VAL) : 0ull);
                        ex_r += wave_sum64(lane < upto ? (v2 & LOOK_VAL) : 0ull);
                        if (incl) break;
                        pos -= 64;
                        continue;
                    }
                    if (++spins > DIRECT_SPIN_LIMIT) {  // somebody in front of this chunk is not running: give up, let the others through
                        if (lane == 0) report(status, 1, DE_DIRECT);  // (record number 1 = "timed out", 0 = "not a job for this kernel": pp_polish_finish)
                        break;
                    }
                    __builtin_amdgcn_s_sleep(2);
                }
            }
            if (lane == 0) {
                s_excl[0] = ex_ab;
                s_excl[1] = ex_r;
                __hip_atomic_store(&look[ch], LOOK_INC | ((ex_ab + agg_ab) & LOOK_VAL), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&look[nchunks + ch], LOOK_INC | ((ex_r + agg_r) & LOOK_VAL), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (ch + 1 == nchunks) {  // the job's totals
                    const u64 tot = ex_ab + agg_ab, ta = tot >> 31, tb = tot & 0x7FFFFFFFull;
                    *total_out = ta + tb;
                    *runs_out = (u32)min(ex_r + agg_r + 1ull, (u64)DIRECT_MAX_RUNS + 1ull);
                    if (ta + tb > (u64)ent_cap) report(status, ta + tb, DE_CAPACITY);
                }
            }
        }
        __syncthreads();
        if (first_round) PP_STAMP(2, 4);
        const u32 baseA = (u32)(s_excl[0] >> 31), baseB = (u32)(s_excl[0] & 0x7FFFFFFFull), baseR = (u32)min(s_excl[1], 0xFFFFull);
        // ---- the items, and where the windows begin (per run) ----
#pragma unroll
        for (u32 u = 0; u < DIRECT_RPT; u++) {
            const u32 i = u * DIRECT_THREADS + tid;
            if (c0 + i >= n) continue;
            u32 pa = baseA + (before[u] & 0xFFFFu), pb = baseB + (before[u] >> 16);
            const u32 run = baseR + rbefore[u] + starts[u];                        // the run this record belongs to
            const u32 hw = s_hw[1 + i], prev = s_hw[i];
            if (run >= DIRECT_MAX_RUNS) { report(status, 0, DE_DIRECT); continue; }  // more runs than the window tables hold
            u32 *wo_r = win_off + (u64)run * stride, *so_r = spill_off + (u64)run * stride;
            if (starts[u]) {  // the run in front of this one ends here: its windows behind `prev`; this one's up to `hw`
                u32 *wo_p = wo_r - stride, *so_p = so_r - stride;
                for (u32 w = prev + 1u; w <= nwin; w++) { wo_p[w] = pa; so_p[w] = pb; }
                for (u32 w = 0; w <= hw; w++) { wo_r[w] = pa; so_r[w] = pb; }
            } else if (prev == NONE || hw > prev)
                for (u32 w = prev == NONE ? 0u : prev + 1u; w <= hw; w++) { wo_r[w] = pa; so_r[w] = pb; }
            if (word[u]) {
                direct_items(g[u], word[u], sof[u], kc[u], fidx[u], nwin, [&](u32 w, const uint4 &e) {
                    if (w == hw) { if (pa < ent_cap) ent[pa] = e; pa++; }
                    else if (w == hw + 1u) { if (pb < ent_cap) ent[ent_cap - 1u - pb] = e; pb++; }
                });
            }
            if (c0 + i == n - 1)  // behind the last record: the windows that are left
                for (u32 w = hw + 1u; w <= nwin; w++) { wo_r[w] = pa; so_r[w] = pb; }
        }
        if (first_round) PP_STAMP(2, 5);
        __syncthreads();  // (s_hw, s_later are the next round's)
    }
    PP_STAMP(2, 6);
    if (__ballot(fast_len > PLAIN_NARROW_MAX)) {  // (as k_prep)
        for (int o = 32; o > 0; o >>= 1) fast_len = max(fast_len, (u32)__shfl_xor((int)fast_len, o, 64));
        if (lane == 0 && fast_len > __hip_atomic_load(maxlen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(maxlen, fast_len);
    }
}

// the heavy-window list (note_heavy) from the stretches' sizes
__global__ __launch_bounds__(256) void k_heavy_direct(u32 nwin, WinSource S, u32 heavy_min, u32 *__restrict__ heavy,
                                                      u8 *__restrict__ win_heavy, const u64 *__restrict__ late, u64 *status) {
    const u32 w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w == 0 && *late) atomicMin(status, ~*late);  // (k_direct's second-rank errors: only if nothing else was wrong)
    if (*status != ~0ull) {  // (the offsets may be incomplete after an error)
        if (w < nwin) win_heavy[w] = 0;
        return;
    }
    if (w < nwin) note_heavy(w, win_item_count(S, w), heavy_min, heavy, win_heavy);
}

}  // namespace pp
