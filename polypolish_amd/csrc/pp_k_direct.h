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
// k_prepd
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024, PP_PREP_WAVES) void k_prepd(u64 n, u64 chunk, const pp_wo_rec *__restrict__ wo,
                                                               const u64 *__restrict__ cig_off, const u32 *__restrict__ n_cig,
                                                               const u32 *__restrict__ cigar, const u8 *__restrict__ seq,
                                                               const u64 *__restrict__ contig_off, u32 n_contigs, u32 nwin,
                                                               const u32 *__restrict__ run_end, u32 n_runs,  // ends of the mirror's runs (ascending, the last one = n)
                                                               u32 *__restrict__ first, u32 *__restrict__ x_cnt, u32 *__restrict__ x_nb,
                                                               uint4 *__restrict__ xent, u32 xcap, u32 *__restrict__ maxlen,
                                                               u64 *__restrict__ x_need, u64 *status) {
    __shared__ u32 later[PREP_LATER_MAX], n_later;
    // The block's extras are STAGED in LDS and get their slots in the windows' rooms at the end, one returning global atomic
    // per window of the block instead of one per wave, window and trip through the loop (a wave waited out thirteen of
    // those round trips, one after the other: k_prepd 0.16 ms where k_prep + k_fill took 0.19).  A block's entries lie in
    // a handful of consecutive windows (the mirror is in window order): XLOCAL counters from the window of its first entry
    // on; an extra for a window outside that range (a block across the end of a run, a long read), or one more than the
    // stage holds, takes its slot from the global counter on the spot.
    constexpr u32 XSTAGE = 2048, XLOCAL = 64;
    __shared__ uint4 st_item[XSTAGE];
    __shared__ u32 st_key[XSTAGE], l_cnt[XLOCAL], l_base[XLOCAL], l_nb[XLOCAL], n_st, s_wbase;
    PP_STAMP(0, 0);
    const u64 lo = (u64)blockIdx.x * chunk, hi = min(n, lo + chunk);
    if (threadIdx.x == 0) {
        n_later = 0;
        n_st = 0;
        u32 wb = 0;
        if (lo < hi) {
            const pp_wo_rec r0 = wo[lo];
            if (r0.contig < n_contigs) wb = wo_home(contig_off[r0.contig], r0.ref_start, nwin);
        }
        s_wbase = wb;
    }
    if (threadIdx.x < XLOCAL) { l_cnt[threadIdx.x] = 0; l_nb[threadIdx.x] = 0; }
    __syncthreads();
    PP_STAMP(0, 1);
    const u32 wbase = s_wbase;
    const u32 lane = threadIdx.x & 63u;
    const u32 stride = nwin + 1u;
    // one extra into its slot of its window's room, or a capacity overflow (the host gives the windows more room and reruns)
    auto put = [&](u32 w, u32 slot, const uint4 &e) {
        if (slot < xcap) xent[(u64)w * xcap + slot] = e;
        else {
            atomicMax(x_need, (u64)slot + 1ull);
            report(status, slot, DE_CAPACITY);
        }
    };
    // one extra of window w: staged, or straight to its window
    auto emit = [&](u32 w, const uint4 &e) {
        const u32 wl = w - wbase;
        if (wl < XLOCAL) {
            const u32 pos = atomicAdd(&n_st, 1u);
            if (pos < XSTAGE) {
                st_key[pos] = (wl << 16) | atomicAdd(&l_cnt[wl], 1u);
                st_item[pos] = e;
                return;
            }
        }
        put(w, atomicAdd(&x_cnt[w], 1u), e);
    };
    // the pieces of a record that is NOT bulk (prep_general's verdict), window by window
    auto cut = [&](u32 g_out, u32 word, u64 so, u32 kc, u32 fi) {
        if (!word) return;
        const u32 cls = word >> 30, ia = (word >> 9) & 0xFFu, idel = (word >> 17) & 1u;
        for_each_piece(g_out, word, [&](u32 piece, u32 g, u32 sp) {
            if (!sp) return;
            const u32 wa = g / (u32)TILE, wb = min((g + sp - 1u) / (u32)TILE, nwin - 1u);
            u64 pso = so;
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
                emit(w, e);
            }
        });
    };
    auto general = [&](const pp_wo_rec &r, u64 c_lo, u64 c_hi) {
        u32 g_out = 0, nk_out = 0;
        u8 fl_out = 0;
        const u32 fi = r.file_idx;
        if (fi >= n) return;  // (reported in the loop: the arrays cannot be read for it)
        if (r.contig >= n_contigs) report(status, fi, DE_BAD_CONTIG);
        else {
            const u32 nc = r.op0 == PP_WO_MULTI_RUN ? n_cig[fi] : 1u;
            if (nc == 0) report(status, fi, DE_BAD_RUN);
            else prep_general(fi, r.ref_start, r.seq_len, r.seq_off, cigar + cig_off[fi], nc, seq, c_lo, c_hi - c_lo, &g_out, &nk_out, &fl_out, status);
        }
        cut(g_out, nk_out | ((u32)fl_out << 30), r.seq_off, kclass_of(r.k), fi);
    };
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
    u32 run0 = 0, run0_lo = 0;  // the run of the block's first entry
    while (run0 + 1u < n_runs && lo >= run_end[run0]) { run0_lo = run_end[run0]; run0++; }
    u32 fast_len = 0;
    constexpr int WU = PP_WO_UNROLL;
    const uint4 *wq = (const uint4 *)wo;
    const u64 trip = (u64)WU * blockDim.x;
    const u64 span = (hi - lo + trip - 1) / trip * trip;  // whole waves and whole trips: the ballots below need every lane
    for (u64 a0 = lo + threadIdx.x; a0 < lo + span; a0 += trip) {
        uint4 qa[WU], qb[WU], pa[WU];
#pragma unroll
        for (int u = 0; u < WU; u++) {
            const u64 a = min(a0 + (u64)u * blockDim.x, n - 1);  // clamped: the loads are unconditional
            qa[u] = wq[2 * a];
            qb[u] = wq[2 * a + 1];
            pa[u] = wq[2 * (a ? a - 1 : 0)];  // the entry in front (its first half: contig, ref_start): where a window begins
        }
#pragma unroll
        for (int u = 0; u < WU; u++) {
            const u64 a = a0 + (u64)u * blockDim.x;
            const bool in = a < hi;
            pp_wo_rec r;
            r.contig = qa[u].x; r.ref_start = qa[u].y; r.k = qa[u].z; r.seq_len = qa[u].w;
            r.seq_off = (u64)qb[u].x | ((u64)qb[u].y << 32); r.op0 = qb[u].z; r.file_idx = qb[u].w;
            const bool c_ok = r.contig < n_contigs, p_ok = pa[u].x < n_contigs;
            const u32 cc = min(r.contig, n_contigs - 1u), pc = min(pa[u].x, n_contigs - 1u);
            const u64 c_lo = contig_off[cc], c_hi = contig_off[cc + 1], p_lo = contig_off[pc];
            const bool bulk = in && wo_bulk(c_ok, r.ref_start, r.seq_len, r.op0, c_hi - c_lo);
            // ---- checks every record gets (k_fill's, and the mirror's own) ----
            if (in) {
                if (r.file_idx >= n) report(status, a, DE_BAD_MIRROR);
                else if (r.k == 0) report(status, r.file_idx, DE_BAD_K);
                else if (r.seq_off + r.seq_len > (1ull << 40)) report(status, r.file_idx, DE_OVERFLOW);
            }
            // ---- where the windows begin in this entry's run ----
            u32 run = run0, run_lo = run0_lo;  // (the block's entries lie in one run, or in a few)
            while (run + 1u < n_runs && a >= run_end[run]) { run_lo = run_end[run]; run++; }
            const u32 run_hi = run_end[run];
            const u32 h = in && c_ok ? wo_home(c_lo, r.ref_start, nwin) : NOHOME;
            const u32 hp = p_ok ? wo_home(p_lo, pa[u].y, nwin) : NOHOME;
            const bool starts = (u32)a == run_lo;
            if (h != NOHOME && !starts && hp != NOHOME && h < hp) report(status, (1ull << 40) - 1ull, DE_MIRROR_ORDER);
            const bool begins = h != NOHOME && (starts || (hp != NOHOME && h > hp));
            fill(begins, run, starts ? 0u : hp + 1u, h, (u32)a);
            fill(h != NOHOME && (u32)a + 1u == run_hi, run, h + 1u, nwin, (u32)a + 1u);
            // ---- a bulk read that reaches into the next window: one extra there ----
            const u64 g = c_lo + r.ref_start;
            const u32 w1 = bulk ? (u32)((g + r.seq_len - 1u) / (u64)TILE) : 0u;
            if (bulk && w1 > h) emit(w1, wo_item(r.seq_off, r.seq_len, kclass_of(r.k), g, w1, r.file_idx));
            if (bulk) fast_len = max(fast_len, r.seq_len);
            else if (in) {
                // (a record that is not bulk: the window it starts in holds its entry AND, among the extras, its pieces --
                // counted, so that k_winplan can tell what the window's work amounts to)
                if (h != NOHOME) { if (h - wbase < XLOCAL) atomicAdd(&l_nb[h - wbase], 1u); else atomicAdd(&x_nb[h], 1u); }
                const u32 slot = atomicAdd(&n_later, 1u);
                if (slot < PREP_LATER_MAX) later[slot] = (u32)(a - lo);
                else general(r, c_lo, c_hi);
            }
        }
    }
    PP_STAMP(0, 2);
    __syncthreads();
    PP_STAMP(0, 3);
    for (u32 i = threadIdx.x; i < min(n_later, PREP_LATER_MAX); i += blockDim.x) {
        const pp_wo_rec r = wo[lo + later[i]];
        const u32 cc = min(r.contig, n_contigs - 1u);
        general(r, contig_off[cc], contig_off[cc + 1]);
    }
    PP_STAMP(0, 4);
    // ---- the staged extras: a stretch of slots per window of the block, then every item to its slot ----
    __syncthreads();
    if (threadIdx.x < XLOCAL) {
        const u32 c = l_cnt[threadIdx.x], nb = l_nb[threadIdx.x];
        l_base[threadIdx.x] = c ? atomicAdd(&x_cnt[wbase + threadIdx.x], c) : 0u;  // (c > 0: a window of the assembly)
        if (nb) atomicAdd(&x_nb[wbase + threadIdx.x], nb);
    }
    __syncthreads();
    PP_STAMP(0, 5);
    for (u32 i = threadIdx.x; i < min(n_st, XSTAGE); i += blockDim.x) {
        const u32 key = st_key[i], wl = key >> 16;
        put(wbase + wl, l_base[wl] + (key & 0xFFFFu), st_item[i]);
    }
    PP_STAMP(0, 6);
    // the job's longest fast-class read (as k_prep)
    if (__ballot(fast_len > PLAIN_NARROW_MAX)) {
        for (int o = 32; o > 0; o >>= 1) fast_len = max(fast_len, (u32)__shfl_xor((int)fast_len, o, 64));
        if (lane == 0 && fast_len > __hip_atomic_load(maxlen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(maxlen, fast_len);
    }
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
