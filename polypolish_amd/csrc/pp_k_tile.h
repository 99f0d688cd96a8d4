// pp_k_tile.h -- k_tile: pileup accumulate (from the units k_stream / k_regroup prepared) + vote for one window.
// Part of pp_kernels.hip (included there, in this order, and nowhere else: it defines __global__ kernels).
#pragma once

namespace pp {

// =============================================================================================
// k_tile: pileup accumulate + vote for one 2048-position window
// =============================================================================================
struct TileArgs {
    const u64 *units;      // regrouped units: per window its PLAIN / SLOW items, then its EVENTs
    const u64 *win_start;
    const u32 *win_nitem, *win_nev;
    u32 nwin;
    const u32 *contig, *ref_start, *kk;   // record arrays (SLOW units only)
    const u32 *nkeep_arr;                 // kept entries of SLOW records (k_stream)
    const u8 *seq;
    const u64 *seq_off;
    const u64 *cig_off;
    const u32 *n_cig;
    const u32 *cigar;
    const u8 *bases;
    u64 G;
    const u64 *contig_off;
    u32 n_contigs;
    u32 min_depth;
    double fv, fi;
    u8 *code;
    u32 *win_len;
    u32 *counters;  // [0] positions on the global replay list, [1] n_multi, [2] all flagged positions
    u32 cap_flag;
    u32 *flag_bits;   // per window: 2048-bit map of flagged positions (64 words)
    u32 *win_nflag;   // per window: number of flagged positions
    u32 *win_slab;    // per window: index of its tally slab (6 x 2048 u32), or ~0
    u32 *slabs;
    u32 cap_slabs;
    u32 *flag_pos;
    u32 *flag_cov;
    u64 *scr_need;  // replay scratch the listed positions will need (sum of their coverage), counted past cap_flag too
    ContigStatsDev *stats;
    const u32 *own;   // optional (lo, hi) emit range per contig, relative to the contig (pp_polish_set_emit)
    double *dbg_depth;
    u32 *dbg_counts;  // 7 planes of G: a, c, g, t, other, valid_thr, invalid_thr
    u8 *dbg_status;
    u64 *status;
    int dbg;
};

__device__ __forceinline__ void tile_add(u32 *cnt, int row, int p, u32 kc) {
    atomicAdd(&cnt[row * TILE + p], 1u);
    if (kc) {
        if (kc == KCLASS_NONDYADIC) atomicOr(&cnt[ROW_DEF * TILE + p], 0x80000000u);
        else atomicAdd(&cnt[ROW_DEF * TILE + p], (1u << DEPTH_FX_BITS) - (1u << (DEPTH_FX_BITS - kc)));
    }
}

__device__ __forceinline__ u32 find_contig(const u64 *contig_off, u32 n_contigs, u64 p) {
    u32 lo = 0, hi = n_contigs;  // contig_off[lo] <= p < contig_off[hi]
    while (hi - lo > 1) {
        u32 mid = (lo + hi) >> 1;
        if (contig_off[mid] <= p) lo = mid; else hi = mid;
    }
    return lo;
}

struct VoteOut {
    u8 out;     // byte to emit (0 = nothing)
    u8 status;  // PP_ST_*
    u32 vthr, ithr;
};

// pileup.rs:67-134 restricted to the keys A,C,G,T and "-"; callers guarantee that no other key
// can reach either threshold.
__device__ __forceinline__ VoteOut vote5(u32 nA, u32 nC, u32 nG, u32 nT, u32 nDel, double depth,
                                         u8 orig, u32 min_depth, double fv, double fi) {
    VoteOut v;
    u32 vt = d_bankers(__dmul_rn(depth, fv));
    v.vthr = max(min_depth, vt);
    v.ithr = d_bankers(__dmul_rn(depth, fi));
    v.out = orig;
    v.status = PP_ST_KEPT;
    if (depth < (double)min_depth) {
        v.status = PP_ST_LOW_DEPTH;
    } else {
        int nv = 0, ni = 0;
        u8 win = 0;
        if (nA >= v.vthr) { nv++; win = 'A'; } else if (nA >= v.ithr) ni++;
        if (nC >= v.vthr) { if (!nv) win = 'C'; nv++; } else if (nC >= v.ithr) ni++;
        if (nG >= v.vthr) { if (!nv) win = 'G'; nv++; } else if (nG >= v.ithr) ni++;
        if (nT >= v.vthr) { if (!nv) win = 'T'; nv++; } else if (nT >= v.ithr) ni++;
        if (nDel > 0) {
            if (nDel >= v.vthr) { if (!nv) win = '-'; nv++; } else if (nDel >= v.ithr) ni++;
        }
        if (nv == 1) {
            if (ni > 0) v.status = PP_ST_TOO_CLOSE;
            else { v.out = win; if (win != orig) v.status = PP_ST_CHANGED; }
        } else if (nv == 0) {
            v.status = PP_ST_NONE;
        } else {
            v.status = PP_ST_MULTIPLE;
        }
    }
    if (v.out == (u8)'-') v.out = 0;  // polish.rs:188
    return v;
}

// LDS copy of the window's assembly bytes (the vote needs the original base of every position)
constexpr int ASM_PAD = 0;
constexpr int ASM_WORDS = TILE / 4;

// exact integer tallies of one window position from the LDS rows: explicit tallies plus, for the
// assembly's own base, the fast-class bases that were never tallied one by one
__device__ __forceinline__ void position_tallies(const u32 *cnt, u8 orig, u32 p, u32 &nA, u32 &nC, u32 &nG, u32 &nT,
                                                 u32 &nDel, u32 &nOth) {
    nA = cnt[ROW_A * TILE + p]; nC = cnt[ROW_C * TILE + p]; nT = cnt[ROW_T * TILE + p];
    nG = cnt[ROW_G * TILE + p]; nDel = cnt[ROW_DEL * TILE + p]; nOth = cnt[ROW_OTH * TILE + p];
    const u32 same = cnt[ROW_COV * TILE + p] - cnt[ROW_MIS * TILE + p];
    const int ro = row_of(orig);
    nA += (ro == ROW_A) ? same : 0u; nC += (ro == ROW_C) ? same : 0u; nT += (ro == ROW_T) ? same : 0u;
    nG += (ro == ROW_G) ? same : 0u; nDel += (ro == ROW_DEL) ? same : 0u; nOth += (ro == ROW_OTH) ? same : 0u;
}

// The units of one window.  PLAIN: two LDS atomics into the coverage difference array (+1 at the first kept
// position, -1 one past the last; prefix-summed before the vote).  EVENT: the row of the differing base + the
// mismatch row.
// one unit, one lane: PLAIN and EVENT units are two LDS atomics each
__device__ __forceinline__ void tile_unit_fast(u32 *cnt, u32 *s_ndbits, u32 lo32) {
    const u32 tag = lo32 & 3u;
    if (tag == UNIT_PLAIN) {
        const int rel = (int)((lo32 >> 2) & 0xFFFu) - UNIT_REL_BIAS;
        const int nkeep = (int)((lo32 >> 14) & 0xFFu);
        const int lo = max(0, -rel), hi = min(nkeep, TILE - rel);
        if (hi > lo) {
            atomicAdd(&cnt[ROW_COV * TILE + rel + lo], 1u);
            if (rel + hi < TILE) atomicAdd(&cnt[ROW_COV * TILE + rel + hi], 0xFFFFFFFFu);
            if ((lo32 >> 22) & 1u) {  // mark [rel+lo, rel+hi) in the window's bitmap of order-dependent positions
                const u32 a = (u32)(rel + lo), b = (u32)(rel + hi);
                for (u32 wd = a >> 5; wd <= (b - 1u) >> 5; wd++) {
                    const u32 from = wd == (a >> 5) ? (a & 31u) : 0u, to = wd == ((b - 1u) >> 5) ? ((b - 1u) & 31u) : 31u;
                    atomicOr(&s_ndbits[wd], (0xFFFFFFFFu >> (31u - to)) & (0xFFFFFFFFu << from));
                }
            }
        }
    } else if (tag == UNIT_EVENT) {
        const u32 p = (lo32 >> 2) & (u32)(TILE - 1), row = (lo32 >> 13) & 7u;
        atomicAdd(&cnt[row * TILE + p], 1u);
        atomicAdd(&cnt[ROW_MIS * TILE + p], 1u);
    }
}

// SLOW units of one wave-load: the record is walked run by run (indels), one wave per unit, every kept base tallied
// explicitly (pileup.rs:56-65,189-200).
__device__ __forceinline__ void tile_units_slow(const TileArgs &A, u32 *cnt, u64 u, u64 w0, u32 lane) {
    const u32 lo32 = (u32)u;
    const bool is_slow = (lo32 & 3u) == UNIT_SLOW;
    u64 slow = __ballot(is_slow);
    if (!slow) return;
    // the record fields of all SLOW units of this wave-load are fetched side by side (one round of latency) ...
    u32 f_nkeep = 0, f_kc = 0, f_nc = 0;
    int f_rel = 0;
    u64 f_so = 0, f_co = 0;
    if (is_slow) {
        const u32 idx = (u32)(u >> 32);
        f_nkeep = A.nkeep_arr[idx];
        f_rel = (int)((long long)(A.contig_off[A.contig[idx]] + A.ref_start[idx]) - (long long)w0);
        f_kc = kclass_of(A.kk[idx]);
        f_so = A.seq_off[idx];
        f_co = A.cig_off[idx];
        f_nc = A.n_cig[idx];
    }
    // ... then every unit is walked by the whole wave
    while (slow) {
        const int j = __ffsll((long long)slow) - 1;
        slow &= slow - 1;
        const u32 fl = ((u32)__builtin_amdgcn_readlane((int)lo32, j) >> 2) & 3u;
        const int nkeep = __builtin_amdgcn_readlane((int)f_nkeep, j), rel = __builtin_amdgcn_readlane(f_rel, j);
        const u32 kc = (u32)__builtin_amdgcn_readlane((int)f_kc, j);
        const u8 *s = A.seq + (((u64)(u32)__builtin_amdgcn_readlane((int)(u32)(f_so >> 32), j) << 32) |
                               (u64)(u32)__builtin_amdgcn_readlane((int)(u32)f_so, j));
        if (!(fl & ENT_COMPLEX)) {
            // no indels, trimmed by k_stream (long read, contig overhang, dyadic share): entry i is base i
            const int lo = max(0, -rel), hi = min(nkeep, TILE - rel);
            for (int q = lo + (int)lane; q < hi; q += 64) tile_add(cnt, row_of(s[q]), rel + q, kc);
        } else {
            const u32 *cg = A.cigar + (((u64)(u32)__builtin_amdgcn_readlane((int)(u32)(f_co >> 32), j) << 32) |
                                       (u64)(u32)__builtin_amdgcn_readlane((int)(u32)f_co, j));
            const u32 nc = (u32)__builtin_amdgcn_readlane((int)f_nc, j);
            int ent0 = 0;
            u64 ro = 0;
            for (u32 r = 0; r < nc && ent0 < nkeep; r++) {
                const u32 op = cg[r], len = op >> 4, o = op & 15u;
                if (o == PP_OP_I) { ro += len; continue; }
                u32 ins = 0;
                for (u32 r2 = r + 1; r2 < nc && (cg[r2] & 15u) == PP_OP_I; r2++) ins += cg[r2] >> 4;
                const int a = max(ent0, -rel), b = min(min(ent0 + (int)len, nkeep), TILE - rel);
                for (int q = a + (int)lane; q < b; q += 64) {
                    const bool ext = (q == ent0 + (int)len - 1) && ins > 0;
                    int row;
                    if (o == PP_OP_D) row = ext ? (ins == 1 ? row_of(s[ro]) : ROW_OTH) : ROW_DEL;
                    else row = ext ? ROW_OTH : row_of(s[ro + (u64)(q - ent0)]);
                    tile_add(cnt, row, rel + q, kc);
                }
                ent0 += (int)len;
                if (o != PP_OP_D) ro += len;
            }
        }
    }
}

constexpr u32 TILE_UNROLL = 4;  // unit loads in flight per lane

__global__ __launch_bounds__(TILE_THREADS, 8) void k_tile(TileArgs A) {
    __shared__ u32 cnt[N_ROWS * TILE];
    __shared__ __attribute__((aligned(16))) u32 asm_w[ASM_WORDS];  // the window's assembly bytes
    __shared__ u32 s_len, s_changed, s_zero, s_c0, s_c1, s_wsum[TILE_THREADS / 64], s_fbits[TILE / 32], s_ndbits[TILE / 32], s_nflag;
    __shared__ u64 s_depth;

    // XCD-aware order: consecutive windows (which share boundary-crossing reads) stay on one XCD
    u32 per = gridDim.x >> 3;
    u32 w = (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
    if (w >= A.nwin || job_state(A.status) == 2) return;
    const u32 tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const u64 w0 = (u64)w * TILE;
    // the first units of every lane are requested before the counters are cleared
    const u64 e0 = A.win_start[w];
    const u32 n_items = A.win_nitem[w], n_units = n_items + A.win_nev[w];
    u64 pre[TILE_UNROLL];
#pragma unroll
    for (u32 q = 0; q < TILE_UNROLL; q++) {
        const u32 i = q * TILE_THREADS + tid;
        pre[q] = i < n_units ? A.units[e0 + i] : (u64)UNIT_NOP;
    }

    for (u32 i = tid; i < (u32)(N_ROWS * TILE); i += TILE_THREADS) cnt[i] = 0;
    if (tid < (u32)(TILE / 32)) { s_fbits[tid] = 0; s_ndbits[tid] = 0; }
    {
        u8 *ab = (u8 *)asm_w;
        for (u32 i = tid; i < (u32)TILE; i += TILE_THREADS) ab[i] = (w0 + i < A.G) ? A.bases[w0 + i] : (u8)0;
    }
    if (tid == 0) {
        s_len = 0; s_changed = 0; s_zero = 0; s_depth = 0; s_nflag = 0;
        s_c0 = find_contig(A.contig_off, A.n_contigs, w0);
        u64 last = min(w0 + TILE, A.G) - 1;
        s_c1 = find_contig(A.contig_off, A.n_contigs, last);
    }
    __syncthreads();

    for (u32 base = 0;; base += TILE_UNROLL * TILE_THREADS) {
        if (base) {
#pragma unroll
            for (u32 q = 0; q < TILE_UNROLL; q++) {
                const u32 i = base + q * TILE_THREADS + tid;
                pre[q] = i < n_units ? A.units[e0 + i] : (u64)UNIT_NOP;
            }
        }
#pragma unroll
        for (u32 q = 0; q < TILE_UNROLL; q++) tile_unit_fast(cnt, s_ndbits, (u32)pre[q]);
#pragma unroll
        for (u32 q = 0; q < TILE_UNROLL; q++) tile_units_slow(A, cnt, pre[q], w0, lane);
        if (base + TILE_UNROLL * TILE_THREADS >= n_units) break;
    }
    __syncthreads();

    // ---- coverage of the fast class: prefix sum of the difference array, in place ----
    {
        u32 *cov = cnt + ROW_COV * TILE;
        const u32 d0 = cov[2 * tid], d1 = cov[2 * tid + 1];
        const u32 sum = d0 + d1;
        u32 inc = sum;
        for (int o = 1; o < 64; o <<= 1) {
            const u32 v = __shfl_up(inc, o, 64);
            if ((int)lane >= o) inc += v;
        }
        if (lane == 63) s_wsum[wave] = inc;
        __syncthreads();
        u32 base = 0;
        for (u32 i = 0; i < wave; i++) base += s_wsum[i];
        const u32 ex = base + inc - sum;
        cov[2 * tid] = ex + d0;
        cov[2 * tid + 1] = ex + d0 + d1;
    }
    __syncthreads();

    // ---- vote: one lane per position ----
    u32 my_len = 0, my_changed = 0, my_zero = 0;
    u64 my_depth = 0;
    const bool one_contig = (s_c0 == s_c1);
    for (u32 p = tid; p < (u32)TILE; p += TILE_THREADS) {
        const u64 gp = w0 + p;
        if (gp >= A.G) break;
        if (A.own) {  // window tiling: halo positions are voted by the rank that owns them
            const u32 c = one_contig ? s_c0 : find_contig(A.contig_off, A.n_contigs, gp);
            const u32 rel = (u32)(gp - A.contig_off[c]);
            if (rel < A.own[2 * c] || rel >= A.own[2 * c + 1]) {
                A.code[gp] = 0;
                continue;
            }
        }
        u32 nA, nC, nG, nT, nDel, nOth;
        const u32 defw = cnt[ROW_DEF * TILE + p];
        const u8 orig = ((const u8 *)asm_w)[ASM_PAD + p];
        position_tallies(cnt, orig, p, nA, nC, nG, nT, nDel, nOth);
        const bool nd = (defw >> 31) != 0 || ((s_ndbits[p >> 5] >> (p & 31u)) & 1u) != 0;
        const u32 deficit = defw & 0x7FFFFFFFu;
        const u32 ntot = nA + nC + nG + nT + nDel + nOth;
        if (orig >= 0x80u) report(A.status, gp, DE_NON_ASCII);
        const u64 dfx = ((u64)ntot << DEPTH_FX_BITS) - deficit;
        const double depth = (double)dfx * (1.0 / (double)(1u << DEPTH_FX_BITS));  // exact
        bool flag = false;
        VoteOut v;
        v.out = (orig == (u8)'-') ? 0 : orig;
        v.status = PP_ST_LOW_DEPTH;
        v.vthr = 0; v.ithr = 0;
        if (nd) {
            // depth is an order-dependent f64 sum: exact only in k_exact.  depth <= ntot always,
            // so ntot < min_depth already decides DepthTooLow.
            if (ntot >= A.min_depth || A.dbg) flag = true;
        } else {
            const u32 ithr = d_bankers(__dmul_rn(depth, A.fi));
            if (!(depth < (double)A.min_depth) && nOth > 0 && nOth >= ithr) flag = true;
            else v = vote5(nA, nC, nG, nT, nDel, depth, orig, A.min_depth, A.fv, A.fi);
        }
        if (A.dbg && nDel + nOth > 0) flag = true;  // --debug lists every key: k_exact writes the records
        if (flag) {
            const bool to_list = A.dbg == 1 || n_items > SORT_MAX;  // dbg 2: test hook, see run_pipeline
            if (!to_list) {
                atomicOr(&s_fbits[p >> 5], 1u << (p & 31u));
                atomicAdd(&s_nflag, 1u);
            } else {
                atomicAdd(&A.counters[2], 1u);
            }
            if (to_list) {
                // bucket too large for the wave-per-position replay: global list for k_exact
                const u32 slot = atomicAdd(&A.counters[0], 1u);
                atomicAdd(A.scr_need, (u64)ntot);
                if (slot < A.cap_flag) {
                    A.flag_pos[slot] = (u32)gp;
                    A.flag_cov[slot] = ntot;
                } else {
                    report(A.status, slot, DE_CAPACITY_LATE);
                }
            }
            A.code[gp] = 0;
            continue;
        }
        A.code[gp] = v.out;
        const u32 l = v.out ? 1u : 0u, ch = (v.status == PP_ST_CHANGED), z = (ntot == 0);
        if (one_contig) {
            my_len += l; my_changed += ch; my_zero += z; my_depth += dfx;
        } else {
            my_len += l;
            const u32 c = find_contig(A.contig_off, A.n_contigs, gp);
            if (ch) atomicAdd(&A.stats[c].changed, 1ull);
            if (z) atomicAdd(&A.stats[c].zero_depth, 1ull);
            if (dfx) atomicAdd(&A.stats[c].depth_fx, dfx);
        }
        if (A.dbg) {
            A.dbg_depth[gp] = depth;
            A.dbg_counts[0 * A.G + gp] = nA;
            A.dbg_counts[1 * A.G + gp] = nC;
            A.dbg_counts[2 * A.G + gp] = nG;
            A.dbg_counts[3 * A.G + gp] = nT;
            A.dbg_counts[4 * A.G + gp] = nDel + nOth;
            A.dbg_counts[5 * A.G + gp] = v.vthr;
            A.dbg_counts[6 * A.G + gp] = v.ithr;
            A.dbg_status[gp] = v.status;
        }
    }
    my_len = wave_sum(my_len);
    my_changed = wave_sum(my_changed);
    my_zero = wave_sum(my_zero);
    my_depth = wave_sum64(my_depth);
    if (lane == 0) {
        if (my_len) atomicAdd(&s_len, my_len);
        if (my_changed) atomicAdd(&s_changed, my_changed);
        if (my_zero) atomicAdd(&s_zero, my_zero);
        if (my_depth) atomicAdd(&s_depth, my_depth);
    }
    __syncthreads();
    if (tid < (u32)(TILE / 32)) A.flag_bits[(u64)w * (TILE / 32) + tid] = s_fbits[tid];
    if (s_nflag && n_items <= SORT_MAX) {
        // the ordered-depth replay needs this window's integer tallies: save them (rare windows only)
        if (tid == 0) {
            const u32 slab = atomicAdd(&A.counters[3], 1u);
            if (slab >= A.cap_slabs) report(A.status, slab, DE_CAPACITY_LATE);
            s_c1 = slab;
            A.win_slab[w] = slab;
        }
        __syncthreads();
        const u32 slab = s_c1;
        if (slab < A.cap_slabs) {
            u32 *dst = A.slabs + (u64)slab * 6u * TILE;
            for (u32 p = tid; p < (u32)TILE; p += TILE_THREADS) {
                u32 nA, nC, nG, nT, nDel, nOth;
                position_tallies(cnt, ((const u8 *)asm_w)[ASM_PAD + p], p, nA, nC, nG, nT, nDel, nOth);
                dst[0 * TILE + p] = nA; dst[1 * TILE + p] = nC; dst[2 * TILE + p] = nG;
                dst[3 * TILE + p] = nT; dst[4 * TILE + p] = nDel; dst[5 * TILE + p] = nOth;
            }
        }
    }
    if (tid == 0) {
        A.win_nflag[w] = s_nflag;
        if (s_nflag) atomicAdd(&A.counters[2], s_nflag);
        A.win_len[w] = s_len;
        if (s_changed) atomicAdd(&A.stats[s_c0].changed, (u64)s_changed);
        if (s_zero) atomicAdd(&A.stats[s_c0].zero_depth, (u64)s_zero);
        if (s_depth) atomicAdd(&A.stats[s_c0].depth_fx, s_depth);
    }
}

}  // namespace pp
