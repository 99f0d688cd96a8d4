// pp_k_prep.h -- prep_general: run validation, spans and the trim of one record that is not a single short M run
// (called by k_stream, one lane per record).
// Part of pp_kernels.hip (included there, in this order, and nowhere else: it defines __global__ kernels).
#pragma once

namespace pp {
#ifndef PP_PREP_INLINE
#define PP_PREP_INLINE __forceinline__
#endif

// every record that is not a single short M run inside its contig
__device__ PP_PREP_INLINE void prep_general(u64 a, u32 rs, u32 sl, u64 so, const u32 *cg, u32 nc, const u8 *seq,
                                               u64 c_lo, u64 c_hi, u32 *g_out, u32 *nk_out, u8 *fl_out, u64 *status) {
    // walk the runs (alignment.rs:178-194): spans and validity
    u64 ref_span = 0, read_span = 0;
    bool indel = false;
    for (u32 r = 0; r < nc; r++) {
        u32 op = cg[r], len = op >> 4, o = op & 15u;
        if (len == 0 || o > 8u) { report(status, a, DE_BAD_RUN); return; }
        if (o == PP_OP_M || o == PP_OP_EQ || o == PP_OP_X) { ref_span += len; read_span += len; }
        else if (o == PP_OP_I) { read_span += len; indel = true; }
        else if (o == PP_OP_D) { ref_span += len; indel = true; }
        else { report(status, a, DE_UNEXPECTED_OP); return; }
    }
    u32 o_first = cg[0] & 15u, o_last = cg[nc - 1] & 15u;
    if (!((o_first == PP_OP_M || o_first == PP_OP_EQ) && (o_last == PP_OP_M || o_last == PP_OP_EQ))) {
        report(status, a, DE_BAD_ENDS);
        return;
    }
    if (read_span != (u64)sl) { report(status, a, DE_LEN_MISMATCH); return; }
    if (ref_span >= 0x3FFFFFFFull) { report(status, a, DE_OVERFLOW); return; }

    const u64 clen = c_hi - c_lo;
    const u8 *s = seq + so;
    u32 n_entries = (u32)ref_span;
    if (!indel && sl <= FAST_MAX_LEN && (u64)rs + ref_span <= clen) {
        // fast class (=/X runs): k_tile loads the whole read anyway and trims it there, so the read
        // bytes are not touched here; bucketed by its untrimmed span
        *g_out = (u32)(c_lo + rs);
        *nk_out = n_entries;
        return;
    }
    // trim_bases_for_homopolymers (alignment.rs:364-378).  The last entry is the single base
    // seq[sl-1] (the last run is M/=).  `run` = number of trailing entries equal to it.
    u8 last = s[sl - 1];
    u32 run = 0;
    if (!indel) {
        run = sl - simple_trim_start(s, sl);
    } else {
        // walk the entries from the right end and stop at the first one that differs from the
        // last base (typically after 2-3 steps): runs in reverse; `pend` = bases inserted right
        // after the run being visited (they extend its last entry)
        u64 ro = sl;
        u32 pend = 0;
        bool stop = false;
        for (u32 r = nc; r-- > 0 && !stop;) {
            const u32 op = cg[r], len = op >> 4, o = op & 15u;
            if (o == PP_OP_I) { ro -= len; pend += len; continue; }
            if (o == PP_OP_D) {
                // last slot of the run: empty, or rewritten to the inserted bases; the others are empty
                if (pend == 1 && s[ro] == last) { run += 1; if (len > 1) stop = true; }
                else stop = true;
            } else {
                for (u32 t = 0; t < len; t++) {
                    const bool extended = (t == 0) && pend > 0;
                    if (!extended && s[ro - 1 - t] == last) run += 1; else { stop = true; break; }
                }
                ro -= len;
            }
            pend = 0;
        }
    }
    u32 nk = (n_entries > run) ? n_entries - run - 1u : 0u;
    if (nk == 0) return;  // contributes nothing; the reference never indexes the pileup for it
    if ((u64)rs + nk > clen) { report(status, a, DE_OUT_OF_BOUNDS); return; }
    *g_out = (u32)(c_lo + rs);
    *nk_out = nk;
    *fl_out = indel ? (u8)ENT_COMPLEX : (u8)ENT_PRETRIM;
}

}  // namespace pp
