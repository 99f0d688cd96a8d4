// pp_shard.cpp -- host side of the multi-GPU polish: which rank polishes what, and how the ranks' polished bytes go
// back together in FASTA order.  No reference counterpart (the reference is one thread on one CPU); the partition
// follows SURVEY.md section 8(e) / BASELINE.json configs[3] and [4]:
//   * whole contigs, longest-processing-time greedy on their alignment counts (configs[3]);
//   * a contig that carries more than one rank's share of the alignments is cut into up to `world` windows on
//     2048-bp tile boundaries, one per rank (configs[4]).
// A rank polishes the records that reach its units (pp_shard_split, pp_shard_dev.hip -- or simply all records) with
// pp_polish_set_emit(ranges of its units): the device works on its windows only, over a compact assembly of what the rank
// owns, and every owned position still sees ALL of its alignments in file order -- the order-dependent f64 depth
// (src/pileup.rs:64) stays exact and no halo bookkeeping is needed.  The only exchange of the polish itself is the
// collection of the polished bytes (pp_comm.hip).
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "polypolish_hip.h"
#include "pp_host.h"

namespace {
constexpr uint64_t WINDOW_ALIGN = 2048;  // the device's tile width: windows start on tile boundaries
}

extern "C" int pp_shard_plan_create(uint32_t n_contigs, const uint64_t *contig_off, const uint64_t *aln_per_contig,
                                    uint32_t world, uint64_t min_window, pp_shard_plan **out) {
    if (!out) return PP_ERR_ARG;
    *out = nullptr;
    if (!contig_off || !aln_per_contig || n_contigs == 0 || world == 0) return PP_ERR_ARG;
    if (min_window == 0) min_window = 1u << 16;
    struct Unit { uint32_t contig; uint64_t lo, hi; double weight; uint32_t rank; };
    std::vector<Unit> units;
    double total = 0;
    for (uint32_t c = 0; c < n_contigs; c++) total += (double)aln_per_contig[c];
    const double share = total / world;
    for (uint32_t c = 0; c < n_contigs; c++) {
        const uint64_t len = contig_off[c + 1] - contig_off[c];
        uint64_t pieces = 1;
        if (world > 1 && share > 0 && (double)aln_per_contig[c] > share) {
            const uint64_t by_share = (uint64_t)((double)aln_per_contig[c] / share + 0.999999);
            pieces = std::min<uint64_t>(std::min<uint64_t>(world, by_share), std::max<uint64_t>(1, len / min_window));
        }
        std::vector<uint64_t> cuts{0};
        for (uint64_t j = 1; j < pieces; j++) {
            const uint64_t x = (len * j / pieces) / WINDOW_ALIGN * WINDOW_ALIGN;
            if (x > cuts.back()) cuts.push_back(x);
        }
        cuts.push_back(len);
        for (size_t j = 0; j + 1 < cuts.size(); j++) {
            Unit u;
            u.contig = c; u.lo = cuts[j]; u.hi = cuts[j + 1]; u.rank = 0;
            u.weight = (len ? (double)aln_per_contig[c] * (double)(u.hi - u.lo) / (double)len : 0.0) + 1e-9 * (double)(u.hi - u.lo);
            units.push_back(u);
        }
    }
    // longest processing time first; the windows of one contig go to different ranks (a rank emits ONE range per contig)
    std::vector<size_t> order(units.size());
    for (size_t i = 0; i < order.size(); i++) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return units[a].weight > units[b].weight; });
    std::vector<double> load(world, 0.0);
    std::vector<std::vector<uint32_t>> taken(n_contigs);  // ranks that already hold a window of the contig
    for (size_t i : order) {
        Unit &u = units[i];
        uint32_t best = world;
        for (uint32_t r = 0; r < world; r++) {
            const auto &t = taken[u.contig];
            if (std::find(t.begin(), t.end(), r) != t.end()) continue;
            if (best == world || load[r] < load[best]) best = r;
        }
        if (best == world) return PP_ERR_ARG;  // cannot happen: a contig has at most `world` windows
        u.rank = best;
        load[best] += u.weight;
        taken[u.contig].push_back(best);
    }
    pp_shard_plan *p = (pp_shard_plan *)calloc(1, sizeof *p);
    if (!p) return PP_ERR_LIMIT;
    const size_t n = units.size();
    p->n_units = (uint32_t)n;
    p->world = world;
    p->n_contigs = n_contigs;
    p->contig = (uint32_t *)malloc((n ? n : 1) * 4);
    p->rank = (uint32_t *)malloc((n ? n : 1) * 4);
    p->lo = (uint64_t *)malloc((n ? n : 1) * 8);
    p->hi = (uint64_t *)malloc((n ? n : 1) * 8);
    if (!p->contig || !p->rank || !p->lo || !p->hi) {  // out of host memory
        pp_shard_plan_free(p);
        return PP_ERR_LIMIT;
    }
    for (size_t i = 0; i < n; i++) {
        p->contig[i] = units[i].contig; p->rank[i] = units[i].rank; p->lo[i] = units[i].lo; p->hi[i] = units[i].hi;
    }
    *out = p;
    return PP_OK;
}

extern "C" void pp_shard_plan_free(pp_shard_plan *p) {
    if (!p) return;
    free(p->contig); free(p->rank); free(p->lo); free(p->hi);
    free(p);
}

// The ranges one rank emits, in the form pp_polish_set_emit takes: [lo, hi) per contig, empty for a contig it has
// no unit of.
extern "C" int pp_shard_emit_ranges(const pp_shard_plan *p, uint32_t rank, uint64_t *emit_lo, uint64_t *emit_hi) {
    if (!p || !emit_lo || !emit_hi || rank >= p->world) return PP_ERR_ARG;
    for (uint32_t c = 0; c < p->n_contigs; c++) emit_lo[c] = emit_hi[c] = 0;
    for (uint32_t u = 0; u < p->n_units; u++)
        if (p->rank[u] == rank) { emit_lo[p->contig[u]] = p->lo[u]; emit_hi[p->contig[u]] = p->hi[u]; }
    return PP_OK;
}

// Put the ranks' polished bytes back in assembly order.  rank_bytes[r] = what rank r's pp_polish_result gave
// (its units in contig order), rank_contig_off[r] = its contig_out_off (n_contigs + 1).  out receives the polished
// assembly (total = sum of all), contig_out_off (n_contigs + 1) the start of every contig in it.
extern "C" int pp_shard_assemble(const pp_shard_plan *p, const uint8_t *const *rank_bytes, const uint64_t *const *rank_contig_off,
                                 uint8_t *out, uint64_t *contig_out_off) {
    if (!p || !rank_bytes || !rank_contig_off || !contig_out_off) return PP_ERR_ARG;
    // units are listed contig by contig, windows in position order
    uint64_t w = 0;
    uint32_t u = 0;
    for (uint32_t c = 0; c < p->n_contigs; c++) {
        contig_out_off[c] = w;
        for (; u < p->n_units && p->contig[u] == c; u++) {
            const uint32_t r = p->rank[u];
            const uint64_t a = rank_contig_off[r][c], b = rank_contig_off[r][c + 1];
            if (out && b > a) memcpy(out + w, rank_bytes[r] + a, b - a);
            w += b - a;
        }
    }
    contig_out_off[p->n_contigs] = w;
    return PP_OK;
}

// ---- the library's own window-order mirrors (pp_internal.h) ------------------------------------------------------------
#include <map>
#include <mutex>
namespace {
std::mutex g_mirror_mu;
std::map<const void *, std::pair<const char *, const char *>> g_mirrors;  // owner -> [lo, hi)
}  // namespace
void pp_mirror_register_(const void *owner, const void *p, size_t bytes) {
    std::lock_guard<std::mutex> lk(g_mirror_mu);
    if (p && bytes) g_mirrors[owner] = {(const char *)p, (const char *)p + bytes};
    else g_mirrors.erase(owner);
}
void pp_mirror_forget_(const void *owner) {
    std::lock_guard<std::mutex> lk(g_mirror_mu);
    g_mirrors.erase(owner);
}
bool pp_mirror_trusted_(const void *p, size_t bytes) {
    std::lock_guard<std::mutex> lk(g_mirror_mu);
    const char *lo = (const char *)p, *hi = lo + bytes;
    for (const auto &kv : g_mirrors)
        if (lo >= kv.second.first && hi <= kv.second.second) return true;
    return false;
}
