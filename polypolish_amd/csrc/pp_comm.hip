// pp_comm.hip -- the one exchange of the multi-GPU polish: the ranks' polished bytes go to rank 0 over RCCL (xGMI).
// One process per GPU (bench.py / polypolish_amd.distributed under torch.distributed.run): rank 0 makes an
// ncclUniqueId, the launcher hands it to every rank, each rank joins with ncclCommInitRank on its context's device.
//   pp_polish_gather = ncclAllGather of (byte count, per-contig output offsets) + ONE group of ncclSend / ncclRecv
//   straight into rank 0's buffer at the exclusive-scan offsets (RCCL has no gatherv).
// The messages are small (<= 31 MB per rank for configs[4]) and point-to-point into rank 0: seven xGMI links are
// used at once, nothing is ring-shaped.
//
// librccl is loaded at run time (dlopen): a process that already holds an RCCL -- PyTorch brings its own -- keeps
// using that one, the CLI loads the system's.  Without a usable librccl pp_comm_* fail with PP_ERR_HIP; nothing
// else in the library depends on it.
#include <dlfcn.h>

#include <chrono>
#include <condition_variable>
#include <cstring>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "pp_internal.h"

// The handful of RCCL declarations this file needs (rccl.h, NCCL 2.x ABI), so that the library builds without the RCCL
// headers and runs without librccl as long as nobody asks for a communicator.
extern "C" {
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclUint8 = 1, ncclUint32 = 3, ncclUint64 = 5 } ncclDataType_t;
}

namespace {

struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};

Rccl &rccl() {
    static Rccl R = [] {
        Rccl r;
        // PP_RCCL_LIB: another library with the same nine entry points (tests/fake_rccl.cpp: a stand-in that runs several ranks
        // on ONE device, so that the one-process driver's RCCL route and a rank that fails to join can be exercised on a
        // one-GPU box)
        if (const char *alt = getenv("PP_RCCL_LIB")) r.lib = dlopen(alt, RTLD_NOW | RTLD_LOCAL);
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            if (r.lib) break;
            r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        }
        if (!r.lib) return r;
#define PP_SYM(field, sym) r.field = (decltype(r.field))dlsym(r.lib, sym)
        PP_SYM(GetUniqueId, "ncclGetUniqueId");
        PP_SYM(CommInitRank, "ncclCommInitRank");
        PP_SYM(CommDestroy, "ncclCommDestroy");
        PP_SYM(AllGather, "ncclAllGather");
        PP_SYM(Send, "ncclSend");
        PP_SYM(Recv, "ncclRecv");
        PP_SYM(GroupStart, "ncclGroupStart");
        PP_SYM(GroupEnd, "ncclGroupEnd");
        PP_SYM(GetErrorString, "ncclGetErrorString");
#undef PP_SYM
        r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllGather && r.Send && r.Recv && r.GroupStart &&
               r.GroupEnd && r.GetErrorString;
        return r;
    }();
    return R;
}

#define PP_NCCLCHK(ctx, expr)                                                                       \
    do {                                                                                            \
        ncclResult_t r__ = (expr);                                                                  \
        if (r__ != ncclSuccess)                                                                     \
            return (ctx)->fail(PP_ERR_HIP, "%s failed: %s", #expr, rccl().GetErrorString(r__));     \
    } while (0)

}  // namespace

static_assert(sizeof(ncclUniqueId) == PP_COMM_ID_BYTES, "PP_COMM_ID_BYTES");  // NCCL_UNIQUE_ID_BYTES

extern "C" int pp_comm_unique_id(void *id) {
    if (!id) return PP_ERR_ARG;
    if (!rccl().ok) return PP_ERR_HIP;
    ncclUniqueId u;
    if (rccl().GetUniqueId(&u) != ncclSuccess) return PP_ERR_HIP;
    memcpy(id, &u, sizeof u);
    return PP_OK;
}

extern "C" int pp_comm_init(pp_ctx *ctx, int rank, int world, const void *id) {
    if (!ctx || !id || world < 1 || rank < 0 || rank >= world) return PP_ERR_ARG;
    if (int rdy = pp_ctx_wait(ctx)) return rdy;
    if (!rccl().ok) return ctx->fail(PP_ERR_HIP, "librccl could not be loaded: the multi-GPU gather is not available");
    if (ctx->comm) return ctx->fail(PP_ERR_ARG, "pp_comm_init: the context already has a communicator");
    PP_HIPCHK(ctx, hipSetDevice(ctx->device));
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    // ncclCommInitRank returns when EVERY rank has joined: a rank that failed before it got here (or was never started)
    // would keep the others inside it for good.  The call runs on a helper thread and is waited for
    // with a deadline (PP_COMM_TIMEOUT seconds, default 120); past it this rank returns an error -- the helper stays behind in
    // the call (there is no communicator yet to abort), which costs a failing process one parked thread.
    struct Join {
        std::mutex m;
        std::condition_variable cv;
        bool done = false;
        ncclResult_t r = ncclSuccess;
        ncclComm_t c = nullptr;
    };
    auto join = std::make_shared<Join>();
    const int device = ctx->device;
    std::thread([join, world, u, rank, device]() {
        (void)hipSetDevice(device);
        ncclComm_t c = nullptr;
        const ncclResult_t r = rccl().CommInitRank(&c, world, u, rank);
        std::lock_guard<std::mutex> lk(join->m);
        join->r = r;
        join->c = c;
        join->done = true;
        join->cv.notify_all();
    }).detach();
    const long limit = getenv("PP_COMM_TIMEOUT") ? std::max(1L, atol(getenv("PP_COMM_TIMEOUT"))) : 120L;
    ncclComm_t c = nullptr;
    {
        std::unique_lock<std::mutex> lk(join->m);
        if (!join->cv.wait_for(lk, std::chrono::seconds(limit), [&] { return join->done; }))
            return ctx->fail(PP_ERR_HIP, "ncclCommInitRank (rank %d of %d) did not return within %ld s: a rank of the job has not joined "
                                         "(it failed before it got there, or was never started)", rank, world, limit);
        if (join->r != ncclSuccess) return ctx->fail(PP_ERR_HIP, "ncclCommInitRank failed: %s", rccl().GetErrorString(join->r));
        c = join->c;
    }
    ctx->comm = c;
    ctx->comm_rank = rank;
    ctx->comm_world = world;
    return PP_OK;
}

extern "C" void pp_comm_destroy(pp_ctx *ctx) {
    if (!ctx || !ctx->comm) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    (void)rccl().CommDestroy((ncclComm_t)ctx->comm);
    ctx->comm = nullptr;
    ctx->comm_world = 1;
    ctx->comm_rank = 0;
}

extern "C" int pp_polish_gather(pp_ctx *ctx, uint8_t *gathered, uint64_t cap, uint64_t *rank_len, uint64_t *rank_contig_off) {
    if (!ctx) return PP_ERR_ARG;
    if (!ctx->comm) return ctx->fail(PP_ERR_ARG, "pp_polish_gather without pp_comm_init");
    if (!ctx->job_done) return ctx->fail(PP_ERR_ARG, "no finished polish job");
    PP_HIPCHK(ctx, hipSetDevice(ctx->device));
    ncclComm_t comm = (ncclComm_t)ctx->comm;
    hipStream_t st = ctx->stream;
    const int world = ctx->comm_world, rank = ctx->comm_rank;
    const uint32_t nc = ctx->n_contigs;
    // ---- every rank learns every rank's byte count, per-contig offsets and buffer size ----
    const size_t words = (size_t)nc + 3;  // byte count, contig_out_off[0..nc], cap (rank 0's is the one that counts)
    if (int rc = pp::dev_ensure(ctx, ctx->b_comm, (size_t)(world + 1) * words * 8)) return rc;
    uint64_t *d_mine = (uint64_t *)ctx->b_comm.p, *d_all = d_mine + words;
    std::vector<uint64_t> mine(words), all((size_t)world * words);
    mine[0] = ctx->total_out;
    for (uint32_t c = 0; c <= nc; c++) mine[1 + c] = ctx->contig_out_off[c];
    mine[nc + 2] = gathered ? cap : 0;
    PP_HIPCHK(ctx, hipMemcpyAsync(d_mine, mine.data(), words * 8, hipMemcpyHostToDevice, st));
    PP_NCCLCHK(ctx, rccl().AllGather(d_mine, d_all, words, ncclUint64, comm, st));
    PP_HIPCHK(ctx, hipMemcpyAsync(all.data(), d_all, (size_t)world * words * 8, hipMemcpyDeviceToHost, st));
    PP_HIPCHK(ctx, hipStreamSynchronize(st));
    uint64_t total = 0;
    std::vector<uint64_t> start(world);
    for (int r = 0; r < world; r++) {
        start[r] = total;
        total += all[(size_t)r * words];
        if (rank_len) rank_len[r] = all[(size_t)r * words];
        if (rank_contig_off) memcpy(rank_contig_off + (size_t)r * (nc + 1), &all[(size_t)r * words + 1], ((size_t)nc + 1) * 8);
    }
    // Whether rank 0's buffer holds it all is decided by EVERY rank from the gathered figures, before anybody posts a
    // send: a rank 0 that backed out on its own would leave the others waiting in ncclSend for good.
    const uint64_t cap0 = all[nc + 2];
    if (total > cap0)
        return ctx->fail(PP_ERR_ARG, "pp_polish_gather: %llu bytes do not fit rank 0's buffer of %llu (every rank returns this)",
                         (unsigned long long)total, (unsigned long long)cap0);
    // ---- the bytes: everybody sends to rank 0, which receives at the exclusive-scan offsets ----
    if (rank == 0) {
        if (ctx->total_out)
            PP_HIPCHK(ctx, hipMemcpyAsync(gathered, ctx->b_out.p, ctx->total_out, hipMemcpyDeviceToDevice, st));
        // a group that was opened is closed on every path: an error inside it is kept and reported afterwards
        ncclResult_t first_bad = ncclSuccess;
        PP_NCCLCHK(ctx, rccl().GroupStart());
        for (int r = 1; r < world && first_bad == ncclSuccess; r++)
            if (all[(size_t)r * words]) first_bad = rccl().Recv(gathered + start[r], all[(size_t)r * words], ncclUint8, r, comm, st);
        const ncclResult_t ended = rccl().GroupEnd();
        if (first_bad != ncclSuccess) return ctx->fail(PP_ERR_HIP, "ncclRecv failed: %s", rccl().GetErrorString(first_bad));
        if (ended != ncclSuccess) return ctx->fail(PP_ERR_HIP, "ncclGroupEnd failed: %s", rccl().GetErrorString(ended));
    } else if (ctx->total_out) {
        PP_NCCLCHK(ctx, rccl().Send(ctx->b_out.p, ctx->total_out, ncclUint8, 0, comm, st));
    }
    PP_HIPCHK(ctx, hipStreamSynchronize(st));
    return PP_OK;
}

// pp_polish_gather for a caller without HIP of its own (the one-process multi-GPU driver, pp_driver.cpp): rank 0 receives
// into a library-owned device buffer and copies the whole FASTA payload to `host_out` (>= cap bytes) in ONE transfer;
// the other ranks pass host_out = nullptr.  Same collective contract as pp_polish_gather: every rank calls it.
extern "C" int pp_polish_gather_to_host_(pp_ctx *ctx, uint8_t *host_out, uint64_t cap, uint64_t *rank_len, uint64_t *rank_contig_off) {
    if (!ctx) return PP_ERR_ARG;
    if (!ctx->comm) return ctx->fail(PP_ERR_ARG, "pp_polish_gather without pp_comm_init");
    PP_HIPCHK(ctx, hipSetDevice(ctx->device));
    const bool root = ctx->comm_rank == 0;
    // A rank 0 that cannot take the bytes (no host buffer, no room on its device) still JOINS the collective, with a buffer
    // of zero bytes: every rank then learns from the gathered figures that nothing fits and returns the same error, where a
    // rank 0 that backed out here would leave the others waiting in the AllGather for good.
    int root_rc = PP_OK;
    if (root && !host_out) root_rc = ctx->fail(PP_ERR_ARG, "pp_polish_gather_to_host_: rank 0 needs a buffer");
    if (root && root_rc == PP_OK) root_rc = pp::dev_ensure(ctx, ctx->b_gather, (size_t)cap + 16);
    const std::string root_err = root_rc ? ctx->err : std::string();
    const bool have = root && root_rc == PP_OK;
    std::vector<uint64_t> lens((size_t)ctx->comm_world, 0);
    if (int rc = pp_polish_gather(ctx, have ? (uint8_t *)ctx->b_gather.p : nullptr, have ? cap : 0, lens.data(), rank_contig_off)) {
        if (root_rc) { ctx->err = root_err; return root_rc; }
        return rc;
    }
    if (root_rc) { ctx->err = root_err; return root_rc; }  // (only when there was nothing to gather at all)
    if (rank_len) memcpy(rank_len, lens.data(), lens.size() * 8);
    if (root) {
        uint64_t total = 0;
        for (uint64_t l : lens) total += l;
        if (total) PP_HIPCHK(ctx, hipMemcpyAsync(host_out, ctx->b_gather.p, total, hipMemcpyDeviceToHost, ctx->stream));
        PP_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    return PP_OK;
}
