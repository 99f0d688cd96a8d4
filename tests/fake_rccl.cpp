// fake_rccl.cpp -- TEST INFRASTRUCTURE: a stand-in for librccl with the nine entry points pp_comm.hip binds (PP_RCCL_LIB), for
// ranks that are THREADS of one process whose contexts may share one device.  It lets a one-GPU box run what RCCL itself
// refuses there: the one-process multi-GPU driver's RCCL route (pp_polish_files_multi with PP_GATHER=rccl: ncclCommInitRank from
// one thread per context, pp_polish_gather's ncclAllGather + one group of ncclSend / ncclRecv), and a rank that never joins
// (FAKE_RCCL_FAIL_RANK=r: ncclCommInitRank of rank r returns an error at once, the others wait for it as RCCL's would --
// what pp_comm_init's deadline is for).  Rendezvous through process-global tables; the bytes move with hipMemcpyAsync on the
// caller's stream, behind a synchronisation of the sender's stream.  Built by the test: hipcc -shared -fPIC tests/fake_rccl.cpp.
#include <hip/hip_runtime.h>

#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

extern "C" {
typedef struct fakeComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;
typedef int ncclDataType_t;
}

namespace {
struct Group {  // the ranks of one communicator
    int world = 0, joined = 0;
    // one collective at a time: every rank posts its buffers, the last one in releases the round
    int round = 0, posted = 0;
    std::vector<const void *> send;
    std::vector<hipStream_t> stream;
    // point to point: a sender's (buffer, bytes) waiting for rank 0's receive
    std::map<int, std::pair<const void *, size_t>> mail;
    std::map<int, bool> taken;
};
std::mutex g_mu;
std::condition_variable g_cv;
std::map<std::string, Group> g_groups;
int g_ids = 0;
size_t dtype_size(ncclDataType_t t) { return t == 1 ? 1 : (t == 3 ? 4 : 8); }
}  // namespace

struct fakeComm {
    Group *g;
    int rank;
};

extern "C" {
ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
    std::lock_guard<std::mutex> lk(g_mu);
    memset(id, 0, sizeof *id);
    snprintf(id->internal, sizeof id->internal, "fake-%d", ++g_ids);
    return 0;
}
ncclResult_t ncclCommInitRank(ncclComm_t *comm, int world, ncclUniqueId id, int rank) {
    if (const char *f = getenv("FAKE_RCCL_FAIL_RANK"))
        if (atoi(f) == rank) return 2;  // (this rank fails before it joins)
    std::unique_lock<std::mutex> lk(g_mu);
    Group &g = g_groups[std::string(id.internal)];
    g.world = world;
    g.send.resize(world);
    g.stream.resize(world);
    g.joined++;
    g_cv.notify_all();
    g_cv.wait(lk, [&] { return g.joined >= world; });  // (as the real one: returns when every rank has joined)
    *comm = new fakeComm{&g, rank};
    return 0;
}
ncclResult_t ncclCommDestroy(ncclComm_t c) { delete c; return 0; }
const char *ncclGetErrorString(ncclResult_t r) { return r ? "fake rccl: the rank failed" : "no error"; }
ncclResult_t ncclGroupStart() { return 0; }
ncclResult_t ncclGroupEnd() { return 0; }
static void barrier(Group &g, std::unique_lock<std::mutex> &lk) {  // (reusable: a generation counter)
    const int gen = g.round;
    if (++g.posted == g.world) {
        g.posted = 0;
        g.round++;
        g_cv.notify_all();
    } else g_cv.wait(lk, [&] { return g.round != gen; });
}
ncclResult_t ncclAllGather(const void *sendbuf, void *recvbuf, size_t count, ncclDataType_t dt, ncclComm_t c, hipStream_t st) {
    (void)hipStreamSynchronize(st);  // (what the caller put into sendbuf on its stream is there)
    Group &g = *c->g;
    std::unique_lock<std::mutex> lk(g_mu);
    g.send[c->rank] = sendbuf;
    barrier(g, lk);                  // every rank has posted its buffer
    const std::vector<const void *> src(g.send);
    lk.unlock();
    const size_t bytes = count * dtype_size(dt);
    for (int r = 0; r < g.world; r++) (void)hipMemcpyAsync((char *)recvbuf + (size_t)r * bytes, src[r], bytes, hipMemcpyDeviceToDevice, st);
    (void)hipStreamSynchronize(st);
    lk.lock();
    barrier(g, lk);                  // every rank has copied: the buffers may change
    return 0;
}
ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t c, hipStream_t st) {
    (void)peer;
    (void)hipStreamSynchronize(st);
    Group &g = *c->g;
    std::unique_lock<std::mutex> lk(g_mu);
    g.mail[c->rank] = {buf, count * dtype_size(dt)};
    g.taken[c->rank] = false;
    g_cv.notify_all();
    g_cv.wait(lk, [&] { return g.taken[c->rank]; });  // (the receiver has copied the bytes out)
    g.mail.erase(c->rank);
    return 0;
}
ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t c, hipStream_t st) {
    Group &g = *c->g;
    std::unique_lock<std::mutex> lk(g_mu);
    g_cv.wait(lk, [&] { return g.mail.count(peer) && !g.taken[peer]; });
    const std::pair<const void *, size_t> m = g.mail[peer];
    lk.unlock();
    (void)hipMemcpyAsync(buf, m.first, std::min(m.second, count * dtype_size(dt)), hipMemcpyDeviceToDevice, st);
    (void)hipStreamSynchronize(st);
    lk.lock();
    g.taken[peer] = true;
    g_cv.notify_all();
    return 0;
}
}
