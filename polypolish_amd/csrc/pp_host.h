// pp_host.h -- host-side utilities shared by the SAM ingest (pp_ingest.cpp) and the filter driver
// (pp_filter_host.cpp): huge-page growable arrays, a fork-join parallel_for, a read-only file mapping.
#pragma once
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <ctime>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

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

namespace pph {

// Growable array for the multi-GB buffers of a large job: anonymous mmap backed by transparent huge
// pages where the kernel allows (512x fewer page faults while many threads first-touch it, and a much
// cheaper teardown), grown with mremap, and never zero-filled by us.
template <typename T>
struct HugeBuf {
    static constexpr size_t HUGE = size_t(2) << 20;
    T *p = nullptr;
    size_t n = 0, cap = 0, mapped = 0;  // elements, elements, bytes
    HugeBuf() = default;
    HugeBuf(const HugeBuf &) = delete;
    HugeBuf &operator=(const HugeBuf &) = delete;
    HugeBuf(HugeBuf &&o) noexcept : p(o.p), n(o.n), cap(o.cap), mapped(o.mapped) { o.p = nullptr; o.n = o.cap = o.mapped = 0; }
    ~HugeBuf() { if (p) munmap(p, mapped); }
    T *data() { return p; }
    const T *data() const { return p; }
    size_t size() const { return n; }
    bool empty() const { return n == 0; }
    T &operator[](size_t i) { return p[i]; }
    const T &operator[](size_t i) const { return p[i]; }
    T &back() { return p[n - 1]; }
    void reserve(size_t m) {
        if (m <= cap) return;
        size_t bytes = std::max(m, cap * 2) * sizeof(T);
        bytes = (bytes + HUGE - 1) / HUGE * HUGE;
        void *q = p ? mremap(p, mapped, bytes, MREMAP_MAYMOVE)
                    : mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (q == MAP_FAILED) throw std::bad_alloc();
        madvise(q, bytes, MADV_HUGEPAGE);
        p = (T *)q;
        mapped = bytes;
        cap = bytes / sizeof(T);
    }
    void resize(size_t m) { reserve(m); n = m; }
    void push_back(const T &v) {
        if (n == cap) reserve(n + 1);
        p[n++] = v;
    }
};

// f(lo, hi, t): contiguous ranges of [0, n) in order, one per thread index t
template <typename F>
void parallel_for(size_t n, unsigned threads, F f) {
    if (threads <= 1 || n < 2) { f((size_t)0, n, 0u); return; }
    std::vector<std::thread> pool;
    const size_t per = (n + threads - 1) / threads;
    for (unsigned t = 0; t < threads; t++) {
        const size_t lo = std::min(n, (size_t)t * per), hi = std::min(n, lo + per);
        if (lo < hi) pool.emplace_back([=] { f(lo, hi, t); });
    }
    for (auto &th : pool) th.join();
}

// The command-line program leaves with _exit as soon as its output is written: unmapping tens of GB of input mappings
// (0.14 s per 15 GB, and every mmap / hipMalloc of the process waits for the lock meanwhile) and releasing device buffers is
// then the kernel's business at exit.  Library callers never set this: their mappings are unmapped on a background thread.
inline bool &process_leaving_soon() { static bool v = false; return v; }

// seconds since this process was started (its start time in /proc/self/stat against CLOCK_BOOTTIME; 10 ms resolution):
// what a PP_TIMING line of a command driver counts from -- exec, dynamic linking and the HIP runtime's start-up included
inline double seconds_since_process_start() {
    static const double started = [] {
        double st = -1.0;
        if (FILE *f = fopen("/proc/self/stat", "r")) {
            char buf[1024];
            const size_t n = fread(buf, 1, sizeof buf - 1, f);
            fclose(f);
            buf[n] = 0;
            if (const char *p = strrchr(buf, ')')) {  // fields after the command name: state is the 3rd, starttime the 22nd
                int field = 2;
                for (p++; *p && field < 22; p++)  // (every space opens the next field: p ends up on the first digit of the 22nd)
                    if (*p == ' ') field++;
                if (field == 22) st = strtod(p, nullptr) / (double)sysconf(_SC_CLK_TCK);
            }
        }
        return st;
    }();
    struct timespec ts;
    if (started < 0 || clock_gettime(CLOCK_BOOTTIME, &ts) != 0) return -1.0;
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec - started;
}

// PP_INGEST_THREADS, else one thread per 4 MiB of text up to min(cores, 64)
inline unsigned host_threads(size_t text_bytes) {
    if (const char *e = getenv("PP_INGEST_THREADS")) return (unsigned)std::max(1, std::min(64, atoi(e)));
    return std::max(1u, std::min({std::thread::hardware_concurrency(), 64u, (unsigned)(text_bytes / (4u << 20)) + 1u}));
}

// Pre-faulting a file mapping (page-cache pages into this process's page tables) with a few threads.  A host-to-device
// copy out of a mapping nobody has touched faults it in page by page, 4 KiB at a time: 13 GB/s instead of the 56 GB/s
// the link delivers (measured on a 1.2 GB SAM file, tools/microbench/h2d2.hip: 89 ms against 3 + 21 ms).
#ifndef MADV_POPULATE_READ
#define MADV_POPULATE_READ 22

#endif
inline void populate_mapping(void *map, size_t size, unsigned threads = 8) {
    if (!map || !size) return;
    threads = std::max(1u, std::min(threads, (unsigned)(size >> 24) + 1u));
    std::vector<std::thread> th;
    for (unsigned t = 0; t < threads; t++)
        th.emplace_back([=] {
            const size_t page = 4096;
            size_t a = (size / threads * t) & ~(page - 1), b = t + 1 == threads ? size : (size / threads * (t + 1)) & ~(page - 1);
            if (b <= a) return;
            if (madvise((char *)map + a, b - a, MADV_POPULATE_READ) == 0) return;
            volatile char sink = 0;  // kernels before 5.14: touch every page
            for (size_t q = a; q < b; q += page) sink += ((const volatile char *)map)[q];
            (void)sink;
        });
    for (auto &x : th) x.join();
}

// Files opened ahead of their use (the driver does this while the HIP runtime initialises): mapped, and pre-faulted by a
// background thread.  FileText::open_file adopts the mapping of a path that was prefetched.
struct PrefetchedFile {
    std::string path;
    void *map = nullptr;
    size_t size = 0;
    int fd = -1;
    const void *owner = nullptr;  // who asked for it (a job's context): prefetch_drop_all releases only its own
    std::thread worker;
};
inline std::mutex &prefetch_mutex() { static std::mutex m; return m; }
inline std::vector<PrefetchedFile *> &prefetch_list() { static std::vector<PrefetchedFile *> v; return v; }
inline void prefetch_file(const char *path, const void *owner) {
    struct stat st;
    if (::stat(path, &st) != 0 || !S_ISREG(st.st_mode) || st.st_size <= 0) return;  // never open a pipe just to look at it
    const int fd = ::open(path, O_RDONLY);
    if (fd < 0) return;
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || st.st_size <= 0) { ::close(fd); return; }
    void *map = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    if (map == MAP_FAILED) { ::close(fd); return; }
    PrefetchedFile *f = new PrefetchedFile;
    f->path = path; f->map = map; f->size = (size_t)st.st_size; f->fd = fd; f->owner = owner;
    f->worker = std::thread([f] { populate_mapping(f->map, f->size); });
    std::lock_guard<std::mutex> lock(prefetch_mutex());
    prefetch_list().push_back(f);
}
inline PrefetchedFile *prefetch_take(const char *path) {  // nullptr if the path was not prefetched
    PrefetchedFile *f = nullptr;
    {
        std::lock_guard<std::mutex> lock(prefetch_mutex());
        auto &v = prefetch_list();
        for (size_t i = 0; i < v.size(); i++)
            if (v[i]->path == path) { f = v[i]; v.erase(v.begin() + (long)i); break; }
    }
    if (f && f->worker.joinable()) f->worker.join();
    return f;
}
inline void prefetch_drop_all(const void *owner) {  // whatever `owner` had prefetched and never opened
    for (;;) {
        PrefetchedFile *f = nullptr;
        {
            std::lock_guard<std::mutex> lock(prefetch_mutex());
            auto &v = prefetch_list();
            for (size_t i = v.size(); i-- > 0;)
                if (v[i]->owner == owner) { f = v[i]; v.erase(v.begin() + (long)i); break; }
            if (!f) return;
        }
        if (f->worker.joinable()) f->worker.join();
        munmap(f->map, f->size);
        ::close(f->fd);
        delete f;
    }
}

// BufRead::lines() yields an error for a line that is not valid UTF-8, and every loader of the reference turns that into
// "unable to load ..." (src/alignment.rs:238-240, src/filter.rs:119-121, src/misc.rs:109-111).  Lines are ASCII in
// practice: eight bytes at a time until a byte with its top bit set shows up, the full check (overlong forms, surrogates,
// code points beyond U+10FFFF as Rust's str::from_utf8 rejects them) only from there.
inline bool valid_utf8(const char *s, size_t n) {
    size_t i = 0;
    for (; i + 8 <= n; i += 8) {
        uint64_t w;
        memcpy(&w, s + i, 8);
        if (w & 0x8080808080808080ull) break;
    }
    while (i < n) {
        const unsigned char c = (unsigned char)s[i];
        if (c < 0x80) { i++; continue; }
        size_t need;
        uint32_t cp;
        if ((c & 0xE0) == 0xC0) { need = 1; cp = c & 0x1Fu; if (cp < 2) return false; }
        else if ((c & 0xF0) == 0xE0) { need = 2; cp = c & 0x0Fu; }
        else if ((c & 0xF8) == 0xF0) { need = 3; cp = c & 0x07u; if (cp > 4) return false; }
        else return false;
        for (size_t k = 1; k <= need; k++) {
            if (i + k >= n) return false;  // truncated sequence
            const unsigned char cc = (unsigned char)s[i + k];
            if ((cc & 0xC0) != 0x80) return false;
            cp = (cp << 6) | (cc & 0x3Fu);
        }
        if (need == 2 && (cp < 0x800 || (cp >= 0xD800 && cp <= 0xDFFF))) return false;
        if (need == 3 && (cp < 0x10000 || cp > 0x10FFFF)) return false;
        i += need + 1;
    }
    return true;
}

// A whole file as read-only bytes: mmap for regular files, read() for pipes.
struct FileText {
    const char *text = nullptr;
    size_t size = 0;
    void *map = nullptr;
    int fd = -1;
    std::vector<char> fallback;
    FileText() = default;
    FileText(const FileText &) = delete;
    FileText &operator=(const FileText &) = delete;
    ~FileText() { close_file(); }
    void close_file() {
        if (map) munmap(map, size);
        if (fd >= 0) ::close(fd);
        map = nullptr;
        fd = -1;
    }
    bool open_file(const char *path) {
        if (PrefetchedFile *f = prefetch_take(path)) {  // opened, mapped and pre-faulted ahead of time
            map = f->map; size = f->size; fd = f->fd;
            text = (const char *)map;
            delete f;
            return true;
        }
        fd = ::open(path, O_RDONLY);
        if (fd < 0) return false;
        struct stat st;
        if (fstat(fd, &st) != 0) return false;
        if (S_ISREG(st.st_mode) && st.st_size > 0) {
            size = (size_t)st.st_size;
            map = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
            if (map == MAP_FAILED) { map = nullptr; return false; }
            madvise(map, size, MADV_SEQUENTIAL);
            text = (const char *)map;
            return true;
        }
        char tmp[1 << 16];
        ssize_t r;
        while ((r = ::read(fd, tmp, sizeof tmp)) > 0) fallback.insert(fallback.end(), tmp, tmp + r);
        if (r < 0) return false;
        text = fallback.data();
        size = fallback.size();
        return true;
    }
};

// [beg, end) of slice t of `threads` line-aligned slices of the text
inline void line_slices(const char *text, size_t size, unsigned threads, std::vector<const char *> &cut) {
    cut.assign(threads + 1, text + size);
    const char *p = text, *end = text + size;
    for (unsigned t = 0; t < threads; t++) {
        cut[t] = p;
        const char *want = (t + 1 == threads) ? end : text + (size / threads) * (t + 1);
        if (want < p) want = p;
        if (want < end) {
            const char *nl = (const char *)memchr(want, '\n', (size_t)(end - want));
            want = nl ? nl + 1 : end;
        }
        p = want;
    }
    cut[threads] = end;
}

// ---- text of the run log (the reference pins these three with unit tests; exported as pp_log_text) ----
// qscore, polish.rs:290-300
inline std::string qscore(double identity) {
    if (identity >= 100.0) return "Q\xE2\x88\x9E";
    if (identity <= 0.0) return "Q0";
    char buf[64];
    snprintf(buf, sizeof buf, "Q%.2f", -10.0 * log10(1.0 - identity / 100.0));
    return buf;
}

// format_duration, misc.rs:195-201
inline std::string format_duration_us(uint64_t us) {
    char buf[64];
    snprintf(buf, sizeof buf, "%llu:%02llu:%02llu.%06llu", (unsigned long long)(us / 1000000 / 3600),
             (unsigned long long)(us / 1000000 / 60 % 60), (unsigned long long)(us / 1000000 % 60),
             (unsigned long long)(us % 1000000));
    return buf;
}
inline std::string format_duration(double seconds) { return format_duration_us((uint64_t)(seconds * 1e6)); }

// f64::to_string: the shortest decimal string that reads back as the same double, never in exponent form
inline std::string shortest_decimal(double v) {
    char b[400];
    for (int decimals = 0; decimals <= 340; decimals++) {
        snprintf(b, sizeof b, "%.*f", decimals, v);
        if (strtod(b, nullptr) == v) break;
    }
    return b;
}

// get_percentile_name, filter.rs:262-270
inline std::string percentile_name(double p) {
    const std::string s = shortest_decimal(p);
    const char *suffix = "th";
    if (s.back() == '1' && p != 11.0) suffix = "st";
    else if (s.back() == '2' && p != 12.0) suffix = "nd";
    else if (s.back() == '3' && p != 13.0) suffix = "rd";
    return s + suffix + " percentile";
}

}  // namespace pph
